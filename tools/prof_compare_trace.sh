#!/bin/bash
# rocprofv3 kernel stats of tools/profile_compare.py (one setwise compare shape); RK_OPTS passes engine options through.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
tag=${1:-base}
OUT=$PWD/gpurun_out/cmp_$tag; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $OLDPWD/tools/profile_compare.py > $OUT/stdout.txt 2> $OUT/stderr.txt )
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.2f us" % (float(r["AverageNs"]) / 1e3), "%5.1f %%" % (100 * float(r["TotalDurationNs"]) / tot))
P
