#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (no CPU baseline, no in-process profiling pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_profile > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
echo "rocprofv3 rc=$?"
ls -R $OUT | head -30
find $OUT -name "*kernel_stats*" | head -3
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
cat $OUT/bench_stdout.txt | tail -2
# drop the big per-dispatch trace, keep stats
find $OUT -name "*kernel_trace.csv" -size +20M -delete
