#!/bin/bash
# rocprofv3 evidence for the bench command: (1) kernel trace + stats, (2) PMC passes (each counter set in its own
# run, with --kernel-trace only — never combined with sys/hip/hsa tracing).  Summaries land in gpurun_out/prof/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
STEPS=${PROF_STEPS:-20}; WARM=${PROF_WARMUP:-10}   # a whole group of 10: every launch of the run then has the same M
CMD="python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup $WARM --no_cpu_baseline --no_profile --no_per_query --no_extras --regions 1 --overlap 0"   # one stream: kernels run one at a time, so durations and counters belong to one kernel
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- $CMD > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
echo "rocprofv3 stats rc=$?"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc_$tag -- $CMD > $OUT/pmc_${tag}_stdout.txt 2>&1
  echo "pmc $tag rc=$?"
done
PROF_STEPS=$STEPS PROF_WARMUP=$WARM python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof"
summ = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/pmc_*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            summ[k][c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
stats = {}
f = out + "/bench_kernel_stats.csv"
if os.path.exists(f):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
import re, subprocess, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
steps = int(os.environ.get("PROF_STEPS", "20"))
try:
    box = open(out + "/../final/box.txt").read()
except Exception:
    box = os.uname().nodename
import __graft_entry__ as ge
res = {"box": box, "source_hash": ge._source_hash(), "command": f"bench.py --steps {steps} --warmup {os.environ.get('PROF_WARMUP', '5')} --overlap 0 (one stream: kernels run one at a time)",
       "tokens_per_launch": bench.auto_group(steps) * 32 * 184,
       "note": "FETCH_SIZE/WRITE_SIZE in KiB as reported; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md HBM section)",
       "kernels": {k: {"stats": stats.get(k), "pmc": v} for k, v in summ.items()}}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["pct"])[:14]:
    p = summ.get(k, {})
    fs = p.get("FETCH_SIZE", {}).get("avg_per_launch"); ws = p.get("WRITE_SIZE", {}).get("avg_per_launch")
    mb = p.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg_per_launch"); ga = p.get("GRBM_GUI_ACTIVE", {}).get("avg_per_launch")
    util = (mb / (ga / 8 * 1024)) if mb and ga else None
    print(f"{v['pct']:6.2f}%  {v['avg_ns']/1e3:8.1f}us x{v['calls']:5d}  fetchKiB={fs and round(fs)} writeKiB={ws and round(ws)} mfma_util={util and round(util,3)}  {k[:70]}")
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -1 $OUT/bench_stdout.txt | cut -c1-300
