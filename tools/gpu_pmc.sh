#!/bin/bash
# rocprofv3 PMC passes (one counter set per run, kernel-trace only) over a short bench run; prints per-kernel averages.
# usage: PMC_SETS="A B C;D E" KERNELS="attn_enc_short|gemm_pp2" bash tools/gpu_pmc.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
CMD="python $PWD/bench.py --steps ${PMC_STEPS:-16} --warmup 8 --no_cpu_baseline --no_profile --no_per_query --no_extras --regions 1"
IFS=';' read -ra SETS <<< "${PMC_SETS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE}"
cd /tmp
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass${i}_stdout.txt 2>&1
  echo "pass $i ($set) rc=$?"
done
KERNELS="${KERNELS:-attn_enc|gemm_pp2|xattn|gemm_skinny}" python - <<'PY'
import csv, glob, os, re, collections, json
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc"
pat = re.compile(os.environ["KERNELS"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["launches"] = max(len(v) for v in cs.values())
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} {sum(v)/len(v):16.1f}")
json.dump(res, open(out + "/pmc_kernels.json", "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
