#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python tools/gemm_bench.py 20 > gpurun_out/gemm_bench.txt 2>&1; echo "gemm_bench rc=$?"; cat gpurun_out/gemm_bench.txt | grep -v amdgpu.ids
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -ciE "SQ_|TCC_|GRBM" $OUT/counters_list.txt
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc_$tag -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py 3 qkv,o,ffn_in_geglu,ffn_out > $OUT/run_$tag.txt 2>&1
  echo "pmc $tag rc=$?"
done
ls $OUT
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc"
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "gemm_f16" not in k: continue
        key = (k[:40], r.get("Grid_Size"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(f))
    for key, cs in agg.items():
        print(key, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
