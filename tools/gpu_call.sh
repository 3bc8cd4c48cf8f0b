cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c4
L=gpurun_out/c4/debug.log; : > $L
run() { echo "== $*" >> $L; RK_ENGINE_LIB=exp/librk_engine_measure.so timeout 120 python tools/chain_debug.py "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
run flan-t5-small 40 0 0
run flan-t5-small 7 0 0 1
grep -v "amdgpu.ids\|^\[tools\]" $L | cut -c1-300
timeout 600 python tools/chain_check.py > gpurun_out/c4/chain_check.jsonl 2> gpurun_out/c4/chain_check.err; echo "chain_check rc=$?"
cut -c1-1200 gpurun_out/c4/chain_check.jsonl
tail -3 gpurun_out/c4/chain_check.err
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "per_sequence or qlm or two_token or chained" 2>&1 | tail -5
timeout 300 python tools/bench_qlm_xl.py 2>/dev/null | tail -1 > gpurun_out/c4/qlm_xl.json; cut -c1-1500 gpurun_out/c4/qlm_xl.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_per_query --no_extras > gpurun_out/c4/bench_chain1.json 2> gpurun_out/c4/bench_chain1.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_per_query --no_extras --opt chain=0 > gpurun_out/c4/bench_chain0.json 2> gpurun_out/c4/bench_chain0.err; echo "bench rc=$?"
python - <<'PY'
import json
for n in ("chain1", "chain0"):
    try:
        j = json.loads(open(f"gpurun_out/c4/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step"], j["config"]["whole_path_frac_of_mfma_peak"], j["config"]["timed_regions_ms"], {k: v["ms_per_step"] for k, v in j["roofline"]["per_class"].items()})
    except Exception as exc:
        print(n, "failed", exc); print(open(f"gpurun_out/c4/bench_{n}.err").read()[-800:])
PY
