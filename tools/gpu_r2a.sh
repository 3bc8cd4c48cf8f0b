#!/bin/bash
# round-2 call A: sanity tests, group-size policy at the driver's --steps 20, stagger sweep, overlapped kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r2a_pytest.log; cat gpurun_out/r2a_pytest.log
timeout 900 python tools/sweep.py \
  "G=5,steps=20,warmup=5,rep=3" "G=10,steps=20,warmup=5,rep=3" "G=20,steps=20,warmup=5,rep=3" "G=4,steps=20,warmup=5,rep=3" \
  "G=8,steps=64,warmup=16,rep=2" \
  "G=8,steps=64,warmup=16,rep=2,gemm_stagger=600" "G=8,steps=64,warmup=16,rep=2,gemm_stagger=1200" \
  "G=8,steps=64,warmup=16,rep=2,gemm_stagger=1800" "G=8,steps=64,warmup=16,rep=2,gemm_stagger=2600" \
  "G=8,steps=64,warmup=16,rep=2,gemm_stagger=600,gemm_stagger_phases=4" "G=8,steps=64,warmup=16,rep=2,gemm_stagger=1000,gemm_stagger_phases=4" \
  "G=8,steps=64,warmup=16,rep=2,gemm_stagger=300,gemm_stagger_phases=8" \
  "G=10,steps=20,warmup=5,rep=3,gemm_stagger=1200" \
  > gpurun_out/r2a_sweep.jsonl 2> gpurun_out/r2a_sweep.err
cat gpurun_out/r2a_sweep.jsonl; tail -3 gpurun_out/r2a_sweep.err
# serial per-kernel durations with and without stagger (one stream)
for st in 0 1200; do
  timeout 300 python bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_per_query --opt gemm_stagger=$st > gpurun_out/r2a_bench_st$st.json 2> gpurun_out/r2a_bench_st$st.err
  python - $st <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r2a_bench_st{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("stagger", sys.argv[1], j["value"], {k: (v["ms_per_step"], v["tflops"]) for k, v in j["roofline"]["per_class"].items()})
PY
done
# kernel trace WITH overlap (two encoder streams + decoder streams): who runs when
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a_trace -o ov -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_profile --no_per_query > $GRAFT_REPO_ROOT/gpurun_out/r2a_trace_stdout.txt 2>&1
echo "trace rc=$?"; ls -la $GRAFT_REPO_ROOT/gpurun_out/r2a_trace/* | head
