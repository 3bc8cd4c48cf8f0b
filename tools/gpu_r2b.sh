#!/bin/bash
# round-2 call B: folded RMSNorm + residual-epilogue prefetch A/B, s_memrealtime tick check, full bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r2b_pytest.log; cat gpurun_out/r2b_pytest.log
timeout 900 python tools/sweep.py \
  "G=8,steps=64,warmup=16,rep=2,fold_norm=0,gemm_epi_depth=0" "G=8,steps=64,warmup=16,rep=2,fold_norm=0,gemm_epi_depth=2" \
  "G=8,steps=64,warmup=16,rep=2,fold_norm=0,gemm_epi_depth=3" "G=8,steps=64,warmup=16,rep=2,fold_norm=1,gemm_epi_depth=0" \
  "G=8,steps=64,warmup=16,rep=2,fold_norm=1,gemm_epi_depth=2" "G=8,steps=64,warmup=16,rep=2,fold_norm=1,gemm_epi_depth=3" \
  "G=10,steps=20,warmup=5,rep=3,fold_norm=1,gemm_epi_depth=2" "G=10,steps=20,warmup=5,rep=3,fold_norm=1,gemm_epi_depth=3" \
  "G=8,steps=16,warmup=8,rep=1,gemm_stagger=100000" "G=8,steps=16,warmup=8,rep=1,gemm_stagger=0" \
  "G=8,steps=64,warmup=16,rep=2,gemm_stagger=1500" "G=8,steps=64,warmup=16,rep=2,gemm_stagger=800,gemm_stagger_phases=4" \
  > gpurun_out/r2b_sweep.jsonl 2> gpurun_out/r2b_sweep.err
cat gpurun_out/r2b_sweep.jsonl; tail -3 gpurun_out/r2b_sweep.err
for cfg in "fold_norm=0 gemm_epi_depth=0" "fold_norm=1 gemm_epi_depth=2" "fold_norm=1 gemm_epi_depth=3"; do
  args=""; for kv in $cfg; do args="$args --opt $kv"; done
  tag=$(echo $cfg | tr ' =' '__')
  timeout 300 python bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_per_query $args > gpurun_out/r2b_bench_$tag.json 2> gpurun_out/r2b_bench_$tag.err
  python - $tag <<'PY'
import json, sys
try:
    j = json.loads(open(f"gpurun_out/r2b_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], j["value"], {k: (v["ms_per_step"], v["tflops"]) for k, v in j["roofline"]["per_class"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/r2b_bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
# the driver's command, complete line (cpu baseline, per-query numbers)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2b_bench_driver.json 2> gpurun_out/r2b_bench_driver.err; echo "driver-style bench rc=$?"
tail -c 6000 gpurun_out/r2b_bench_driver.json; tail -5 gpurun_out/r2b_bench_driver.err
