#!/bin/bash
# A/B prebuilt engine libraries (variants/librk_<name>.so): per-shape GEMM throughput at the G=8 shapes + one bench line each
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cp llm-rankers_amd/lib/librk_engine.so /tmp/librk_orig.so
for n in ${VARIANTS}; do
  cp variants/librk_$n.so llm-rankers_amd/lib/librk_engine.so; touch llm-rankers_amd/lib/librk_engine.so
  echo "== $n"
  RK_BENCH_M=47104 RK_GEMM_VARIANTS=5 timeout 300 python tools/gemm_bench.py 20 qkv,o,ffn_in_geglu,ffn_out 2>&1 | grep -v amdgpu.ids | grep -v JSON
  timeout 300 python bench.py --no_cpu_baseline --no_profile 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'])"
done
cp /tmp/librk_orig.so llm-rankers_amd/lib/librk_engine.so
