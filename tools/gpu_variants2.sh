#!/bin/bash
# A/B prebuilt engine libraries through bench.py only (variants/librk_<name>.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cp llm-rankers_amd/lib/librk_engine.so /tmp/librk_orig.so
for n in ${VARIANTS}; do
  cp variants/librk_$n.so llm-rankers_amd/lib/librk_engine.so; touch llm-rankers_amd/lib/librk_engine.so
  timeout 300 python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); pc=j['roofline']['per_class']; print('$n', j['value'], j['ms_per_step'], {k:pc[k]['ms_per_step'] for k in ('enc_gemm_qkv','enc_gemm_ffn_in','enc_gemm_o')})"
done
cp /tmp/librk_orig.so llm-rankers_amd/lib/librk_engine.so
