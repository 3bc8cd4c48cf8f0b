#!/bin/bash
# small-M baseline: per-variant GEMM times at the one-compare / one-rank-share / Llama-compare shapes, and the per-class profiles
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/smallm; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" 2> $O/build.log
for M in 1450 2392; do
  RK_BENCH_M=$M RK_GEMM_VARIANTS=${RK_GEMM_VARIANTS:-0,1,4,5,6} timeout 300 python tools/gemm_bench.py 50 qkv,o,ffn_in_geglu,ffn_out > $O/gemm_M$M.log 2>&1
done
RK_BENCH_M=1536 RK_BENCH_SHAPES=1536x6144x4096x0,1536x4096x4096x1,1536x28672x4096x5,1536x4096x14336x1 RK_GEMM_VARIANTS=${RK_GEMM_VARIANTS:-0,1,4,5,6} timeout 300 python tools/gemm_bench.py 20 none > $O/gemm_llama.log 2>&1
RK_L=1450 RK_B=1 timeout 300 python tools/profile_compare.py > $O/compare_profile.json 2> $O/compare_profile.err
RK_L=184 RK_B=13 timeout 300 python tools/profile_compare.py > $O/share_profile.json 2> $O/share_profile.err
tail -n 30 $O/*.log $O/*.json
