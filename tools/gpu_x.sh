#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
: > $O/epi_ko_smallm.txt
for M in 2048 8192 16384; do
for n in base ko1 ko6 ko7 d3; do
  echo "== $n M=$M" >> $O/epi_ko_smallm.txt
  RK_ENGINE_LIB=exp/librk_$n.so RK_BENCH_M=$M RK_BENCH_FOLD=1 RK_GEMM_VARIANTS=5 timeout 300 python tools/gemm_bench.py 50 o,ffn_out 2>&1 | grep -v "JSON\|amdgpu\|tools\]" >> $O/epi_ko_smallm.txt
done
done
cat $O/epi_ko_smallm.txt
