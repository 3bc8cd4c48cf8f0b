#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/sk1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" 2> $O/build.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stream_k or gemm_vs_numpy" > $O/pytest.log 2>&1
tail -n 15 $O/pytest.log
for M in 1450 2392; do
  RK_BENCH_M=$M RK_GEMM_VARIANTS=0,7 timeout 300 python tools/gemm_bench.py 50 qkv,o,ffn_in_geglu,ffn_out 2>&1 | grep -v "JSON\|amdgpu" > $O/gemm_M$M.log
done
RK_BENCH_M=1536 RK_BENCH_SHAPES=1536x6144x4096x0,1536x4096x4096x1,1536x28672x4096x5,1536x4096x14336x1 RK_GEMM_VARIANTS=0,7 timeout 300 python tools/gemm_bench.py 20 none 2>&1 | grep -v "JSON\|amdgpu" > $O/gemm_llama.log
cat $O/gemm_*.log
