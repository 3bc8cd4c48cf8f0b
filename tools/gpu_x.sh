#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rerank.py -x -q -m gpu -k "greedy or setwise or fused_decoder or toy_logits or config3 or monot5 or pipelined or rerank_many or lockstep or pairwise or flan_t5_large_dims" 2>&1 | tail -n 6
RK_L=1450 RK_B=1 timeout 300 python tools/profile_compare.py 2>/dev/null | tee gpurun_out/x/compare_profile_few.json
O=$PWD/gpurun_out/x/trace2; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/compare_trace.py run > $O/stdout.txt 2>&1
cd $R && python tools/compare_trace.py summarize $O | tee gpurun_out/x/compare_trace2.txt
find $O -name "*.csv" -size +4M -delete
