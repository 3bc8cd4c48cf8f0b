#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" 2> $O/build.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rerank.py -q -m gpu -k "long_sequence_attention or setwise_shape or config3" 2>&1 | tail -3
for rep in 1 2 3; do for n in prev new; do for B in 8 1; do RK_ENGINE_LIB=exp/librk_$n.so RK_B=$B RK_L=1560 timeout 300 python tools/profile_compare.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n B', j['B'], 'likelihood_ms', j['likelihood_ms'], 'enc_attn', j['likelihood_classes_ms']['enc_attn'])"; done; done; done | tee $O/attl_time2.txt
