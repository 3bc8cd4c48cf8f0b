#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "decoder_cross_attention or decoder_attention_per_sequence or qlm or full_batch_vs_hf" 2>&1 | grep -v "^$" | tail -n 14
timeout 600 python tools/bench_qlm_xl.py 2>/dev/null | tail -2 | tee $O/qlm_xl.json
