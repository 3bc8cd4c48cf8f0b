#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/sweep.py "G=8,steps=64,warmup=16,rep=2" "G=8,steps=64,warmup=16,rep=2,attn_heads_per_wg=8" "G=8,steps=64,warmup=16,rep=2,attn_heads_per_wg=4" "G=8,steps=64,warmup=16,rep=2,attn_heads_per_wg=2" "G=8,steps=64,warmup=16,rep=2,attn_heads_per_wg=1" "G=10,steps=20,warmup=5,rep=3,attn_heads_per_wg=8" > gpurun_out/r2h_sweep.jsonl 2> gpurun_out/r2h_sweep.err
cat gpurun_out/r2h_sweep.jsonl; tail -3 gpurun_out/r2h_sweep.err
python - <<'PY'
import sys, os
sys.path[:0] = ["llm-rankers_amd", "."]
import torch
from llmrankers import _synth
from llmrankers._engine import RkEngine
eng = RkEngine(_synth.TOY_GATED_UNTIED, 0, max_tokens=256, max_seqs=4, max_dec_len=4).load_state(_synth.synth_state_dict(_synth.TOY_GATED_UNTIED, 1).items())
print("occupancy [attn_short4, attn_pair, gemm_f16, pp2, attn_enc, rmsnorm]:", eng.debug_read("occupancy", 6))
PY
