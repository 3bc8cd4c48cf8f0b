#!/usr/bin/env python
"""Resident workgroups per CU as computed by the HIP runtime for the engine's main kernels."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth
from llmrankers._engine import RkEngine
dims = _synth.TOY_GATED_UNTIED
eng = RkEngine(dims, 0, max_tokens=256, max_seqs=4, max_dec_len=4).load_state(_synth.synth_state_dict(dims, 1).items())
v = eng.debug_read("occupancy", 6)
print(dict(zip(["attn_enc_short<4>(256thr,54KB)", "attn_enc_pair(768thr,106KB)", "gemm_f16 128x128(256thr,64KB)", "gemm_pp2(512thr,128KB)", "attn_enc tiled(256thr)", "rmsnorm<4>(256thr)"], [int(x) for x in v])))
