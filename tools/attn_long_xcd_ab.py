#!/usr/bin/env python
"""A/B of the long-sequence attention kernel's workgroup -> XCD mapping (engine option attn_long_xcd) in one process:
flan-t5-large dims, 1 / 2 / 4 / 8 prompts of 1560 tokens, alternating the two mappings; same bits required."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
from llmrankers import _synth
from llmrankers._engine import RkEngine

dims = _synth.FLAN_T5_LARGE
eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(_synth.synth_tensors(dims, seed=929, threads=32))
prefix, labels = [0, 5454], list(range(71, 82))
for B in (1, 2, 4, 8):
    seqs = _synth.synth_token_batch(B, 1560, 1560, dims.vocab, seed=7)
    sc = {}
    for flag in (0, 1, 0, 1):
        eng.set_option("attn_long_xcd", flag)
        for _ in range(3):
            sc[flag] = eng.score(seqs, prefix, labels)
        eng.profile(True); eng.profile_reset()
        for _ in range(5):
            eng.score(seqs, prefix, labels)
        eng.sync()
        rep = eng.profile_report(); eng.profile(False)
        print(json.dumps({"B": B, "attn_long_xcd": flag, "enc_attn_us_per_layer": round(rep["enc_attn"]["ms"] / 5 / 24 * 1e3, 1)}), flush=True)
    print(json.dumps({"B": B, "same_bits": bool(np.array_equal(sc[0], sc[1]))}), flush=True)
eng.close()
