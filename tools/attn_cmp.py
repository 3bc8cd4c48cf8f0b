#!/usr/bin/env python
"""Compare the encoder attention kernel variants (attn_short = 0 tiled, 1 pipelined, 2 six-wave, 3 pair, 4 four-wave) on ragged input."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth
from llmrankers._engine import RkEngine

for dims, lens in ((_synth.TOY_GATED_UNTIED, [109, 5, 64, 192, 130, 1]), (_synth.FLAN_T5_SMALL, [184, 20, 77, 192, 65])):
    state = _synth.synth_state_dict(dims, 3)
    eng = RkEngine(dims, 0, max_tokens=2048, max_seqs=16, max_dec_len=4).load_state(state.items())
    rs = np.random.RandomState(1)
    seqs = [rs.randint(2, dims.vocab, size=n).tolist() for n in lens]
    T, I = sum(lens), dims.n_heads * dims.d_kv
    ref = None
    for mode in (0, 2, 4, 1):
        eng.set_option("attn_short", mode)
        eng.score(seqs, [0], [3, 4])
        ctx = eng.debug_read("ctx", T * I).reshape(T, I)
        if ref is None:
            ref = ctx
            continue
        d = np.abs(ctx - ref).reshape(T, dims.n_heads, 64).max(axis=2)     # [T, H]
        off = np.cumsum([0] + lens)
        print(f"{dims.n_heads} heads, mode {mode}: max diff {d.max():.3g}; per (seq, head) max:",
              [[float(f"{d[off[b]:off[b+1], h].max():.2g}") for h in range(dims.n_heads)] for b in range(len(lens))])
    eng.close()
