#!/usr/bin/env python
"""Diagnostic: do all GEMM tile variants produce the same bits with the folded RMSNorm on / off?  1-layer model at
flan-t5-large widths; compares the qkv / hidden / enc_out buffers and the scores of every forced variant with variant 5."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
from llmrankers import _synth
from llmrankers._engine import RkEngine

dims = _synth.T5Dims(vocab=512, d_model=1024, n_heads=16, d_kv=64, d_ff=2816, n_enc=int(os.environ.get("NL", "1")), n_dec=1)
state = _synth.synth_state_dict(dims, seed=5, threads=8)
eng = RkEngine(dims, 0, max_tokens=8192, max_seqs=32, max_dec_len=4).load_state(state.items())
batch = _synth.synth_token_batch(32, 184, 184, dims.vocab, seed=9)
T = 32 * 184
for fold in (0, 1):
    eng.set_option("fold_norm", fold)
    ref = None
    for v in (5, 0, 1, 2, 3, 4):
        eng.set_option("gemm_variant", v)
        sc = eng.score(batch, [0], [10, 20])
        bufs = {n: eng.debug_read(n, T * w) for n, w in (("qkv", 3072), ("ctx", 1024), ("enc_hidden", 1024), ("enc_out", 1024))}
        if ref is None:
            ref = (sc, bufs)
            print(f"fold={fold} variant 5: scores[0]={sc[0]}")
            continue
        line = [f"fold={fold} v{v}: score maxdiff {np.abs(sc - ref[0]).max():.3e}"]
        for n in bufs:
            d = np.abs(bufs[n] - ref[1][n])
            bad = np.argwhere(d.reshape(T, -1) > 0)
            line.append(f"{n} {d.max():.3e} nbad {len(bad)}" + (f" first {bad[0].tolist()} rows {np.unique(bad[:,0])[:6].tolist()} cols {np.unique(bad[:,1])[:8].tolist()}" if len(bad) else ""))
        print(" | ".join(line), flush=True)
eng.set_option("gemm_variant", 0)
