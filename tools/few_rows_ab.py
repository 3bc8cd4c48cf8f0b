#!/usr/bin/env python
"""Cross-over of the few-row GEMV family (gemv_rows.h) against the weight-streaming MFMA kernels, per decoder row count:
wall time of the setwise call shapes on flan-t5-large dims with option dec_gemv_rows = 1 (family off for >= 2 rows) .. 16.
  likelihood: B prompts x 2 decoder positions (2B rows);  greedy: two greedy steps of ONE prompt (2, then 3 rows);
  tree: rk_t5_greedy2 with ten candidate first tokens (13 rows).  Alternating settings, median of 5 x 10 calls."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
from llmrankers import _synth
from llmrankers._engine import RkEngine

L = int(os.environ.get("RK_L", "1450"))
dims = _synth.FLAN_T5_LARGE
eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(_synth.synth_tensors(dims, seed=929, threads=32))
prefix, labels = [0, 5454], list(range(71, 81))
cases = {}
for B in (1, 2, 3, 4, 6, 8):
    seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=7)
    cases[f"likelihood_B{B}_rows{2 * B}"] = (2 * B, lambda s=seqs: eng.score(s, prefix, labels))
one = _synth.synth_token_batch(1, L, L, dims.vocab, seed=7)
cases["greedy2steps_rows2_3"] = (2, lambda: eng.greedy(one, prefix, 2))
cases["tree_rows13"] = (13, lambda: eng.greedy(one, prefix, 2, candidates=labels))


def timed(fn, n=10):
    eng.sync(); t = time.perf_counter()
    for _ in range(n):
        fn()
    eng.sync()
    return (time.perf_counter() - t) / n * 1e3


out = {}
for name, (rows, fn) in cases.items():
    res = {0: [], 1: []}
    for rep in range(6):
        for on in (0, 1):
            eng.set_option("dec_gemv_rows", 16 if on else 1)
            if rep == 0:
                fn(); fn(); continue
            res[on].append(timed(fn))
    out[name] = {"rows": rows, "mfma_ms": round(float(np.median(res[0])), 3), "gemv_ms": round(float(np.median(res[1])), 3)}
    print(name, out[name], flush=True)
eng.set_option("dec_gemv_rows", 16)
print("JSON " + json.dumps(out))
