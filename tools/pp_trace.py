#!/usr/bin/env python
"""Cycle-stamp timeline of the ping-pong GEMM's phases (debug variants 58..60 of gemm_variant; see csrc/gemm.h KO bit 8)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth
from llmrankers._engine import RkEngine

dims = _synth.TOY_GATED_UNTIED
eng = RkEngine(dims, 0, max_tokens=256, max_seqs=4, max_dec_len=4).load_state(_synth.synth_state_dict(dims, 1).items())
n = int(os.environ.get("PP_N", "4096"))
rs = np.random.RandomState(0)
a = rs.standard_normal((n, n)).astype(np.float16)
w = rs.standard_normal((n, n)).astype(np.float16)
for variant in [int(v) for v in os.environ.get("PP_VARIANTS", "58,59,60").split(",")]:
    eng.set_option("gemm_variant", variant)
    eng.debug_gemm(a, w, use_glds=True)
    c = eng.debug_gemm(a, w, use_glds=True)
    t = c[0, :256].view(np.uint32).reshape(8, 32)[:, :24].astype(np.int64)
    t0 = t.min()
    print(f"variant {variant}: stamps relative to the earliest (cycles); per phase: top, after B1+frag wait, after last MFMA issue")
    for wv in range(8):
        row = (t[wv] - t0) & 0xFFFFFFFF
        print(f" wave {wv}: " + " | ".join(" ".join(f"{int(x):5d}" for x in row[q * 3:q * 3 + 3]) for q in range(8)))
    d = (t - t0) & 0xFFFFFFFF
    print(" mean load+B1 (1-0):", np.mean(d[:, 1::3] - d[:, 0::3]).round(1), " mean MFMA issue (2-1):", np.mean(d[:, 2::3] - d[:, 1::3]).round(1),
          " mean B2 (next0-2):", np.mean(d[:, 3::3] - d[:, 2:-1:3]).round(1), " pair span:", int(d[:, 21].mean() - d[:, 0].mean()))
