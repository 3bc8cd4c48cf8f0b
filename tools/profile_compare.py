#!/usr/bin/env python
"""Where one setwise compare spends its time (configs[2] call shape: ONE ~1.45k-token prompt, flan-t5-large dims):
wall time of `likelihood` / `generation` calls and the engine's per-class event times (eager, serial)."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B build (tools only)
from llmrankers import _synth
from llmrankers._engine import RkEngine

L = int(os.environ.get("RK_L", "1450"))
B = int(os.environ.get("RK_B", "1"))
dims = _synth.FLAN_T5_LARGE
eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(_synth.synth_tensors(dims, seed=929, threads=32))
for kv in os.environ.get("RK_OPTS", "").split(","):
    if kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=7)
prefix, labels = [0, 5454], list(range(71, 82))
out = {"L": L, "B": B}
for name, fn in (("likelihood", lambda: eng.score(seqs, prefix, labels)), ("generation", lambda: eng.greedy(seqs, prefix, 2))):
    for _ in range(3):
        fn()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    out[name + "_ms"] = round((time.perf_counter() - t) / 10 * 1e3, 3)
    eng.profile(True); eng.profile_reset()
    for _ in range(5):
        fn()
    eng.sync()
    rep = eng.profile_report(); eng.profile(False)
    out[name + "_classes_ms"] = {k: [round(v["ms"] / 5, 3), int(v["launches"] / 5)] for k, v in rep.items() if v["launches"]}
print(json.dumps(out))
