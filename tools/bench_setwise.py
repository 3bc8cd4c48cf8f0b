#!/usr/bin/env python
"""Setwise call-shape timing (BASELINE.json configs[2]): flan-t5-large dims, ONE prompt of (num_child+1) passages,
`likelihood` (label logits at decoder position 1) and `generation` (2 greedy steps).  29 compares per query at
hits=100, num_child=10, k=10."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth                      # noqa: E402
from llmrankers._engine import RkEngine            # noqa: E402


def main():
    dims = _synth.FLAN_T5_LARGE
    eng = RkEngine(dims, 0, max_tokens=16384, max_seqs=8, max_dec_len=8)
    eng.load_state(_synth.synth_tensors(dims, seed=929, threads=min(32, os.cpu_count() or 8)))
    L = 11 * 134 + 32 + 30
    seq = _synth.synth_token_batch(1, L, L, dims.vocab, seed=1)
    prefix, labels = [0, 5454], list(range(71, 82))
    out = {"L_e": L}
    for name, fn in (("likelihood", lambda: eng.score(seq, prefix, labels)), ("generation", lambda: eng.greedy(seq, prefix, 2))):
        for _ in range(3):
            fn()
        t = time.perf_counter()
        n = 20
        for _ in range(n):
            fn()
        ms = (time.perf_counter() - t) / n * 1e3
        out[name] = {"ms_per_compare": round(ms, 3), "compares_per_s": round(1e3 / ms, 1), "ms_per_query_29_compares": round(29 * ms, 1),
                     "tflops": round(1.35e12 / (ms * 1e-3) / 1e12, 1)}
    # where a single likelihood compare spends its time (serial timeline, HIP events per launch)
    eng.set_option("overlap", 0)
    eng.profile(True)
    eng.profile_reset()
    n = 5
    for _ in range(n):
        eng.score(seq, prefix, labels)
    rep = eng.profile_report()
    eng.profile(False)
    eng.set_option("overlap", 1)
    out["likelihood_ms_by_class"] = {k: round(v["ms"] / n, 3) for k, v in rep.items() if v["launches"]}
    out["likelihood_launches_per_compare"] = int(sum(v["launches"] for v in rep.values()) / n)
    # independent compares batched into one engine call (what a level-wise heapify could submit)
    for nb in (2, 4, 8):
        seqs = _synth.synth_token_batch(nb, L, L, dims.vocab, seed=2)
        for _ in range(2):
            eng.score(seqs, prefix, labels)
        t = time.perf_counter()
        n = 10
        for _ in range(n):
            eng.score(seqs, prefix, labels)
        ms = (time.perf_counter() - t) / n * 1e3
        out[f"likelihood_batched_{nb}"] = {"ms_per_call": round(ms, 3), "ms_per_compare": round(ms / nb, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
