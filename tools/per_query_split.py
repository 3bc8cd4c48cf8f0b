#!/usr/bin/env python
"""One query of hits=100 candidates (32-token query / 128-token passage -> 184 tokens per prompt) through the engine's two batch
slots, cut into chunks in different ways: the decoder chain of a chunk overlaps the encoder of the next one, at the price of
smaller GEMMs.  Prints ms per query for every cut given as argument, e.g.  100  64,36  50,50  36,64  (passages per chunk)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def main():
    import numpy as np
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B or measurement build (tools only)
    import bench
    import __graft_entry__ as ge
    ge.build()
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    dims = _synth.FLAN_T5_LARGE
    L = int(os.environ.get("RK_L", "184"))
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=49152, max_seqs=256, max_dec_len=4).load_state(state.items())
    del state
    for kv in os.environ.get("RK_OPTS", "").split(","):
        if kv:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
    seqs = _synth.synth_token_batch(100, L, L, dims.vocab, seed=4242)
    ref = None
    for arg in sys.argv[1:]:
        cuts = [int(x) for x in arg.split(",")]
        assert sum(cuts) == 100
        chunks, pos = [], 0
        for c in cuts:
            chunks.append(seqs[pos:pos + c])
            pos += c
        ts = []
        for rep in range(8):
            t = time.perf_counter()
            parts, pending = [], []
            for k, ch in enumerate(chunks):
                slot = k % eng.num_slots
                if len(pending) == eng.num_slots:
                    parts.append(eng.read_scores(pending.pop(0)))
                eng.stage(ch, slot=slot)
                eng.score_staged([0], [bench.YES_ID, bench.NO_ID], slot=slot)
                pending.append(slot)
            for s0 in pending:
                parts.append(eng.read_scores(s0))
            sc = np.concatenate(parts)
            ts.append(time.perf_counter() - t)
        ref = sc if ref is None else ref
        ms = float(np.median(ts[3:])) * 1e3
        print(json.dumps({"cuts": cuts, "ms_per_query": round(ms, 3), "passages_per_s": round(1e5 / ms, 1),
                          "identical_to_first": bool((sc == ref).all())}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
