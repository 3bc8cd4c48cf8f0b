#!/usr/bin/env python
"""Chained GEMM launches (csrc/gemm_chain.h) on the GPU: (1) scores with the chained form == scores with the separate launches,
bit for bit, on the bench shape, one query, a ragged batch and - with the size gate lowered - small and odd shapes (fewer row
panels than queues, a partial last panel); repeated and pipelined calls (epochs, both slots); (2) time per launch sequence and
per-class event times of both forms on the bench shape.  One JSON object per line."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def main():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
    import bench
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    dims = _synth.NAMED_DIMS[os.environ.get("RK_MODEL", "flan-t5-large")]
    B, L, G = 32, 184, 10
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=G * B * L, max_seqs=G * B, max_dec_len=4)
    eng.load_state(state.items())
    out_ids = [bench.YES_ID, bench.NO_ID]

    def score(seqs, chain, min_panels=None, lead=None):
        eng.set_option("chain", chain)
        if min_panels is not None:
            eng.set_option("chain_min_panels", min_panels)
        if lead is not None:
            eng.set_option("chain_lead", lead)
        return eng.score(seqs, [0], out_ids)

    shapes = [("bench 320x184", _synth.synth_token_batch(320, L, L, dims.vocab, seed=1), 64),
              ("one query 100x184", _synth.synth_token_batch(100, L, L, dims.vocab, seed=2), 64),
              ("ragged 320 x U{96..184}", _synth.synth_token_batch(320, 96, 184, dims.vocab, seed=3), 64),
              ("40x184 (29 panels)", _synth.synth_token_batch(40, L, L, dims.vocab, seed=4), 1),
              ("7 x U{100..180} (~4 panels, partial last)", _synth.synth_token_batch(7, 100, 180, dims.vocab, seed=5), 1),
              ("1 x 150 (one partial panel)", _synth.synth_token_batch(1, 150, 150, dims.vocab, seed=6), 1),
              ("13x184 (10 panels)", _synth.synth_token_batch(13, L, L, dims.vocab, seed=7), 1),
              ("64x184 (46 panels)", _synth.synth_token_batch(64, L, L, dims.vocab, seed=8), 1)]
    ok_all = True
    for name, seqs, mp in shapes:
        try:
            ref = score(seqs, 0)
            res = {"shape": name, "tokens": int(sum(len(s) for s in seqs))}
            for lead in (3, 1, 2):
                got = score(seqs, 1, mp, lead)
                same = bool(np.array_equal(ref, got))
                res[f"lead{lead}_bit_identical"] = same
                if not same:
                    d = np.abs(ref - got)
                    res[f"lead{lead}_max_abs_diff"] = float(d.max()); res[f"lead{lead}_rows_differing"] = int((d.max(axis=1) > 0).sum())
                ok_all &= same
            again = score(seqs, 1, mp, 3)
            res["repeat_bit_identical"] = bool(np.array_equal(ref, again))
            ok_all &= res["repeat_bit_identical"]
        except Exception as exc:
            res = {"shape": name, "error": repr(exc)[:300]}
            ok_all = False
        print(json.dumps(res), flush=True)
    # pipelined: both slots in flight, decoder graphs beside the next encoder (the bench's timed region), chained form
    eng.set_option("chain_min_panels", 64); eng.set_option("chain_lead", 3)
    slot_seqs = [_synth.synth_token_batch(320, L, L, dims.vocab, seed=20 + s) for s in range(eng.num_slots)]
    try:
        eng.set_option("chain", 0)
        refs = [eng.score(q, [0], out_ids) for q in slot_seqs]
        eng.set_option("chain", 1)
        for s, q in enumerate(slot_seqs):
            eng.stage(q, slot=s)
        same = True
        for rep in range(6):
            for s in range(eng.num_slots):
                eng.score_staged([0], out_ids, slot=s)
            for s in range(eng.num_slots):
                same &= bool(np.array_equal(eng.read_scores(s), refs[s]))
        print(json.dumps({"pipelined_two_slots_x6_bit_identical": same}), flush=True)
        ok_all &= same
    except Exception as exc:
        print(json.dumps({"pipelined": "error", "msg": repr(exc)[:300]}), flush=True)
        ok_all = False
    # timing
    seqs = slot_seqs[0]
    for chain in (0, 1, 0, 1):
        eng.set_option("chain", chain)
        eng.stage(seqs, slot=0)
        for _ in range(3):
            eng.score_staged([0], out_ids, slot=0)
        eng.sync()
        t = time.perf_counter()
        n = 10
        for i in range(n):
            eng.score_staged([0], out_ids, slot=i % eng.num_slots) if False else eng.score_staged([0], out_ids, slot=0)
        eng.sync()
        ms = (time.perf_counter() - t) / n * 1e3
        eng.profile(True); eng.profile_reset(); eng.set_option("overlap", 0)
        for _ in range(3):
            eng.score_staged([0], out_ids, slot=0)
        eng.sync()
        rep = eng.profile_report(); eng.profile(False); eng.set_option("overlap", 1)
        print(json.dumps({"chain": chain, "ms_per_launch_sequence_320x184": round(ms, 3), "passages_per_s_one_slot": round(320 / ms * 1e3, 1),
                          "classes_us_per_launch": {k: round(v["ms"] * 1e3 / v["launches"], 1) for k, v in rep.items() if v["launches"]},
                          "classes_ms_per_sequence": {k: round(v["ms"] / 3, 3) for k, v in rep.items() if v["launches"]}}), flush=True)
    print(json.dumps({"all_bit_identical": ok_all}))
    eng.close()


if __name__ == "__main__":
    main()
