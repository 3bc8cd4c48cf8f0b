#!/usr/bin/env python
"""GPU check of the DMA-staged encoder attention kernel (attn_short = 5 / 6) in ONE process:
  1. bit-exactness of the context rows against the tiled kernel (attn_short = 0, which walks the key tiles of a short
     sequence twice to reproduce the whole-row softmax) on ragged batches (toy dims with 3 / 4 heads, flan-t5-small with 6
     heads, lengths 1 .. 192, several heads_per_wg);
  2. per-launch time of the attention kernel class (HIP events around every launch, serial stream) at the bench shape
     (flan-t5-large dims, 320 x 184 tokens) for the tiled kernel and for the DMA kernel at several heads_per_wg;
  3. passages/s of the bench pipeline for the best variants.
Prints one JSON line per measurement."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def ctx_of(eng, seqs, T, I):
    eng.score(seqs, [0], [3, 4])
    return eng.debug_read("ctx", T * I).reshape(T, I).copy()


def exactness():
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    ok = True
    cases = ((_synth.TOY_GATED_UNTIED, [109, 5, 64, 192, 130, 1, 65, 128, 184, 191, 2, 33]),
             (_synth.FLAN_T5_SMALL, [184, 20, 77, 192, 65, 129, 96, 1, 184, 184]))
    for dims, lens in cases:
        state = _synth.synth_state_dict(dims, 3, gain=2.0)
        eng = RkEngine(dims, 0, max_tokens=4096, max_seqs=32, max_dec_len=4).load_state(state.items())
        rs = np.random.RandomState(1)
        seqs = [rs.randint(2, dims.vocab, size=n).tolist() for n in lens]
        T, I = sum(lens), dims.n_heads * dims.d_kv
        eng.set_option("attn_short", 0)
        ref = ctx_of(eng, seqs, T, I)
        off = np.cumsum([0] + lens)
        for mode, hpw in [(m, h) for m in (5, 6) for h in (1, 2, 3, 16)]:
            eng.set_option("attn_short", mode)
            eng.set_option("attn_heads_per_wg", hpw)
            got = ctx_of(eng, seqs, T, I)
            d = np.abs(got - ref).reshape(T, dims.n_heads, 64).max(axis=2)
            same = bool(np.array_equal(got, ref))
            ok &= same
            rec = {"check": "dma_vs_tiled", "mode": mode, "heads": dims.n_heads, "heads_per_wg": hpw, "bit_identical": same,
                   "max_abs_diff": float(d.max())}
            if not same:
                rec["per_seq_head_max"] = [[float(f"{d[off[b]:off[b + 1], h].max():.3g}") for h in range(dims.n_heads)] for b in range(len(lens))]
                bad = np.argwhere(np.abs(got - ref) > 0)
                rec["first_bad"] = bad[:6].tolist()
                rec["nan"] = int(np.isnan(got).sum())
            print(json.dumps(rec), flush=True)
        eng.set_option("attn_heads_per_wg", 0)
        eng.close()
    return ok


# extra timing variants with the timing-only knock-outs of a measurement build (hipcc ... -DRK_MEASURE, RK_ENGINE_LIB=<that .so>)
KO = [int(a[5:]) for a in sys.argv[1:] if a.startswith("--ko=")]


def timing():
    import bench
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    dims = _synth.FLAN_T5_LARGE
    B, L, G = 32, 184, 10
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=G * B * L, max_seqs=G * B, max_dec_len=4)
    eng.load_state(state.items())
    del state
    slot_seqs = [[s for j in range(G) for s in _synth.synth_token_batch(B, L, L, dims.vocab, seed=929 + 8 * sl + j)]
                 for sl in range(eng.num_slots)]
    ref_scores = None
    variants = [("tiled", 0, 0)] + [("dma1", 6, 0)] + [("dma2", 5, h) for h in (0, 2, 5, 20)] + [("dma2_ko%d" % k, 5, 2) for k in KO]
    for name, mode, hpw in variants:
        eng.set_option("attn_ko", int(name.split("_ko")[1]) if "_ko" in name else 0) if KO else None
        eng.set_option("attn_short", mode)
        eng.set_option("attn_heads_per_wg", hpw)
        eng.stage(slot_seqs[0], slot=0)
        eng.set_option("overlap", 0)
        eng.profile(True)
        eng.profile_reset()
        for _ in range(2):
            eng.score_staged([0], [bench.YES_ID, bench.NO_ID], slot=0)
        eng.sync()
        rep = eng.profile_report()
        eng.profile(False)
        eng.set_option("overlap", 1)
        sc = eng.read_scores(0)
        if ref_scores is None:
            ref_scores = sc
        a = rep["enc_attn"]
        print(json.dumps({"timing": name, "heads_per_wg": hpw, "attn_us_per_launch": round(a["ms"] * 1e3 / a["launches"], 1),
                          "launches": a["launches"], "scores_identical_to_tiled": bool(np.array_equal(sc, ref_scores)),
                          "total_ms_per_step": round(sum(v["ms"] for v in rep.values()) / (2 * G), 3)}), flush=True)
    # whole-pipeline passages/s, interleaved A/B (two rounds)
    for rnd in range(0 if "--no-pipeline" in sys.argv else 2):
        for name, mode, hpw in [("dma1", 6, 0), ("dma2", 5, 0), ("dma2", 5, 5)] + [("dma2_ko%d" % k, 5, 2) for k in KO]:
            eng.set_option("attn_ko", int(name.split("_ko")[1]) if "_ko" in name else 0) if KO else None
            eng.set_option("attn_short", mode)
            eng.set_option("attn_heads_per_wg", hpw)
            pipe = bench.GroupPipeline(eng, slot_seqs, B, G, [0], [bench.YES_ID, bench.NO_ID])
            pipe.stage_all()
            elapsed, ev_ms = bench.timed_run(eng, pipe, 40, 10, eng.sync)
            print(json.dumps({"pipeline": name, "heads_per_wg": hpw, "round": rnd, "passages_per_s": round(40 * B / elapsed, 1),
                              "ms_per_step": round(elapsed / 40 * 1e3, 3)}), flush=True)
    eng.close()


if __name__ == "__main__":
    import torch  # noqa: F401  (its HIP runtime first)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B or measurement build (tools only)
    import __graft_entry__ as ge
    if not os.environ.get("RK_ENGINE_LIB"):
        ge.build()
    if "--no-exact" not in sys.argv:
        ok = exactness()
        print(json.dumps({"exactness_ok": ok}), flush=True)
    if "--no-timing" not in sys.argv:
        timing()
