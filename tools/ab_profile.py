#!/usr/bin/env python
"""A/B of engine options on the bench workload in ONE process: for every argument `name:opt=val,opt=val` (or `name:` for the
defaults) print (a) the per-kernel-class milliseconds per step of a serial profiled pass and (b) the pipeline's passages/s,
two interleaved rounds.  Usage: python tools/ab_profile.py base: pair:attn_short=1"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def main():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B or measurement build (tools only)
    import bench
    import __graft_entry__ as ge
    if not os.environ.get("RK_ENGINE_LIB"):
        ge.build()
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    cfgs = []
    for a in sys.argv[1:]:
        name, _, opts = a.partition(":")
        cfgs.append((name, {k: int(v) for k, v in (x.split("=") for x in opts.split(",") if x)}))
    dims = _synth.FLAN_T5_LARGE
    B, L, G = 32, 184, int(os.environ.get("RK_G", "10"))
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=G * B * L, max_seqs=G * B, max_dec_len=4)
    eng.load_state(state.items())
    del state
    slot_seqs = [[s for j in range(G) for s in _synth.synth_token_batch(B, L, L, dims.vocab, seed=929 + 8 * sl + j)]
                 for sl in range(eng.num_slots)]
    touched = {}
    ref = None

    def apply(opts):
        for k, v in touched.items():
            if k not in opts:
                eng.set_option(k, v)
        for k, v in opts.items():
            touched.setdefault(k, DEFAULTS.get(k, 0))
            eng.set_option(k, v)
    for name, opts in cfgs:
        apply(opts)
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
        ref = sc if ref is None else ref
        print(json.dumps({"cfg": name, "identical_to_first": bool((sc == ref).all()),
                          "ms_per_step": {k: round(v["ms"] / (2 * G), 4) for k, v in rep.items() if v["launches"]},
                          "total": round(sum(v["ms"] for v in rep.values()) / (2 * G), 3)}), flush=True)
    for rnd in range(2):
        for name, opts in cfgs:
            apply(opts)
            pipe = bench.GroupPipeline(eng, slot_seqs, B, G, [0], [bench.YES_ID, bench.NO_ID])
            pipe.stage_all()
            elapsed, _ = bench.timed_run(eng, pipe, 4 * G, G, eng.sync)
            print(json.dumps({"cfg": name, "round": rnd, "passages_per_s": round(4 * G * B / elapsed, 1), "ms_per_step": round(elapsed / (4 * G) * 1e3, 3)}), flush=True)
    eng.close()


DEFAULTS = {"dec_fuse": 1, "gemm_split": 1, "dec_ffn_tiled": 1, "xattn_mfma": 1, "attn_short": 5, "attn_heads_per_wg": 0, "gemm_persistent": 1, "overlap": 1, "dec_graph": 1,
            "fold_norm": 1, "dec_fold_norm": 1, "xattn_direct": 1, "gemm_variant": 0}

if __name__ == "__main__":
    main()
