#!/usr/bin/env python
"""A/B sweeps of the bench workload in ONE process (one engine, one weight upload): every argument is a configuration
`G=8,steps=64,warmup=16[,rep=2][,engine_option=value ...]`; prints one JSON line per configuration with passages/s
measured exactly like bench.py's timed region (same GroupPipeline / timed_run).  GPU-minutes are scarce: a sweep of ten
variants costs one engine build instead of ten.

    python tools/sweep.py "G=8,steps=64,warmup=16" "G=8,steps=64,warmup=16,gemm_split=0"
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def main():
    import torch  # noqa: F401  (its HIP runtime first)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B or measurement build (tools only)
    import bench
    import __graft_entry__ as ge
    ge.build()
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    cfgs = []
    for a in sys.argv[1:]:
        kv = dict(x.split("=") for x in a.split(","))
        cfgs.append({k: int(v) for k, v in kv.items()})
    dims = _synth.NAMED_DIMS[os.environ.get("RK_MODEL", "flan-t5-large")]
    B, L = 32, 184
    gmax = max(c.get("G", 8) for c in cfgs)
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=max(8192, gmax * B * L), max_seqs=max(128, gmax * B), max_dec_len=4)
    eng.load_state(state.items())
    del state
    known = {"gemm_persistent": 1, "attn_short": 5, "gemm_variant": 0,
             "xattn_direct": 1, "overlap": 1, "gemm_glds": 1, "attn_heads_per_wg": 0, "fold_norm": 1, "dec_graph": 1, 
             "dec_fold_norm": 1, "gemm_s64_stages": 0, "greedy_spec": 160, "gemm_split": 1, "dec_fuse": 1, "dec_fuse_rows": 0,
             "xattn_mfma": 1, "dec_ffn_tiled": 1, "consumer_stats": 1, 
             "dec_attn_seq": 1, "attn_long": 1, "attn_long_nw": 0, "attn_long_xcd": 1,
             "llama_attn_dma": 1, "llama_attn_nw": 0}
    defaults = {}
    for c in cfgs:
        G, steps, warmup, rep = c.get("G", 8), c.get("steps", 64), c.get("warmup", 16), c.get("rep", 1)
        opts = {k: v for k, v in c.items() if k not in ("G", "steps", "warmup", "rep")}
        for k in defaults:                                   # options of earlier configurations go back to their defaults
            if k not in opts:
                eng.set_option(k, defaults[k])
        for k, v in opts.items():
            defaults.setdefault(k, known[k])
            eng.set_option(k, v)
        slot_seqs = [[s for j in range(G) for s in _synth.synth_token_batch(B, L, L, dims.vocab, seed=929 + 8 * sl + j)]
                     for sl in range(eng.num_slots)]
        vals = []
        for _ in range(rep):
            pipe = bench.GroupPipeline(eng, slot_seqs, B, G, [0], [bench.YES_ID, bench.NO_ID])
            pipe.stage_all()
            elapsed, ev_ms = bench.timed_run(eng, pipe, steps, warmup, eng.sync)
            vals.append(round(steps * B / elapsed, 1))
        print(json.dumps({"cfg": c, "passages_per_s": vals, "ms_per_step": round(elapsed / steps * 1e3, 3)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
