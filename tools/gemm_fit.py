#!/usr/bin/env python
"""Per-tile cost model of the ping-pong GEMM: time per launch at M = 47104 for K in a sweep -> fixed cost per tile
(epilogue + fill) and cost per K tile, per epilogue kind; and the same with half the workgroups (is the epilogue bound
per CU or by the chip's HBM?)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
from llmrankers import _synth
from llmrankers._engine import RkEngine
dims = _synth.TOY_GATED_UNTIED
eng = RkEngine(dims, 0, max_tokens=256, max_seqs=4, max_dec_len=4).load_state(_synth.synth_state_dict(dims, 1).items())
eng.set_option("gemm_variant", 5)
M = int(os.environ.get("RK_BENCH_M", "47104"))
for name, N, epi in (("qkv", 3072, 0), ("o", 1024, 1), ("ffn_in", 5632, 2), ("store_f32", 1024, 4)):
    for wgs in (1, 128):
        eng.set_option("gemm_persistent", wgs)
        rows = []
        for K in (128, 256, 512, 1024, 2048):
            ms = eng.gemm_bench(M, N, K, epi, 8)
            rows.append((K, round(ms * 1e3, 1)))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        rounds = -(-tiles // (256 if wgs == 1 else wgs))
        (k0, t0), (k1, t1) = rows[0], rows[-1]
        per_kt = (t1 - t0) / ((k1 - k0) / 64) / rounds
        fixed = t0 / rounds - per_kt * (k0 / 64)
        print(f"{name:9s} wgs={256 if wgs == 1 else wgs:3d} tiles={tiles} rounds={rounds} us_by_K={rows}  per K tile {per_kt:.2f} us, fixed per tile {fixed:.1f} us", flush=True)
eng.set_option("gemm_persistent", 1)
