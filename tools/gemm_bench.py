#!/usr/bin/env python
"""Per-shape throughput of the engine GEMM at the encoder shapes of BASELINE config 1 (T = 32 x 184 = 5888)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth                      # noqa: E402
from llmrankers._engine import RkEngine            # noqa: E402

SHAPES = [  # name, M, N, K, epi
    ("qkv", 5888, 3072, 1024, 0), ("o", 5888, 1024, 1024, 1), ("ffn_in_geglu", 5888, 5632, 1024, 2),
    ("ffn_out", 5888, 1024, 2816, 1), ("cross_kv", 5888, 49152, 1024, 0), ("square4k", 4096, 4096, 4096, 0),
    ("square8k", 8192, 8192, 8192, 0), ("sq4k_k4224", 4096, 4096, 4224, 0), ("sq4k_k3968", 4096, 4096, 3968, 0),
]


def main():
    import torch  # noqa: F401  (its HIP runtime first)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B or measurement build (tools only)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    m_override = int(os.environ.get("RK_BENCH_M", "0"))
    if m_override:
        for i, (name, m, n, k, epi) in enumerate(SHAPES[:5]):
            SHAPES[i] = (name, m_override, n, k, epi)
        del SHAPES[5:]
    extra = os.environ.get('RK_BENCH_SHAPES')
    if extra:
        SHAPES.extend((f'x{i}', *[int(v) for v in t.split('x')]) for i, t in enumerate(extra.split(',')))
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    dims = _synth.TOY_GATED_UNTIED
    eng = RkEngine(dims, 0, max_tokens=256, max_seqs=4, max_dec_len=4).load_state(_synth.synth_state_dict(dims, 1).items())
    for kv in os.environ.get("RK_OPTS", "").split(","):       # engine options for the whole run, e.g. RK_OPTS=gemm_glds=0
        if kv:
            eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    out = {}
    variants = [int(v) for v in os.environ.get("RK_GEMM_VARIANTS", "0,1,2,3,4,5,6").split(",")]
    for name, m, n, k, epi in SHAPES:
        if only and name not in only:
            continue
        row = {}
        for v in variants:
            if v == 3 and epi == 2:
                continue
            eng.set_option("gemm_variant", v)
            ms = eng.gemm_bench(m, n, k, epi, iters if n < 40000 and m * n * k < 2e11 else max(3, iters // 5))
            row[v] = (round(ms * 1e3, 1), round(2.0 * m * n * k / (ms * 1e-3) / 1e12, 1))
        out[name] = {"M": m, "N": n, "K": k, "us_tflops_by_variant": row}
        print(f"{name:14s} M={m:5d} N={n:5d} K={k:5d}  " + "  ".join(f"v{v}: {us:7.1f}us {tf:6.1f}TF" for v, (us, tf) in row.items()), flush=True)
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
