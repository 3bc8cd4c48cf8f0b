#!/usr/bin/env python
"""Per-class GEMM time of a Llama-3-8B-width prefill (RK_LAYERS layers, timing-only pool weights) with every tile variant
forced in turn (engine option gemm_variant; all variants give the same bits): which variant should choose_variant pick at
M = 1536 (one setwise prompt) and M = 6144 (four)?"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
from llmrankers import _synth
from llmrankers._engine import RkLlamaEngine

layers, L = int(os.environ.get("RK_LAYERS", "8")), int(os.environ.get("RK_L", "1536"))
d = _synth.LLAMA_3_8B
dims = _synth.LlamaDims(vocab=d.vocab, hidden=d.hidden, n_heads=d.n_heads, n_kv_heads=d.n_kv_heads, head_dim=d.head_dim,
                        intermediate=d.intermediate, n_layers=layers, bos_token_id=d.bos_token_id, eos_token_id=d.eos_token_id)
eng = RkLlamaEngine(dims, 0, max_tokens=8192, max_seqs=8).load_state(_synth.synth_tensors_pool(dims, seed=929))
for B in (1, 2, 4):
    seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=3)
    for v in (0, 1, 2, 3, 4, 5, 6):
        eng.set_option("gemm_variant", v)
        for _ in range(2):
            eng.greedy1(seqs)
        eng.profile(True); eng.profile_reset()
        for _ in range(3):
            eng.greedy1(seqs)
        eng.sync()
        rep = eng.profile_report(); eng.profile(False)
        print(json.dumps({"M": B * L, "gemm_variant": v, **{k: round(x["ms"] / 3 / layers * 1e3, 1) for k, x in rep.items() if k.startswith("enc_gemm") and x["launches"]}}), flush=True)
eng.set_option("gemm_variant", 0)
eng.close()
