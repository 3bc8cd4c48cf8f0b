#!/usr/bin/env python
"""Where the time of attn_causal128_dma_kernel goes: timing-only knock-outs of a measurement build
(hipcc ... -DRK_MEASURE -o exp/librk_engine_measure.so; RK_ENGINE_LIB=exp/librk_engine_measure.so python tools/llama_attn_ko.py).
Llama-3-8B widths, RK_LAYERS layers (pool weights), one and four 1 536-token prompts; results of knocked-out runs are garbage."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
from llmrankers import _synth
from llmrankers._engine import RkLlamaEngine

NAMES = {256: "full, 112 KiB of LDS per workgroup (one per CU for certain)", 0: "full", 64: "return at once (launch only)", 128: "prologue only (Q rows, chunk 0, one barrier)", 1: "no DMA in the loop", 2: "no score MFMAs", 4: "no softmax", 8: "no P V", 16: "no barrier", 32: "no stores",
         2 | 8: "no MFMAs at all", 2 | 4 | 8: "DMA + barriers only", 1 | 2 | 4 | 8: "barriers only", 1 | 16: "no DMA, no barrier",
         63: "nothing (launch, prologue, loop skeleton)"}
layers, L = int(os.environ.get("RK_LAYERS", "4")), 1536
d = _synth.LLAMA_3_8B
dims = _synth.LlamaDims(vocab=d.vocab, hidden=d.hidden, n_heads=d.n_heads, n_kv_heads=d.n_kv_heads, head_dim=d.head_dim,
                        intermediate=d.intermediate, n_layers=layers, bos_token_id=d.bos_token_id, eos_token_id=d.eos_token_id)
eng = RkLlamaEngine(dims, 0, max_tokens=8192, max_seqs=8).load_state(_synth.synth_tensors_pool(dims, seed=929))
NW = int(os.environ.get("RK_NW", "0"))
if os.environ.get("RK_KO"):
    NAMES = {int(k): NAMES.get(int(k), "") for k in os.environ["RK_KO"].split(",")}
for B in (1, 4):
    seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=3)
    for ko, name in NAMES.items():
        eng.set_option("attn_ko", ko); eng.set_option("llama_attn_nw", NW)
        eng.greedy1(seqs)
        eng.profile(True); eng.profile_reset()
        for _ in range(3):
            eng.greedy1(seqs)
        eng.sync()
        rep = eng.profile_report(); eng.profile(False)
        print(json.dumps({"B": B, "nw": NW or 4, "ko": ko, "what": name, "attn_us_per_layer": round(rep["enc_attn"]["ms"] / 3 / layers * 1e3, 1)}), flush=True)
eng.set_option("attn_ko", 0)
eng.close()
