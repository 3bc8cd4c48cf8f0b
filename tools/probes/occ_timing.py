"""Do two workgroups of the DMA attention kernel really share a CU?  heads_per_wg = 16 -> one workgroup per sequence:
256 sequences = one workgroup per CU, 512 = two.  Co-residency shows as time(512) well below 2 x time(256)."""
import sys, os, json
sys.path[:0] = ["/root/repo/llm-rankers_amd", "/root/repo"]
import torch
import numpy as np
from llmrankers import _synth
from llmrankers._engine import RkEngine
dims = _synth.T5Dims(vocab=32128, d_model=1024, n_heads=16, d_kv=64, d_ff=2816, n_enc=2, n_dec=1)
eng = RkEngine(dims, 0, max_tokens=768 * 184, max_seqs=768, max_dec_len=4).load_state(_synth.synth_state_dict(dims, 3, threads=16).items())
print("occupancy:", eng.debug_read("occupancy", 10).tolist())
for mode, hpw in ((5, 16), (5, 4), (1, 0)):
    for n in (128, 256, 384, 512, 768):
        seqs = _synth.synth_token_batch(n, 184, 184, dims.vocab, seed=7)
        eng.set_option("attn_short", mode); eng.set_option("attn_heads_per_wg", hpw)
        eng.stage(seqs, slot=0)
        eng.set_option("overlap", 0); eng.profile(True); eng.profile_reset()
        for _ in range(3):
            eng.score_staged([0], [3, 4], slot=0)
        eng.sync()
        a = eng.profile_report()["enc_attn"]
        eng.profile(False)
        print(json.dumps({"mode": mode, "hpw": hpw, "n_seq": n, "attn_us": round(a["ms"] * 1e3 / a["launches"], 1)}), flush=True)
