#!/usr/bin/env python
"""Llama causal attention by LDS-DMA (llama_kernels.h: attn_causal128_dma_kernel, engine option llama_attn_dma) on the GPU:
(1) toy Llama (2 layers, grouped-query): the last-position logits of MANY PREFIXES of one 700-token sequence - i.e. the logits
of many positions, every chunk / diagonal / block-boundary case - with either kernel against the fp32 oracle and against each
other; (2) batch independence: every prefix alone gives the bits it has inside the batch; (3) Llama-3-8B widths (RK_LAYERS
layers, timing-only pool weights): attention time per layer and ms per compare at one and four 1 536-token prompts, both kernels."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
from llmrankers import _synth
from llmrankers._engine import RkLlamaEngine
from oracle.llama_numpy import LlamaOracle


def main():
    dims = _synth.TOY_LLAMA
    state = _synth.synth_state_dict(dims, seed=929)
    base = _synth.synth_token_batch(1, 700, 700, dims.vocab, seed=17)[0]
    lens = sorted(set([1, 2, 3, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 161, 191, 192, 193, 255, 256, 257, 383, 384, 385,
                       511, 512, 513, 639, 640, 641, 700] + list(range(7, 700, 37))))
    seqs = [base[:n] for n in lens]
    ids = list(range(64))
    eng = RkLlamaEngine(dims, device=0, max_tokens=32768, max_seqs=128).load_state(state.items())
    want = LlamaOracle(dims, state).last_logits(seqs)[:, ids]
    scale = float(np.abs(want).max())
    res = {}
    for flag in (0, 1):
        eng.set_option("llama_attn_dma", flag)
        res[flag] = eng.last_logits(seqs, ids)
        err = np.abs(res[flag] - want).max(axis=1)
        print(json.dumps({"toy_llama_attn_dma": flag, "n_prefixes": len(lens), "max_abs_logit_err_vs_oracle": float(err.max()),
                          "worst_prefix_len": int(lens[int(err.argmax())]), "logit_scale": scale, "finite": bool(np.isfinite(res[flag]).all())}), flush=True)
    print(json.dumps({"toy_max_abs_diff_between_kernels": float(np.abs(res[0] - res[1]).max())}), flush=True)
    eng.set_option("llama_attn_dma", 1)
    eng.set_option("llama_attn_nw", 8)
    print(json.dumps({"eight_waves_per_workgroup_same_bits": bool(np.array_equal(eng.last_logits(seqs, ids), res[1]))}), flush=True)
    eng.set_option("llama_attn_nw", 0)
    alone_ok = all(np.array_equal(eng.last_logits([s], ids)[0], res[1][i]) for i, s in enumerate(seqs))
    rev = eng.last_logits(seqs[::-1], ids)[::-1]
    print(json.dumps({"bit_identical_alone_vs_batch": bool(alone_ok), "bit_identical_in_reversed_batch": bool(np.array_equal(rev, res[1]))}), flush=True)
    eng.close()

    layers, L = int(os.environ.get("RK_LAYERS", "8")), 1536
    d = _synth.LLAMA_3_8B
    dims = _synth.LlamaDims(vocab=d.vocab, hidden=d.hidden, n_heads=d.n_heads, n_kv_heads=d.n_kv_heads, head_dim=d.head_dim,
                            intermediate=d.intermediate, n_layers=layers, bos_token_id=d.bos_token_id, eos_token_id=d.eos_token_id)
    eng = RkLlamaEngine(dims, 0, max_tokens=8192, max_seqs=8).load_state(_synth.synth_tensors_pool(dims, seed=929))
    for B in (1, 4):
        seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=3)
        toks = {}
        for flag, nw in ((0, 0), (1, 4), (1, 8), (1, 4), (1, 8)):
            eng.set_option("llama_attn_dma", flag); eng.set_option("llama_attn_nw", nw)
            for _ in range(2):
                toks[flag] = eng.greedy1(seqs)
            t = time.perf_counter()
            for _ in range(5):
                eng.greedy1(seqs)
            ms = (time.perf_counter() - t) / 5 * 1e3
            eng.profile(True); eng.profile_reset()
            eng.greedy1(seqs); eng.sync()
            rep = eng.profile_report()
            eng.profile(False)
            flop = layers * 2.0 * L * L * dims.n_heads * 128 * B
            us = rep["enc_attn"]["ms"] / layers * 1e3
            print(json.dumps({"B": B, "L": L, "layers": layers, "llama_attn_dma": flag, "nw": nw, "ms_per_call": round(ms, 2),
                              "attn_us_per_layer": round(us, 1), "attn_tflops": round(flop / layers / us / 1e6, 1)}), flush=True)
        eng.set_option("llama_attn_nw", 0)
        print(json.dumps({"B": B, "same_greedy_tokens": bool(np.array_equal(toks[0], toks[1]))}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
