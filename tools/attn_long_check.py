#!/usr/bin/env python
"""Long-sequence encoder attention (attention.h: attn_enc_long_kernel, engine option attn_long) on the GPU: (1) the context rows
of one layer and the final scores against the tiled kernel (tolerance: both are fp16-P flash forms with different rounding
points) and against the fp32 oracle on a small model; (2) batch independence - a long sequence alone, among other long ones and
in a batch with short sequences gives the SAME BITS; short sequences in such a batch keep the bits of an all-short batch;
(3) time of a one-prompt and an eight-prompt setwise call with either kernel."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
from llmrankers import _synth
from llmrankers._engine import RkEngine
from oracle.t5_numpy import T5Oracle

def main():
    # (1) small model against the oracle: flan-t5-small dims, prompts of 200 .. 900 tokens
    dims = _synth.FLAN_T5_SMALL
    state = _synth.synth_state_dict(dims, seed=929, threads=8)
    eng = RkEngine(dims, device=0, max_tokens=16384, max_seqs=32, max_dec_len=4).load_state(state.items())
    seqs = _synth.synth_token_batch(3, 200, 900, dims.vocab, seed=41) + _synth.synth_token_batch(1, 385, 385, dims.vocab, seed=42) + _synth.synth_token_batch(1, 193, 193, dims.vocab, seed=43)
    ids = [5, 6, 7, 8]
    want = T5Oracle(dims, state).score_last(seqs, [0, 9], ids)
    res = {}
    for flag in (0, 1):
        eng.set_option("attn_long", flag)
        got = eng.score(seqs, [0, 9], ids)
        res[flag] = got
        print(json.dumps({"small_model_attn_long": flag, "lens": [len(s) for s in seqs], "max_abs_logit_err_vs_oracle": float(np.abs(got - want).max()),
                          "logit_scale": float(np.abs(want).max())}), flush=True)
    print(json.dumps({"small_model_max_abs_diff_between_kernels": float(np.abs(res[0] - res[1]).max())}), flush=True)
    # (2) batch independence with the new kernel
    eng.set_option("attn_long", 1)
    short = _synth.synth_token_batch(4, 30, 192, dims.vocab, seed=44)
    alone = [eng.score([s], [0, 9], ids) for s in seqs]
    mixed = eng.score(short + seqs + short[:2], [0, 9], ids)
    ok_long = all(np.array_equal(mixed[4 + i], alone[i][0]) for i in range(len(seqs)))
    eng.set_option("attn_long", 0)
    short_only = eng.score(short, [0, 9], ids)
    eng.set_option("attn_long", 1)
    ok_short = bool(np.array_equal(mixed[:4], short_only)) and bool(np.array_equal(mixed[4 + len(seqs):], short_only[:2]))
    print(json.dumps({"long_sequences_bit_identical_alone_vs_mixed_batch": bool(ok_long), "short_sequences_keep_the_short_kernel_bits": ok_short}), flush=True)
    eng.close()
    # (3) flan-t5-large dims: 1 / 2 / 8 prompts of 1560 tokens, the tiled kernel and every workgroup size of the new one (same bits)
    dims = _synth.FLAN_T5_LARGE
    eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(_synth.synth_tensors(dims, seed=929, threads=32))
    prefix, labels = [0, 5454], list(range(71, 82))
    for B in (1, 2, 8):
        seqs = _synth.synth_token_batch(B, 1560, 1560, dims.vocab, seed=7)
        sc = {}
        for flag, nw in ((0, 0), (1, 12), (1, 6), (1, 4), (1, 3), (1, 0)):
            eng.set_option("attn_long", flag); eng.set_option("attn_long_nw", nw)
            for _ in range(3):
                sc[(flag, nw)] = eng.score(seqs, prefix, labels)
            t = time.perf_counter()
            for _ in range(10):
                eng.score(seqs, prefix, labels)
            ms = (time.perf_counter() - t) / 10 * 1e3
            eng.profile(True); eng.profile_reset()
            for _ in range(3):
                eng.score(seqs, prefix, labels)
            eng.sync()
            rep = eng.profile_report(); eng.profile(False)
            print(json.dumps({"B": B, "attn_long": flag, "nw": nw, "likelihood_ms": round(ms, 3), "enc_attn_us_per_layer": round(rep["enc_attn"]["ms"] / 3 / 24 * 1e3, 1)}), flush=True)
        print(json.dumps({"B": B, "label_logits_max_abs_diff_between_kernels": float(np.abs(sc[(0, 0)] - sc[(1, 12)]).max()),
                          "same_bits_for_every_workgroup_size": bool(all(np.array_equal(sc[(1, 12)], sc[(1, n)]) for n in (6, 4, 3, 0)))}), flush=True)
    eng.close()

if __name__ == "__main__":
    main()
