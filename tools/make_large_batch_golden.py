#!/usr/bin/env python
"""BASELINE.json configs[1] at full size through the reference's own arithmetic: HF T5ForConditionalGeneration fp32 on the
CPU (oracle/hf_path.py - output-identical to the imported reference on the fixtures) over the bench's batch of 32 x 184
tokens, flan-t5-large dimensions, the synthetic weights of seed 929 -> the two label logits of every passage.  Written once
in the build container to tests/golden/config2_large_batch.npz (a few hundred bytes);
tests/test_gpu_kernels.py::test_flan_t5_large_full_batch_vs_hf_golden replays it on the engine."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth          # noqa: E402
from oracle import hf_path             # noqa: E402

YES_ID, NO_ID = 2163, 465


def main():
    import torch
    torch.set_num_threads(os.cpu_count() or 8)
    dims = _synth.FLAN_T5_LARGE
    state = _synth.synth_state_dict(dims, seed=929, threads=os.cpu_count() or 8)
    model = hf_path.build_hf_model(dims, state)
    out = {"seed": np.array(929), "ids": np.array([YES_ID, NO_ID])}
    for tag, (n, lo, hi, seed) in {"uniform": (32, 184, 184, 929), "ragged": (32, 96, 184, 930)}.items():
        seqs = [list(s) for s in _synth.synth_token_batch(n, lo, hi, dims.vocab, seed=seed)]
        t0 = time.time()
        logits = hf_path.pointwise_yes_no(model, seqs, 32, YES_ID, NO_ID)
        print(f"[{tag}] HF fp32 forward of {n} sequences in {time.time() - t0:.1f}s; logit range {logits.min():.3f}..{logits.max():.3f}", flush=True)
        out[f"{tag}.args"] = np.array([n, lo, hi, seed])
        out[f"{tag}.logits"] = logits.astype(np.float32)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "config2_large_batch.npz"), **out)


if __name__ == "__main__":
    main()
