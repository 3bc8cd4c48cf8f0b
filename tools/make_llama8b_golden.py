#!/usr/bin/env python
"""Llama-3-8B at FULL depth (32 layers, BASELINE.json configs[4] dimensions) through the fp32 numpy oracle on the build
container's host: one ~700-token setwise-sized prompt -> last-position logits (a block of label rows + the top of the
vocabulary).  The 8 G synthetic weights (llmrankers._synth, counter-based: the GPU box regenerates the same bits) are
generated layer by layer and dropped again, so the run needs a few GB, not 32.  Writes tests/golden/llama8b_full_depth.json
(a few KB); tests/test_gpu_kernels.py::test_llama_3_8b_full_depth_vs_oracle_golden replays it on the engine.

    python tools/make_llama8b_golden.py            (about ten minutes on 8 cores)
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth                                   # noqa: E402
from oracle.llama_numpy import LlamaOracle                      # noqa: E402

SEED, PROMPT_LEN, PROMPT_SEED = 929, 700, 800
LABELS = list(range(32, 32 + 64))


class LazyWeights:
    """name -> tensor, generated on access from its (stream, seed) counters and not kept; the tensors of one layer are
    produced together by a thread pool when the first of them is asked for."""

    def __init__(self, dims, seed, threads):
        self.jobs = {name: (stream, name, shape, std, is_norm, seed, 1.0)
                     for stream, (name, shape, std, is_norm) in enumerate(_synth.llama_tensor_specs(dims))}
        self.pool = ThreadPoolExecutor(max_workers=threads)
        self.group, self.cache = None, {}

    def __getitem__(self, name):
        grp = name.rsplit(".", 3)[0] if name.startswith("model.layers.") else name
        if grp != self.group:
            self.cache = {}
            names = [n for n in self.jobs if (n.rsplit(".", 3)[0] if n.startswith("model.layers.") else n) == grp]
            for n, arr in self.pool.map(_synth._make_tensor, [self.jobs[n] for n in names]):
                self.cache[n] = arr
            self.group = grp
        return self.cache[name]


def main():
    dims = _synth.LLAMA_3_8B
    ids = _synth.synth_token_batch(1, PROMPT_LEN, PROMPT_LEN, dims.vocab, seed=PROMPT_SEED)[0]
    orc = LlamaOracle.__new__(LlamaOracle)
    orc.d, orc.w = dims, LazyWeights(dims, SEED, os.cpu_count() or 8)
    t0 = time.time()
    h = orc.hidden_states(ids)[-1]
    print(f"[llama8b] 32 layers in {time.time() - t0:.0f}s; |h_last| = {np.linalg.norm(h):.4f}", flush=True)
    head = orc.w["lm_head.weight"]
    logits = (head @ h).astype(np.float32)
    top = np.argsort(-logits)[:8]
    out = {"dims": "llama-3-8b", "seed": SEED, "prompt_len": PROMPT_LEN, "prompt_seed": PROMPT_SEED,
           "label_ids": LABELS, "label_logits": [float(x) for x in logits[LABELS]],
           "top_ids": [int(i) for i in top], "top_logits": [float(logits[i]) for i in top],
           "logit_abs_max": float(np.abs(logits).max()), "h_last_norm": float(np.linalg.norm(h)),
           "generator": "tools/make_llama8b_golden.py (oracle/llama_numpy.py fp32, weights streamed layer by layer)"}
    with open(os.path.join(REPO, "tests", "golden", "llama8b_full_depth.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("top_ids", "top_logits", "logit_abs_max")}))


if __name__ == "__main__":
    main()
