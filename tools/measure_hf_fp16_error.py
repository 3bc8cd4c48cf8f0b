#!/usr/bin/env python
"""What is the REFERENCE's own accelerator-precision error?  BASELINE.md section 2 sets "every score within 1e-3 of the reference
HF path (fp32 CPU is ground truth)"; the reference's CUDA path itself is fp16 weights with fp32 `wo` / residual stream (ref:
llmrankers/pointwise.py:22-23; hf: modeling_t5.py `_keep_in_fp32_modules = ["wo"]`).  This tool runs HF
T5ForConditionalGeneration on the host in BOTH precisions on the same synthetic checkpoint and ragged batch and prints how far
the fp16 path is from fp32 for the two pointwise scores: yes_no (probability) and qlm (sum of label log-probabilities).
The engine's error against the same fp32 ground truth is measured by the GPU tests; DESIGN.md section 4 quotes both.

    python tools/measure_hf_fp16_error.py [dims-name]        (default flan-t5-small; CPU fp16 matmuls are slow)
"""
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth   # noqa: E402


def main():
    import torch
    from transformers import T5ForConditionalGeneration
    name = sys.argv[1] if len(sys.argv) > 1 else "flan-t5-small"
    dims = _synth.NAMED_DIMS[name]
    seqs = _synth.synth_token_batch(16, 100, 184, dims.vocab, seed=77)
    labels = [0] + np.random.RandomState(6).randint(3, dims.vocab - 28, size=32).tolist()
    L = max(len(s) for s in seqs)
    ids = torch.zeros((len(seqs), L), dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for b, s in enumerate(seqs):
        ids[b, :len(s)] = torch.as_tensor(np.asarray(s, dtype=np.int64))
        mask[b, :len(s)] = 1
    yes_id, no_id = 2163 % dims.vocab, 465 % dims.vocab
    out = {}
    with tempfile.TemporaryDirectory() as d:
        _synth.write_checkpoint(d, {"dims": name, "seed": 929, "gain": 1.0})
        for tag, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
            m = T5ForConditionalGeneration.from_pretrained(d, torch_dtype=dt).eval()
            with torch.no_grad():
                lg = m(input_ids=ids, attention_mask=mask, decoder_input_ids=torch.zeros((len(seqs), 1), dtype=torch.long)).logits[:, 0].float()
                p = torch.softmax(lg[:, [yes_id, no_id]], dim=-1)[:, 0].numpy()
                lab = torch.as_tensor(labels).unsqueeze(0).repeat(len(seqs), 1)
                dec_in = m._shift_right(lab)
                lq = m(input_ids=ids, attention_mask=mask, decoder_input_ids=dec_in).logits.float()
                ce = torch.nn.functional.cross_entropy(lq.reshape(-1, lq.shape[-1]), lab.reshape(-1), reduction="none").reshape(len(seqs), -1)
                q = (-ce.sum(dim=1)).numpy()
            out[tag] = (p, q, lg[:, [yes_id, no_id]].numpy())
    p32, q32, l32 = out["fp32"]
    p16, q16, l16 = out["fp16"]
    res = {"dims": name, "passages": len(seqs), "label_positions": len(labels),
           "yes_no_max_abs_prob_error_fp16_vs_fp32": float(np.abs(p16 - p32).max()),
           "yes_no_max_abs_logit_error_fp16_vs_fp32": float(np.abs(l16 - l32).max()),
           "qlm_score_magnitude": float(np.abs(q32).max()),
           "qlm_max_abs_error_fp16_vs_fp32": float(np.abs(q16 - q32).max()),
           "qlm_max_rel_error_fp16_vs_fp32": float(np.abs(q16 - q32).max() / np.abs(q32).max()),
           "what": "HF T5ForConditionalGeneration torch_dtype=float16 (wo kept fp32 by HF) against float32 on the host CPU, synthetic "
                   "checkpoint (llmrankers._synth seed 929), 16 ragged passages of 100..184 tokens; = the reference's own 'cuda' precision"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
