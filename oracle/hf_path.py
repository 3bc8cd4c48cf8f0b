"""ORACLE — TEST / BASELINE INFRASTRUCTURE ONLY.  Not a product path, not a fallback.

The reference's CPU path restated against the SAME third-party library it calls: HuggingFace transformers'
T5ForConditionalGeneration in fp32 (ref: llmrankers/pointwise.py:20-24 picks fp32 when device != 'cuda').
The reference's Python cannot travel to the GPU box, transformers (a wheel in the image) can; this file is the
"port" of ref: llmrankers/pointwise.py:84-127 (tokenised batch -> right-pad to longest -> llm(...) -> logits of
yes/no -> softmax) used (a) by bench.py's cpu_baseline leg on the GPU box's host cores and (b) by tests to
cross-check the numpy oracle at sizes beyond the committed goldens.  Shown output-identical to the real
reference import on the fixtures by tests/test_oracle_golden.py::test_hf_path_equals_reference_goldens.
"""
from __future__ import annotations

import time
from typing import Dict, List, Sequence

import numpy as np


def build_hf_model(dims, state: Dict[str, np.ndarray]):
    """fp32 eval T5ForConditionalGeneration on CPU holding exactly `state` (HF names)."""
    import torch
    from transformers import T5Config, T5ForConditionalGeneration
    cfg_d = {k: v for k, v in dims.to_hf_config().items() if k not in ("architectures", "model_type")}
    cfg = T5Config(**cfg_d)
    with torch.device("meta"):
        model = T5ForConditionalGeneration(cfg)
    model = model.to_empty(device="cpu")
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in state.items()}
    tsd["encoder.embed_tokens.weight"] = tsd["shared.weight"]
    tsd["decoder.embed_tokens.weight"] = tsd["shared.weight"]
    if "lm_head.weight" not in tsd:
        tsd["lm_head.weight"] = tsd["shared.weight"]
    res = model.load_state_dict(tsd, strict=False, assign=True)
    assert not res.unexpected_keys, res.unexpected_keys
    if not dims.tied_head:
        # transformers 5.x ties lm_head to shared on construction; flan-style checkpoints are untied and unscaled
        model.lm_head.weight = torch.nn.Parameter(tsd["lm_head.weight"].clone(), requires_grad=False)
    model.config.scale_decoder_outputs = bool(dims.tied_head)
    return model.eval()


def pad_batch(chunk: Sequence[Sequence[int]], pad_id: int = 0):
    L = max(len(s) for s in chunk)
    ids = np.full((len(chunk), L), pad_id, dtype=np.int64)
    mask = np.zeros((len(chunk), L), dtype=np.int64)
    for b, s in enumerate(chunk):
        ids[b, :len(s)] = s
        mask[b, :len(s)] = 1
    return ids, mask


def pointwise_yes_no(model, seqs: List[Sequence[int]], batch_size: int, yes_id: int, no_id: int) -> np.ndarray:
    """ref: pointwise.py:102-127 on pre-tokenised prompts: returns [n, 2] = (logit_yes, logit_no)."""
    import torch
    out = np.zeros((len(seqs), 2), dtype=np.float32)
    with torch.no_grad():
        for s0 in range(0, len(seqs), batch_size):
            chunk = seqs[s0:s0 + batch_size]
            ids, mask = pad_batch(chunk)
            dec = torch.zeros((len(chunk), 1), dtype=torch.long)
            logits = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                           decoder_input_ids=dec).logits
            out[s0:s0 + len(chunk), 0] = logits[:, 0, yes_id].numpy()
            out[s0:s0 + len(chunk), 1] = logits[:, 0, no_id].numpy()
    return out


def time_cpu_baseline(dims, state, seqs: List[Sequence[int]], batch_size: int, yes_id: int, no_id: int,
                      max_seconds: float = 30.0) -> dict:
    """Time the HF fp32 CPU path on a bounded sample of the bench workload (whole batches, <= max_seconds)."""
    import torch
    model = build_hf_model(dims, state)
    threads = torch.get_num_threads()
    times, done, scores = [], 0, []
    t_all = time.time()
    for s0 in range(0, len(seqs), batch_size):
        chunk = seqs[s0:s0 + batch_size]
        t0 = time.time()
        scores.append(pointwise_yes_no(model, chunk, batch_size, yes_id, no_id))
        times.append(time.time() - t0)
        done += len(chunk)
        if time.time() - t_all > max_seconds * 0.5:
            break
    per_batch = float(np.median(times))
    return {"value": len(seqs[:batch_size]) / per_batch, "unit": "passages/s", "cores": int(threads), "kind": "port",
            "sample": f"{len(times)} batch(es) of {batch_size} x L={len(seqs[0])} through HF transformers fp32 "
                      f"(torch {torch.__version__} CPU, {threads} threads), median {per_batch:.2f} s/batch",
            "logits": np.concatenate(scores, axis=0)}
