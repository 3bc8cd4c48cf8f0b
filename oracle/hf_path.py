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


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    import os
    return os.cpu_count() or 1


LOADER_PROBE = r"""
import sys, time, json
import torch
from torch.utils.data import DataLoader, Dataset
from transformers import DataCollatorWithPadding, T5Tokenizer
tok = T5Tokenizer.from_pretrained(sys.argv[1])
hits, L, bs = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
class DS(Dataset):                                   # ref: llmrankers/pairwise.py:17-26 (already tokenised here)
    def __init__(self, rows): self.rows = rows
    def __len__(self): return len(self.rows)
    def __getitem__(self, i): return {"input_ids": self.rows[i], "attention_mask": [1] * len(self.rows[i])}
rows = [[3 + (7 * i + j) % 150 for j in range(L - 1)] + [1] for i in range(hits)]
ts = []
for rep in range(4):
    t = time.perf_counter()
    loader = DataLoader(DS(rows), batch_size=bs, collate_fn=DataCollatorWithPadding(tok, max_length=512, padding="longest"),
                        shuffle=False, drop_last=False, num_workers=4)          # ref: llmrankers/pointwise.py:90-101
    n = 0
    for batch in loader:
        n += batch["input_ids"].shape[0]
    del loader
    ts.append(time.perf_counter() - t)
assert n == hits
print(json.dumps({"s_per_query": sorted(ts[1:])[len(ts[1:]) // 2], "first_s": ts[0]}))
"""


def dataloader_overhead(tokenizer_dir: str, hits: int, L: int, batch_size: int, timeout_s: float = 120.0):
    """The reference re-creates a 4-worker DataLoader inside every rerank() call (ref: pointwise.py:90-101): time that
    fixed cost alone, in a fresh interpreter (no model, no HIP context to fork)."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", LOADER_PROBE, tokenizer_dir, str(hits), str(L), str(batch_size)],
                         capture_output=True, text=True, timeout=timeout_s)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-500:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def time_cpu_baseline(dims, state, seqs: List[Sequence[int]], yes_id: int, no_id: int, sample_batch: int = 32,
                      n_timed: int = 3, tokenizer_dir: str = None, budget_s: float = 75.0) -> dict:
    """HF fp32 on the host CPU, torch threads = physical cores, 1 warm-up batch, median of
    `n_timed` batches of `sample_batch` sequences (32 = the bench's batch, the procedure BASELINE.md names).  The leg is
    bounded: if the warm-up batch shows that the timed batches would not fit `budget_s`, fewer are timed (at least one) and
    `sample` says so.  Also reports the reference-faithful per-query figure: the same forward time plus the fixed cost of
    the 4-worker DataLoader the reference forks in every rerank() call."""
    import torch
    cores = _physical_cores()
    torch.set_num_threads(cores)
    model = build_hf_model(dims, state)
    chunk = [list(s) for s in seqs[:sample_batch]]
    t0 = time.perf_counter()
    logits = pointwise_yes_no(model, chunk, len(chunk), yes_id, no_id)            # warm-up (also the parity sample)
    warm = time.perf_counter() - t0
    n_eff = max(1, min(n_timed, int((budget_s - warm) // max(warm, 1e-3))))
    times = []
    for _ in range(n_eff):
        t0 = time.perf_counter()
        pointwise_yes_no(model, chunk, len(chunk), yes_id, no_id)
        times.append(time.perf_counter() - t0)
    per_batch = float(np.median(times))
    value = len(chunk) / per_batch
    res = {"value": value, "unit": "passages/s", "cores": cores, "cpu_model": _cpu_model(), "kind": "port",
           "sample": f"1 warm-up ({warm:.1f} s) + median of {n_eff} batch(es) of {len(chunk)} x L={len(chunk[0])} "
                     f"(BASELINE.md section 3: B=32, median of >= 3" + ("" if n_eff >= 3 and len(chunk) == 32 else f"; bounded here to {budget_s:.0f} s of CPU work") +
                     f") through HF transformers fp32, torch {torch.__version__} CPU, "
                     f"{cores} threads = physical cores; {per_batch:.2f} s/batch (min {min(times):.2f}, max {max(times):.2f})",
           "logits": logits}
    if tokenizer_dir:
        try:
            ov = dataloader_overhead(tokenizer_dir, 100, len(chunk[0]), 32)
            per_query = 100.0 / value + ov["s_per_query"]
            res["reference_faithful"] = {
                "value": round(100.0 / per_query, 3), "unit": "passages/s",
                "dataloader_s_per_query": round(ov["s_per_query"], 3), "dataloader_first_call_s": round(ov["first_s"], 3),
                "note": "hits=100, batch_size=32: forward time at the rate above + the 4-worker DataLoader the reference "
                        "re-creates in every rerank() (ref: pointwise.py:90-101), timed alone in a fresh interpreter"}
        except Exception as exc:
            res["reference_faithful"] = {"error": repr(exc)[:300]}
    return res
