#!/usr/bin/env python
"""Annotate tests/golden/rerank_cases.json with the reference's DECISION MARGINS (run after make_goldens.py).

Setwise parity is "identical docid rank order", which is only a fair demand where the reference's own decision is
not a numerical coin-flip: for every compare of every setwise case this replays the prompt through the fp32 oracle
and records the smallest gap between the winning logit and the runner-up (label logits for `likelihood`, full
vocabulary top-1 vs top-2 at each greedy step for `generation`).  The GPU test asserts the margin exceeds the fp16
noise floor before demanding an identical decision sequence."""
import contextlib
import io
import json
import os
import random
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO, os.path.join(REPO, "tests")]
from llmrankers import _synth                                   # noqa: E402
from llmrankers.rankers import SearchResult                      # noqa: E402
from llmrankers.setwise import SetwiseLlmRanker                  # noqa: E402
from llmrankers.pointwise import PointwiseLlmRanker              # noqa: E402
from _stub import OracleRuntime                                  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


class MarginRuntime(OracleRuntime):
    def __init__(self, dims, state):
        super().__init__(dims, state)
        self.margins = []

    def score(self, seqs, dec_prefix, out_ids):
        lg = super().score(seqs, dec_prefix, out_ids)
        if lg.shape[1] >= 2 and len(out_ids) > 2:          # likelihood compare over label ids
            top = np.sort(lg, axis=1)
            self.margins.extend((top[:, -1] - top[:, -2]).tolist())
        return lg

    def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0):
        for ids in seqs:
            enc = self.orc.encode(ids)
            cur = list(dec_prefix)
            for _ in range(max_new):
                lg = np.sort(self.orc.decode(enc, cur)[-1])
                self.margins.append(float(lg[-1] - lg[-2]))
                nxt = int(np.argmax(self.orc.decode(enc, cur)[-1]))
                cur.append(nxt)
                if nxt == eos_id:
                    break
        return super().greedy(seqs, dec_prefix, max_new, eos_id, pad_id)


def main():
    import tempfile
    from safetensors.numpy import load_file
    from transformers import T5Tokenizer
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        specs = json.load(f)
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        data = json.load(f)
    tmp = tempfile.mkdtemp(prefix="rk_margin_")
    rts = {}
    for name, spec in specs.items():
        path = os.path.join(tmp, name)
        _synth.write_checkpoint(path, spec, os.path.join(GOLD, "tok"))
        dims = _synth.NAMED_DIMS[spec["dims"]]
        rts[name] = (MarginRuntime(dims, load_file(os.path.join(path, "model.safetensors"))), T5Tokenizer.from_pretrained(path))
    for case in data["cases"]:
        rt, tok = rts[case["ckpt"]]
        rt.margins = []
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        if case["kind"] == "setwise":
            rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=case["num_child"], k=case["k"], scoring=case["scoring"],
                                  method=case["method"], num_permutation=case["num_permutation"])
            random.seed(929)
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    rk.rerank(case["query"], ranking)
                except IndexError:
                    pass
            case["min_margin"] = float(min(rt.margins)) if rt.margins else None
        else:
            sc = sorted((s for _, s in case["result"]), reverse=True)
            case["min_margin"] = float(min(a - b for a, b in zip(sc, sc[1:]))) if len(sc) > 1 else None
    with open(os.path.join(GOLD, "rerank_cases.json"), "w") as f:
        json.dump(data, f)
    ms = [(c["kind"], c.get("scoring", c.get("method")), round(c["min_margin"], 5)) for c in data["cases"] if c["min_margin"] is not None]
    print("min margins:", ms)


if __name__ == "__main__":
    main()
