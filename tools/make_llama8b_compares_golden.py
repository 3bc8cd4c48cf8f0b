#!/usr/bin/env python
"""Llama-3-8B at FULL depth, compares of the configs[4] heapsort query: the first N prompts of the one-by-one heapsort of
tests/golden/llama_setwise_query.json's query (hits=100, num_child=10, k=10; the leaf-level compares of the build phase - their
prompts do not depend on any earlier decision) through the fp32 numpy oracle at all 32 layers with the lm_head rows of the label
tokens boosted as in the GPU test: greedy token, its margin over the runner-up and the label logits per prompt -> appended to
tests/golden/llama8b_full_depth.json as "query_compares" (token ids of the prompts included: the GPU box rebuilds nothing but
the weights).  The round-5 review asked for it: the full-depth query was compared with the oracle on ONE synthetic prompt only.

    python tools/make_llama8b_compares_golden.py [N=4]        (about ten host minutes per prompt on 8 cores)
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")]
from llmrankers import _synth                                   # noqa: E402
from llmrankers.rankers import SearchResult                     # noqa: E402
from llmrankers.setwise import SetwiseLlmRanker                 # noqa: E402
from oracle.llama_numpy import LlamaOracle                      # noqa: E402
from make_llama8b_golden import LazyWeights                     # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


class Done(Exception):
    pass


class RecordingRuntime:
    """Collects the prompts of the first `want` compares; answers the first label (never reached beyond `want`)."""
    model_type = "llama"

    def __init__(self, dims, want, first_label):
        self.dims, self.config, self.want, self.first, self.prompts = dims, dims.to_hf_config(), want, first_label, []

    def greedy1(self, seqs):
        for ids in seqs:
            self.prompts.append([int(t) for t in ids])
            if len(self.prompts) >= self.want:
                raise Done()
        return np.full(len(seqs), self.first, dtype=np.int32)


def main():
    from transformers import AutoTokenizer
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tok_llama"))
    with open(os.path.join(GOLD, "llama_setwise_query.json")) as f:
        q = json.load(f)
    with open(os.path.join(GOLD, "llama8b_full_depth.json")) as f:
        gold = json.load(f)
    dims = _synth.NAMED_DIMS[gold["dims"]]
    rt = RecordingRuntime(dims, want, q["boost_ids"][0])
    rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring="generation", method="heapsort")
    rk.batch_independent_compares = False
    ranking = [SearchResult(docid=f"d{i}", score=float(100 - i), text=t) for i, t in enumerate(q["docs"])]
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            rk.rerank(q["query"], ranking)
        except Done:
            pass
    assert len(rt.prompts) == want
    orc = LlamaOracle.__new__(LlamaOracle)
    orc.d, orc.w = dims, LazyWeights(dims, gold["seed"], int(os.environ.get("RK_THREADS", os.cpu_count() or 8)))
    boost_ids = np.asarray(q["boost_ids"], dtype=np.int64)
    out = []
    for i, ids in enumerate(rt.prompts):
        t0 = time.time()
        h = orc.hidden_states(np.asarray(ids, dtype=np.int32))[-1]
        head = np.array(orc.w["lm_head.weight"], dtype=np.float32, copy=True)
        head[boost_ids] = (head[boost_ids] * np.float32(q["boost"])).astype(np.float16).astype(np.float32)
        logits = (head @ h).astype(np.float32)
        two = np.partition(logits, -2)[-2:]
        rec = {"prompt": ids, "token": int(np.argmax(logits)), "margin": float(two[1] - two[0]),
               "label_logits": [float(x) for x in logits[boost_ids]], "logit_abs_max": float(np.abs(logits).max())}
        out.append(rec)
        print(f"[llama8b compares] prompt {i}: {len(ids)} tokens, {time.time() - t0:.0f}s, token {rec['token']}, margin {rec['margin']:.3f}, "
              f"scale {rec['logit_abs_max']:.2f}", flush=True)
        gold["query_compares"] = {"query_fixture": "llama_setwise_query.json", "boost": q["boost"], "boost_ids": q["boost_ids"],
                                  "generator": "tools/make_llama8b_compares_golden.py", "compares": out}
        with open(os.path.join(GOLD, "llama8b_full_depth.json"), "w") as f:
            json.dump(gold, f)


if __name__ == "__main__":
    main()
