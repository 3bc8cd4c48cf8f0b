#!/usr/bin/env python
"""BASELINE.json configs[4] call shape through the ranker, pinned: ONE setwise heapsort query (hits=100, num_child=10, k=10,
scoring generation) with Llama-3-8B WIDTHS (hidden 4096, 32 query / 8 kv heads x 128, SwiGLU 14336, vocabulary 128256, rope
theta 5e5) and TWO layers - the depth the build container's host affords for ~50 compares of ~700-token prompts - through the
build's SetwiseLlmRanker (Llama branch: chat template + " Passage:", prefill, ONE greedy token; ref:
llmrankers/setwise.py:60-69, 159-177) driven by the numpy fp32 oracle, with every compare's decision and the margin of its
greedy token over the runner-up recorded -> tests/golden/llama_setwise_query.json.  The GPU test replays the whole query on the
HIP engine (same ranking, compare sequence and counters: the smallest recorded margin must exceed the fp16 noise floor) and
then runs the SAME query at the full 32 layers for determinism, batch independence and counter consistency (no oracle at that
depth: 32 GB of fp32 weights, ten host minutes per prompt).  The lm_head rows of the 23 label tokens are boosted (x BOOST, as
tests/golden/ckpts.json's ckpt_llama) so that generations are labels, as with a trained checkpoint; nothing but the recipe and
the texts are stored.  Several corpora are tried, the one with the largest minimum margin is kept.

usage: python tools/make_llama_setwise_golden.py [doc_seed ...]
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO, os.path.join(REPO, "tests")]
from llmrankers import _synth                          # noqa: E402
from llmrankers.rankers import SearchResult            # noqa: E402
from llmrankers.setwise import SetwiseLlmRanker        # noqa: E402
from oracle.llama_numpy import LlamaOracle             # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED, BOOST, N_LAYERS = 929, 6.0, 2
WORDS = ("neural ranking model search engine index retrieval document answer question relevant topic passage language "
         "large small fast slow memory compute kernel matrix vector token score sort heap bubble list set point pair wise "
         "zero shot prompt label output input batch size length water river mountain city country history science").split()


def dims_8b(n_layers):
    return _synth.LlamaDims(vocab=128256, hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, n_layers=n_layers,
                            bos_token_id=128000, eos_token_id=128001)


def boost_rows(arr, ids, boost):
    w = np.array(arr, dtype=np.float32, copy=True)
    ids = np.asarray(ids, dtype=np.int64)
    w[ids] = (w[ids] * np.float32(boost)).astype(np.float16).astype(np.float32)
    return w


class MarginLlamaRuntime:
    model_type = "llama"

    def __init__(self, dims, state):
        self.dims, self.orc, self.config = dims, LlamaOracle(dims, state), dims.to_hf_config()
        self.records = []

    def greedy1(self, seqs):
        out = []
        for ids in seqs:
            lg = self.orc.last_logits([ids])[0]
            two = np.partition(lg, -2)[-2:]
            tok = int(np.argmax(lg))
            self.records.append({"token": tok, "margin": float(two[1] - two[0]), "prompt_tokens": len(ids),
                                 "logit_abs_max": float(np.abs(lg).max())})
            out.append(tok)
        return np.asarray(out, dtype=np.int32)


def main():
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tok_llama"))
    label_ids = [tok.encode(c, add_special_tokens=False)[0] for c in SetwiseLlmRanker.CHARACTERS]
    dims = dims_8b(N_LAYERS)
    t0 = time.time()
    state = _synth.synth_state_dict(dims, seed=WEIGHT_SEED, threads=8)
    state["lm_head.weight"] = boost_rows(state["lm_head.weight"], label_ids, BOOST)
    print(f"weights in {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    seeds = [int(a) for a in sys.argv[1:]] or [7101, 7102, 7103]
    best = None
    for doc_seed in seeds:
        rs = np.random.RandomState(doc_seed)
        query = " ".join(rs.choice(WORDS, 8))
        docs = [" ".join(rs.choice(WORDS, int(rs.randint(18, 30)))) for _ in range(100)]
        rt = MarginLlamaRuntime(dims, state)
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring="generation", method="heapsort")
        rk.batch_independent_compares = False                  # the reference's one-by-one order: the recorded sequence
        ranking = [SearchResult(docid=f"d{i}", score=float(100 - i), text=t) for i, t in enumerate(docs)]
        t0 = time.time()
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            res = rk.rerank(query, ranking)
        unexpected = sink.getvalue().count("Unexpected output")
        mm = min(r["margin"] for r in rt.records)
        print(f"seed {doc_seed}: {len(rt.records)} compares in {time.time() - t0:.0f}s, min margin {mm:.3f}, unexpected {unexpected}, "
              f"prompt tokens {min(r['prompt_tokens'] for r in rt.records)}..{max(r['prompt_tokens'] for r in rt.records)}", file=sys.stderr, flush=True)
        cand = {"what": "Llama-3-8B widths, 2 layers, setwise heapsort hits=100 num_child=10 k=10 generation, numpy fp32 oracle through "
                        "SetwiseLlmRanker (tools/make_llama_setwise_golden.py)",
                "weight_seed": WEIGHT_SEED, "boost": BOOST, "boost_ids": label_ids, "n_layers": N_LAYERS, "doc_seed": doc_seed,
                "query": query, "docs": docs, "compares": rt.records, "min_margin": mm, "unexpected_outputs": unexpected,
                "ranking": [[r.docid, r.score] for r in res], "caller_list_after": [d.docid for d in ranking],
                "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]}
        if best is None or (unexpected, -mm) < (best["unexpected_outputs"], -best["min_margin"]):
            best = cand
    with open(os.path.join(GOLD, "llama_setwise_query.json"), "w") as f:
        json.dump(best, f)
    print(json.dumps({"doc_seed": best["doc_seed"], "compares": len(best["compares"]), "min_margin": best["min_margin"],
                      "unexpected": best["unexpected_outputs"], "counters": best["counters"]}))


if __name__ == "__main__":
    main()
