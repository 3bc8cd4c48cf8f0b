#!/usr/bin/env python
"""BASELINE.json configs[3] at full size, pinned: ONE pointwise qlm query (hits=100, flan-t5-xl dimensions: d_model 2048, 32
heads, d_ff 5120, 24+24 layers) through the build's PointwiseLlmRanker driven by the numpy fp32 oracle on the CPU (about
40 TFLOP: minutes), with every passage's score, the final ranking and the three counters recorded
-> tests/golden/xl_qlm_query.json.  The GPU test (tests/test_gpu_rerank.py) runs the same query through
PointwiseLlmRanker(shard_candidates=True) on the HIP engine with a one-rank RCCL communicator (the candidate-sharding path of
ref: llmrankers/pointwise.py:41-82 replaced; shares that need several engine calls) and compares scores,
order and counters.  The checkpoint is the deterministic synthetic one (llmrankers._synth, seed below): nothing but the recipe
and the texts are stored.  The sort / counter logic's parity with the REFERENCE is pinned separately (rerank_cases.json).

usage: python tools/make_xl_qlm_golden.py [n_docs]      (default 100)
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO, os.path.join(REPO, "tests")]
from llmrankers import _synth                          # noqa: E402
from llmrankers.pointwise import PointwiseLlmRanker    # noqa: E402
from llmrankers.rankers import SearchResult            # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED, DOC_SEED, BATCH = 929, 4404, 32
WORDS = ("neural ranking model search engine index retrieval document answer question relevant topic passage language "
         "large small fast slow memory compute kernel matrix vector token score sort heap bubble list set point pair wise "
         "zero shot prompt label output input batch size length water river mountain city country history science").split()


def main():
    from transformers import T5Tokenizer
    from _stub import OracleRuntime
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dims = _synth.FLAN_T5_XL
    tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))
    rs = np.random.RandomState(DOC_SEED)
    query = " ".join(rs.choice(WORDS, 14))
    docs = [" ".join(rs.choice(WORDS, int(rs.randint(60, 100)))) for _ in range(n_docs)]
    t0 = time.time()
    state = _synth.synth_state_dict(dims, seed=WEIGHT_SEED, threads=8)
    print(f"weights in {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    rk = PointwiseLlmRanker.from_runtime(OracleRuntime(dims, state), tok, method="qlm", batch_size=BATCH)
    ranking = [SearchResult(docid=f"d{i}", score=float(n_docs - i), text=t) for i, t in enumerate(docs)]
    t0 = time.time()
    res = rk.rerank(query, ranking)
    print(f"oracle query in {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    by_doc = {r.docid: r.score for r in res}
    out = {"what": "flan-t5-xl dims, pointwise qlm, one query, numpy fp32 oracle through PointwiseLlmRanker (tools/make_xl_qlm_golden.py)",
           "dims": "flan-t5-xl", "weight_seed": WEIGHT_SEED, "doc_seed": DOC_SEED, "batch_size": BATCH, "method": "qlm",
           "query": query, "docs": docs,
           "scores": [by_doc[f"d{i}"] for i in range(n_docs)], "ranking": [r.docid for r in res],
           "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens],
           "label_tokens": len(tok.encode(f"<pad> {query}", add_special_tokens=False))}
    path = os.path.join(GOLD, "xl_qlm_query.json" if n_docs == 100 else f"xl_qlm_query_{n_docs}.json")
    with open(path, "w") as f:
        json.dump(out, f)
    sc = np.sort(np.asarray(out["scores"]))
    print(json.dumps({"path": path, "counters": out["counters"], "label_tokens": out["label_tokens"],
                      "score_range": [float(sc[0]), float(sc[-1])], "min_adjacent_gap": float(np.diff(sc).min())}))


if __name__ == "__main__":
    main()
