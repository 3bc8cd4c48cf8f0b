#!/usr/bin/env python
"""End-to-end ranker timing at BASELINE.json configs[1]: PointwiseLlmRanker.rerank on ONE query with hits=100,
batch_size=32 (batches 32,32,32,4), flan-t5-large dimensions with synthetic weights and the fixture tokenizer.
Includes everything the reference's timed loop includes (ref: run.py:183-196): prompt building, tokenisation,
batching, engine calls (pipelined over the batch slots), score extraction and the sort."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth                              # noqa: E402
from llmrankers._engine import RkEngine                    # noqa: E402
from llmrankers._runtime import T5Runtime                  # noqa: E402
from llmrankers.pointwise import PointwiseLlmRanker        # noqa: E402
from llmrankers.rankers import SearchResult                # noqa: E402


def main():
    from transformers import T5Tokenizer
    tok = T5Tokenizer.from_pretrained(os.path.join(REPO, "tests", "golden", "tok"))
    dims = _synth.FLAN_T5_LARGE
    rt = T5Runtime.__new__(T5Runtime)
    rt.dims, rt.config, rt.model_type, rt.decoder_start_token_id = dims, dims.to_hf_config(), "t5", 0
    rt.max_tokens, rt.max_seqs = 8192, 32
    rt.engine = RkEngine(dims, 0, max_tokens=8192, max_seqs=32, max_dec_len=4)
    rt.engine.load_state(_synth.synth_tensors(dims, seed=929, threads=min(32, os.cpu_count() or 8)))
    ranker = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=32)
    words = "neural ranking model search engine index retrieval document answer question relevant topic passage".split()
    rs = np.random.RandomState(0)
    query = " ".join(rs.choice(words, 30))
    docs = [" ".join(rs.choice(words, 126)) for _ in range(100)]
    lens = []
    times = []
    for it in range(8):
        ranking = [SearchResult(docid=str(i), score=float(100 - i), text=d) for i, d in enumerate(docs)]
        t = time.perf_counter()
        res = ranker.rerank(query, ranking)
        times.append(time.perf_counter() - t)
    t = time.perf_counter()
    prompts = [f"Passage: {d}\nQuery: {query}\nDoes the passage answer the query? Answer 'Yes' or 'No'" for d in docs]
    ids = tok(prompts)["input_ids"]
    t_tok = time.perf_counter() - t
    ms = float(np.median(times[2:])) * 1e3
    print(json.dumps({"ms_per_query": round(ms, 2), "passages_per_s": round(100 / ms * 1e3, 1), "tokenize_ms": round(t_tok * 1e3, 2),
                      "prompt_tokens_avg": float(np.mean([len(x) for x in ids])), "compares": ranker.total_compare,
                      "note": "hits=100, batch_size=32, one query at a time, host tokenisation included"}))


if __name__ == "__main__":
    main()
