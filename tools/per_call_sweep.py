#!/usr/bin/env python
"""How many queries should run.py hand to PointwiseLlmRanker.rerank_many at once?  hits=100 candidates of ~170 tokens per prompt
(fixture tokenizer), text in, rankings out; ms per query for 1 (= rerank), 2 ... 8 queries per call.  An argument `6n` measures
six queries per call WITHOUT the streaming launch (T5Runtime.score_stream hidden: every query is tokenised before the first
launch, the behaviour up to round 4's last day) - `6 6n 6 6n` is the same-box A/B."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def main():
    import numpy as np
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
    import __graft_entry__ as ge
    ge.build()
    from transformers import T5Tokenizer
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    from llmrankers._runtime import T5Runtime
    from llmrankers.pointwise import PointwiseLlmRanker
    from llmrankers.rankers import SearchResult
    dims = _synth.FLAN_T5_LARGE
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=10 * 32 * 184, max_seqs=320, max_dec_len=4).load_state(state.items())
    del state
    rt = T5Runtime.from_engine(eng, dims)
    tok = T5Tokenizer.from_pretrained(os.path.join(REPO, "tests", "golden", "tok"))
    ranker = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=32)
    words = "neural ranking model search engine index retrieval document answer question relevant topic passage".split()
    rs = np.random.RandomState(0)
    docs = [" ".join(rs.choice(words, 126)) for _ in range(100)]
    class NoStream:                                       # the runtime without score_stream: rerank_many falls back to score_batches
        def __init__(self, inner):
            self._inner = inner

        def __getattr__(self, name):
            if name == "score_stream":
                raise AttributeError(name)
            return getattr(self._inner, name)
    ranker_ns = PointwiseLlmRanker.from_runtime(NoStream(rt), tok, method="yes_no", batch_size=32)
    for arg in sys.argv[1:] or ["1", "2", "3", "4", "5", "6", "8"]:
        n, rk = int(arg.rstrip("n")), (ranker_ns if arg.endswith("n") else ranker)
        queries = [" ".join(rs.choice(words, 30)) for _ in range(n)]
        ts = []
        for _ in range(6):
            items = [(q, [SearchResult(docid=str(i), score=float(100 - i), text=d) for i, d in enumerate(docs)]) for q in queries]
            t = time.perf_counter()
            if n == 1:
                rk.rerank(*items[0])
            else:
                rk.rerank_many(items)
            ts.append(time.perf_counter() - t)
        ms = float(np.median(ts[2:])) * 1e3 / n
        print(json.dumps({"queries_per_call": n, "streaming": not arg.endswith("n"), "ms_per_query": round(ms, 2), "passages_per_s": round(1e5 / ms, 1)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
