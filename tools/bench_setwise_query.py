#!/usr/bin/env python
"""One setwise heapsort query end to end (BASELINE.json configs[2] shape: hits=100, num_child=10, k=10, flan-t5-large
dims, ~128-token passages, synthetic weights and the test tokenizer): SetwiseLlmRanker.rerank() wall time with the
build phase one compare at a time (the reference's order) and with the level-batched build phase."""
import contextlib
import io
import json
import os
import random
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
from llmrankers import _synth                      # noqa: E402
from llmrankers._engine import RkEngine            # noqa: E402
from llmrankers.rankers import SearchResult        # noqa: E402
from llmrankers.setwise import SetwiseLlmRanker    # noqa: E402


class EngineRuntime:
    """T5Runtime's interface on an engine with synthetic weights (no checkpoint directory needed)."""
    def __init__(self, eng, dims):
        self.engine, self.config = eng, dims.to_hf_config()

    def score(self, seqs, dec_prefix, out_ids):
        return self.engine.score(seqs, dec_prefix, out_ids)

    def score_async(self, seqs, dec_prefix, out_ids, slot):      # T5Runtime.score_async (capacity: this tool sizes the engine itself)
        self.engine.stage(seqs, slot=slot)
        self.engine.score_staged(dec_prefix, out_ids, slot=slot)
        return slot

    def score_collect(self, handle):
        return self.engine.read_scores(handle)

    supports_greedy_candidates = True

    def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0, candidates=None):
        toks, steps = self.engine.greedy(seqs, dec_prefix, max_new, eos_id, pad_id,
                                         candidates if os.environ.get("RK_GREEDY_SPEC", "1") == "1" else None)
        toks = toks.copy()
        toks[:, steps:] = -1
        return toks


def run(state=None, reps=3, many=None, one_by_one=True, words=None, query_words=None):
    """-> dict of timings.  `state`: the synthetic flan-t5-large state dict if the caller already has it (bench.py).
    `words` per passage before truncate(128) (RK_WORDS, default 60 -> ~84-token passages, ~0.9k-token prompts with the fixture
    tokenizer; 140 -> every passage is cut to the full 128 tokens of run.py's default --passage_length, prompts of ~1.5k
    tokens = SURVEY 8d S3); `query_words` (RK_QUERY_WORDS, default 7; ~22 words = the 32-token query of S3)."""
    from transformers import T5Tokenizer
    tok = T5Tokenizer.from_pretrained(os.path.join(REPO, "tests", "golden", "tok"))
    dims = _synth.FLAN_T5_LARGE
    nq_cap = int(os.environ.get("RK_MANY", "4")) if many is None else many
    words = int(os.environ.get("RK_WORDS", "60")) if words is None else words
    query_words = int(os.environ.get("RK_QUERY_WORDS", "7")) if query_words is None else query_words
    # (this tool's runtime wrapper does not chunk: the build phase of NQ queries in lockstep is 9 NQ prompts of ~900 tokens)
    eng = RkEngine(dims, 0, max_tokens=max(32768, (20000 if words > 80 else 12000) * nq_cap), max_seqs=max(256, 10 * nq_cap), max_dec_len=8)
    # random weights would generate arbitrary tokens ("Unexpected output" on every compare): like the goldens
    # (tests/golden/setwise_large.json) the lm_head rows of the passage labels a prompt can hold (A .. K at num_child = 10)
    # and EOS are scaled x6, so that a generation is "<label> </s>" as with a trained checkpoint - the case the product
    # path is built for
    if state is None:
        state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    state = dict(state)
    label_ids = [tok.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1] for c in SetwiseLlmRanker.CHARACTERS[:11]]
    head = state["lm_head.weight"].copy()
    ids = np.asarray(sorted(set(label_ids + [tok.eos_token_id])), dtype=np.int64)
    head[ids] = (head[ids] * np.float32(6.0)).astype(np.float16).astype(np.float32)
    state["lm_head.weight"] = head
    eng.load_state(state.items())
    del state, head
    eng.set_option("dec_graph", int(os.environ.get("RK_DEC_GRAPH", "1")))
    for kv in os.environ.get("RK_OPTS", "").split(","):
        if kv:
            eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    rt = EngineRuntime(eng, dims)
    import bench
    rs = random.Random(3)
    vocab = [tok.convert_ids_to_tokens(i).replace("\u2581", "") for i in range(10, 200)]
    vocab = [w for w in vocab if w.isalpha()] or ["a", "b", "c"]
    docs = [(f"d{i}", float(100 - i), " ".join(rs.choice(vocab) for _ in range(words))) for i in range(100)]
    query0 = "which passage mentions the most relevant words" if query_words == 7 else " ".join(rs.choice(vocab) for _ in range(query_words))
    out = {}
    for scoring in ("likelihood", "generation"):
        for batched in ((False, True) if one_by_one else (True,)):
            rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring=scoring, method="heapsort")
            rk.batch_independent_compares = batched
            best, res0 = None, None
            for rep in range(reps):
                ranking = [SearchResult(docid=d, score=s, text=rk.truncate(t, 128)) for d, s, t in docs]
                rk.total_compare = rk.total_prompt_tokens = rk.total_completion_tokens = 0
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    res = rk.rerank(query0, ranking)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
                res0 = [r.docid for r in res][:10]
            avg_len = rk.total_prompt_tokens / max(rk.total_compare, 1)
            tf = rk.total_compare * bench.algorithmic_gflop_per_passage(dims, avg_len, 2) / 1e3 / best     # SURVEY 8d formula, L_d = 2
            out[f"{scoring}_{'batched' if batched else 'one_by_one'}"] = {
                "ms_per_query": round(best * 1e3, 1), "compares": rk.total_compare,
                "algorithmic_tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / 2500.0, 4),
                "avg_prompt_tokens": round(rk.total_prompt_tokens / max(rk.total_compare, 1), 1), "top10": res0}
    # several queries at once (SetwiseLlmRanker.rerank_many / run.py --queries_per_call): the dependency chains of NQ queries
    # advance in lockstep, their pending compares share an engine call; identical rankings, amortised time per query
    NQ = int(os.environ.get("RK_MANY", "4")) if many is None else many
    qtexts = [query0] + [" ".join(rs.choice(vocab) for _ in range(query_words)) for _ in range(NQ - 1)]
    variants = [("likelihood", False), ("likelihood", True), ("generation", True)] if NQ > 1 else []
    for scoring, alternate in variants:
        # (likelihood: the chains as two groups alternating over the engine's slots - the default - and as one group)
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring=scoring, method="heapsort")
        rk.alternate_groups = alternate
        best, res0, n_tok = None, None, 0
        for rep in range(reps):
            items = [(q, [SearchResult(docid=d, score=s, text=rk.truncate(t, 128)) for d, s, t in docs]) for q in qtexts]
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                res, counters = rk.rerank_many(items)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            res0 = [r.docid for r in res[0]][:10]
        assert res0 == out[f"{scoring}_batched"]["top10"], "rerank_many changed the ranking of the first query"
        n_cmp = sum(c[0] for c in counters)
        avg_len = sum(c[1] for c in counters) / max(n_cmp, 1)
        tf = n_cmp * bench.algorithmic_gflop_per_passage(dims, avg_len, 2) / 1e3 / best
        out[f"{scoring}_many{NQ}" + ("" if alternate else "_one_group")] = {"ms_per_query": round(best * 1e3 / NQ, 1), "queries_per_call": NQ, "compares": n_cmp,
                                      "algorithmic_tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / 2500.0, 4)}
    if os.environ.get("RK_HOSTPROF"):                       # where the host time of one query goes (stderr)
        import cProfile, pstats
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring="generation", method="heapsort")
        ranking = [SearchResult(docid=d, score=s, text=rk.truncate(t, 128)) for d, s, t in docs]
        pr = cProfile.Profile(); pr.enable()
        with contextlib.redirect_stdout(io.StringIO()):
            rk.rerank("which passage mentions the most relevant words", ranking)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
    if one_by_one:
        for scoring in ("likelihood", "generation"):
            assert out[f"{scoring}_batched"]["top10"] == out[f"{scoring}_one_by_one"]["top10"], "batched build phase changed the ranking"
    eng.close()
    return out


def main():
    print(json.dumps(run()))


if __name__ == "__main__":
    main()
