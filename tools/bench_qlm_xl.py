#!/usr/bin/env python
"""BASELINE.json configs[3] on one MI355X: flan-t5-xl dimensions, pointwise qlm (ref: llmrankers/pointwise.py:41-82), hits=100,
128-token passages, a ~32-token query (= ~32 label tokens), through PointwiseLlmRanker.rerank on text.  Passages/s, the
algorithmic TFLOP/s (SURVEY 8d formula with L_d = the label count, plus the full-vocabulary head 2 d V L_d the reference
computes) as a fraction of the MFMA peak, and the engine's per-class event times of one query.  Weights: the timing-only
pool weights (llmrankers._synth.synth_tensors_pool); tokenizer: the fixture tokenizer."""
import json, os, random, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def run(model="flan-t5-xl", hits=100, batch_size=32, reps=5, shard=0):
    """shard > 0: only the first ceil(hits / shard) candidates are ranked per call - what ONE of `shard` ranks scores of a
    doc-sharded query (the gather itself is a 1-rank no-op here)."""
    import numpy as np
    import torch  # noqa  (its HIP runtime first)
    from transformers import T5Tokenizer
    import bench
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    from llmrankers._runtime import T5Runtime
    from llmrankers.pointwise import PointwiseLlmRanker
    from llmrankers.rankers import SearchResult
    dims = _synth.NAMED_DIMS[model]
    t0 = time.time()
    eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=128, max_dec_len=48).load_state(_synth.synth_tensors_pool(dims, seed=929))
    load_s = time.time() - t0
    rt = T5Runtime.from_engine(eng, dims)
    tok = T5Tokenizer.from_pretrained(os.path.join(REPO, "tests", "golden", "tok"))
    ranker = PointwiseLlmRanker.from_runtime(rt, tok, method="qlm", batch_size=batch_size)
    rs = random.Random(3)
    vocab = [tok.convert_ids_to_tokens(i).replace("▁", "") for i in range(10, 200)]
    vocab = [w for w in vocab if w.isalpha()] or ["a", "b", "c"]
    n = hits if not shard else -(-hits // shard)
    docs = [ranker.truncate(" ".join(rs.choice(vocab) for _ in range(140)), 128) for _ in range(n)]
    query = " ".join(rs.choice(vocab) for _ in range(24))
    n_labels = len(tok.encode(f"<pad> {query}", add_special_tokens=False))
    ts = []
    for _ in range(reps):
        ranking = [SearchResult(docid=str(i), score=float(n - i), text=d) for i, d in enumerate(docs)]
        t = time.perf_counter()
        res = ranker.rerank(query, ranking)
        ts.append(time.perf_counter() - t)
    ms = float(np.median(ts[2:])) * 1e3
    assert all(np.isfinite(r.score) for r in res)
    lens = [len(x) for x in tok([f"Passage: {d}\nPlease write a question based on this passage." for d in docs])["input_ids"]]
    gfl = sum(bench.algorithmic_gflop_per_passage(dims, L, n_labels) for L in lens) + n * 2.0 * dims.d_model * dims.vocab * n_labels / 1e9
    out = {"workload": f"{model} dims, pointwise qlm, {n} passages per call" + (f" (one of {shard} ranks of hits={hits})" if shard else f" (hits={hits})") +
                       f", batch_size={batch_size}, 128-token passages (prompts of {sum(lens) / len(lens):.0f} tokens), {n_labels} label tokens, "
                       "PointwiseLlmRanker.rerank on text (fixture tokenizer), full-vocabulary log-sum-exp head",
           "ms_per_query": round(ms, 2), "passages_per_s": round(n / ms * 1e3, 1), "n_labels": n_labels,
           "algorithmic_gflop_per_passage": round(gfl / n, 1), "algorithmic_tflops": round(gfl / ms, 1),
           "frac_of_mfma_peak": round(gfl / ms / bench.MFMA_PEAK_TFLOPS, 4), "weights": "pool (timing only)", "load_s": round(load_s, 1)}
    eng.profile(True); eng.profile_reset()
    ranker.rerank(query, [SearchResult(docid=str(i), score=float(n - i), text=d) for i, d in enumerate(docs)])
    eng.sync()
    rep = eng.profile_report(); eng.profile(False)
    out["classes_ms"] = {k: [round(v["ms"], 3), int(v["launches"])] for k, v in rep.items() if v["launches"]}
    eng.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run(shard=int(os.environ.get("RK_SHARD", "0")))))
