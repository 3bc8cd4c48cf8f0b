#!/usr/bin/env python
"""BASELINE.json configs[2] at full size, pinned: ONE setwise heapsort query (hits=100, num_child=10, k=10, both
scorings) at flan-t5-large dimensions through the build's SetwiseLlmRanker driven by the numpy fp32 oracle (CPU, slow:
~1.2 TFLOP per compare), with every compare, its label logits and its decision margin recorded.
-> tests/golden/setwise_large.json.  The GPU test replays every recorded compare on the HIP engine (label logits within
tolerance, same decision wherever the recorded margin is above the fp16 noise floor) and the whole query end to end.
With random weights the smallest of 58 top-2 label gaps is small for any corpus, so several corpora are tried and the
one with the largest minimum margin is kept; the margins are part of the fixture.

The checkpoint is the deterministic synthetic one (llmrankers._synth, seed below) with the lm_head rows of the 23 label
tokens and EOS boosted (as tests/golden/ckpts.json's ckpt_labelboost) so that generation emits labels; nothing but the
recipe is stored.  The sort driver's parity with the REFERENCE is pinned separately (rerank_cases.json, sort_traces.json);
this fixture pins the engine's arithmetic at the real problem size.

usage: python tools/make_setwise_large_golden.py [doc_seed ...]   (tries seeds until min margin > FLOOR)
       python tools/make_setwise_large_golden.py --generation-only [doc_seed ...]
           keeps the committed likelihood run and searches a corpus of its own for the `generation` run (its margins are top-2
           gaps over the FULL vocabulary at both greedy steps, so a corpus that is decisive for likelihood need not be for
           generation); the run then carries its own doc_seed / query / docs.
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
from oracle.t5_numpy import T5Oracle                   # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
FLOOR = 0.25
WEIGHT_SEED, BOOST = 929, 6.0
WORDS = ("neural ranking model search engine index retrieval document answer question relevant topic passage language "
         "large small fast slow memory compute kernel matrix vector token score sort heap bubble list set point pair wise "
         "zero shot prompt label output input batch size length water river mountain city country history science").split()


def boosted_state(dims, label_ids):
    state = _synth.synth_state_dict(dims, seed=WEIGHT_SEED, threads=8)
    w = state["lm_head.weight"].copy()
    ids = np.asarray(label_ids + [1], dtype=np.int64)
    w[ids] = (w[ids] * np.float32(BOOST)).astype(np.float16).astype(np.float32)
    state["lm_head.weight"] = w
    return state


class MarginRuntime:
    """Oracle-backed runtime that records, per compare, the logits the decision was taken on and its margin."""

    def __init__(self, dims, state):
        self.dims, self.orc = dims, T5Oracle(dims, state)
        self.decoder_start_token_id, self.config = 0, dims.to_hf_config()
        self.records = []

    def score(self, seqs, dec_prefix, out_ids):
        out = self.orc.score_last(seqs, dec_prefix, out_ids)
        for row in out:
            top = np.sort(row)[::-1]
            self.records.append({"logits": [round(float(x), 5) for x in row], "margin": float(top[0] - top[1])})
        return out

    def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0):
        res = np.full((len(seqs), max_new), -1, dtype=np.int32)
        for b, ids in enumerate(seqs):
            enc = self.orc.encode(ids)
            cur, margins, tops = list(dec_prefix), [], []
            for t in range(max_new):
                lg = self.orc.decode(enc, cur)[-1]
                nxt = int(np.argmax(lg))
                two = np.partition(lg, -2)[-2:]
                margins.append(float(two[1] - two[0]))
                tops.append(nxt)
                res[b, t] = nxt
                cur.append(nxt)
                if nxt == eos_id:
                    break
            self.records.append({"tokens": tops, "margin": min(margins)})
        return res


def main():
    from transformers import T5Tokenizer
    tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))
    dims = _synth.FLAN_T5_LARGE
    label_ids = [tok.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1] for c in SetwiseLlmRanker.CHARACTERS]
    state = boosted_state(dims, label_ids)
    rt = MarginRuntime(dims, state)
    gen_only = "--generation-only" in sys.argv[1:]
    seeds = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [2, 3, 4, 5]

    def corpus(seed):
        rs = np.random.RandomState(seed)
        query = " ".join(rs.choice(WORDS, 10))
        return query, [[f"L{i}", " ".join(rs.choice(WORDS, int(rs.randint(110, 124))))] for i in range(100)]

    def run(seed, scoring):
        query, docs = corpus(seed)
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring=scoring, method="heapsort")
        rk.batch_independent_compares = False          # the reference's one-by-one order
        del rt.records[:]
        log = []
        orig = rk.compare

        def logged(q, d):
            out = orig(q, d)
            log.append([[x.docid for x in d], out])
            return out

        rk.compare = logged
        ranking = [SearchResult(docid=d, score=float(100 - i), text=t) for i, (d, t) in enumerate(docs)]
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            res = rk.rerank(query, ranking)
        assert len(rt.records) == len(log)
        for c, r in zip(log, rt.records):
            c.append(dict(r))
        mm = min(r["margin"] for r in rt.records)
        print(f"seed {seed} {scoring}: {len(log)} compares in {time.time() - t0:.0f}s, min margin {mm:.3f}, "
              f"margins sorted {sorted(round(r['margin'], 3) for r in rt.records)[:4]}, prompt tokens {rk.total_prompt_tokens}", flush=True)
        return {"compares": log, "result": [[r.docid, r.score] for r in res],
                "caller_list_after": [r.docid for r in ranking], "min_margin": mm,
                "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]}

    if gen_only:
        path = os.path.join(GOLD, "setwise_large.json")
        with open(path) as f:
            out = json.load(f)
        best = None
        for seed in seeds:
            r = run(seed, "generation")
            if best is None or r["min_margin"] > best[1]["min_margin"]:
                best = (seed, r)
            if r["min_margin"] > FLOOR:
                break
        seed, gen = best
        if gen["min_margin"] <= out["runs"]["generation"]["min_margin"]:
            print("no better corpus found; fixture unchanged")
            return
        query, docs = corpus(seed)
        gen.update({"doc_seed": seed, "query": query, "docs": docs})
        out["runs"]["generation"] = gen
        with open(path, "w") as f:
            json.dump(out, f)
        print("written tests/golden/setwise_large.json (generation corpus seed %d, min margin %.3f)" % (seed, gen["min_margin"]))
        return
    best = None
    for seed in seeds:                                   # likelihood first: keep the corpus with the largest minimum margin
        r = run(seed, "likelihood")
        if best is None or r["min_margin"] > best[1]["min_margin"]:
            best = (seed, r)
        if r["min_margin"] > FLOOR:
            break
    seed, lik = best
    query, docs = corpus(seed)
    out = {"doc_seed": seed, "weights": {"dims": "flan-t5-large", "seed": WEIGHT_SEED, "boost_ids": label_ids + [1], "boost": BOOST},
           "query": query, "docs": docs, "num_child": 10, "k": 10, "floor": FLOOR,
           "runs": {"likelihood": lik, "generation": run(seed, "generation")}}
    with open(os.path.join(GOLD, "setwise_large.json"), "w") as f:
        json.dump(out, f)
    print("written tests/golden/setwise_large.json (corpus seed %d)" % seed)


if __name__ == "__main__":
    main()
