#!/usr/bin/env python
"""Generate tests/golden/ — runs ONLY in the build container (needs /root/reference + transformers).

Imports the real reference (ielab/llm-rankers @ /root/reference, read-only, never copied) against the
installed HuggingFace transformers, on synthetic checkpoints + a synthetic tokenizer, and records
inputs / activations / logits / rankings / counters as small data fixtures.  The reference ships no
tests or golden vectors, so these are the parity pins for oracle/ and for the HIP
engine.  Nothing Python from the reference travels to the GPU box; only the data written here does.

Shims needed to import the reference under transformers 5.x:
  * `openai`, `tiktoken` stub modules (only used by the OpenAI ranker classes),
  * `T5Tokenizer.batch_encode_plus` (removed in 5.x; used at ref: llmrankers/setwise.py:55).

Usage:  python tools/make_goldens.py            (rewrites tests/golden/ except setwise_large.json, which
                                                 tools/make_setwise_large_golden.py makes; then run tools/annotate_margins.py)
        python tools/make_goldens.py --only-monot5 | --only-pairwise | --only-llama   (add one fixture family in place)
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import random
import shutil
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "llm-rankers_amd"))
sys.path.insert(0, REPO)

from llmrankers import _synth  # noqa: E402  (the build's own generator; not the reference package)

REF = "/root/reference"

WORDS = (
    "neural ranking model search engine index retrieval document answer question relevant topic "
    "passage language large small fast slow memory compute kernel matrix vector token score sort heap "
    "bubble list set point pair wise zero shot prompt label output input batch size length water river "
    "mountain city country history science physics chemistry biology medicine doctor patient virus "
    "vaccine covid treatment symptom economy market price trade bank money law court judge music art "
    "film book author").split()
PROMPT_WORDS = (
    "Passage: Query: Does the passage answer query? Answer 'Yes' or 'No' Given a query which of "
    "following passages is most relevant one to Output only label Please write question based on this "
    "Document: Relevant: Passage Yes No").split()
LABELS = [chr(ord("A") + i) for i in range(23)]


def build_vocab():
    pieces = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), ("▁", -3.0)]
    seen = {p for p, _ in pieces}

    def add(p, s):
        if p not in seen:
            seen.add(p)
            pieces.append((p, s))

    for w in PROMPT_WORDS:
        add("▁" + w, -2.0)
    for c in LABELS:
        add("▁" + c, -2.5)
    for w in WORDS:
        add("▁" + w, -2.0)
    for c in "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789:?'\",.-":
        add(c, -7.0)
    assert len(pieces) <= 256, len(pieces)
    return pieces


def make_tokenizer(path):
    from transformers import T5Tokenizer
    tok = T5Tokenizer(vocab=build_vocab(), extra_ids=0)
    tok.save_pretrained(path)
    return T5Tokenizer.from_pretrained(path)


def write_ckpt(path, spec, tok_dir):
    """spec = the regeneration recipe committed to tests/golden/ckpts.json (weights themselves are not committed)."""
    _synth.write_checkpoint(path, spec, tok_dir)
    return _synth.checkpoint_sha256(path)


def rand_text(rs, lo, hi):
    return " ".join(rs.choice(WORDS, size=int(rs.randint(lo, hi + 1))))


def import_reference():
    for m in ("openai", "tiktoken"):
        sys.modules.setdefault(m, types.ModuleType(m))
    from transformers import T5Tokenizer
    if not hasattr(T5Tokenizer, "batch_encode_plus"):
        T5Tokenizer.batch_encode_plus = lambda self, texts, **kw: self(texts, **kw)
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        # NB: `llmrankers` would resolve to the build's own package first on sys.path, so load the
        # reference's package under its real name from its own directory explicitly.
        for k in [k for k in sys.modules if k == "llmrankers" or k.startswith("llmrankers.")]:
            del sys.modules[k]
        sys.path.remove(os.path.join(REPO, "llm-rankers_amd"))
        import llmrankers.rankers as ref_rankers
        import llmrankers.pointwise as ref_pointwise
        import llmrankers.setwise as ref_setwise
    assert ref_rankers.__file__.startswith(REF), ref_rankers.__file__
    return ref_rankers, ref_pointwise, ref_setwise


def hf_model_goldens(ckpt_dir, name, tok):
    """Padded HF batch forward with hooks -> activations + logits for oracle pinning."""
    import torch
    from transformers import T5ForConditionalGeneration
    model = T5ForConditionalGeneration.from_pretrained(ckpt_dir, torch_dtype=torch.float32).eval()
    cfg = model.config
    rs = np.random.RandomState(4242)
    lens = [7, 24, 13, 1, 19]
    seqs = [rs.randint(3, cfg.vocab_size, size=n).astype(np.int64) for n in lens]
    for s in seqs:
        s[-1] = 1
    L = max(lens)
    ids = np.zeros((len(seqs), L), dtype=np.int64)
    mask = np.zeros((len(seqs), L), dtype=np.int64)
    for b, s in enumerate(seqs):
        ids[b, :len(s)] = s
        mask[b, :len(s)] = 1
    passage_id = tok.encode("<pad> Passage", add_special_tokens=False)
    out = {"lens": np.array(lens), "input_ids": ids, "attention_mask": mask}
    caps = {}

    def hook(key):
        def fn(mod, args, output):
            caps[key] = (output[0] if isinstance(output, tuple) else output).detach().numpy().copy()
        return fn

    hs = []
    for i, blk in enumerate(model.encoder.block):
        hs.append(blk.layer[0].register_forward_hook(hook(f"enc.{i}.attn")))
        hs.append(blk.register_forward_hook(hook(f"enc.{i}.ffn")))
    hs.append(model.encoder.embed_tokens.register_forward_hook(hook("enc.embed")))
    hs.append(model.encoder.final_layer_norm.register_forward_hook(hook("enc.final")))
    for i, blk in enumerate(model.decoder.block):
        hs.append(blk.layer[0].register_forward_hook(hook(f"dec.{i}.self")))
        hs.append(blk.layer[1].register_forward_hook(hook(f"dec.{i}.cross")))
        hs.append(blk.register_forward_hook(hook(f"dec.{i}.ffn")))
    hs.append(model.decoder.final_layer_norm.register_forward_hook(hook("dec.final_norm")))

    for tag, dec in (("d1", [0]), ("d2", passage_id), ("d5", [0, 9, 17, 4, 30])):
        caps.clear()
        dec_ids = torch.tensor([dec] * len(seqs))
        with torch.no_grad():
            o = model(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask), decoder_input_ids=dec_ids)
        out[f"{tag}.dec_ids"] = np.array(dec)
        out[f"{tag}.logits"] = o.logits.numpy()
        if tag == "d2":      # per-sublayer activations for one decoder shape only (keeps the fixture small)
            for k, v in caps.items():
                out[f"{tag}.{k}"] = v
    for h in hs:
        h.remove()
    # qlm style labels path (HF shifts right internally)
    labels = np.array([0, 12, 7, 99, 45, 3], dtype=np.int64)
    with torch.no_grad():
        o = model(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask),
                  labels=torch.tensor(labels)[None].repeat(len(seqs), 1))
    out["qlm.labels"] = labels
    out["qlm.logits"] = o.logits.numpy()
    # greedy generation (batched, prefix [0, Passage])
    with torch.no_grad():
        g = model.generate(torch.tensor(ids), attention_mask=torch.tensor(mask),
                           decoder_input_ids=torch.tensor([passage_id] * len(seqs)), max_new_tokens=2,
                           do_sample=False)
    out["gen.prefix"] = np.array(passage_id)
    out["gen.output_ids"] = g.numpy()
    singles = []
    for s in seqs:   # the reference's own call shape: one sequence, no mask (ref: setwise.py:93-95)
        with torch.no_grad():
            g1 = model.generate(torch.tensor(s)[None], decoder_input_ids=torch.tensor([passage_id]),
                                max_new_tokens=2)
        singles.append(g1[0].numpy().tolist())
    out["gen.single_json"] = np.frombuffer(json.dumps(singles).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, f"model_{name}.npz"), **out)
    print(f"[model_{name}] logits d1 range {out['d1.logits'].min():.3f}..{out['d1.logits'].max():.3f}")


def bucket_goldens():
    from transformers.models.t5.modeling_t5 import T5Attention
    import torch
    rel = torch.arange(-1024, 1025)
    out = {"rel": rel.numpy()}
    for bidir in (True, False):
        for nb, md in ((32, 128), (32, 64), (16, 128)):
            out[f"b{int(bidir)}_{nb}_{md}"] = T5Attention._relative_position_bucket(
                rel, bidirectional=bidir, num_buckets=nb, max_distance=md).numpy()
    np.savez_compressed(os.path.join(GOLD, "rel_buckets.npz"), **out)


def sort_trace_goldens(ref_rankers, ref_setwise):
    """Reference sort drivers with a deterministic fake comparator (SURVEY.md section 3.2 probe recipe)."""
    traces = []
    for (n, c, k) in ((100, 10, 10), (100, 3, 10), (100, 2, 10), (20, 3, 5), (7, 3, 10), (1, 3, 5), (24, 22, 3)):
        for method in ("heapsort", "bubblesort"):
            for mode in ("truth", "garbage"):
                rk = ref_setwise.SetwiseLlmRanker.__new__(ref_setwise.SetwiseLlmRanker)
                rk.num_child, rk.k, rk.method, rk.num_permutation = c, k, method, 1
                rs = np.random.RandomState(n * 131 + c * 17 + k)
                rel = rs.permutation(n).tolist()           # hidden relevance of doc i
                calls = []

                def fake_compare(query, docs, _rel=rel, _calls=calls, _mode=mode):
                    rk.total_compare += 1
                    idx = [int(d.docid[1:]) for d in docs]
                    _calls.append(idx)
                    if not idx:          # k > n in bubblesort: the reference really calls compare([])
                        return "A"
                    best = max(range(len(docs)), key=lambda j: _rel[idx[j]])
                    if _mode == "garbage" and len(_calls) % 3 == 0:
                        return ["?", "Z", "zz"][len(_calls) % 9 // 3]   # ValueError -> 0 fallback paths
                    if _mode == "garbage" and len(_calls) % 7 == 0 and rk.method == "heapsort":
                        return "W"     # IndexError -> keep parent (bubblesort has no such guard: it raises)
                    return ref_setwise.SetwiseLlmRanker.CHARACTERS[best]

                rk.compare = fake_compare
                ranking = [ref_rankers.SearchResult(docid=f"d{i}", score=float(n - i), text=f"t{i}") for i in range(n)]
                with contextlib.redirect_stdout(io.StringIO()):
                    res = rk.rerank("q", ranking)
                traces.append({"n": n, "c": c, "k": k, "method": method, "mode": mode, "rel": rel,
                               "calls": calls, "total_compare": rk.total_compare,
                               "result": [[r.docid, r.score] for r in res],
                               "caller_list_after": [r.docid for r in ranking]})
    with open(os.path.join(GOLD, "sort_traces.json"), "w") as f:
        json.dump(traces, f)
    print(f"[sort_traces] {len(traces)} traces")


def rerank_goldens(ref_rankers, ref_pointwise, ref_setwise, ckpts):
    rs = np.random.RandomState(77)
    cases = []
    queries = [rand_text(rs, 3, 8) for _ in range(3)]
    doc_pool = [rand_text(rs, 8, 40) for _ in range(40)]

    def make_ranking(n, off):
        return [ref_rankers.SearchResult(docid=f"D{off + i}", score=float(100 - i), text=doc_pool[(off + i) % 40])
                for i in range(n)]

    sink = io.StringIO()
    # pointwise (ref: pointwise.py:36-130); batch sizes chosen so the last batch is ragged
    for ck in ("ckpt_gated_untied", "ckpt_relu_tied"):
        for method, bs, n in (("yes_no", 4, 20), ("yes_no", 32, 13), ("qlm", 4, 10), ("qlm", 3, 7)):
            with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
                rk = ref_pointwise.PointwiseLlmRanker(ckpts[ck], ckpts[ck], device="cpu", method=method, batch_size=bs)
                for qi, q in enumerate(queries[:2]):
                    ranking = make_ranking(n, 5 * qi)
                    inp = [[r.docid, r.score, r.text] for r in ranking]
                    res = rk.rerank(q, ranking)
                    cases.append({"kind": "pointwise", "ckpt": ck, "method": method, "batch_size": bs, "query": q,
                                  "input": inp, "result": [[r.docid, r.score] for r in res],
                                  "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]})
    # monoT5 (ref: pointwise.py:136-186): its fixed token ids 6136/1176 exceed the toy vocabulary, so it has its own
    # checkpoint and fixture file (monot5_goldens below)

    # setwise (ref: setwise.py:79-313)
    for ck, scoring, method, c, k, nperm, n in (
        ("ckpt_gated_untied", "likelihood", "heapsort", 3, 5, 1, 20),
        ("ckpt_gated_untied", "likelihood", "bubblesort", 3, 5, 1, 20),
        ("ckpt_gated_untied", "generation", "heapsort", 3, 5, 1, 20),
        ("ckpt_labelboost", "generation", "heapsort", 3, 5, 1, 20),
        ("ckpt_labelboost", "generation", "bubblesort", 4, 4, 1, 14),
        ("ckpt_labelboost", "likelihood", "bubblesort", 4, 4, 1, 14),
        ("ckpt_labelboost", "generation", "heapsort", 2, 3, 3, 12),
        ("ckpt_labelboost", "likelihood", "heapsort", 10, 10, 1, 30),
        ("ckpt_relu_tied", "likelihood", "heapsort", 3, 5, 1, 16),
    ):
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            rk = ref_setwise.SetwiseLlmRanker(ckpts[ck], ckpts[ck], device="cpu", num_child=c, k=k, scoring=scoring,
                                              method=method, num_permutation=nperm)
            compare_log = []
            orig_compare = rk.compare

            def logged(query, docs, _o=orig_compare, _l=compare_log):
                out = _o(query, docs)
                _l.append([[d.docid for d in docs], out])
                return out

            rk.compare = logged
            for qi, q in enumerate(queries[:2]):
                ranking = make_ranking(n, 3 * qi)
                inp = [[r.docid, r.score, r.text] for r in ranking]
                random.seed(929)
                del compare_log[:]
                raises = None
                try:
                    res = rk.rerank(q, ranking)
                except IndexError:     # the reference's bubblesort has no guard for an out-of-window label
                    raises, res = "IndexError", []
                cases.append({"kind": "setwise", "ckpt": ck, "scoring": scoring, "method": method, "num_child": c, "k": k,
                              "num_permutation": nperm, "query": q, "input": inp, "raises": raises,
                              "result": [[r.docid, r.score] for r in res], "compares": list(compare_log),
                              "caller_list_after": [r.docid for r in ranking],
                              "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]})
    # truncate() (ref: pointwise.py:132-133)
    from transformers import T5Tokenizer
    tok = T5Tokenizer.from_pretrained(ckpts["ckpt_gated_untied"])
    trunc = []
    for t in doc_pool[:6] + ["Hello World-42, unknown é chars?", ""]:
        for n in (0, 1, 5, 128):
            trunc.append([t, n, tok.convert_tokens_to_string(tok.tokenize(t)[:n])])
    with open(os.path.join(GOLD, "rerank_cases.json"), "w") as f:
        json.dump({"cases": cases, "truncate": trunc}, f)
    n_unexp = sink.getvalue().count("Unexpected output")
    print(f"[rerank_cases] {len(cases)} cases; reference printed 'Unexpected output' {n_unexp}x")


def monot5_goldens(ref_rankers, ref_pointwise, ckpt_dir):
    """The reference's MonoT5LlmRanker (ref: pointwise.py:136-186) on a relu / tied-head checkpoint whose vocabulary
    covers the hard-coded 'false' / 'true' ids 6136 / 1176: prompt, decoder_start_token_id input, false/true ordering,
    counters, stable sort.  -> tests/golden/monot5_cases.json"""
    rs = np.random.RandomState(91)
    queries = [rand_text(rs, 3, 8) for _ in range(2)]
    doc_pool = [rand_text(rs, 8, 40) for _ in range(30)]
    cases = []
    sink = io.StringIO()
    for bs, n in ((4, 13), (32, 9), (1, 3)):
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            rk = ref_pointwise.MonoT5LlmRanker(ckpt_dir, ckpt_dir, device="cpu", method="yes_no", batch_size=bs)
            for qi, q in enumerate(queries):
                ranking = [ref_rankers.SearchResult(docid=f"M{7 * qi + i}", score=float(50 - i), text=doc_pool[(7 * qi + i) % 30])
                           for i in range(n)]
                inp = [[r.docid, r.score, r.text] for r in ranking]
                res = rk.rerank(q, ranking)
                cases.append({"kind": "monot5", "ckpt": "ckpt_monot5", "batch_size": bs, "query": q, "input": inp,
                              "result": [[r.docid, r.score] for r in res],
                              "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]})
    with open(os.path.join(GOLD, "monot5_cases.json"), "w") as f:
        json.dump({"cases": cases}, f)
    sc = [s for c in cases for _, s in c["result"]]
    print(f"[monot5_cases] {len(cases)} cases, P(true) range {min(sc):.4f}..{max(sc):.4f}")


def pairwise_goldens(ref_rankers, ref_pairwise, ckpts):
    """The reference's PairwiseLlmRanker (ref: pairwise.py:29-295) — allpair / heapsort / bubblesort on the label-boosted
    checkpoint (generations are 'Passage A' / 'Passage B') and on a plain one (mostly other tokens: the conflict and
    'not a win' paths).  -> tests/golden/pairwise_cases.json"""
    rs = np.random.RandomState(55)
    queries = [rand_text(rs, 3, 8) for _ in range(2)]
    doc_pool = [rand_text(rs, 8, 30) for _ in range(20)]
    cases, sink = [], io.StringIO()
    for ck, method, bs, k, n in (("ckpt_abboost", "allpair", 4, 3, 6), ("ckpt_abboost", "allpair", 3, 10, 5),
                                 ("ckpt_abboost", "heapsort", 2, 3, 9), ("ckpt_abboost", "bubblesort", 2, 3, 8),
                                 ("ckpt_gated_untied", "allpair", 4, 2, 5), ("ckpt_gated_untied", "heapsort", 2, 2, 6),
                                 ("ckpt_gated_untied", "bubblesort", 2, 4, 6)):
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            rk = ref_pairwise.PairwiseLlmRanker(ckpts[ck], ckpts[ck], device="cpu", method=method, batch_size=bs, k=k)
            log = []
            orig = rk.compare

            def logged(query, docs, _o=orig, _l=log):
                out = _o(query, docs)
                _l.append([list(docs), list(out)])
                return out

            rk.compare = logged
            for qi, q in enumerate(queries):
                ranking = [ref_rankers.SearchResult(docid=f"P{4 * qi + i}", score=float(30 - i), text=doc_pool[(4 * qi + i) % 20])
                           for i in range(n)]
                inp = [[r.docid, r.score, r.text] for r in ranking]
                del log[:]
                res = rk.rerank(q, ranking)
                cases.append({"kind": "pairwise", "ckpt": ck, "method": method, "batch_size": bs, "k": k, "query": q,
                              "input": inp, "result": [[r.docid, r.score] for r in res], "compares": list(log),
                              "caller_list_after": [r.docid for r in ranking],
                              "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]})
    with open(os.path.join(GOLD, "pairwise_cases.json"), "w") as f:
        json.dump({"cases": cases}, f)
    outs = [o for c in cases for _, out in c["compares"] for o in out]
    print(f"[pairwise_cases] {len(cases)} cases, {len(outs)} logged generations, distinct: {sorted(set(outs))[:8]}")


def add_pairwise():
    """Incremental: pairwise fixtures on the existing checkpoint recipes, without regenerating the other fixtures."""
    import tempfile
    tok_dir = os.path.join(GOLD, "tok")
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        specs = json.load(f)
    tmp = tempfile.mkdtemp(prefix="rk_goldens_")
    ckpts = {}
    if "ckpt_abboost" not in specs:        # only 'A', 'B' and EOS boosted: generations are 'Passage A' / 'Passage B'
        from transformers import T5Tokenizer
        tok = T5Tokenizer.from_pretrained(tok_dir)
        ab = [tok.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1] for c in "AB"]
        specs["ckpt_abboost"] = {"dims": "toy-gated-untied", "seed": 31, "gain": 2.0, "boost_ids": ab, "boost": 8.0, "boost2_ids": [1], "boost2": 5.0}
        specs["ckpt_abboost"]["sha256"] = write_ckpt(os.path.join(tmp, "ckpt_abboost"), specs["ckpt_abboost"], tok_dir)
        with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
            json.dump(specs, f, indent=1)
    for name in ("ckpt_abboost", "ckpt_gated_untied"):
        ckpts[name] = os.path.join(tmp, name)
        assert write_ckpt(ckpts[name], specs[name], tok_dir) == specs[name]["sha256"]
    ref_rankers, _, _ = import_reference()
    import llmrankers.pairwise as ref_pairwise
    assert ref_pairwise.__file__.startswith(REF)
    pairwise_goldens(ref_rankers, ref_pairwise, ckpts)
    shutil.rmtree(tmp)


LLAMA_CHAT_TEMPLATE = ("{{ bos_token }}{% for message in messages %}{{ '<|start_header_id|>' + message['role'] + '<|end_header_id|>\n\n' "
                       "+ message['content'] | trim + '<|eot_id|>' }}{% endfor %}"
                       "{% if add_generation_prompt %}{{ '<|start_header_id|>assistant<|end_header_id|>\n\n' }}{% endif %}")


def make_llama_tokenizer(path):
    """A synthetic Llama-3-style tokenizer (no real vocabulary exists offline): Unigram pieces of the T5 fixture plus the
    chat special tokens, BOS prepended by the post-processor, Llama-3's chat template."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    specials = ["<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>", "<unk>"]
    pieces = [(t, 0.0) for t in specials] + [p for p in build_vocab() if p[0] not in ("<pad>", "</s>", "<unk>")]
    assert len(pieces) <= 256, len(pieces)
    tk = Tokenizer(models.Unigram(pieces, unk_id=5, byte_fallback=False))
    tk.pre_tokenizer = pre_tokenizers.Metaspace(replacement="\u2581", prepend_scheme="always")
    tk.decoder = decoders.Metaspace(replacement="\u2581", prepend_scheme="always")
    tk.post_processor = processors.TemplateProcessing(single="<|begin_of_text|> $A", special_tokens=[("<|begin_of_text|>", 0)])
    tk.add_special_tokens(specials)
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<|begin_of_text|>", eos_token="<|end_of_text|>", unk_token="<unk>",
                                   additional_special_tokens=["<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"])
    fast.chat_template = LLAMA_CHAT_TEMPLATE
    os.makedirs(path, exist_ok=True)
    fast.save_pretrained(path)
    return fast


def llama_goldens(ref_rankers, ref_setwise, ckpt_dir):
    """HF LlamaForCausalLM logits (oracle pin) and the reference's SetwiseLlmRanker on a Llama checkpoint
    (ref: setwise.py:60-69, 159-177): chat-template prompt + " Passage:", one greedy token, counters.
    -> tests/golden/model_llama.npz, tests/golden/llama_cases.json"""
    import torch
    from transformers import AutoModelForCausalLM, AutoTokenizer
    model = AutoModelForCausalLM.from_pretrained(ckpt_dir, torch_dtype=torch.float32).eval()
    tok = AutoTokenizer.from_pretrained(ckpt_dir)
    rs = np.random.RandomState(808)
    lens = [9, 33, 1, 140, 64]
    seqs = [rs.randint(6, model.config.vocab_size, size=n).astype(np.int64) for n in lens]
    out = {"lens": np.array(lens), "tokens": np.concatenate(seqs)}
    last = []
    with torch.no_grad():
        for i, s_ in enumerate(seqs):
            lg = model(input_ids=torch.tensor(s_)[None]).logits[0].numpy()
            last.append(lg[-1])
            if i == 1:
                out["full_logits_seq1"] = lg
    out["last_logits"] = np.stack(last)
    np.savez_compressed(os.path.join(GOLD, "model_llama.npz"), **out)
    print(f"[model_llama] last-position logits range {out['last_logits'].min():.3f}..{out['last_logits'].max():.3f}")

    rs = np.random.RandomState(99)
    queries = [rand_text(rs, 3, 8) for _ in range(2)]
    doc_pool = [rand_text(rs, 8, 40) for _ in range(40)]
    cases, sink = [], io.StringIO()
    for scoring, method, c, k, nperm, n in (("generation", "heapsort", 3, 5, 1, 20), ("generation", "bubblesort", 4, 4, 1, 14),
                                            ("generation", "heapsort", 10, 10, 1, 30), ("generation", "heapsort", 2, 3, 3, 12),
                                            ("likelihood", "heapsort", 3, 5, 1, 8)):
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            rk = ref_setwise.SetwiseLlmRanker(ckpt_dir, ckpt_dir, device="cpu", num_child=c, k=k, scoring=scoring, method=method,
                                              num_permutation=nperm)
            log, orig = [], rk.compare

            def logged(query, docs, _o=orig, _l=log):
                out_ = _o(query, docs)
                _l.append([[d.docid for d in docs], out_])
                return out_

            rk.compare = logged
            for qi, q in enumerate(queries):
                ranking = [ref_rankers.SearchResult(docid=f"Q{3 * qi + i}", score=float(100 - i), text=doc_pool[(3 * qi + i) % 40]) for i in range(n)]
                inp = [[r.docid, r.score, r.text] for r in ranking]
                random.seed(929)
                del log[:]
                raises = None
                try:
                    res = rk.rerank(q, ranking)
                except NotImplementedError:
                    raises, res = "NotImplementedError", []
                except IndexError:
                    raises, res = "IndexError", []
                cases.append({"kind": "setwise-llama", "ckpt": "ckpt_llama", "scoring": scoring, "method": method, "num_child": c, "k": k,
                              "num_permutation": nperm, "query": q, "input": inp, "raises": raises,
                              "result": [[r.docid, r.score] for r in res], "compares": list(log),
                              "caller_list_after": [r.docid for r in ranking],
                              "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]})
    prompt = tok.apply_chat_template([{"role": "user", "content": "hello world"}], tokenize=False, add_generation_prompt=True) + " Passage:"
    with open(os.path.join(GOLD, "llama_cases.json"), "w") as f:
        json.dump({"cases": cases, "prompt_probe": {"text": prompt, "ids": tok(prompt)["input_ids"]}}, f)
    outs = sorted({o for c_ in cases for _, o in c_["compares"]})
    print(f"[llama_cases] {len(cases)} cases, outputs seen: {outs[:12]}; reference printed 'Unexpected output' {sink.getvalue().count('Unexpected output')}x")


def add_llama_pairwise():
    """Incremental: the reference's PairwiseLlmRanker on the Llama fixture checkpoint (ref: pairwise.py:60-77, 104-129: chat
    template + " Passage:", one greedy token per ordering) - heapsort and bubblesort queries with every compare logged, and
    `allpair` (T5-only in the reference: AttributeError) -> tests/golden/llama_pairwise_cases.json"""
    import tempfile
    tok_dir = os.path.join(GOLD, "tok_llama")
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        specs = json.load(f)
    tmp = tempfile.mkdtemp(prefix="rk_goldens_")
    ck = os.path.join(tmp, "ckpt_llama")
    assert write_ckpt(ck, specs["ckpt_llama"], tok_dir) == specs["ckpt_llama"]["sha256"]
    ref_rankers, _, _ = import_reference()
    import llmrankers.pairwise as ref_pairwise
    assert ref_pairwise.__file__.startswith(REF)
    rs = np.random.RandomState(77)
    queries = [rand_text(rs, 3, 8) for _ in range(2)]
    doc_pool = [rand_text(rs, 8, 30) for _ in range(24)]
    cases, sink = [], io.StringIO()
    for method, k, n in (("heapsort", 4, 9), ("bubblesort", 3, 7), ("allpair", 3, 4)):
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            rk = ref_pairwise.PairwiseLlmRanker(ck, ck, device="cpu", method=method, batch_size=2, k=k)
            log, orig = [], rk.compare

            def logged(query, docs, _o=orig, _l=log):
                out_ = _o(query, docs)
                _l.append([list(docs), out_])
                return out_

            rk.compare = logged
            for qi, q in enumerate(queries):
                ranking = [ref_rankers.SearchResult(docid=f"P{5 * qi + i}", score=float(50 - i), text=doc_pool[(5 * qi + i) % 24]) for i in range(n)]
                inp = [[r.docid, r.score, r.text] for r in ranking]
                del log[:]
                raises = None
                try:
                    res = rk.rerank(q, ranking)
                except AttributeError:
                    raises, res = "AttributeError", []
                cases.append({"kind": "pairwise-llama", "ckpt": "ckpt_llama", "method": method, "k": k, "query": q, "input": inp,
                              "raises": raises, "result": [[r.docid, r.score] for r in res], "compares": list(log),
                              "counters": [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]})
    with open(os.path.join(GOLD, "llama_pairwise_cases.json"), "w") as f:
        json.dump({"cases": cases}, f)
    outs = sorted({o for c_ in cases for _, out in c_["compares"] for o in out})
    print(f"[llama_pairwise_cases] {len(cases)} cases, {sum(len(c_['compares']) for c_ in cases)} compares, outputs seen: {outs[:10]}")
    shutil.rmtree(tmp)


def add_llama3rope():
    """Incremental: HF LlamaForCausalLM logits of the toy checkpoint with rope type "llama3" (Llama-3.1 / 3.2's scaling,
    hf: modeling_rope_utils.py _compute_llama3_parameters) -> tests/golden/model_llama3rope.npz.  Pins oracle/llama_numpy.py's
    rope_tables(scaling=...) and, through it, the engine's rk_llama_set_rope_scaling."""
    import tempfile
    import torch
    from transformers import AutoModelForCausalLM
    spec = {"dims": "toy-llama3rope", "seed": 23, "gain": 2.0}
    tmp = tempfile.mkdtemp(prefix="rk_gold_")
    ck = os.path.join(tmp, "ckpt_llama3rope")
    _synth.write_checkpoint(ck, spec)
    model = AutoModelForCausalLM.from_pretrained(ck, torch_dtype=torch.float32).eval()
    rp = getattr(model.config, "rope_parameters", None) or getattr(model.config, "rope_scaling", None)
    assert rp and rp.get("rope_type") == "llama3", rp
    inv = model.model.rotary_emb.inv_freq.numpy()
    rs = np.random.RandomState(909)
    lens = [9, 40, 1, 150, 64]
    seqs = [rs.randint(6, model.config.vocab_size, size=n).astype(np.int64) for n in lens]
    out = {"lens": np.array(lens), "tokens": np.concatenate(seqs), "inv_freq": inv, "seed": np.array(spec["seed"]), "gain": np.array(spec["gain"])}
    last = []
    with torch.no_grad():
        for i, s_ in enumerate(seqs):
            lg = model(input_ids=torch.tensor(s_)[None]).logits[0].numpy()
            last.append(lg[-1])
            if i == 3:
                out["full_logits_seq3"] = lg
    out["last_logits"] = np.stack(last)
    np.savez_compressed(os.path.join(GOLD, "model_llama3rope.npz"), **out)
    print(f"[model_llama3rope] inv_freq[:4]={inv[:4]}, [-4:]={inv[-4:]}; last-position logits range {out['last_logits'].min():.3f}..{out['last_logits'].max():.3f}")


def add_llama():
    """Incremental: Llama tokenizer, checkpoint recipe, HF logits and reference setwise cases."""
    import tempfile
    tok_dir = os.path.join(GOLD, "tok_llama")
    tok = make_llama_tokenizer(tok_dir)
    label_ids = [tok.encode(" Passage: " + c, add_special_tokens=False)[-1] for c in LABELS]
    assert len(set(label_ids)) == 23, label_ids
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        specs = json.load(f)
    spec = {"dims": "toy-llama", "seed": 21, "gain": 2.0, "boost_ids": label_ids, "boost": 6.0, "tokenizer": "tok_llama"}
    tmp = tempfile.mkdtemp(prefix="rk_goldens_")
    ck = os.path.join(tmp, "ckpt_llama")
    spec["sha256"] = write_ckpt(ck, spec, tok_dir)
    specs["ckpt_llama"] = spec
    with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
        json.dump(specs, f, indent=1)
    ref_rankers, _, ref_setwise = import_reference()
    llama_goldens(ref_rankers, ref_setwise, ck)
    shutil.rmtree(tmp)


def add_monot5():
    """Incremental: adds the monoT5 checkpoint recipe and cases without regenerating the other fixtures."""
    import tempfile
    tok_dir = os.path.join(GOLD, "tok")
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        specs = json.load(f)
    spec = {"dims": "toy-monot5", "seed": 14, "gain": 1.0}
    tmp = tempfile.mkdtemp(prefix="rk_goldens_")
    ck = os.path.join(tmp, "ckpt_monot5")
    spec["sha256"] = write_ckpt(ck, spec, tok_dir)
    specs["ckpt_monot5"] = spec
    with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
        json.dump(specs, f, indent=1)
    ref_rankers, ref_pointwise, _ = import_reference()
    monot5_goldens(ref_rankers, ref_pointwise, ck)
    shutil.rmtree(tmp)


def config1_golden(ref_rankers, ref_pointwise, tok_dir):
    """BASELINE.json configs[0]: flan-t5-small shape, pointwise yes_no, hits=20, batch_size=4, CPU HF reference.
    Weights come from the counter generator (regenerable on the GPU box); prompts are pre-tokenised ids."""
    import torch
    from transformers import T5ForConditionalGeneration, T5Config
    dims = _synth.FLAN_T5_SMALL
    cfg = T5Config(**{k: v for k, v in dims.to_hf_config().items() if k not in ("architectures", "model_type")})
    model = T5ForConditionalGeneration(cfg).eval()
    sd = _synth.synth_state_dict(dims, seed=929)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    tsd["encoder.embed_tokens.weight"] = tsd["shared.weight"]
    tsd["decoder.embed_tokens.weight"] = tsd["shared.weight"]
    model.lm_head.weight = torch.nn.Parameter(tsd["lm_head.weight"].clone())     # untie like flan (hf 5.x ties by default)
    missing = model.load_state_dict(tsd, strict=False)
    assert not missing.unexpected_keys, missing
    model.config.scale_decoder_outputs = False
    assert model.lm_head.weight.data_ptr() != model.shared.weight.data_ptr()
    seqs = _synth.synth_token_batch(20, 60, 184, dims.vocab, seed=930)
    yes_id, no_id = 2163, 465       # real flan-t5 ids quoted from memory; any two ids work for parity
    logits = np.zeros((20, 2), dtype=np.float32)
    full0 = None
    for s0 in range(0, 20, 4):                     # batch_size 4, right padded like DataCollatorWithPadding
        chunk = seqs[s0:s0 + 4]
        L = max(len(s) for s in chunk)
        ids = np.zeros((len(chunk), L), dtype=np.int64)
        mask = np.zeros_like(ids)
        for b, s in enumerate(chunk):
            ids[b, :len(s)] = s
            mask[b, :len(s)] = 1
        with torch.no_grad():
            lg = model(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask),
                       decoder_input_ids=torch.zeros((len(chunk), 1), dtype=torch.long)).logits[:, 0]
        logits[s0:s0 + len(chunk), 0] = lg[:, yes_id].numpy()
        logits[s0:s0 + len(chunk), 1] = lg[:, no_id].numpy()
        if full0 is None:
            full0 = lg[0].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "config1_flan_t5_small.npz"),
                        lens=np.array([len(s) for s in seqs]), tokens=np.concatenate(seqs),
                        yes_no_ids=np.array([yes_id, no_id]), logits=logits, full_logits_seq0=full0,
                        seed=np.array(929), token_seed=np.array(930))
    print(f"[config1] yes/no logit diff range {np.ptp(logits[:, 0] - logits[:, 1]):.3f}")


def main():
    if "--only-monot5" in sys.argv:
        return add_monot5()
    if "--only-pairwise" in sys.argv:
        return add_pairwise()
    if "--only-llama-pairwise" in sys.argv:
        return add_llama_pairwise()
    if "--only-llama3rope" in sys.argv:
        return add_llama3rope()
    if "--only-llama" in sys.argv:
        return add_llama()
    keep = {}                                  # fixtures made by other tools survive a full regeneration
    for fn in ("setwise_large.json",):         # tools/make_setwise_large_golden.py (30 CPU-minutes)
        if os.path.exists(os.path.join(GOLD, fn)):
            with open(os.path.join(GOLD, fn), "rb") as f:
                keep[fn] = f.read()
    if os.path.isdir(GOLD):
        shutil.rmtree(GOLD)
    os.makedirs(GOLD)
    for fn, data in keep.items():
        with open(os.path.join(GOLD, fn), "wb") as f:
            f.write(data)
    tok_dir = os.path.join(GOLD, "tok")
    tok = make_tokenizer(tok_dir)
    label_ids = [tok.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1] for c in LABELS]
    assert len(set(label_ids)) == 23 and all(i > 3 for i in label_ids), label_ids
    import tempfile
    tmp = tempfile.mkdtemp(prefix="rk_goldens_")
    specs = {
        "ckpt_gated_untied": {"dims": "toy-gated-untied", "seed": 11, "gain": 1.0},
        "ckpt_relu_tied": {"dims": "toy-relu-tied", "seed": 12, "gain": 1.0},
        "ckpt_labelboost": {"dims": "toy-gated-untied", "seed": 13, "gain": 1.0,
                            "boost_ids": label_ids + [1], "boost": 6.0},
    }
    ckpts = {}
    for name, spec in specs.items():
        ckpts[name] = os.path.join(tmp, name)
        spec["sha256"] = write_ckpt(ckpts[name], spec, tok_dir)
    with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
        json.dump(specs, f, indent=1)
    bucket_goldens()
    hf_model_goldens(ckpts["ckpt_gated_untied"], "gated_untied", tok)
    hf_model_goldens(ckpts["ckpt_relu_tied"], "relu_tied", tok)
    ref_rankers, ref_pointwise, ref_setwise = import_reference()
    sort_trace_goldens(ref_rankers, ref_setwise)
    rerank_goldens(ref_rankers, ref_pointwise, ref_setwise, ckpts)
    config1_golden(ref_rankers, ref_pointwise, tok_dir)
    spec = {"dims": "toy-monot5", "seed": 14, "gain": 1.0}
    ck = os.path.join(tmp, "ckpt_monot5")
    spec["sha256"] = write_ckpt(ck, spec, tok_dir)
    specs["ckpt_monot5"] = spec
    with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
        json.dump(specs, f, indent=1)
    monot5_goldens(ref_rankers, ref_pointwise, ck)
    import llmrankers.pairwise as ref_pairwise
    ab = [tok.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1] for c in "AB"]
    spec = {"dims": "toy-gated-untied", "seed": 31, "gain": 2.0, "boost_ids": ab, "boost": 8.0, "boost2_ids": [1], "boost2": 5.0}
    ckpts["ckpt_abboost"] = os.path.join(tmp, "ckpt_abboost")
    spec["sha256"] = write_ckpt(ckpts["ckpt_abboost"], spec, tok_dir)
    specs["ckpt_abboost"] = spec
    with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
        json.dump(specs, f, indent=1)
    pairwise_goldens(ref_rankers, ref_pairwise, ckpts)
    # Llama family: own tokenizer, checkpoint recipe, HF logits, reference setwise cases
    ltok_dir = os.path.join(GOLD, "tok_llama")
    ltok = make_llama_tokenizer(ltok_dir)
    llabels = [ltok.encode(" Passage: " + c, add_special_tokens=False)[-1] for c in LABELS]
    assert len(set(llabels)) == 23, llabels
    spec = {"dims": "toy-llama", "seed": 21, "gain": 2.0, "boost_ids": llabels, "boost": 6.0, "tokenizer": "tok_llama"}
    ck = os.path.join(tmp, "ckpt_llama")
    spec["sha256"] = write_ckpt(ck, spec, ltok_dir)
    specs["ckpt_llama"] = spec
    with open(os.path.join(GOLD, "ckpts.json"), "w") as f:
        json.dump(specs, f, indent=1)
    llama_goldens(ref_rankers, ref_setwise, ck)
    with open(os.path.join(GOLD, "PROVENANCE.json"), "w") as f:
        import transformers, torch
        json.dump({"generator": "tools/make_goldens.py", "reference": "ielab/llm-rankers @ /root/reference (2025-07-18)",
                   "transformers": transformers.__version__, "torch": torch.__version__,
                   "numpy": np.__version__}, f, indent=1)
    shutil.rmtree(tmp)
    total = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(GOLD) for fn in fns)
    print(f"tests/golden: {total / 1e6:.2f} MB  (now run tools/annotate_margins.py to add the decision margins)")


if __name__ == "__main__":
    main()
