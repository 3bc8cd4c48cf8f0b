"""Pin the oracle (oracle/t5_numpy.py) to the HuggingFace/reference goldens (CPU, no GPU).

Goldens come from tools/make_goldens.py: HF transformers 5.15.0 fp32 forward of padded batches — the
arithmetic the reference executes at ref: llmrankers/pointwise.py:117-119 / setwise.py:93-95,184."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, load_state
from oracle.t5_numpy import T5Oracle, relative_position_bucket

TOL = 1e-4   # fp32 vs fp32: different summation order, ragged vs padded; logits reach |12|


def test_rel_bucket_tables():
    g = np.load(os.path.join(GOLD, "rel_buckets.npz"))
    rel = g["rel"]
    for key in g.files:
        if key == "rel":
            continue
        b, nb, md = key[1:].split("_")
        got = relative_position_bucket(rel, bidirectional=bool(int(b)), num_buckets=int(nb), max_distance=int(md))
        np.testing.assert_array_equal(got, g[key], err_msg=key)


@pytest.mark.parametrize("name", ["gated_untied", "relu_tied"])
def test_forward_matches_hf(name, ckpt_dirs):
    dims, state = load_state(ckpt_dirs["ckpt_" + name])
    g = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    orc = T5Oracle(dims, state)
    lens = g["lens"]
    seqs = [g["input_ids"][b, :n] for b, n in enumerate(lens)]
    for tag in ("d1", "d2", "d5"):
        dec = g[f"{tag}.dec_ids"].tolist()
        for b, ids in enumerate(seqs):
            orc.capture = {}
            logits = orc.decode(orc.encode(ids), dec)
            np.testing.assert_allclose(logits, g[f"{tag}.logits"][b], atol=TOL, rtol=TOL)
            if tag == "d2":
                n = len(ids)
                for k, v in orc.capture.items():
                    gk = f"d2.{k}"
                    if gk not in g.files:
                        continue
                    ref = g[gk][b]
                    ref = ref[:n] if k.startswith("enc.") else ref
                    np.testing.assert_allclose(v, ref, atol=TOL, rtol=TOL, err_msg=f"{k} seq {b}")
        orc.capture = None


@pytest.mark.parametrize("name", ["gated_untied", "relu_tied"])
def test_qlm_and_greedy_match_hf(name, ckpt_dirs):
    dims, state = load_state(ckpt_dirs["ckpt_" + name])
    g = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    orc = T5Oracle(dims, state)
    seqs = [g["input_ids"][b, :n] for b, n in enumerate(g["lens"])]
    labels = g["qlm.labels"]
    lg = g["qlm.logits"].astype(np.float64)
    m = lg.max(-1, keepdims=True)
    lse = (m + np.log(np.exp(lg - m).sum(-1, keepdims=True)))[..., 0]
    want = -(lse - np.take_along_axis(lg, labels[None, :, None].repeat(len(seqs), 0), -1)[..., 0]).sum(-1)
    np.testing.assert_allclose(orc.qlm(seqs, labels), want, atol=1e-4, rtol=1e-5)
    prefix = g["gen.prefix"].tolist()
    got = orc.greedy(seqs, prefix, max_new=2)
    gen = g["gen.output_ids"]
    assert gen.shape[1] <= len(prefix) + 2
    want_new = np.zeros_like(got)
    want_new[:, :gen.shape[1] - len(prefix)] = gen[:, len(prefix):]
    np.testing.assert_array_equal(got, want_new)
    singles = json.loads(bytes(g["gen.single_json"]).decode())
    for b, s in enumerate(singles):       # the reference's own B=1 call shape stops at EOS
        new = s[len(prefix):]
        assert got[b, :len(new)].tolist() == new


def test_config1_flan_t5_small_logits():
    """BASELINE.json configs[0] plumbing case: flan-t5-small shape, 20 passages, HF fp32 CPU logits."""
    from llmrankers import _synth
    g = np.load(os.path.join(GOLD, "config1_flan_t5_small.npz"))
    dims = _synth.FLAN_T5_SMALL
    orc = T5Oracle(dims, _synth.synth_state_dict(dims, seed=int(g["seed"])))
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    seqs = [g["tokens"][off[i]:off[i + 1]] for i in range(len(g["lens"]))]
    regen = _synth.synth_token_batch(20, 60, 184, dims.vocab, seed=int(g["token_seed"]))
    assert all(np.array_equal(a, b) for a, b in zip(seqs, regen))
    sub = [0, 7, 19]                              # 3 of 20 keeps the CPU suite quick; GPU test does all 20
    got = orc.score_last([seqs[i] for i in sub], [0], g["yes_no_ids"])
    np.testing.assert_allclose(got, g["logits"][sub], atol=5e-5, rtol=1e-5)
    full = orc.score_last([seqs[0]], [0])[0]
    np.testing.assert_allclose(full, g["full_logits_seq0"], atol=5e-5, rtol=1e-5)


def test_llama_oracle_matches_hf_llama_logits():
    """oracle/llama_numpy.py vs HF LlamaForCausalLM (tools/make_goldens.py --only-llama): last-position logits of five
    ragged prompts and the full logits of one, toy GQA / RoPE / SwiGLU checkpoint with hot (gain 2) weights."""
    import json
    from llmrankers import _synth
    from oracle.llama_numpy import LlamaOracle
    g = np.load(os.path.join(GOLD, "model_llama.npz"))
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        spec = json.load(f)["ckpt_llama"]
    dims = _synth.NAMED_DIMS[spec["dims"]]
    state = _synth.synth_state_dict(dims, seed=spec["seed"], gain=spec["gain"])
    w = state["lm_head.weight"].copy()
    ids = np.asarray(spec["boost_ids"], dtype=np.int64)
    w[ids] = (w[ids] * np.float32(spec["boost"])).astype(np.float16).astype(np.float32)
    state["lm_head.weight"] = w
    orc = LlamaOracle(dims, state)
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    seqs = [g["tokens"][off[i]:off[i + 1]] for i in range(len(g["lens"]))]
    got = orc.last_logits(seqs)
    assert np.abs(got - g["last_logits"]).max() < 2e-4 * max(1.0, np.abs(g["last_logits"]).max())
    head = orc.w["lm_head.weight"]
    full = orc.hidden_states(seqs[1]) @ head.T
    assert np.abs(full - g["full_logits_seq1"]).max() < 2e-4 * max(1.0, np.abs(g["full_logits_seq1"]).max())
    np.testing.assert_array_equal(orc.greedy1(seqs), np.argmax(g["last_logits"], axis=-1))


def test_llama3_rope_scaling_oracle_matches_hf():
    """rope type "llama3" (Llama-3.1 / 3.2 checkpoints the reference loads through AutoModelForCausalLM, ref: setwise.py:65-69):
    oracle/llama_numpy.py's scaled inverse frequencies equal HF's rotary_emb.inv_freq, and its logits equal HF
    LlamaForCausalLM's on a toy checkpoint whose frequencies fall in all three bands (tools/make_goldens.py
    --only-llama3rope); the config round trip keeps the scaling."""
    from llmrankers import _synth
    from oracle.llama_numpy import LlamaOracle, rope_inv_freq
    g = np.load(os.path.join(GOLD, "model_llama3rope.npz"))
    dims = _synth.NAMED_DIMS["toy-llama3rope"]
    assert _synth.LlamaDims.from_hf_config(dims.to_hf_config()).rope_scaling == dims.rope_scaling
    assert _synth.LlamaDims.from_hf_config(_synth.TOY_LLAMA.to_hf_config()).rope_scaling is None
    np.testing.assert_allclose(rope_inv_freq(dims.head_dim, dims.rope_theta, dims.rope_scaling), g["inv_freq"], rtol=2e-6)
    assert not np.allclose(g["inv_freq"], rope_inv_freq(dims.head_dim, dims.rope_theta), rtol=1e-3)    # the scaling is not a no-op
    state = _synth.synth_state_dict(dims, seed=int(g["seed"]), gain=float(g["gain"]))
    orc = LlamaOracle(dims, state)
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    seqs = [g["tokens"][off[i]:off[i + 1]] for i in range(len(g["lens"]))]
    got = orc.last_logits(seqs)
    assert np.abs(got - g["last_logits"]).max() < 2e-4 * max(1.0, np.abs(g["last_logits"]).max())
    full = orc.hidden_states(seqs[3]) @ orc.w["lm_head.weight"].T
    assert np.abs(full - g["full_logits_seq3"]).max() < 2e-4 * max(1.0, np.abs(g["full_logits_seq3"]).max())
    # the default rope type on the same weights gives different logits: the test would notice a scaling that is ignored
    plain = LlamaOracle(_synth.LlamaDims(**{**dims.__dict__, "rope_scaling": None}), state).last_logits(seqs)
    assert np.abs(plain - g["last_logits"]).max() > 50 * 2e-4



def test_hf_path_equals_reference_goldens(ckpt_dirs):
    """The timed CPU baseline of bench.py (oracle/hf_path.py: build_hf_model + pointwise_yes_no, the restatement of
    ref: llmrankers/pointwise.py:84-127 on pre-tokenised prompts) gives the scores the REAL reference recorded on the fixture
    checkpoints (tests/golden/rerank_cases.json, tools/make_goldens.py imports /root/reference): every yes_no case, both
    checkpoints (gated / untied head and relu / tied + scaled head), the reference's own batches (batch_size 4 and 32 - padded
    to the batch's longest prompt like DataCollatorWithPadding does) - scores to 1e-6, order and counters exactly."""
    from transformers import T5Tokenizer
    from llmrankers.pointwise import PointwiseLlmRanker
    from llmrankers.rankers import SearchResult
    from oracle import hf_path

    class HfPathRuntime:                       # PointwiseLlmRanker's runtime interface over the baseline port, batch by batch
        model_type = "t5"
        decoder_start_token_id = 0

        def __init__(self, dims, state):
            self.model, self.config, self.calls = hf_path.build_hf_model(dims, state), dims.to_hf_config(), 0

        def score(self, seqs, dec_prefix, out_ids):
            assert list(dec_prefix) == [0] and len(out_ids) == 2
            self.calls += 1
            return hf_path.pointwise_yes_no(self.model, [list(s) for s in seqs], len(seqs), out_ids[0], out_ids[1])

    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["kind"] == "pointwise" and c["method"] == "yes_no"]
    assert len(cases) >= 8 and {c["ckpt"] for c in cases} == {"ckpt_gated_untied", "ckpt_relu_tied"}
    rts = {}
    for case in cases:
        ck = ckpt_dirs[case["ckpt"]]
        if case["ckpt"] not in rts:
            rts[case["ckpt"]] = (HfPathRuntime(*load_state(ck)), T5Tokenizer.from_pretrained(ck))
        rt, tok = rts[case["ckpt"]]
        ranker = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=case["batch_size"])
        before = rt.calls
        res = ranker.rerank(case["query"], [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]])
        assert [r.docid for r in res] == [d for d, _ in case["result"]]
        np.testing.assert_allclose([r.score for r in res], [s for _, s in case["result"]], atol=1e-6, rtol=0)
        assert [ranker.total_compare, ranker.total_prompt_tokens, ranker.total_completion_tokens] == case["counters"]
        assert rt.calls - before == case["counters"][0]            # one HF forward per reference batch
