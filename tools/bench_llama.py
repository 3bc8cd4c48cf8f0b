#!/usr/bin/env python
"""BASELINE.json configs[4] call shape on one MI355X: Llama-3-8B dimensions (synthetic weights), setwise `generation`
compares = prefill of one ~1.5k-token prompt + the arg-max of the last position over the 128k vocabulary
(ref: llmrankers/setwise.py:159-177).  ms per compare for 1 and 4 prompts per call and the algorithmic TFLOP/s
(2 x parameters x tokens + causal attention).  RK_POOL_WEIGHTS=1 (bench.py's leg): the timing-only pool weights of
llmrankers._synth.synth_tensors_pool (seconds instead of two minutes of host time)."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]


def run(layers=32, L=1536, pool=False, iters=5, profile=True):
    import torch  # noqa  (its HIP runtime first)
    from llmrankers import _synth
    from llmrankers._engine import RkLlamaEngine
    d = _synth.LLAMA_3_8B
    dims = _synth.LlamaDims(vocab=d.vocab, hidden=d.hidden, n_heads=d.n_heads, n_kv_heads=d.n_kv_heads, head_dim=d.head_dim,
                            intermediate=d.intermediate, n_layers=layers, bos_token_id=d.bos_token_id, eos_token_id=d.eos_token_id)
    t0 = time.time()
    gen = _synth.synth_tensors_pool(dims, seed=929) if pool else _synth.synth_tensors(dims, seed=929, threads=min(48, os.cpu_count() or 8))
    eng = RkLlamaEngine(dims, 0, max_tokens=8192, max_seqs=8).load_state(gen)
    load_s = time.time() - t0
    print(f"[bench_llama] {layers} layers generated + loaded in {load_s:.0f}s", file=sys.stderr)
    q, kv, h, f = dims.n_heads * 128, dims.n_kv_heads * 128, dims.hidden, dims.intermediate
    flop_tok = 2.0 * layers * (h * (q + 2 * kv) + q * h + 3 * h * f)
    out = {"workload": f"Llama-3-8B dims, {layers} layers, setwise generation compare = prefill of one {L}-token prompt + full-vocabulary "
                       "arg-max of the last position (rk_llama_greedy1); B1 = one compare per call, B4 = four prompts per call",
           "layers": layers, "L": L, "weights": "pool (timing only)" if pool else "synth_tensors", "load_s": round(load_s, 1)}
    for B in (1, 4):
        seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=3)
        for _ in range(2):
            eng.greedy1(seqs)
        t = time.perf_counter()
        for _ in range(iters):
            eng.greedy1(seqs)
        ms = (time.perf_counter() - t) / iters * 1e3
        flop = B * (L * flop_tok + layers * 2.0 * L * L * q) + B * 2.0 * h * dims.vocab
        out[f"B{B}"] = {"ms_per_call": round(ms, 2), "ms_per_compare": round(ms / B, 2), "algorithmic_tflops": round(flop / ms / 1e9, 1),
                        "frac_of_mfma_peak": round(flop / ms / 1e9 / 2500.0, 4)}
    if profile:
        eng.profile(True); eng.profile_reset()
        eng.greedy1(_synth.synth_token_batch(1, L, L, dims.vocab, seed=3)); eng.sync()
        out["classes_ms_B1"] = {k: round(v["ms"], 2) for k, v in eng.profile_report().items() if v["launches"]}
        eng.profile(False)
    eng.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run(layers=int(os.environ.get("RK_LAYERS", "32")), L=int(os.environ.get("RK_L", "1536")),
                         pool=os.environ.get("RK_POOL_WEIGHTS", "0") == "1")))
