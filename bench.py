#!/usr/bin/env python
"""bench.py — passages/sec of the pointwise yes_no hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without torchrun: re-executes itself under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   torch.distributed.run)

Workload (BASELINE.json configs[1]): flan-t5-large dimensions, pointwise yes_no, batch_size=32, synthetic
32-token-query / 128-token-passage prompts -> L_e = 184 encoder tokens per passage, decoder
input [pad], scores = logits of two vocabulary rows.  Weights are the deterministic synthetic checkpoint
(llmrankers._synth, HF init scales) — there are no real weights offline.  One "step" = one batch of 32
passages through encoder + decoder + head with the token ids already resident in HBM (rk_t5_stage before the
timed region; the host->device copy of 32x184 int32 = 23.5 KB is noted in DESIGN.md, never part of `value`).

Multi-GPU: passages of a query are independent, so the candidate list shards across ranks (weak scaling: each
rank scores its own batches) and every group's scores are collected with ONE RCCL all_gather over xGMI issued by the
ENGINE itself (rk_comm_all_gather_slot: straight from the device score buffer, on the stream that produces it).
torch.distributed ('gloo') only carries the host-side control plane: the RCCL id, the barrier, the max over ranks.
value = passages all ranks scored / max-over-ranks time.

`--mode shard` is the other multi-GPU mode north_star names: ONE query's candidate list (hits=100) is cut into contiguous
shares over the ranks (13,13,13,13,12,12,12,12 at 8 GPUs), every rank scores its share and ONE engine-issued all_gather per
query collects the scores; a step is then one QUERY, value = 100 x steps / time ("scaling": "strong" - the work per step is
fixed as N grows), and after the timed region every rank re-scores the whole query locally and asserts that the gathered
scores are bit-identical.  Latency-bound by design (M = 12-13 passages per GPU); the weak mode is the throughput mode.

The timed region is repeated `--regions` times (default 3, each exactly --steps steps between fences, max over ranks) and the
MEDIAN region is reported; all of them are listed in config.timed_regions_ms.

Extra objects on the JSON line: `roofline` (the dominant kernel — the FFN-in + GEGLU GEMM — MFMA-bound: algorithmic
2MNK flops per launch / average launch duration measured with HIP events on the engine's stream in a second, serial,
profiled pass; the whole tiled-GEMM family beside it), `cpu_baseline` (the reference's CPU path — HF transformers
fp32 — timed on this box's host cores, rank 0, N=1 only, on a bounded sample of the same workload) and
`config.per_query` (what the drop-in API delivers one query at a time).
"""
import argparse
import json
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "llm-rankers_amd"))
sys.path.insert(0, REPO)

MFMA_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md chip table
YES_ID, NO_ID = 2163, 465     # flan-t5 "Yes"/"No" ids quoted from memory; any two rows cost the same
GEMM_CLASSES = ["enc_gemm_qkv", "enc_gemm_o", "enc_gemm_ffn_in", "enc_gemm_ffn_out", "gemm_cross_kv"]
KERNEL_OF_CLASS = {
    "enc_gemm_ffn_in": "gemm_pp2_kernel (256x256x64 ping-pong fp16 MFMA GEMM, fused GEGLU epilogue), M={M} N={N2F} K={D}",
}


def algorithmic_gflop_per_passage(d, L_e, L_d=1):
    """SURVEY.md section 8(d) formula (2 FLOP per MAC), head excluded."""
    dm, I, F = d.d_model, d.inner, d.d_ff
    ffn = 6 if d.gated else 4
    enc = d.n_enc * (L_e * (8 * dm * I + ffn * dm * F) + 4 * L_e * L_e * I)
    dec = d.n_dec * (L_d * (8 * dm * I + 4 * dm * I + ffn * dm * F) + 4 * L_e * dm * I + 4 * L_d * L_e * I + 4 * L_d * L_d * I)
    return (enc + dec) / 1e9


def auto_group(steps: int) -> int:
    """Batches per engine launch sequence.  Around 8-10 batches (256-320 passages, 47-59 k tokens) every encoder GEMM
    is an almost whole number of rounds over the 256 CUs and the decoder chain is amortised; the group size must not
    put a ragged group inside the timed region, so it is the first preferred size that divides --steps."""
    for g in (8, 10, 9, 12, 7, 11, 6, 5, 16, 4):
        if steps % g == 0:
            return g
    return 8


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="flan-t5-large")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--seq_len", type=int, default=184)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_profile", action="store_true")
    ap.add_argument("--no_per_query", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (rk_engine_set_option), repeatable")
    ap.add_argument("--glds", type=int, default=1)
    ap.add_argument("--overlap", type=int, default=1, help="1: decoder chain of step i overlaps encoder of step i+1 (two HIP streams)")
    ap.add_argument("--group", type=int, default=0, help="batches (steps) per engine launch sequence; 0 = auto (see auto_group)")
    ap.add_argument("--mode", default="weak", choices=["weak", "shard"],
                    help="weak: every rank scores its own batches (throughput); shard: one query's hits=100 candidates cut over the ranks, one gather per query")
    ap.add_argument("--hits", type=int, default=100, help="shard mode: candidates per query")
    ap.add_argument("--regions", type=int, default=3, help="timed regions of exactly --steps steps each; the median is reported")
    ap.add_argument("--no_extras", action="store_true", help="skip the legs after the timed region: ragged (S2), one rank's shard share, setwise (S3), flan-t5-xl qlm (configs[3]), Llama-3-8B compare (configs[4])")
    ap.add_argument("--cpu_batch", type=int, default=32, help="batch size of the cpu_baseline leg (BASELINE.md section 3: 32)")
    ap.add_argument("--dry_ranks", action="store_true",
                    help="no GPU: N gloo ranks over tools/dry_engine.py walk the multi-rank control flow (communicator bring-up, "
                         "sharding, one gather per launch sequence, gather checks, the JSON line); timings are meaningless")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` (no torchrun): start the N ranks ourselves, exactly as the driver's own launch line."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: re-executing as\n        %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


class GroupPipeline:
    """A step is one batch of B passages (the reference's batch_size).  The engine consumes steps in GROUPS of G batches
    per launch sequence: the candidates of a query are all known up front, the reference's batch_size only shapes its
    host loop, and results do not depend on batch composition (ragged execution, bit-exact — tests).  Group i runs in
    slot i % n_slots; its decoder chain (own stream) overlaps the encoder of group i+1.  Steps that do not fill a last
    group run as a smaller group."""

    def __init__(self, eng, slot_seqs, B, G, dec, out_ids, world=1):
        self.eng, self.slot_seqs, self.B, self.G, self.dec, self.out_ids, self.world = eng, slot_seqs, B, G, dec, out_ids, world
        self.n_slots = len(slot_seqs)
        self.i = self.launched = 0
        self.last_slot = None

    def stage_all(self):
        for s in range(self.n_slots):
            self.eng.stage(self.slot_seqs[s], slot=s)          # inputs resident in HBM before the timed region

    def _launch_group(self):
        slot = self.launched % self.n_slots
        self.launched += 1
        self.eng.score_staged(self.dec, self.out_ids, slot=slot)
        if self.world > 1:                                     # ONE RCCL all_gather per group, behind the head kernel
            self.eng.comm_all_gather(self.G * self.B * len(self.out_ids), slot=slot)
        self.last_slot = slot

    def run(self, steps):
        for _ in range(steps):
            self.i += 1
            if self.i % self.G == 0:                           # the G-th batch of a group completes it: launch
                self._launch_group()
        rem = self.i % self.G
        if rem:                                                # leftover steps: a smaller group (re-stage a shorter slot)
            slot = self.launched % self.n_slots
            self.eng.sync()
            self.eng.stage(self.slot_seqs[slot][:rem * self.B], slot=slot)
            self._launch_group()
        self.i = 0


def timed_run(eng, pipe, steps, warmup, fence):
    """W untimed warm-up steps, then exactly `steps` steps between two fences.  Returns (wall seconds, HIP-event ms)."""
    pipe.run(warmup)
    fence()
    pipe.stage_all()
    fence()
    t_start = time.perf_counter()
    eng.timer_begin()
    pipe.run(steps)
    ev_ms = eng.timer_end()                                    # HIP events on the engine's own streams
    fence()
    return time.perf_counter() - t_start, ev_ms


class ShardPipeline:
    """--mode shard: a step is one QUERY of `hits` candidates.  This rank's contiguous share of query q (llmrankers/_dist.py:
    shard_bounds) is resident in slot q % n_slots; a step enqueues encoder + decoder + head for the share and, with more than
    one rank, ONE all_gather of the slot's score buffer (share width x 2 label rows) behind them.  Slots alternate, so the
    decoder chain and the gather of query i overlap the encoder of query i+1."""

    def __init__(self, eng, slot_seqs, dec, out_ids, world, width):
        self.eng, self.slot_seqs, self.dec, self.out_ids, self.world, self.width = eng, slot_seqs, dec, out_ids, world, width
        self.n_slots = len(slot_seqs)
        self.i = 0
        self.last_slot = None

    def stage_all(self):
        for s in range(self.n_slots):
            self.eng.stage(self.slot_seqs[s], slot=s)

    def run(self, steps):
        for _ in range(steps):
            slot = self.i % self.n_slots
            self.i += 1
            self.eng.score_staged(self.dec, self.out_ids, slot=slot)
            if self.world > 1:
                self.eng.comm_all_gather(self.width * len(self.out_ids), slot=slot)
            self.last_slot = slot


def ragged_leg(eng, dims, B, G, n_slots):
    """lengths ~ U{96..184} (seed 930), same grouped pipeline; passages/s, tokens/s and the fraction of the MFMA
    peak by the algorithmic FLOPs of the actual lengths."""
    from llmrankers import _synth
    slot_seqs = [[s for j in range(G) for s in _synth.synth_token_batch(B, 96, 184, dims.vocab, seed=930 + 8 * sl + j)]
                 for sl in range(n_slots)]
    pipe = GroupPipeline(eng, slot_seqs, B, G, [0], [YES_ID, NO_ID])
    pipe.stage_all()
    steps = 2 * G
    elapsed, _ = timed_run(eng, pipe, steps, G, eng.sync)
    done = [slot_seqs[k % n_slots] for k in range(steps // G)]
    n_pass = sum(len(g) for g in done)
    n_tok = sum(len(s) for g in done for s in g)
    gfl = sum(algorithmic_gflop_per_passage(dims, len(s)) for g in done for s in g)
    return {"workload": f"S2 ragged: lengths U{{96..184}}, batch_size={B}, {G} batches per launch sequence, {steps} steps",
            "passages_per_s": round(n_pass / elapsed, 1), "tokens_per_s": round(n_tok / elapsed, 1),
            "mean_tokens_per_passage": round(n_tok / n_pass, 1), "ms_per_step": round(elapsed / steps * 1e3, 3),
            "algorithmic_tflops": round(gfl / 1e3 / elapsed, 1), "frac_of_mfma_peak": round(gfl / 1e3 / elapsed / MFMA_PEAK_TFLOPS, 4)}


def setwise_leg(state):
    """SURVEY 8d S3 / BASELINE configs[2]: one setwise heapsort query (hits=100, num_child=10, k=10) end to end through
    SetwiseLlmRanker.rerank on its own engine (label rows of the head boosted so that generations are labels, as with a
    trained checkpoint), both scorings; and the CLI's default number of queries in lockstep (rerank_many; 32 since the end of round 6).
    tools/bench_setwise_query.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rk_bench_setwise", os.path.join(REPO, "tools", "bench_setwise_query.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # 140-word passages are cut to the full 128 tokens of run.py's default --passage_length and the query is ~31 tokens: prompts
    # of ~1.56k tokens = S3's 11 x 134 + 32 + 30 (rounds 1-4 ran this leg on 60-word passages: ~0.9k-token prompts)
    from llmrankers._batching import default_queries_per_call
    nq = default_queries_per_call("setwise", 100)
    res = mod.run(state=state, reps=2, many=nq, one_by_one=False, words=140, query_words=24)
    for v in res.values():
        v.pop("top10", None)
    res["workload"] = "S3 at its stated size: flan-t5-large dims, setwise heapsort, hits=100 num_child=10 k=10, 128-token passages + " \
                      "~31-token query (fixture tokenizer; avg_prompt_tokens per compare is reported), level-batched build phase; " \
                      f"*_many{nq} = {nq} queries ranked in lockstep (SetwiseLlmRanker.rerank_many = run.py's default for setwise; sixteen until the end of round 6: 47.1 / 51.3 ms per query in the leases of that round)"
    return res


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("rk_" + name, os.path.join(REPO, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def qlm_xl_leg():
    """BASELINE configs[3]: flan-t5-xl dims, pointwise qlm, hits=100 through PointwiseLlmRanker.rerank (tools/bench_qlm_xl.py), and
    the 13-passage share one of eight ranks scores of the doc-sharded query."""
    mod = _tool("bench_qlm_xl")
    res = mod.run(reps=5)
    try:
        share = mod.run(reps=5, shard=8)
        res["share_of_8_ranks"] = {k: share[k] for k in ("workload", "ms_per_query", "passages_per_s", "frac_of_mfma_peak")}
        res["share_of_8_ranks"]["predicted_8gpu_passages_per_s"] = round(100.0 / share["ms_per_query"] * 1e3, 1)
    except Exception as exc:
        res["share_of_8_ranks"] = {"error": repr(exc)[:300]}
    return res


def llama_leg():
    """BASELINE configs[4]: Llama-3-8B dims at full depth, one setwise generation compare (tools/bench_llama.py)."""
    return _tool("bench_llama").run(layers=32, L=1536, pool=True, iters=5, profile=True)


def shard_share_leg(eng, dims, L, hits=100, world=8, steps=40, queries_per_step=1):
    """What ONE of `world` ranks does per query in --mode shard (strong scaling): its 13-passage share of a hits=100 query per
    step through the two slots, no collective (one rank).  100 / (ms per step) is the 1-GPU prediction of the 8-GPU
    strong-scaling number (the gather adds one ~20 us collective per query).  queries_per_step = 16: what run.py does by default
    under --shard_candidates 1 since round 6 (PointwiseLlmRanker.rerank_many -> _rerank_sharded_many: every rank's share of
    sixteen queries in ONE launch sequence, one gather for all of them) - 208 passages per rank and step instead of 13."""
    from llmrankers import _synth
    from llmrankers._dist import shard_bounds
    lo, hi = shard_bounds(hits, world)[0]
    n_slots = eng.num_slots
    nq = queries_per_step
    queries = [[_synth.synth_token_batch(hits, L, L, dims.vocab, seed=4000 + q * nq + j) for j in range(nq)] for q in range(n_slots)]
    pipe = ShardPipeline(eng, [[s for qq in qs for s in qq[lo:hi]] for qs in queries], [0], [YES_ID, NO_ID], 1, (hi - lo) * nq)
    pipe.stage_all()
    elapsed, _ = timed_run(eng, pipe, steps, 8, eng.sync)
    ms = elapsed / steps * 1e3 / nq                              # per query
    gfl = algorithmic_gflop_per_passage(dims, L) * (hi - lo)
    what = "ONE query" if nq == 1 else f"{nq} queries (one launch sequence, one gather)"
    return {"workload": f"one rank's share of {what} per step: {hi - lo} of hits={hits} candidates each (world = {world}), L_e={L}, {steps} steps",
            "queries_per_step": nq, "ms_per_query": round(ms, 3), "ms_per_step": round(ms * nq, 3),
            "share_passages_per_s": round((hi - lo) / ms * 1e3, 1),
            "predicted_8gpu_strong_scaling_passages_per_s": round(hits / ms * 1e3, 1),
            "frac_of_mfma_peak_per_gpu": round(gfl / ms / MFMA_PEAK_TFLOPS, 4)}


def profile_pass(eng, pipe, G, M_tokens):
    """Second, serial pass with an event pair around every launch -> per-class time, the roofline object."""
    eng.profile(True)
    eng.profile_reset()
    eng.set_option("overlap", 0)                               # per-kernel events need a serial timeline
    n_prof = 3
    for _ in range(n_prof):
        eng.score_staged(pipe.dec, pipe.out_ids, slot=0)
    eng.sync()
    rep = eng.profile_report()
    eng.profile(False)
    eng.set_option("overlap", 1)
    classes = [c for c in GEMM_CLASSES if c in rep]
    fam_ms = sum(rep[c]["ms"] for c in classes)
    fam_fl = sum(rep[c]["flops"] for c in classes)
    fam_n = sum(rep[c]["launches"] for c in classes)
    total_ms = sum(v["ms"] for v in rep.values())
    dom = max(classes, key=lambda c: rep[c]["ms"])            # the kernel with the largest share of GPU time
    d = rep[dom]
    dd = eng.dims
    kernel_desc = KERNEL_OF_CLASS.get(dom, "gemm_pp2_kernel (256x256x64 ping-pong fp16 MFMA GEMM)").format(
        M=M_tokens, D=dd.d_model, I=dd.inner, F=dd.d_ff, N2F=(2 if dd.gated else 1) * dd.d_ff, N3I=3 * dd.inner)
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    # HBM-side traffic of the dominant kernel comes from rocprofv3 PMC passes of THIS command (tools/gpu_prof.sh:
    # FETCH_SIZE and WRITE_SIZE each in their own --pmc run; gfx950: FETCH_SIZE x2 for wide coalesced reads, KiB units;
    # MI355X_MICROARCH.md HBM section), summarised per launch in profiles/pmc_summary_latest.json.  It is only quoted
    # when that summary was taken at this run's GEMM M (= tokens per launch sequence); otherwise null.
    traffic, traffic_src = None, "no PMC summary for M=%d tokens per launch" % M_tokens
    pm = {}
    # The committed summaries are quoted only when they were taken from THESE kernel sources (tools/gpu_prof.sh stamps them with
    # __graft_entry__._source_hash(), the hash the in-tree library is built from): after a kernel change without a new profiling
    # lease the rocprofv3-derived fields are null instead of stale (round-5 advisor finding).
    try:
        import __graft_entry__ as _ge
        src_hash = _ge._source_hash()
    except Exception:
        src_hash = None
    try:
        with open(os.path.join(REPO, "profiles", "pmc_summary_latest.json")) as f:
            pm = json.load(f)
        if pm.get("source_hash") != src_hash or src_hash is None:
            traffic_src = "profiles/pmc_summary_latest.json was taken from other kernel sources (source_hash differs): not quoted"
            pm = {}
        if pm.get("tokens_per_launch") == M_tokens:
            want = "gemm_pp2"
            gk = [(v["stats"]["pct"], k, v) for k, v in pm["kernels"].items()
                  if want in k and v.get("stats") and "FETCH_SIZE" in v["pmc"] and "WRITE_SIZE" in v["pmc"]]
            if gk:
                _, kname, v = max(gk)
                traffic = int((2 * v["pmc"]["FETCH_SIZE"]["avg_per_launch"] + v["pmc"]["WRITE_SIZE"]["avg_per_launch"]) * 1024)
                traffic_src = f"bytes per launch of {kname}, rocprofv3 PMC at the same M, profiles/pmc_summary_latest.json"
    except Exception:
        pass
    # the same fraction from the committed rocprofv3 kernel statistics (tools/gpu_final.sh runs this command un-profiled and under
    # rocprofv3 --kernel-trace --stats back to back in ONE lease and the summaries are committed as profiles/bench_kernel_stats_latest.csv
    # + pmc_summary_latest.json): algorithmic flops per launch / the profiler's average duration of the dominant kernel.  A profiled
    # run clocks ~1-2 % lower than the event-timed one, and a driver-run line comes from another box (+-3 %): both are stated.
    frac_rocprof = None
    try:
        import csv
        want = {"enc_gemm_ffn_in": "gemm_pp2_kernel<2, 0, true", "enc_gemm_qkv": "gemm_pp2_kernel<0, 0, true"}.get(dom, "gemm_pp2_kernel<1, 0, false")
        with open(os.path.join(REPO, "profiles", "bench_kernel_stats_latest.csv")) as f:
            rows = [r for r in csv.DictReader(f) if want in r["Name"]]
        if rows and pm.get("tokens_per_launch") == M_tokens:
            r = max(rows, key=lambda r: float(r["Percentage"]))
            avg_us = float(r["AverageNs"]) / 1e3
            fl = d["flops"] / max(d["launches"], 1)
            frac_rocprof = {"frac": round(fl / (avg_us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "avg_launch_us": round(avg_us, 1), "calls": int(r["Calls"]),
                            "kernel": r["Name"], "source": "profiles/bench_kernel_stats_latest.csv: rocprofv3 --kernel-trace --stats of this command at the same M and from the same "
                                                           "kernel sources (source_hash " + str(src_hash)[:12] + "), taken in the builder's evidence lease (not on the box of a driver-run line)"}
    except Exception:
        pass
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "frac_rocprof": frac_rocprof, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": f"{dom}: {kernel_desc}",
            "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 2), "launches": int(d["launches"]),
            "algorithmic_gflop_per_launch": round(d["flops"] / max(d["launches"], 1) / 1e9, 1),
            "family": {"kernel": "all tiled encoder GEMM launches (qkv, o, ffn_in+GEGLU, ffn_out; chained or separate)",
                       "achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 1) if fam_ms > 0 else None,
                       "frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if fam_ms > 0 else None,
                       "launches": int(fam_n), "share_of_gpu_time": round(fam_ms / total_ms, 3) if total_ms else None},
            "per_class": {k: {"ms_per_step": round(v["ms"] / (n_prof * G), 4),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None}
                          for k, v in rep.items() if v["launches"]}}


def per_query_numbers(eng, dims, B, L):
    """What the drop-in API delivers ONE query at a time (hits=100, batch_size=32 -> batches 32,32,32,4): (a) the
    engine part on pre-tokenised prompts of the bench shape through T5Runtime.score_batches (what
    PointwiseLlmRanker.rerank calls), (b) PointwiseLlmRanker.rerank on text, host tokenisation included (fixture
    tokenizer, so the prompt length differs from 184)."""
    import numpy as np
    from llmrankers import _synth
    from llmrankers._runtime import T5Runtime
    rt = T5Runtime.from_engine(eng, dims)
    out = {}
    seqs = _synth.synth_token_batch(100, L, L, dims.vocab, seed=4242)
    chunks = [seqs[s:s + B] for s in range(0, 100, B)]
    ts = []
    for _ in range(7):
        t = time.perf_counter()
        rt.score_batches(chunks, [0], [YES_ID, NO_ID])
        ts.append(time.perf_counter() - t)
    ms = float(np.median(ts[2:])) * 1e3
    out["pretokenised"] = {"ms_per_query": round(ms, 2), "passages_per_s": round(100 / ms * 1e3, 1), "tokens_per_prompt": L}
    try:
        from transformers import T5Tokenizer
        from llmrankers.pointwise import PointwiseLlmRanker
        from llmrankers.rankers import SearchResult
        tok = T5Tokenizer.from_pretrained(os.path.join(REPO, "tests", "golden", "tok"))
        ranker = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=B)
        words = "neural ranking model search engine index retrieval document answer question relevant topic passage".split()
        rs = np.random.RandomState(0)
        query = " ".join(rs.choice(words, 30))
        docs = [" ".join(rs.choice(words, 126)) for _ in range(100)]
        ts = []
        for _ in range(6):
            ranking = [SearchResult(docid=str(i), score=float(100 - i), text=d) for i, d in enumerate(docs)]
            t = time.perf_counter()
            ranker.rerank(query, ranking)
            ts.append(time.perf_counter() - t)
        ms = float(np.median(ts[2:])) * 1e3
        out["rerank_api"] = {"ms_per_query": round(ms, 2), "passages_per_s": round(100 / ms * 1e3, 1),
                             "tokens_per_prompt": round(ranker.total_prompt_tokens / 100.0, 1),
                             "note": "PointwiseLlmRanker.rerank incl. prompt building, host tokenisation (fixture tokenizer), sort"}
        try:
            # (c) the same through PointwiseLlmRanker.rerank_many with run.py's DEFAULT queries per call (--queries_per_call 0 =
            # auto: enough queries for >= 1 600 passages, 16 at hits=100; launches start while later queries are still being
            # tokenised, T5Runtime.score_stream) - identical rankings and counters, the engine's grouped
            # throughput instead of its one-query-at-a-time one: this is what `python run.py run ... pointwise` delivers
            from llmrankers._batching import default_queries_per_call
            queries = [" ".join(rs.choice(words, 30)) for _ in range(default_queries_per_call("pointwise", 100))]
            ts = []
            for _ in range(5):
                items = [(q, [SearchResult(docid=str(i), score=float(100 - i), text=d) for i, d in enumerate(docs)]) for q in queries]
                t = time.perf_counter()
                ranker.rerank_many(items)
                ts.append(time.perf_counter() - t)
            ms = float(np.median(ts[2:])) * 1e3 / len(queries)
            out["rerank_many_api"] = {"queries_per_call": len(queries), "ms_per_query": round(ms, 2), "passages_per_s": round(100 / ms * 1e3, 1),
                                      "note": "run.py's default (--queries_per_call 0 = auto): PointwiseLlmRanker.rerank_many"}
        except Exception as exc:
            out["rerank_many_api"] = {"error": repr(exc)}
    except Exception as exc:                                   # never take the headline number down
        out["rerank_api"] = {"error": repr(exc)}
    return out


def main():
    args = parse_args()
    self_spawn(args)
    if args.group <= 0:
        args.group = auto_group(args.steps)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch                        # first: its bundled HIP runtime must be the one in the process
    import torch.distributed as dist
    dry = args.dry_ranks
    if dry:
        args.no_profile = args.no_per_query = args.no_cpu_baseline = args.no_extras = True
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    elif local_rank >= torch.cuda.device_count():
        if torch.cuda.device_count() == 1 and os.environ.get("RK_BENCH_MASKED_DEVICES", "1") == "1":
            # a launcher that masks the devices per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = one GPU each): this rank's
            # GPU is device 0.  (Ranks that really share one GPU fail loudly further down: RCCL refuses duplicate devices.)
            print(f"[bench] rank {rank}: LOCAL_RANK {local_rank}, one GPU visible: using device 0 (devices masked per rank)", file=sys.stderr)
            local_rank = 0
            torch.cuda.set_device(0)
        else:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    else:
        torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="gloo")                # host-side control plane only; scores travel over RCCL
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from llmrankers import _synth
    from llmrankers._runtime import T5Runtime
    if dry:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from dry_engine import DryEngine as RkEngine
    else:
        import __graft_entry__ as ge
        ge.build()
        from llmrankers._engine import RkEngine

    dims = _synth.NAMED_DIMS[args.model]
    B, L, G = args.batch_size, args.seq_len, max(1, args.group)
    t0 = time.time()
    state = {} if dry else _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=local_rank, max_tokens=max(8192, G * B * L, 100 * (L + 72)), max_seqs=max(128, G * B), max_dec_len=4)
    eng.load_state(state.items())
    eng.set_option("gemm_glds", args.glds)
    eng.set_option("overlap", args.overlap)
    for kv in args.opt:                                    # engine A/B switches for experiments, e.g. --opt gemm_variant=2
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    rccl = None
    if world > 1:
        # engine-owned RCCL communicator (K9), built exactly as the rankers build theirs on their first sharded query
        # (T5Runtime.ensure_comm -> comm_init_from_process_group: rank 0's id travels over the gloo group, nothing else does)
        rt = T5Runtime.from_engine(eng, dims)
        rt.comm_init_from_process_group(max(G * B * 2, args.hits * 2))
        assert rt.comm_ready() and rt.comm_capacity >= G * B * 2
    try:
        rccl = eng.comm_library_info()                     # which librccl this process binds (path | ncclGetVersion code)
    except Exception as exc:                               # never take a 1-GPU number down: RCCL is only needed for N > 1
        rccl = f"unavailable: {exc}"
    if rank == 0:
        print(f"[bench] weights generated + engine finalized in {time.time() - t0:.1f}s; group = {G} batches; rccl = {rccl}", file=sys.stderr)

    def group_batch(n_batches, seed):
        return [s for j in range(n_batches) for s in _synth.synth_token_batch(B, L, L, dims.vocab, seed=seed + j)]
    n_slots = eng.num_slots
    shard = args.mode == "shard"
    if shard:
        from llmrankers._dist import shard_bounds
        bounds = shard_bounds(args.hits, world)
        width = max(b - a for a, b in bounds)
        lo, hi = bounds[rank]
        queries = [_synth.synth_token_batch(args.hits, L, L, dims.vocab, seed=4000 + q) for q in range(n_slots)]   # same on every rank
        pipe = ShardPipeline(eng, [q[lo:hi] for q in queries], [0], [YES_ID, NO_ID], world, width)
        seqs = queries[0][:B]
    else:
        slot_seqs = [group_batch(G, 929 + rank * 64 + 8 * s) for s in range(n_slots)]
        seqs = slot_seqs[0][:B]
        pipe = GroupPipeline(eng, slot_seqs, B, G, [0], [YES_ID, NO_ID], world)
    pipe.stage_all()

    def fence():
        eng.sync()
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    regions, ev_regions = [], []
    for r in range(max(1, args.regions)):
        el, ev_ms = timed_run(eng, pipe, args.steps, args.warmup if r == 0 else 0, fence)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        regions.append(el)
        ev_regions.append(ev_ms)
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    mid = order[len(order) // 2]
    elapsed, ev_ms = regions[mid], ev_regions[mid]
    gather_check = None
    pipeline_check = None
    if shard:
        # every rank re-scores the WHOLE last query itself: the gathered scores must be those bits (batch independence)
        q = queries[pipe.last_slot]
        if world > 1:
            got = eng.comm_read_gathered(pipe.last_slot)       # [world, width * 2]
            gathered = np.concatenate([got[r_, :(b_ - a_) * 2] for r_, (a_, b_) in enumerate(bounds)]).reshape(-1, 2)
        else:
            gathered = eng.read_scores(pipe.last_slot)
        local_all = eng.score(q, [0], [YES_ID, NO_ID])
        assert np.array_equal(gathered, local_all), f"rank {rank}: gathered scores differ from the locally recomputed ones"
        gather_check = f"gathered == locally recomputed scores of all {args.hits} candidates on every rank (bit-exact)"
        scores = local_all[:B]
    else:
        if world > 1:
            got = eng.comm_read_gathered(pipe.last_slot)       # [world, G*B*2]: every rank holds every rank's scores
            mine = eng.read_scores(pipe.last_slot).reshape(-1)
            assert got.shape[0] == world and np.array_equal(got[rank, :len(mine)], mine), f"rank {rank}: its own gathered row differs from its scores"
            gather_check = "every rank's gathered row of its own scores == its local score buffer (bit-exact)"
        scores = eng.read_scores(0)[:B]
        assert all(np.isfinite(eng.read_scores(s)).all() for s in range(n_slots))
        # the pipelined launch sequences (two slots in flight, decoder graphs beside the next encoder) must give the bits of a
        # blocking call on the same batch: batch independence under the conditions of the timed region itself (round 4 found a
        # counted-wait race that only showed here)
        if not dry:
            for s_ in range(n_slots):
                piped = eng.read_scores(s_)
                again = eng.score(pipe.slot_seqs[s_][:len(piped)], [0], [YES_ID, NO_ID])   # (a leftover group re-staged a prefix)
                assert np.array_equal(piped, again), f"rank {rank}: slot {s_} of the pipeline differs from a blocking call on the same batch"
            pipeline_check = "scores of every slot's last pipelined launch sequence == a blocking call on the same batch (bit-exact)"
    if shard and not args.no_profile and rank == 0:
        print("[bench] --mode shard: the roofline pass profiles the share-sized launches (M = %d tokens)" % ((hi - lo) * L), file=sys.stderr)

    roofline = None
    if not args.no_profile:
        pipe.stage_all()
        roofline = profile_pass(eng, pipe, 1 if shard else G, (hi - lo) * L if shard else G * B * L)

    extras = {}
    if rank == 0 and world == 1 and not shard and not args.no_extras:
        try:
            extras["ragged"] = ragged_leg(eng, dims, B, G, n_slots)
        except Exception as exc:                      # never take the headline number down
            extras["ragged"] = {"error": repr(exc)[:300]}

        try:
            extras["shard_share"] = shard_share_leg(eng, dims, L, args.hits)
            extras["shard_share"]["grouped16"] = shard_share_leg(eng, dims, L, args.hits, steps=10, queries_per_step=16)
        except Exception as exc:
            extras["shard_share"] = {"error": repr(exc)[:300]}

    per_query = None
    if rank == 0 and world == 1 and not args.no_per_query:
        per_query = per_query_numbers(eng, dims, B, L)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import hf_path
            cpu = hf_path.time_cpu_baseline(dims, state, [list(s) for s in seqs], YES_ID, NO_ID, sample_batch=args.cpu_batch,
                                              tokenizer_dir=os.path.join(REPO, "tests", "golden", "tok"))
            ref_logits = cpu.pop("logits")
            p_cpu = 1 / (1 + np.exp(-(ref_logits[:, 0] - ref_logits[:, 1])))
            p_gpu = 1 / (1 + np.exp(-(scores[:len(p_cpu), 0] - scores[:len(p_cpu), 1])))
            cpu["max_abs_score_diff_vs_gpu"] = float(np.abs(p_cpu - p_gpu).max())
            cpu["value"] = round(cpu["value"], 3)
        except Exception as exc:                      # the baseline must never take the GPU number down with it
            cpu = {"value": None, "unit": "passages/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {exc!r}"}

    if rank == 0 and world == 1 and not shard and not args.no_extras:
        eng.close()                                   # the setwise leg builds its own engine (other head rows, other capacities)
        try:
            extras["setwise_query"] = setwise_leg(state)
        except Exception as exc:
            extras["setwise_query"] = {"error": repr(exc)[:300]}
        del state
        for key, leg in (("qlm_xl", qlm_xl_leg), ("llama_compare", llama_leg)):
            t_leg = time.time()
            try:
                extras[key] = leg()
            except Exception as exc:
                extras[key] = {"error": repr(exc)[:300]}
            print(f"[bench] leg {key}: {time.time() - t_leg:.0f}s", file=sys.stderr)

    if rank == 0:
        passages = args.steps * (args.hits if shard else B * world)
        gfl = algorithmic_gflop_per_passage(dims, L)
        value = passages / elapsed
        # What the first real 8-GPU lease should show, predicted from THIS 1-GPU run (no curve is claimed: SCALE records have been
        # skipped for want of an 8-GPU node): --mode weak = 8 replicas of this pipeline, each adding ONE engine-issued all_gather of
        # its launch sequence's scores (G * B * 2 floats per rank; latency-bound, ASSUMED 25 us - the N > 1 collective never ran on
        # hardware); --mode shard = one query per step, every rank scoring its 12-13 passage share (config.shard_share).
        predicted_8gpu = None
        if world == 1 and not shard and not dry:
            GATHER_US = 25.0
            seq_ms = elapsed / args.steps * 1e3 * G
            predicted_8gpu = {"weak_passages_per_s": round(8 * value * seq_ms / (seq_ms + GATHER_US * 1e-3), 1),
                              "weak_basis": f"8 x this line's value, one assumed {GATHER_US:.0f}-us all_gather per {seq_ms:.1f}-ms launch sequence and rank",
                              "shard_passages_per_s": (extras.get("shard_share") or {}).get("predicted_8gpu_strong_scaling_passages_per_s"),
                              "shard_basis": "hits / (ms per 13-passage share on one GPU), config.shard_share; + one gather per query",
                              "shard_grouped16_passages_per_s": ((extras.get("shard_share") or {}).get("grouped16") or {}).get("predicted_8gpu_strong_scaling_passages_per_s"),
                              "shard_grouped16_basis": "the same with every rank's share of 16 queries per launch sequence (run.py's default under --shard_candidates 1)",
                              "north_star_target_passages_per_s": 10000}
        line = {
            "metric": "passages/sec (pointwise yes_no reranking, flan-t5-large shape)", "value": round(value, 1),
            "unit": "passages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "vs_baseline": None, "dtype": "f16 (MFMA inputs), f32 accumulate + residual stream",
            "data": "synthetic" if not dry else "DRY RUN (no GPU: tools/dry_engine.py, gloo) - control flow only, timings meaningless",
            "config": {"engine_group": (f"one query's share ({hi - lo} of {args.hits} candidates) per engine launch sequence" if shard
                                        else f"{G} batches ({G * B} passages) per engine launch sequence"),
                       "workload": (f"{args.model} pointwise yes_no, ONE query of hits={args.hits} candidates per step, cut over {world} rank(s) "
                                    f"({', '.join(str(b_ - a_) for a_, b_ in bounds)}), L_e={L}, L_d=1, 2 label rows" if shard else
                                    f"{args.model} pointwise yes_no, hits=100 batch_size={B} (one step = one batch), "
                                    f"L_e={L} (128-token passage + 32-token query + template), L_d=1, 2 label rows"),
                       "mode": args.mode, "global_batch": args.hits if shard else B * world, "seq_len": L,
                       "parallelism": (f"dp{world}, candidates of each query sharded; one engine-issued RCCL all_gather per query" if shard else
                                       f"dp{world} (every rank scores its own batches; one engine-issued RCCL all_gather per launch sequence)"),
                       "gather_check": gather_check, "pipeline_check": pipeline_check, "rccl": rccl,
                       "timed_regions_ms": [round(x * 1e3, 2) for x in regions], "region_reported": "median",
                       "weights": "synthetic N(0, HF-init std), seed 929", "engine_stream_ms_per_step": round(ev_ms / args.steps, 3),
                       "algorithmic_gflop_per_passage": round(gfl, 2),
                       "note_executed_flops": "throughput fractions use the reference's algorithmic FLOPs (SURVEY 8d); the engine skips "
                                              "the dead decoder q/k at L_d=1 and replaces the 18.5 GFLOP/passage cross-K/V projections by "
                                              "the exact query-side form (DESIGN.md section 3)",
                       "whole_path_tflops_per_gpu": round(value / world * gfl / 1e3, 1),
                       "whole_path_frac_of_mfma_peak": round(value / world * gfl / 1e3 / MFMA_PEAK_TFLOPS, 4),
                       "per_query": per_query, "ragged": extras.get("ragged"), "shard_share": extras.get("shard_share"),
                       "predicted_8gpu": predicted_8gpu,
                       "setwise_query": extras.get("setwise_query"), "qlm_xl": extras.get("qlm_xl"), "llama_compare": extras.get("llama_compare")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if eng.h:
        eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
