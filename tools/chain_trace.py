#!/usr/bin/env python
"""Timeline of ONE chained GEMM launch from a measurement build (RK_ENGINE_LIB=exp/librk_engine_measure.so): every workgroup
stamps the 100 MHz wall clock at the start of a tile, at the end of its main loop and at the end of its epilogue (ChainArgs::trace;
the LAST chained launch of an encoder pass: layer 23's {O -> FFN-in}, or with RK_ONLY=2 layer 22's {FFN-out -> QKV}).  Prints, per
tile kind, the main-loop / epilogue / hand-over times, how many tiles found their panel unpublished, tiles per workgroup, the
launch span and the tail.  RK_DEBUG = knock-out bits (timing only), RK_LEAD = producer lead."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
import bench
from llmrankers import _synth
from llmrankers._engine import RkEngine
dims = _synth.FLAN_T5_LARGE
B, L, G = 32, 184, 10
state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
eng = RkEngine(dims, device=0, max_tokens=G * B * L, max_seqs=G * B, max_dec_len=4).load_state(state.items())
seqs = [s for j in range(G) for s in _synth.synth_token_batch(B, L, L, dims.vocab, seed=929 + j)]
eng.stage(seqs, slot=0)
eng.set_option("overlap", 0)
for only in [int(x) for x in os.environ.get("RK_ONLY", "1,2").split(",")]:
    for dbg in [int(x) for x in os.environ.get("RK_DEBUG", "0").split(",")]:
        for lead in [int(x) for x in os.environ.get("RK_LEAD", "3").split(",")]:
            eng.set_option("chain", 1); eng.set_option("chain_only", only); eng.set_option("chain_debug", dbg); eng.set_option("chain_lead", lead)
            for _ in range(2):
                eng.score_staged([0], [bench.YES_ID, bench.NO_ID], slot=0)
            eng.sync()
            # one launch of the middle of the pass: with both pairs chained launch 2 l is layer l's {O -> FFN-in}, 2 l + 1 its
            # {FFN-out -> QKV}; with one pair chained launch l is layer l's
            which = 12 if only else (24 if os.environ.get("RK_PAIR", "A") == "A" else 25)
            eng.set_option("chain_trace", 1 + which)             # (clears the buffer)
            eng.score_staged([0], [bench.YES_ID, bench.NO_ID], slot=0)
            eng.sync()
            raw = eng.debug_read("chain_trace", 256 * 64 * 4 * 2).view(np.uint64).reshape(256, 64, 4)
            eng.set_option("chain_debug", dbg)
            eng.set_option("chain_trace", 0)
            eng.profile(True); eng.profile_reset()
            eng.score_staged([0], [bench.YES_ID, bench.NO_ID], slot=0); eng.sync()
            rep = eng.profile_report(); eng.profile(False)
            t0 = raw[:, :, 0].astype(np.float64); valid = t0 > 0
            base = t0[valid].min()
            start = (t0 - base) / 100.0; main = (raw[:, :, 1].astype(np.float64) - base) / 100.0; end = (raw[:, :, 2].astype(np.float64) - base) / 100.0
            code = (raw[:, :, 3] & 0xFFFFFFFF).astype(np.int64); ready = (raw[:, :, 3] >> 32).astype(np.int64)
            prod = ((code >> 30) & 1) == 1
            n_tiles = valid.sum(axis=1)
            res = {"only": only, "debug": dbg, "lead": lead, "tiles": int(valid.sum()), "tiles_per_wg_min_max": [int(n_tiles.min()), int(n_tiles.max())],
                   "span_us": round(float(end[valid].max()), 1), "first_wg_done_us": round(float(np.array([end[w, n_tiles[w] - 1] for w in range(256) if n_tiles[w]]).min()), 1),
                   "launch_us_events": {k: round(v["ms"] * 1e3 / v["launches"], 1) for k, v in rep.items() if v["launches"] and k.startswith("enc_chain")}}
            for name, sel in (("producer", valid & prod), ("consumer", valid & ~prod)):
                if not sel.any():
                    continue
                gap = []
                for w in range(256):
                    for k in range(1, n_tiles[w]):
                        if sel[w, k - 1]:
                            gap.append(start[w, k] - end[w, k - 1])
                res[name] = {"n": int(sel.sum()), "main_us_mean": round(float((main - start)[sel].mean()), 2), "main_us_p90": round(float(np.percentile((main - start)[sel], 90)), 2),
                             "epilogue_us_mean": round(float((end - main)[sel].mean()), 2), "epilogue_us_p90": round(float(np.percentile((end - main)[sel], 90)), 2),
                             "gap_to_next_tile_us_mean": round(float(np.mean(gap)), 2) if gap else None,
                             "next_not_ready_when_main_ended": int((sel & (ready == 0)).sum())}
            # how many producer tiles are in their epilogue at a time (mean over the launch, sampled every 2 us)
            ts = np.arange(0, end[valid].max(), 2.0)
            in_epi = [(valid & prod & (main <= t) & (end > t)).sum() for t in ts]
            res["producer_epilogues_in_flight_mean_max"] = [round(float(np.mean(in_epi)), 1), int(np.max(in_epi))]
            print(json.dumps(res), flush=True)
eng.close()
