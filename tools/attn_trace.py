#!/usr/bin/env python
"""Phase timeline of the DMA attention kernel from a measurement build (hipcc ... -DRK_MEASURE -o exp/librk_engine_measure.so;
RK_ENGINE_LIB=exp/librk_engine_measure.so python tools/attn_trace.py): the twelve waves of one workgroup stamp the 100 MHz wall
clock at the phase boundaries of every item (attention.h: ATTD_STAMP); prints the mean duration of every phase in microseconds
per wave group, for the bench shape (flan-t5-large dims, 320 x 184 tokens)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
PHASES = ["dma_issue", "scores(QK)", "bias+max", "exp+pack", "issue_q+wait_V", "barrier_A", "P.V", "ctx_convert", "wait_Q+table",
          "stores", "barrier_B"]


def main():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()   # RK_ENGINE_LIB: A/B or measurement build (tools only)
    import bench
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    dims = _synth.FLAN_T5_LARGE
    B, L, G = 32, 184, 10
    state = _synth.synth_state_dict(dims, seed=929, threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, device=0, max_tokens=G * B * L, max_seqs=G * B, max_dec_len=4)
    eng.load_state(state.items())
    seqs = [s for j in range(G) for s in _synth.synth_token_batch(B, L, L, dims.vocab, seed=929 + j)]
    eng.set_option("overlap", 0)
    eng.set_option("attn_trace", 1)
    eng.stage(seqs, slot=0)
    for _ in range(2):
        eng.score_staged([0], [bench.YES_ID, bench.NO_ID], slot=0)
    eng.sync()
    t = eng.debug_read("attn_trace", 12 * 16 * 16).reshape(12, 16, 16) / 100.0      # microseconds since the wave's start
    n_items = 10
    for grp in range(2):
        w = t[6 * grp:6 * grp + 6, 1:n_items - 1, :12]              # steady-state items (not the first, not the last)
        d = np.diff(w, axis=2)                                      # [wave, item, phase]
        item = (w[:, 1:, 0] - w[:, :-1, 0]).mean()
        print(json.dumps({"group": grp, "us_per_item": round(float(item), 2),
                          "phase_us_mean": {p: round(float(d[:, :, i].mean()), 2) for i, p in enumerate(PHASES)},
                          "phase_us_max_over_waves": {p: round(float(d[:, :, i].mean(axis=1).max()), 2) for i, p in enumerate(PHASES)}}))
    print(json.dumps({"item_start_us_wave0_group0": [round(float(x), 1) for x in t[0, :n_items, 0]],
                      "item_start_us_wave0_group1": [round(float(x), 1) for x in t[6, :n_items, 0]]}))
    eng.close()


if __name__ == "__main__":
    main()
