#!/usr/bin/env python
"""Phase timeline of the long-sequence attention kernel from a measurement build (hipcc ... -DRK_MEASURE -o exp/librk_engine_measure.so;
RK_ENGINE_LIB=exp/librk_engine_measure.so python tools/attn_long_trace.py): the four waves of one workgroup stamp the shader clock at
the phase boundaries of every 128-key chunk (attention.h: ATTL_STAMP); prints mean cycles per phase, near and far chunks apart,
for one and eight 1 560-token prompts at flan-t5-large dims."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lib; _lib.use_env_library()
from llmrankers import _synth
from llmrankers._engine import RkEngine
PHASES = ["scores(QK)", "bias+max", "exp+pack+rescale", "PV tile 0", "PV tile 1", "waitcnt(dma,lds)", "barrier", "issue next chunk/table"]
dims = _synth.FLAN_T5_LARGE
eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(_synth.synth_tensors(dims, seed=929, threads=32))
eng.set_option("overlap", 0)
eng.set_option("attn_trace", 1)
prefix, labels = [0, 5454], list(range(71, 82))
for B in (8, 1):
    seqs = _synth.synth_token_batch(B, 1560, 1560, dims.vocab, seed=7)
    for _ in range(2):
        eng.score(seqs, prefix, labels)
    eng.sync()
    t = eng.debug_read("attn_trace", 12 * 16 * 16).reshape(12, 16, 16)[:4, :13, :8]      # [wave][chunk][stamp 0..7]
    # phase k = stamp k+1 - stamp k (k = 0..6); phase 7 = next chunk's stamp 0 - this chunk's stamp 7
    d = np.diff(t, axis=2)                                                            # [wave][chunk][7]
    nxt = t[:, 1:, 0] - t[:, :-1, 7]                                                   # [wave][12]
    out = {"B": B, "cycles_per_chunk_mean": round(float((t[:, 1:, 0] - t[:, :-1, 0]).mean()), 0)}
    for name, sel in (("all", slice(1, 12)), ("near(4..6)", slice(4, 7)), ("far(8..11)", slice(8, 12))):
        out[name] = {p: round(float(d[:, sel, i].mean()), 0) for i, p in enumerate(PHASES[:7])}
        out[name][PHASES[7]] = round(float(nxt[:, sel].mean()), 0)
    out["per_wave_chunk_cycles"] = [[int(x) for x in (t[w, 1:, 0] - t[w, :-1, 0])] for w in range(4)]
    print(json.dumps(out), flush=True)
eng.close()
