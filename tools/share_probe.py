#!/usr/bin/env python
"""One rank's 13-passage share per step (bench.py shard_share_leg) under engine options: RK_OPTS sets, ';'-separated."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
import torch  # noqa
import bench
from llmrankers import _synth
from llmrankers._engine import RkEngine
dims = _synth.FLAN_T5_LARGE
eng = RkEngine(dims, device=0, max_tokens=32 * 184, max_seqs=32, max_dec_len=4)
eng.load_state(_synth.synth_state_dict(dims, seed=929, threads=32).items())
DEF = {"overlap": 1, "dec_graph": 1, "gemm_variant": 0}
for cfg in os.environ.get("RK_OPTS", "").split(";"):
    opts = {k: int(v) for k, v in (x.split("=") for x in cfg.split(",") if x)}
    for k, v in {**DEF, **opts}.items():
        eng.set_option(k, v)
    r = bench.shard_share_leg(eng, dims, 184, steps=int(os.environ.get("RK_STEPS", "40")))
    print(json.dumps({"opts": cfg, "ms_per_step": r["ms_per_step"], "frac": r["frac_of_mfma_peak_per_gpu"]}), flush=True)
