"""bench.py's multi-rank control flow without a GPU: `--gpus N --dry_ranks` spawns N gloo ranks over tools/dry_engine.py and walks
communicator bring-up (the SAME T5Runtime.comm_init_from_process_group the rankers use), sharding, one gather per launch
sequence, the gather checks and the JSON line - for N in {2, 4, 8} in both modes, so that the first real 8-GPU run of the
driver is not a debugging session.  (The engine-side RCCL calls themselves run with one rank in the GPU suite.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


@pytest.mark.parametrize("mode", ["weak", "shard"])
@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_dry_ranks(n, mode):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--dry_ranks", "--steps", "4", "--warmup", "2",
                        "--regions", "2", "--mode", mode], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in j and isinstance(j[k], t), (k, j.get(k))
    assert j["vs_baseline"] is None and j["n_gpus"] == n and j["steps"] == 4 and j["warmup"] == 2 and j["value"] > 0
    assert j["scaling"] == ("weak" if mode == "weak" else "strong") and "DRY RUN" in j["data"]
    cfg = j["config"]
    assert cfg["mode"] == mode and cfg["gather_check"] and cfg["rccl"] and len(cfg["timed_regions_ms"]) == 2
    if mode == "shard":
        import re
        shares = [int(x) for x in re.search(r"\((\d+(?:, \d+)*)\)", cfg["workload"]).group(1).split(",")]
        assert sum(shares) == 100 and len(shares) == n and max(shares) - min(shares) <= 1
        assert cfg["global_batch"] == 100
    else:
        assert cfg["global_batch"] == 32 * n
