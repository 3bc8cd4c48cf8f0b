"""pytest wiring: `gpu` marker, import paths, and fixture-checkpoint materialisation.

tests/golden/ckpts.json holds regeneration recipes (dims, seed, gain, boost) — the toy weights are
recreated bit-identically from llmrankers._synth and verified against the recorded sha256."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
for p in (os.path.join(REPO, "llm-rankers_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLD


@pytest.fixture(scope="session")
def ckpt_dirs(tmp_path_factory):
    """name -> HF-layout checkpoint dir (config.json, model.safetensors, tokenizer files)."""
    from llmrankers import _synth
    with open(os.path.join(GOLD, "ckpts.json")) as f:
        specs = json.load(f)
    root = tmp_path_factory.mktemp("ckpts")
    out = {}
    for name, spec in specs.items():
        path = str(root / name)
        _synth.write_checkpoint(path, spec, os.path.join(GOLD, spec.get("tokenizer", "tok")))
        assert _synth.checkpoint_sha256(path) == spec["sha256"], f"{name}: regenerated weights differ from goldens"
        out[name] = path
    return out


def load_state(path):
    from safetensors.numpy import load_file
    from llmrankers import _synth
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    dims = _synth.LlamaDims.from_hf_config(cfg) if cfg.get("model_type") == "llama" else _synth.T5Dims.from_hf_config(cfg)
    return dims, load_file(os.path.join(path, "model.safetensors"))
