"""C-ABI checks that need no GPU: the in-tree library loads, exports every symbol include/rk_engine.h
declares, and its host-only helpers agree with the goldens."""
import os
import re

import numpy as np

from conftest import GOLD, REPO


def _header_functions():
    src = open(os.path.join(REPO, "include", "rk_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from llmrankers import _engine
    lib = _engine.load_library()
    declared = _header_functions()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rk_engine.h but not exported"
    assert sorted(_engine.ABI) == declared, "ctypes table and header disagree"
    assert lib.rk_abi_version() == 1


def test_rel_bucket_matches_hf_tables():
    from llmrankers import _engine
    g = np.load(os.path.join(GOLD, "rel_buckets.npz"))
    rel = g["rel"]
    for key in g.files:
        if key == "rel":
            continue
        b, nb, md = key[1:].split("_")
        got = np.array([_engine.rel_bucket(int(r), bool(int(b)), int(nb), int(md)) for r in rel])
        np.testing.assert_array_equal(got, g[key], err_msg=key)


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a gfx950 device engine creation must raise, not degrade."""
    import ctypes
    from llmrankers import _engine, _synth
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        return
    try:
        _engine.RkEngine(_synth.TOY_GATED_UNTIED, device=0, max_tokens=256, max_seqs=4, max_dec_len=4)
    except _engine.RkError as e:
        assert e.code in (-2, -3)
    else:
        raise AssertionError("engine creation succeeded without a GPU")
