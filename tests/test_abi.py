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


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """The boundary is a C ABI: include/rk_engine.h compiles as C99 (no C++ or torch types), and a C translation unit that
    takes the address of every declared entry point links against the in-tree library - what a cgo / JNI / ctypes binding
    on the reference's side needs.  Nothing is executed (no GPU here)."""
    import shutil
    import subprocess
    import __graft_entry__ as g
    g.build()
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("gcc not available")
    inc = os.path.join(REPO, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "rk_engine.h")], check=True)
    names = _header_functions()
    src = tmp_path / "bind.c"
    src.write_text('#include "rk_engine.h"\n#include <stdio.h>\nint main(void) {\n  const void* f[] = {\n'
                   + "".join(f"    (const void*)&{n},\n" for n in names)
                   + '  };\n  printf("%d entry points\\n", (int)(sizeof f / sizeof f[0]));\n  return f[0] == 0;\n}\n')
    obj = tmp_path / "bind.o"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-Wno-pedantic", "-I", inc, "-c", str(src), "-o", str(obj)], check=True)
    out = subprocess.run(["nm", "-u", str(obj)], check=True, capture_output=True, text=True).stdout
    undefined = {line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("rk_")}
    assert undefined == set(names)
    lib = os.path.join(REPO, "llm-rankers_amd", "lib", "librk_engine.so")
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in exported.splitlines() if line.split()}
    assert undefined <= exported
