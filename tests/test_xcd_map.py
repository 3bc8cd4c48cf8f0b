"""Workgroup -> (group, item) mapping of the XCD-aware attention launches (llm-rankers_amd/csrc/xcd_map.h), checked on the host:
the header is plain integer arithmetic shared with the kernels (attn_causal128_dma_kernel, attn_enc_long_kernel), compiled here
with g++.  For every (groups, W): the grid covers every (group, item) exactly once, the padding workgroups are rejected, all
workgroups of one group are congruent mod 8 (= run on one XCD under the round-robin dispatch) and come in item order."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#include "xcd_map.h"
int main(int argc, char** argv) {
  const int groups = atoi(argv[1]), W = atoi(argv[2]);
  const unsigned n = xcd_grid(groups, W);
  if (n % (RK_XCDS * W) || n < (unsigned)groups * W || n >= (unsigned)(groups + RK_XCDS) * W) { printf("grid %u\n", n); return 1; }
  std::set<long> seen;
  std::map<int, int> xcd_of, last_item;
  unsigned padding = 0;
  for (unsigned i = 0; i < n; ++i) {
    int g, w;
    if (!xcd_decode((int)i, groups, W, g, w)) { ++padding; continue; }
    if (g < 0 || g >= groups || w < 0 || w >= W) { printf("range (%d, %d) at %u\n", g, w, i); return 1; }
    if (!seen.insert((long)g * W + w).second) { printf("duplicate (%d, %d) at %u\n", g, w, i); return 1; }
    if (xcd_of.count(g) && xcd_of[g] != (int)(i % RK_XCDS)) { printf("group %d on two XCDs\n", g); return 1; }
    xcd_of[g] = (int)(i % RK_XCDS);
    if (last_item.count(g) && last_item[g] + 1 != w) { printf("group %d: item %d after %d\n", g, w, last_item[g]); return 1; }
    last_item[g] = w;
  }
  if ((long)seen.size() != (long)groups * W || padding != n - (unsigned)groups * W) { printf("coverage %zu, padding %u\n", seen.size(), padding); return 1; }
  printf("ok %u\n", n);
  return 0;
}
"""


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("xcd")
    (d / "t.cpp").write_text(SRC)
    exe = d / "t"
    subprocess.run([gxx, "-O1", "-std=c++17", "-I", os.path.join(REPO, "llm-rankers_amd", "csrc"), str(d / "t.cpp"), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("groups,W", [(1, 1), (1, 13), (2, 48), (8, 48), (9, 48), (32, 48), (36, 7), (104, 24), (128, 13), (5, 1), (16, 3), (255, 2)])
def test_every_group_item_once_and_a_group_on_one_xcd(prog, groups, W):
    r = subprocess.run([prog, str(groups), str(W)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
