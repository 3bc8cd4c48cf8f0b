"""Tile order of the chained GEMM launch (llm-rankers_amd/csrc/gemm_chain_sched.h), checked on the host: the header is plain
integer arithmetic shared with the kernel, compiled here with g++.  For every shape: each queue lists each of its tiles
exactly once, the queues partition the output, and every consumer tile comes after all producer tiles of its row panel IN ITS
OWN QUEUE (so one workgroup pulling a queue alone could finish it: no residency assumption, no deadlock), by at least the
promised lead."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include "gemm_chain_sched.h"
int main(int argc, char** argv) {
  const int tiles_m = atoi(argv[1]), NP = atoi(argv[2]), NC = atoi(argv[3]), lead = atoi(argv[4]);
  std::set<int> seen;
  long min_gap = 1 << 30;
  for (int q = 0; q < CHAIN_QUEUES; ++q) {
    int p0, np; chain_queue(q, tiles_m, p0, np);
    const int len = chain_queue_len(q, tiles_m, NP, NC);
    if (len != np * (NP + NC)) { printf("len\n"); return 1; }
    std::vector<int> done(np, 0), last(np, -1);
    for (int u = 0; u < len; ++u) {
      const int c = chain_decode(q, u, tiles_m, NP, NC, lead);
      const int tm = chain_tm(c), tn = chain_tn(c);
      if (tm < p0 || tm >= p0 + np) { printf("panel %d outside queue %d at %d\n", tm, q, u); return 1; }
      if (tn < 0 || tn >= (chain_is_prod(c) ? NP : NC)) { printf("column\n"); return 1; }
      if (!seen.insert(c).second) { printf("duplicate tile %x in queue %d at %d\n", c, q, u); return 1; }
      if (chain_is_prod(c)) { done[tm - p0]++; last[tm - p0] = u; }
      else {
        if (done[tm - p0] != NP) { printf("consumer (%d,%d) of queue %d at %d before its producers (%d of %d)\n", tm, tn, q, u, done[tm - p0], NP); return 1; }
        if (u - last[tm - p0] < min_gap) min_gap = u - last[tm - p0];
      }
    }
  }
  if ((long)seen.size() != (long)tiles_m * (NP + NC)) { printf("coverage %zu\n", seen.size()); return 1; }
  printf("ok min_gap %ld\n", min_gap);
  return 0;
}
"""


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("chain")
    (d / "t.cpp").write_text(SRC)
    exe = d / "t"
    subprocess.run([gxx, "-O1", "-std=c++17", "-I", os.path.join(REPO, "llm-rankers_amd", "csrc"), str(d / "t.cpp"), "-o", str(exe)], check=True)
    return str(exe)


# (row panels, producer column tiles, consumer column tiles): flan-t5-large O -> FFN-in (4, 22) and FFN-out -> QKV (4, 12) at the
# bench's M = 58 880 (230 panels) and one query (72), flan-t5-xl (8, 40 / 8, 24), relu models (4, 11), tiny and degenerate shapes
@pytest.mark.parametrize("tiles_m,NP,NC", [(230, 4, 22), (230, 4, 12), (72, 4, 22), (72, 4, 12), (115, 8, 40), (64, 8, 24), (230, 4, 11),
                                            (7, 4, 22), (8, 4, 12), (9, 2, 3), (1, 4, 22), (33, 1, 1), (17, 3, 9), (40, 4, 8)])
@pytest.mark.parametrize("lead", [1, 2, 3])
def test_every_queue_is_a_dependency_ordered_list_of_its_tiles(prog, tiles_m, NP, NC, lead):
    out = subprocess.run([prog, str(tiles_m), str(NP), str(NC), str(lead)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout
    gap = int(out.stdout.split()[-1])
    if tiles_m >= 8 * 3 * 4:
        # the tightest spot is the start of a queue: the first consumer tiles follow the `lead` producer tiles directly, i.e. a
        # consumer starts at least (lead - 1) blocks of PRODUCER tiles behind the last producer tile of its panel
        assert gap >= (lead - 1) * 4 * NP + 1, gap
