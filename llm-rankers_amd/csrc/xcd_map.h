// Workgroup -> (group, item) mapping of the attention kernels whose workgroups SHARE data per group (attn_causal128_dma_kernel:
// group = (sequence, kv head), items = heads-per-kv-head x query blocks; attn_enc_long_kernel: group = (sequence, head), items =
// query blocks).  Consecutive workgroups of a 1-D grid go to the 8 XCDs of an MI355X in turn, each XCD with its own L2: workgroup
// i therefore takes group 8 (i / 8 / W) + i % 8, item (i / 8) % W - all W workgroups of a group run on ONE XCD, in item order,
// and read the group's K / V rows through one L2.  The grid is padded to whole rows of 8 groups; workgroups of the padding
// return at once.  Plain integer arithmetic shared with a host-side test (tests/test_xcd_map.py compiles this header with g++:
// every (group, item) exactly once, a group's workgroups all congruent mod 8).
#pragma once

#if defined(__HIPCC__)
#define RK_XCD_HD __host__ __device__ __forceinline__
#else
#define RK_XCD_HD inline
#endif

#define RK_XCDS 8

// workgroups to launch for `groups` groups of W items
RK_XCD_HD unsigned xcd_grid(int groups, int W) { return (unsigned)((groups + RK_XCDS - 1) / RK_XCDS * RK_XCDS) * (unsigned)W; }
// workgroup i -> group / item; false: a workgroup of the padding
RK_XCD_HD bool xcd_decode(int i, int groups, int W, int& group, int& item) {
  group = (i >> 3) / W * RK_XCDS + (i & 7);
  item = (i >> 3) % W;
  return group < groups;
}
