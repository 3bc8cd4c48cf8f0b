// How many bytes per microsecond does ONE CU pull out of L2 when all CUs pull - by LDS-DMA (global_load_lds_dwordx4) and by
// ordinary 16-byte loads into registers?  512-thread workgroups, one per CU (160 KiB of LDS requested), every wave streams a
// 64-KiB region of its workgroup (L2-resident: 32 workgroups x 64 KiB = 2 MiB per XCD) again and again, eight 1-KiB
// instructions in flight per wave.  Also with half / a quarter of the CUs active, and with a busy MFMA loop in four of the
// eight waves (the clock the matrix cores allow).  Every wave records how long IT ran: prints the loader waves' rate and the
// MFMA waves' time separately.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int MFMA>   // MODE 0: LDS-DMA, 1: register loads, 2: LDS-DMA with GEMM-tile addressing (8 rows x 128 B at a 2 KiB stride, XOR-swizzled chunks)
__global__ __launch_bounds__(512) void pull(const char* src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* base = src + (size_t)blockIdx.x * 65536;
  const long long t_begin = wall_clock64();
  f32x16 acc = {};
  half8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.37f + 0.01f * lane); b[i] = (_Float16)(0.11f * (i + 1)); }
  if (MFMA && wave >= 4) {                                    // four waves keep the matrix cores busy for the same number of rounds
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 123.f) sink[0] = acc[1];
    if (lane == 0) sink[16 + blockIdx.x * 8 + wave] = (float)(wall_clock64() - t_begin) * 0.01f;   // microseconds (100 MHz)
    return;
  }
  const int nload = MFMA ? 4 : 8;                             // loader waves
  if (wave >= nload) return;
  u32x4 r[8];
  unsigned keep = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned off = (unsigned)(((it * 8 + u) * nload + wave) & 63) * 1024u + lane * 16u;   // walks the 64-KiB region
      if (MODE == 2) {
        // one instruction = tile rows 8j .. 8j+7 of a [512 rows][1024 halfs] operand slice (row stride 2 KiB), k-step from the round
        const int j = ((it * 8 + u) * nload + wave) & 63, ktile = (it >> 3) & 15;
        const int row = j * 8 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        const unsigned o2 = (unsigned)row * 2048u + (unsigned)ktile * 128u + (unsigned)chunk * 16u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(blockIdx.x & 31) * (512 * 2048) + o2),
                                         (__attribute__((address_space(3))) void*)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
      } else if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                         (__attribute__((address_space(3))) void*)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
      } else {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[u]) : "v"(off), "s"(base) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(r[u])); keep ^= r[u][0]; }
    }
  }
  if (MODE != 1) keep = ((unsigned*)lds)[threadIdx.x];
  if (keep == 0x12345678u) sink[1] = 1.f;
  if (lane == 0) sink[16 + blockIdx.x * 8 + wave] = (float)(wall_clock64() - t_begin) * 0.01f;
}

template <int MODE, int MFMA>
void run(const char* name, const char* src, float* sink, int grid) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)pull<MODE, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((pull<MODE, MFMA>), dim3(grid), dim3(512), 163840, 0, src, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const int nload = MFMA ? 4 : 8;
  const double kb_per_wg = (double)iters * 8 * nload;       // 1 KiB per instruction
  static float host[16 + 256 * 8];
  hipMemcpy(host, sink, sizeof(host), hipMemcpyDeviceToHost);
  double tl = 0, tm = 0;
  for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) (w < nload ? tl : tm) += host[16 + b * 8 + w];
  tl /= (double)grid * nload; if (MFMA) tm /= (double)grid * 4;
  printf("%-28s grid %3d: kernel %7.1f us; loader waves %7.1f us = %6.1f KB/us per CU (%5.2f TB/s aggregate)", name, grid, ms * 1e3, tl,
         kb_per_wg / tl, kb_per_wg * 1024 * grid / (tl * 1e-6) / 1e12);
  if (MFMA) printf("; MFMA waves %7.1f us", tm);
  printf("\n");
}

int main() {
  char* src; float* sink;
  hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20); hipMalloc(&sink, (16 + 256 * 8) * 4);
  for (int grid : {256, 128, 64}) {
    run<0, 0>("LDS-DMA, 8 loader waves", src, sink, grid);
    run<2, 0>("tile-addressed DMA, 8 waves", src, sink, grid);
    run<2, 1>("tile-addressed DMA 4 + MFMA 4", src, sink, grid);
    run<1, 0>("register loads, 8 waves", src, sink, grid);
    run<0, 1>("LDS-DMA 4 waves + MFMA 4", src, sink, grid);
    run<1, 1>("register loads 4 + MFMA 4", src, sink, grid);
  }
  printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
