// Residency census: how many 384-thread workgroups share a CU, as a function of dynamic LDS bytes and VGPR count?
// Each workgroup spins ~40 us; the grid is 2 x CUs; wall time ~40 us => two per CU, ~80 us => one per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VG>
__global__ __launch_bounds__(384, 3) void spin(long long cycles, int* out) {
  extern __shared__ int lds[];
  if (VG >= 160) asm volatile("v_mov_b32 v160, 0" ::: "v160");
  if (VG >= 166) asm volatile("v_mov_b32 v165, 0" ::: "v165");
  if (VG >= 128 && VG < 160) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  lds[threadIdx.x] = threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (lds[threadIdx.x] == -1) out[0] = 1;
}
template <int VG>
void run(int lds, int threads) {
  hipFuncSetAttribute((const void*)spin<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int* d; hipMalloc(&d, 4);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<VG>, threads, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(spin<VG>, dim3(512), dim3(threads), lds, 0, 4000000LL / 100, d);   // 40k ticks of the 100 MHz counter = 400 us? see ratio
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms1; hipEventElapsedTime(&ms1, a, b);
  hipEventRecord(a);
  hipLaunchKernelGGL(spin<VG>, dim3(256), dim3(threads), lds, 0, 4000000LL / 100, d);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms0; hipEventElapsedTime(&ms0, a, b);
  printf("VGPR>=%3d threads %d LDS %6d B: api occupancy %d, time 256 WGs %.1f us, 512 WGs %.1f us -> ratio %.2f (%s)\n", VG, threads, lds, occ, ms0 * 1e3, ms1 * 1e3,
         ms1 / ms0, ms1 / ms0 < 1.5 ? "two per CU" : "ONE per CU");
  hipFree(d);
}
int main() {
  for (int lds : {16384, 49152, 65536, 76800, 81920}) { run<64>(lds, 384); run<128>(lds, 384); run<160>(lds, 384); run<166>(lds, 384); }
  run<166>(76800, 256); run<166>(76800, 320); run<166>(65536, 256);
  printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
