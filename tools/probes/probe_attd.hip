// Hardware probes for the addressing assumptions of attn_enc_dma_kernel (tools/probes/run_probe.sh builds + runs it):
//  1. ds_read_b64_tr_b16 with per-lane addresses: which element does lane l, register element j receive?
//  2. v_permlane32_swap: which half-wave gets what?
//  3. global_load_lds_dwordx4: where do the 64 lanes' 16-byte pieces land?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half_t;
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define SWZ(r) (((((r) >> 1) & 1) << 2) | ((((r) >> 3) & 1) << 1) | (((r) >> 2) & 1))

// V image [64 keys][64 d] halfs with the kernel's swizzle; value(key, d) = key * 64 + d (exact in fp16 up to 2048: use key<32)
__global__ void probe_tr(float* out) {
  __shared__ __attribute__((aligned(16))) half_t sV[64 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 64; i += 64) {
    const int key = i >> 6, d = i & 63;
    const int c = (d >> 3) ^ SWZ(key);
    sV[key * 64 + c * 8 + (d & 7)] = (half_t)(float)((key & 31) * 64 + d);
  }
  __syncthreads();
  const int hh = lane >> 5, i16 = lane & 15, g1 = (lane >> 4) & 1;
  const int vfo0 = (4 * hh + (i16 >> 2)) * 64 + (((2 * g1 + ((i16 & 3) >> 1)) ^ ((((i16 >> 3) & 1) << 2) | hh)) << 3) + 4 * (i16 & 1);
  // the four reads of k16 step 0: {o0 first, o0 second, o1 first, o1 second}
  for (int v = 0; v < 4; ++v) {
    const half_t* a = sV + (v & 1) * 8 * 64 + (vfo0 ^ (16 * v));
    half4 r = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)a));
    for (int j = 0; j < 4; ++j) out[(v * 64 + lane) * 4 + j] = (float)r[j];
  }
}

__global__ void probe_swap(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane * 2] = r[0]; out[lane * 2 + 1] = r[1];
}

__global__ void probe_dma(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned s[2 * 256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 512; i += blockDim.x) s[i] = 0xdeadbeefu;
  __syncthreads();
  // lane l fetches the 16 bytes at src + (63 - l) * 16 (reversed), destination base = s + wave * 256 dwords
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)src + wave * 1024 + (63 - lane) * 16),
                                   (__attribute__((address_space(3))) void*)(s + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = s[i];
}

int main() {
  float* d; hipMalloc(&d, 4 * 64 * 4 * sizeof(float));
  probe_tr<<<1, 64>>>(d);
  std::vector<float> h(4 * 64 * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int v = 0; v < 4; ++v)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        const int hh = lane >> 5, l31 = lane & 31;
        const int key = (v & 1) * 8 + 4 * hh + j, dd = (v >> 1) * 32 + l31;
        const float want = (float)(key * 64 + dd), got = h[(v * 64 + lane) * 4 + j];
        if (want != got) { if (bad < 24) printf("tr mismatch v=%d lane=%d j=%d want key=%d d=%d got key=%d d=%d\n", v, lane, j, key, dd, (int)got / 64, (int)got % 64); ++bad; }
      }
  printf("probe_tr: %d mismatches of %d\n", bad, 4 * 64 * 4);
  if (bad) { for (int lane = 0; lane < 64; lane += 1) { printf("lane %2d:", lane); for (int j = 0; j < 4; ++j) { int g = (int)h[lane * 4 + j]; printf(" (k%d,d%d)", g / 64, g % 64); } printf("\n"); } }

  unsigned* du; hipMalloc(&du, 128 * 4);
  probe_swap<<<1, 64>>>(du);
  std::vector<unsigned> hu(128);
  hipMemcpy(hu.data(), du, 128 * 4, hipMemcpyDeviceToHost);
  printf("swap: lane0 -> (%u,%u) lane31 -> (%u,%u) lane32 -> (%u,%u) lane63 -> (%u,%u)   [a=1000+lane, b=2000+lane]\n",
         hu[0], hu[1], hu[62], hu[63], hu[64], hu[65], hu[126], hu[127]);

  unsigned *ds, *dout; hipMalloc(&ds, 2048); hipMalloc(&dout, 2048);
  std::vector<unsigned> hs(512); for (int i = 0; i < 512; ++i) hs[i] = i;
  hipMemcpy(ds, hs.data(), 2048, hipMemcpyHostToDevice);
  probe_dma<<<1, 128>>>(ds, dout);
  std::vector<unsigned> ho(512);
  hipMemcpy(ho.data(), dout, 2048, hipMemcpyDeviceToHost);
  int badd = 0;
  for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
    const unsigned want = w * 256 + (63 - l) * 4 + k, got = ho[w * 256 + l * 4 + k];
    if (want != got) { if (badd < 8) printf("dma mismatch wave %d lane %d dword %d want %u got %u\n", w, l, k, want, got); ++badd; }
  }
  printf("probe_dma: %d mismatches of 512\n", badd);
  printf("hip error: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
