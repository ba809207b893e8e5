// L2 -> LDS through the DMA form of a buffer load (buffer_load_dword[x4] ... lds) against the same bytes through VGPRs:
// every wave issues NLOAD wave-wide loads of 64 x BYTES consecutive bytes at scattered bases inside `span`
// (span <= 4 MB: L2 hits; 16 MB+: beyond one XCD's L2).     hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, bool DMA>
__global__ __launch_bounds__(256) void k(const char* p, unsigned* out, int iters, unsigned span) {
  __shared__ __attribute__((aligned(16))) char lds[4][8 * 1024];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, span, 0x00020000);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned wave = (blockIdx.x * 4 + wv) * 977u;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned base = __builtin_amdgcn_readfirstlane(((wave + it * 131u) * 4096u) % (span - 64 * 64 * 16));
    if (DMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if constexpr (BYTES == 16)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds[wv] + i * 1024), 16, lane * 16,
                                                   base + i * 1024, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds[wv] + i * 256), 4, lane * 4,
                                                   base + i * 256, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += *reinterpret_cast<unsigned*>(lds[wv] + lane * 4);
    } else {
      unsigned v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (BYTES == 4) v[i] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * BYTES, base + i * 64 * BYTES, 0);
        else { auto t = __builtin_amdgcn_raw_buffer_load_b128(r, lane * BYTES, base + i * 64 * BYTES, 0); v[i] = t[0] ^ t[1] ^ t[2] ^ t[3]; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i] * (i + 1);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int B, bool DMA> void run(const char* p, unsigned* o, unsigned span, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 256;
  k<B, DMA><<<blocks, 256>>>(p, o, iters, span); hipDeviceSynchronize();
  hipEventRecord(e0); k<B, DMA><<<blocks, 256>>>(p, o, iters, span); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double loads = (double)blocks * 4 * iters * 8;
  printf("  %2d B/lane %s, %4d workgroups: %.2f ns per wave-load per CU (=%.1f cycles at 2.4 GHz), %.2f TB/s\n", B,
         DMA ? "-> LDS (DMA) " : "-> VGPRs     ", blocks, ms * 1e6 / (loads / 256), ms * 1e6 / (loads / 256) * 2.4, loads * 64 * B / ms / 1e9);
}
int main() {
  char* p; unsigned* o; hipMalloc(&p, 64u << 20); hipMalloc(&o, 4096 * 256 * 4); hipMemset(p, 1, 64u << 20);
  for (unsigned span : {2u << 20, 64u << 20}) {
    printf("span %u KB\n", span >> 10);
    for (int blocks : {512, 768, 2048}) {
      run<16, false>(p, o, span, blocks); run<16, true>(p, o, span, blocks);
      run<4, false>(p, o, span, blocks); run<4, true>(p, o, span, blocks);
    }
  }
  return 0;
}
