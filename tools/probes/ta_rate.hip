// TA / L1 cost of a wave-level load by width: every wave issues NLOAD loads whose 64 lanes cover consecutive addresses
// (one 128-B line for 2-byte loads, two for 4-byte, ...), data L2-resident.    hipcc --offload-arch=gfx950 -O3 ta_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int BYTES>
__global__ __launch_bounds__(256) void k(const char* p, unsigned* out, int iters, unsigned span) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, span, 0x00020000);
  const int lane = threadIdx.x & 63;
  unsigned wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    // L1 mode (span <= 32 KB): every wave re-reads one small region; otherwise scattered over `span`
    unsigned base = span <= (32u << 10) ? (it & 1) * 64u : ((wave + it * 131u) * 4096u) % (span - 64 * 64 * 16);
    unsigned v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const unsigned off = span <= (32u << 10) ? base + (i & 3) * 64 * BYTES + lane * BYTES
                                               : base + i * 128 * (BYTES >= 4 ? BYTES / 2 : 1) + lane * BYTES;
      if (BYTES == 2) v[i] = __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
      else if (BYTES == 4) v[i] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
      else if (BYTES == 8) { auto t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); v[i] = t[0] ^ t[1]; }
      else { auto t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); v[i] = t[0] ^ t[1] ^ t[2] ^ t[3]; }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += v[i] * (i + 1);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int B> void run(const char* p, unsigned* o, unsigned span) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 64, blocks = 2048;
  k<B><<<blocks, 256>>>(p, o, iters, span); hipDeviceSynchronize();
  hipEventRecord(e0); k<B><<<blocks, 256>>>(p, o, iters, span); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double loads = (double)blocks * 4 * iters * 32;
  printf("%2d B/lane: %.1f us, %.2f ns per wave-load per CU (=%.1f cycles at 2.4 GHz), %.2f TB/s\n", B, ms * 1e3,
         ms * 1e6 / (loads / 256), ms * 1e6 / (loads / 256) * 2.4, loads * 64 * B / ms / 1e9);
}
int main() {
  char* p; unsigned* o; hipMalloc(&p, 16u << 20); hipMalloc(&o, 2048 * 256 * 4); hipMemset(p, 1, 16u << 20);
  for (unsigned span : {16u << 10, 512u << 10, 2u << 20, 4u << 20, 16u << 20}) {
    printf("span %u KB\n", span >> 10);
    run<2>(p, o, span); run<4>(p, o, span); run<8>(p, o, span); run<16>(p, o, span);
  }
  return 0;
}
