// TA / L1 cost of a 64-lane dword load whose lanes are split into G groups that read 256/G-byte slices of G different
// 256-byte lines (G = 1: one full line - the displacement-major lookup's gather; G = 2 / 4: a wave that serves 32 / 16
// pixels on 2 / 4 pyramid levels at once; the G waves of a workgroup that share a tile read the G slices of the same lines).
//     hipcc --offload-arch=gfx950 -O3 ta_groups.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int G>
__global__ __launch_bounds__(256) void k(const char* p, unsigned* out, int iters, unsigned span) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, span, 0x00020000);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int LPG = 64 / G;                       // lanes per group
  const int g = lane / LPG, l = lane % LPG;
  const int slice = wv % G;                         // which 256/G-byte slice of the lines this wave reads
  const unsigned team = (blockIdx.x * 4 + wv) / G;  // the G waves of a team walk the same lines
  const unsigned region = span / 4;                 // group g reads region g (a "level")
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned v[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) {
      const unsigned line = (team * 2654435761u + (unsigned)(it * 40 + i) * 40503u) % (region / 256);
      const unsigned off = g * region + line * 256u + slice * (256u / G) + l * 4u;
      v[i] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) acc += v[i] * (i + 1);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int G> void run(const char* p, unsigned* o, unsigned span, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 32;
  k<G><<<blocks, 256>>>(p, o, iters, span); hipDeviceSynchronize();
  hipEventRecord(e0); k<G><<<blocks, 256>>>(p, o, iters, span); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double loads = (double)blocks * 4 * iters * 40;
  printf("  G=%d blocks %d: %.1f us, %.2f ns per wave-load per CU (%.1f cycles at 2.4 GHz), useful %.2f TB/s\n", G, blocks,
         ms * 1e3, ms * 1e6 / (loads / 256), ms * 1e6 / (loads / 256) * 2.4, loads * 256 / ms / 1e9);
}
int main() {
  char* p; unsigned* o; const size_t big = 2048u << 20;
  hipMalloc(&p, big); hipMalloc(&o, 8192 * 256 * 4); hipMemset(p, 1, big);
  for (unsigned span : {2u << 20, 64u << 20, 2048u << 20}) {
    printf("span %u MB\n", span >> 20);
    for (int blocks : {512, 2048}) { run<1>(p, o, span, blocks); run<2>(p, o, span, blocks); run<4>(p, o, span, blocks); }
  }
  return 0;
}
