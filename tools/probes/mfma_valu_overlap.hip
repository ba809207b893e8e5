// Probe: do fp32 MFMAs (16x16x4) and VALU / transcendental work overlap on gfx950?
//   mode 0: MFMA only      mode 1: VALU only (softplus-like: exp + log + 5 simple ops)
//   mode 2: both, interleaved inside every wave (8 MFMA then 8 softplus, independent data)
//   mode 3: both, split across waves (even waves MFMA only, odd waves VALU only; same totals per SIMD pair)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef VALU_KIND
#define VALU_KIND 0
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float sp(float x) {
#if VALU_KIND == 0   // softplus(beta = 100): 2 transcendentals + ~7 simple ops
  const float t = 100.0f * x;
  return t > 20.0f ? x : 0.01f * __logf(1.0f + __expf(t));
#elif VALU_KIND == 1 // 9 dependent fmas
  float y = x;
#pragma unroll
  for (int i = 0; i < 9; ++i) y = fmaf(y, 0.999f, 0.001f);
  return y;
#else                // 2 transcendentals only
  return __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(x));
#endif
}

template <int MODE>
__global__ __launch_bounds__(512, 4) void probe(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc[8];
  float v[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = seed * (lane + i) * 1e-3f; }
  const float a = seed + lane, b = seed * 0.5f;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wv & 4) == 0);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wv & 4) != 0);
  // in mode 3 each wave does twice the iterations of its own kind so that per-SIMD totals equal mode 2
  const int n = MODE == 3 ? 2 * iters : iters;
  for (int it = 0; it < n; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#ifdef MFMA_F16
          f16x8 ah, bh;
          for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)a; bh[j] = (_Float16)b; }
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
#else
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#endif
        }
    }
    if (do_valu) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = sp(v[i] - 0.05f);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(512), 0, 0, out, iters, 0.37f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(512), 0, 0, out, iters, 0.37f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  const int iters = 4000;
  // per SIMD: 512 WGs * 8 waves / (256 CUs * 4 SIMDs) = 4 waves, each: iters * 32 MFMAs of 32 cycles
  const double mfma_cycles = 4.0 * iters * 32 * 32;
  float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters), t3 = run<3>(out, iters);
  printf("MFMA only      %.3f ms  (=> %.2f GHz if the pipe is saturated)\n", t0, mfma_cycles / (t0 * 1e-3) / 1e9);
  printf("VALU only      %.3f ms  (16 softplus / iter / wave)\n", t1);
  printf("both, in-wave  %.3f ms  (sum %.3f, max %.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
  printf("both, by wave  %.3f ms\n", t3);
  return 0;
}
