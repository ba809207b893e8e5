// First convolution of the flow encoder (scope row A4): 7x7, zero padding 3, 4 -> 128 channels,
// + bias + ReLU  (/root/reference/src/modules/droid_net/droid_net.py:79-83).
//
// With only 4 input channels a GEMM library sees K = 196 and pads / transposes its way to 90 us
// for 10 GFLOP.  Here a kernel ROW of the stencil is one MFMA K-slice: 7 taps x 4 channels = 28
// values, padded to 8 taps = 32 -- and in a channels-last fp32 motion map [pixel][4] those 32
// values are 128 contiguous bytes.  A lane builds its v_mfma_f32_16x16x32_f16 pixel fragment with
// two 16-byte loads (2 taps x 4 channels) and a conversion; the [128 x 224] weight panel sits in
// LDS (row stride padded to 232 halfs: conflict-free b128 reads) and is the MFMA "A" operand, so a
// lane ends up with 4 consecutive output channels of one pixel (8-byte stores).
// A wave owns 16 pixels x 128 channels (8 accumulator blocks), 7 K-slices = 56 MFMAs per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kFlowK = 224;       // 7 rows x (8 taps x 4 channels)
constexpr int kFlowLd = 232;      // LDS row stride in halfs

__global__ __launch_bounds__(256) void flow_conv7_kernel(const float* __restrict__ flow,
                                                         const _Float16* __restrict__ wp,
                                                         const float* __restrict__ bias,
                                                         _Float16* __restrict__ out, int os, long P, int H,
                                                         int W) {
  __shared__ __attribute__((aligned(16))) _Float16 wl[128 * kFlowLd];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  for (int i = tid; i < 128 * (kFlowK / 8); i += 256) {
    const int n = i / (kFlowK / 8), c8 = i - n * (kFlowK / 8);
    *reinterpret_cast<f16x8*>(wl + n * kFlowLd + c8 * 8) = *reinterpret_cast<const f16x8*>(wp + n * kFlowK + c8 * 8);
  }
  __syncthreads();
  float4 b[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) b[mi] = *reinterpret_cast<const float4*>(bias + mi * 16 + kg * 4);

  const long ntiles = (P + 15) / 16;
  for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long p = tile * 16 + col;
    const bool pv = p < P;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    f32x4 acc[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int x0 = x - 3 + 2 * kg;                 // this lane's two taps of every stencil row
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int yy = y + ky - 3;
      const bool rv = pv && (unsigned)yy < (unsigned)H;
      const float* src = flow + (p + (long)(ky - 3) * W - 3 + 2 * kg) * 4;
      float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
      if (rv && (unsigned)x0 < (unsigned)W) f0 = *reinterpret_cast<const float4*>(src);
      if (rv && (unsigned)(x0 + 1) < (unsigned)W) f1 = *reinterpret_cast<const float4*>(src + 4);
      const f16x8 xf = {(_Float16)f0.x, (_Float16)f0.y, (_Float16)f0.z, (_Float16)f0.w,
                        (_Float16)f1.x, (_Float16)f1.y, (_Float16)f1.z, (_Float16)f1.w};
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const f16x8 wf = *reinterpret_cast<const f16x8*>(wl + (mi * 16 + col) * kFlowLd + ky * 32 + kg * 8);
        acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[mi], 0, 0, 0);
      }
    }
    if (pv) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        f16x4 o;
        o[0] = (_Float16)fmaxf(acc[mi][0] + b[mi].x, 0.0f);
        o[1] = (_Float16)fmaxf(acc[mi][1] + b[mi].y, 0.0f);
        o[2] = (_Float16)fmaxf(acc[mi][2] + b[mi].z, 0.0f);
        o[3] = (_Float16)fmaxf(acc[mi][3] + b[mi].w, 0.0f);
        *reinterpret_cast<f16x4*>(out + p * os + mi * 16 + kg * 4) = o;
      }
    }
  }
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_flow_conv7(const float* flow, const void* w_packed, const float* bias, void* out,
                                 int out_stride, int N, int H, int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || (out_stride & 3) || out_stride < 128) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!flow || !w_packed || !bias || !out) return GLORIE_EINVAL;
  const long P = (long)N * H * W, ntiles = (P + 15) / 16;
  // 2 workgroups per CU (59 KB of LDS each): the weight panel is loaded once per workgroup, so few,
  // long-running workgroups amortise it
  const unsigned grid = (unsigned)((ntiles + 3) / 4 < 512 ? (ntiles + 3) / 4 : 512);
  hipLaunchKernelGGL(flow_conv7_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, flow,
                     reinterpret_cast<const _Float16*>(w_packed), bias, reinterpret_cast<_Float16*>(out),
                     out_stride, P, H, W);
  return check_launch();
}
