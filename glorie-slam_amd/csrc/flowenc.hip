// First convolution of the flow encoder (scope row A4): 7x7, zero padding 3, 4 -> 128 channels,
// + bias + ReLU  (/root/reference/src/modules/droid_net/droid_net.py:79-83).
//
// With only 4 input channels a GEMM library sees K = 196 and pads / transposes its way to 90 us
// for 10 GFLOP.  Here a kernel ROW of the stencil is one MFMA K-slice: 7 taps x 4 channels = 28
// values, padded to 8 taps = 32 -- and in a channels-last fp32 motion map [pixel][4] those 32
// values are 128 contiguous bytes.  A lane builds its v_mfma_f32_16x16x32_f16 pixel fragment with
// two 16-byte loads (2 taps x 4 channels) and a conversion; the weights are the MFMA "A" operand,
// so a lane ends up with 4 consecutive output channels of one pixel (8-byte stores).  A workgroup
// walks 16-pixel tiles; its 4 waves split the 128 output channels and keep their 32 x 224 slice of
// the weight panel in registers (a first version staged the panel in LDS and re-read it per tile:
// 56 exposed LDS round trips per tile, 60 us; this one: see profiles).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kFlowK = 224;       // 7 rows x (8 taps x 4 channels)

// 4 waves per workgroup; wave w owns output channels 32w .. 32w+31 of EVERY pixel tile of the
// workgroup and keeps its slice of the weight panel (2 blocks x 7 K slices = 14 fragments, 56 VGPRs)
// in registers for the whole kernel: no LDS, no barrier, occupancy limited by registers only.
__global__ __launch_bounds__(256) void flow_conv7_kernel(const float* __restrict__ flow,
                                                         const _Float16* __restrict__ wp,
                                                         const float* __restrict__ bias,
                                                         _Float16* __restrict__ out, int os, long P, int H,
                                                         int W) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  f16x8 wf[2][7];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
      wf[mi][ky] = *reinterpret_cast<const f16x8*>(wp + (size_t)(wv * 32 + mi * 16 + col) * kFlowK + ky * 32 + kg * 8);
  float4 b[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) b[mi] = *reinterpret_cast<const float4*>(bias + wv * 32 + mi * 16 + kg * 4);

  const int ntiles = (int)((P + 15) / 16);       // P < 2^31 (checked by the host)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p = tile * 16 + col;
    const bool pv = p < (int)P;
    const int rowi = p / W;
    const int x = p - rowi * W, y = rowi % H;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int x0 = x - 3 + 2 * kg;                 // this lane's two taps of every stencil row
    const bool c0 = (unsigned)x0 < (unsigned)W, c1 = (unsigned)(x0 + 1) < (unsigned)W;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int yy = y + ky - 3;
      const bool rv = pv && (unsigned)yy < (unsigned)H;
      const float* src = flow + ((long)p + (ky - 3) * W - 3 + 2 * kg) * 4;
      // unconditional loads from a clamped address + select: the loads of all 7 stencil rows can be
      // in flight together
      const bool v0 = rv && c0, v1 = rv && c1;
      const float4 f0 = *reinterpret_cast<const float4*>(v0 ? src : flow);
      const float4 f1 = *reinterpret_cast<const float4*>(v1 ? src + 4 : flow);
      // (component-wise selects: a float4 ternary is lowered to a select through scratch memory)
      const f16x8 xf = {(_Float16)(v0 ? f0.x : 0.f), (_Float16)(v0 ? f0.y : 0.f), (_Float16)(v0 ? f0.z : 0.f),
                        (_Float16)(v0 ? f0.w : 0.f), (_Float16)(v1 ? f1.x : 0.f), (_Float16)(v1 ? f1.y : 0.f),
                        (_Float16)(v1 ? f1.z : 0.f), (_Float16)(v1 ? f1.w : 0.f)};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][ky], xf, acc[mi], 0, 0, 0);
    }
    if (pv) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f16x4 o;
        o[0] = (_Float16)fmaxf(acc[mi][0] + b[mi].x, 0.0f);
        o[1] = (_Float16)fmaxf(acc[mi][1] + b[mi].y, 0.0f);
        o[2] = (_Float16)fmaxf(acc[mi][2] + b[mi].z, 0.0f);
        o[3] = (_Float16)fmaxf(acc[mi][3] + b[mi].w, 0.0f);
        *reinterpret_cast<f16x4*>(out + (long)p * os + wv * 32 + mi * 16 + kg * 4) = o;
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
  if (P >= 0x7fffffffL) return GLORIE_EUNSUPPORTED;
  const unsigned grid = (unsigned)(ntiles < 1024 ? ntiles : 1024);   // 4 workgroups per CU: ~10 tiles each amortise the weight load
  hipLaunchKernelGGL(flow_conv7_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, flow,
                     reinterpret_cast<const _Float16*>(w_packed), bias, reinterpret_cast<_Float16*>(out),
                     out_stride, P, H, W);
  return check_launch();
}
