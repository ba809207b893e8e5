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
#include <stdlib.h>
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

// Second form (the one launched): the motion map as ZERO-PADDED fp16, [N][H+6][W+8][4] halfs (3 zero rows above and below
// every map, 3 zero pixels left and 5 right; written by glorie_motion_padded or glorie_flow_pad).  The first form is bound
// by its vector ALU and load-path work, not by the 14 MFMAs of a tile: address arithmetic, zero-padding selects and
// fp32 -> fp16 conversions of the pixel fragment, repeated by all four channel-slice waves, and 14 KB of 16-byte loads per
// wave and tile through the texture path (stand-alone ablation at 36x60x80: 15 us of ALU + MFMA, +12 loads, +20 stores).  Here
//  * a lane's two taps x 4 channels of a stencil row are 16 contiguous bytes of fp16 that ARE its MFMA B fragment: one
//    load per stencil row, no conversion, no select - the padding supplies the zeros;
//  * a wave owns 64 output channels (4 blocks x 7 K slices = 28 weight fragments, 112 VGPRs) and the wave PAIRS of a
//    workgroup walk different pixel tiles: half the redundant loads;
//  * the pixel position advances incrementally from tile to tile (no division in the loop), the loads of tiles t+1 and
//    t+2 are in flight during the MFMAs of tile t;
//  * the MFMA rows are assigned to channels so that a lane ends up with two runs of 8 consecutive channels (row i of block
//    mi is channel 32*(mi/2) + 8*(i/4) + 4*(mi%2) + i%4 of the wave's 64): two 16-byte stores per lane, the four lanes of
//    a pixel cover 64 contiguous bytes with each; the stores are unconditional buffer stores (pixels past the end are
//    dropped by the range check) so that the compiler can count them instead of waiting for vmcnt(0).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
constexpr int kFlowPadY = 3, kFlowPadL = 3, kFlowPadW = 8;     // rows above/below, pixels left, extra pixels per row

__global__ __launch_bounds__(256, 2) void flow_conv7_padded_kernel(const _Float16* __restrict__ fp,
                                                                   const _Float16* __restrict__ wp,
                                                                   const float* __restrict__ bias,
                                                                   _Float16* __restrict__ out, int os, long P, int H,
                                                                   int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int ch0 = (wv & 1) * 64;
  f16x8 wf[4][7];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int ch = ch0 + (mi >> 1) * 32 + (col >> 2) * 8 + (mi & 1) * 4 + (col & 3);       // MFMA row `col` of block mi
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
      wf[mi][ky] = *reinterpret_cast<const f16x8*>(wp + (size_t)ch * kFlowK + ky * 32 + kg * 8);
  }
  float4 b[4];                                       // the lane's channels ch0 + 32*(mi/2) + 8*kg + 4*(mi%2) .. +3
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
    b[mi] = *reinterpret_cast<const float4*>(bias + ch0 + (mi >> 1) * 32 + kg * 8 + (mi & 1) * 4);

  const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)fp, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
  const int ntiles = (int)((P + 15) / 16);
  const int stride = gridDim.x * 2;
  const int WP = W + kFlowPadW, HP = H + 2 * kFlowPadY;
  // tile -> tile + stride moves the pixel by stride*16 = (dn*H + dy)*W + dx
  const int step = stride * 16;
  const int drow = step / W, dx = step - drow * W, dn = drow / H, dy = drow - dn * H;

  int tile = blockIdx.x * 2 + (wv >> 1);
  if (tile >= ntiles) return;
  int p = tile * 16 + col;
  int x, y, n;
  {
    const int pc = min(p, (int)P - 1);               // lanes past the end read (and drop) the last pixel
    const int rowi = pc / W;
    x = pc - rowi * W; n = rowi / H; y = rowi - n * H;
  }
  // byte offset of the lane's taps (x - 3 + 2kg, x - 2 + 2kg) of stencil row 0 (= map row y - 3): padded row y, pixel x + 2kg
  auto offset = [&]() { return (unsigned)((((n * HP + y) * WP) + x + 2 * kg) * 8); };
  auto issue = [&](u32x4_ (&r)[7]) {
    const unsigned off = offset();
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) r[ky] = __builtin_amdgcn_raw_buffer_load_b128(rF, off, ky * WP * 8, 0);
  };
  auto advance = [&]() {
    p += step;
    x += dx; y += dy; n += dn;
    if (x >= W) { x -= W; y += 1; }
    if (y >= H) { y -= H; n += 1; }
    if (p >= (int)P) { x = 0; y = 0; n = 0; }        // past the end: any valid address
  };
  auto finish = [&](int pp, const u32x4_ (&r)[7]) {
    f32x4 acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const f16x8 xf = __builtin_bit_cast(f16x8, r[ky]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][ky], xf, acc[mi], 0, 0, 0);
    }
    f16x8 o[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      o[mi >> 1][(mi & 1) * 4 + 0] = (_Float16)fmaxf(acc[mi][0] + b[mi].x, 0.0f);
      o[mi >> 1][(mi & 1) * 4 + 1] = (_Float16)fmaxf(acc[mi][1] + b[mi].y, 0.0f);
      o[mi >> 1][(mi & 1) * 4 + 2] = (_Float16)fmaxf(acc[mi][2] + b[mi].z, 0.0f);
      o[mi >> 1][(mi & 1) * 4 + 3] = (_Float16)fmaxf(acc[mi][3] + b[mi].w, 0.0f);
    }
    const unsigned so = pp < (int)P ? (unsigned)(((long)pp * os + ch0 + kg * 8) * 2) : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, o[0]), rO, so, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, o[1]), rO, so, 64, 0);
  };

  // three register sets in a ring: the taps of the next TWO tiles are in flight while this one runs its MFMAs (one tile
  // of MFMAs is ~0.25 us, a load round trip ~1 us; two waves per SIMD).  The loads past the last tile are issued too
  // (valid address, result unused): no branch, no register copies around it.
  u32x4_ r[3][7];
  int pq[3];
  pq[0] = p; issue(r[0]); advance();
  pq[1] = p; issue(r[1]); advance();
  while (true) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      constexpr int nx[3] = {2, 0, 1};
      pq[nx[k]] = p; issue(r[nx[k]]); advance();
      finish(pq[k], r[k]);
      tile += stride;
      if (tile >= ntiles) return;
    }
  }
#endif
}

// fp32 channels-last motion map [N][H][W][4] -> the zero-padded fp16 form (interior only: the borders of `padded` must be
// zero, they are never written)
__global__ __launch_bounds__(256) void flow_pad_kernel(const float4* __restrict__ flow, _Float16* __restrict__ padded,
                                                       long P, int H, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const long rowi = i / W;
  const int x = (int)(i - rowi * W);
  const long n = rowi / H;
  const int y = (int)(rowi - n * H);
  const float4 f = flow[i];
  const f16x4 h = {(_Float16)f.x, (_Float16)f.y, (_Float16)f.z, (_Float16)f.w};
  *reinterpret_cast<f16x4*>(padded + (((n * (H + 2 * kFlowPadY) + y + kFlowPadY) * (W + kFlowPadW)) + x + kFlowPadL) * 4) = h;
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

extern "C" int glorie_flow_pad(const float* flow, void* padded, int N, int H, int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!flow || !padded) return GLORIE_EINVAL;
  const long P = (long)N * H * W;
  hipLaunchKernelGGL(flow_pad_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(flow), reinterpret_cast<_Float16*>(padded), P, H, W);
  return check_launch();
}

extern "C" int glorie_flow_conv7_padded(const void* padded, const void* w_packed, const float* bias, void* out,
                                        int out_stride, int N, int H, int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || (out_stride & 7) || out_stride < 128) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!padded || !w_packed || !bias || !out) return GLORIE_EINVAL;
  const long P = (long)N * H * W, ntiles = (P + 15) / 16;
  // 31-bit byte offsets into the padded map and the output
  if ((long)N * (H + 6) * (W + 8) * 8 >= 0x7fffffffL || P * (long)out_stride * 2 >= 0x7fffffffL) return GLORIE_EUNSUPPORTED;
  const long wgs = (ntiles + 1) / 2;
  const unsigned grid = (unsigned)(wgs < 512 ? wgs : 512);             // 2 workgroups per CU (register-limited)
  hipLaunchKernelGGL(flow_conv7_padded_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(padded), reinterpret_cast<const _Float16*>(w_packed), bias,
                     reinterpret_cast<_Float16*>(out), out_stride, P, H, W);
  return check_launch();
}
