// On-the-fly correlation lookup ("alt" implementation, scope row A3) for gfx950.
//
// Replaces droid_backends.altcorr_forward
//   (/root/reference/src/lib/altcorr_kernel.cu:27-149,290-319): for every source pixel the
//   (2r+2)^2 dot products <f1[px], f2[window]> over C channels are formed without ever
//   materialising the correlation volume, then blended bilinearly into (2r+1)^2 outputs.
//
// Mapping (r = 3): one wave64 per source pixel, lane = ix*8 + iy owns window position
// (ix, iy); it streams its fmap2 row (C contiguous floats) with 16-byte loads while the
// source feature is wave-uniform.  The 4 bilinear neighbours are lane+0/+8/+1/+9, fetched
// with shuffles.  A workgroup of 4 waves serves 64 consecutive pixels and stages its
// 49 x 64 outputs in LDS so that the channel-major output is written in 256-byte rows
// instead of 4-byte scatters.
//
// Accumulation order follows the reference: channels in chunks of 32 (CHANNEL_STRIDE), and
// inside a chunk the four corners in the order (iy,ix), (iy,ix+1), (iy+1,ix), (iy+1,ix+1).
#include <hip/hip_runtime.h>
#include "common.hiph"

namespace glorie {

constexpr int kAltPix = 64;  // pixels per workgroup

__global__ __launch_bounds__(256) void altcorr_r3_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2,
    const float* __restrict__ coords, float* __restrict__ out, int S, int H, int W, int H2,
    int W2, int C) {
  __shared__ float obuf[49][kAltPix + 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int ix = lane >> 3, iy = lane & 7;
  const int HW = H * W;
  const int b = blockIdx.z;
  const int sidx = blockIdx.y;
  const int p0 = blockIdx.x * kAltPix;

  for (int q = 0; q < kAltPix / 4; ++q) {
    const int pl = wv * (kAltPix / 4) + q;  // pixel slot inside the workgroup
    const int p = p0 + pl;
    if (p >= HW) break;  // wave-uniform
    const float x0 = coords[(((size_t)b * S + sidx) * HW + p) * 2 + 0];
    const float y0 = coords[(((size_t)b * S + sidx) * HW + p) * 2 + 1];
    const float fx = floorf(x0), fy = floorf(y0);
    const float dx = x0 - fx, dy = y0 - fy;
    const int w2 = static_cast<int>(fx) - 3 + ix;
    const int h2 = static_cast<int>(fy) - 3 + iy;
    const bool inb = (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2);
    const float4* f2 = reinterpret_cast<const float4*>(
        fmap2 + (((size_t)b * H2 + (inb ? h2 : 0)) * W2 + (inb ? w2 : 0)) * C);
    const float4* f1 = reinterpret_cast<const float4*>(fmap1 + ((size_t)b * HW + p) * C);

    const float w_se = (1.0f - dy) * (1.0f - dx);
    const float w_sw = (1.0f - dy) * dx;
    const float w_ne = dy * (1.0f - dx);
    const float w_nw = dy * dx;
    float o = 0.0f;
    for (int c0 = 0; c0 < C; c0 += 32) {
      float s = 0.0f;
      const int cend = min(32, C - c0) / 4;
#pragma unroll 8
      for (int c4 = 0; c4 < cend; ++c4) {
        const float4 a = f1[c0 / 4 + c4];
        const float4 v = inb ? f2[c0 / 4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        s = fmaf(a.x, v.x, s);
        s = fmaf(a.y, v.y, s);
        s = fmaf(a.z, v.z, s);
        s = fmaf(a.w, v.w, s);
      }
      const float s_x = __shfl_down(s, 8, 64);   // (ix+1, iy)
      const float s_y = __shfl_down(s, 1, 64);   // (ix, iy+1)
      const float s_xy = __shfl_down(s, 9, 64);  // (ix+1, iy+1)
      o = fmaf(s, w_se, o);
      o = fmaf(s_x, w_sw, o);
      o = fmaf(s_y, w_ne, o);
      o = fmaf(s_xy, w_nw, o);
    }
    if (ix < 7 && iy < 7) obuf[ix * 7 + iy][pl] = o;  // channel = iy + 7*ix
  }
  __syncthreads();
  // coalesced write-out: 49 rows of up to 64 consecutive pixels
  const int npx = min(kAltPix, HW - p0);
  for (int idx = tid; idx < 49 * kAltPix; idx += 256) {
    const int ch = idx / kAltPix, pl = idx % kAltPix;
    if (pl < npx) out[(((size_t)b * S + sidx) * 49 + ch) * HW + p0 + pl] = obuf[ch][pl];
  }
}

// generic radius: one thread per (pixel), serial window walk (rarely used)
__global__ __launch_bounds__(256) void altcorr_generic_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2,
    const float* __restrict__ coords, float* __restrict__ out, int S, int H, int W, int H2,
    int W2, int C, int r) {
  const int HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int sidx = blockIdx.y, b = blockIdx.z;
  if (p >= HW) return;
  const int rd = 2 * r + 1;
  const float x0 = coords[(((size_t)b * S + sidx) * HW + p) * 2 + 0];
  const float y0 = coords[(((size_t)b * S + sidx) * HW + p) * 2 + 1];
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  const float* f1 = fmap1 + ((size_t)b * HW + p) * C;
  float* o = out + (((size_t)b * S + sidx) * rd * rd) * HW + p;
  for (int c = 0; c < rd * rd; ++c) o[(size_t)c * HW] = 0.0f;
  for (int c0 = 0; c0 < C; c0 += 32)
    for (int iy = 0; iy <= rd; ++iy)
      for (int ix = 0; ix <= rd; ++ix) {
        const int h2 = static_cast<int>(fy) - r + iy, w2 = static_cast<int>(fx) - r + ix;
        float s = 0.0f;
        if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
          const float* f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C;
          for (int c = c0; c < min(c0 + 32, C); ++c) s = fmaf(f1[c], f2[c], s);
        }
        if (iy > 0 && ix > 0) o[(size_t)((iy - 1) + rd * (ix - 1)) * HW] += s * (dy * dx);
        if (iy > 0 && ix < rd) o[(size_t)((iy - 1) + rd * ix) * HW] += s * (dy * (1 - dx));
        if (iy < rd && ix > 0) o[(size_t)(iy + rd * (ix - 1)) * HW] += s * ((1 - dy) * dx);
        if (iy < rd && ix < rd) o[(size_t)(iy + rd * ix) * HW] += s * ((1 - dy) * (1 - dx));
      }
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_altcorr_fwd(const float* fmap1, const float* fmap2, const float* coords,
                                  float* out, int B, int S, int H, int W, int H2, int W2, int C,
                                  int radius, void* stream) {
  if (B < 0 || S < 0 || H < 0 || W < 0 || H2 < 0 || W2 < 0 || C < 0 || radius < 0)
    return GLORIE_EINVAL;
  if (B == 0 || S == 0 || H * W == 0) return GLORIE_OK;
  if (!fmap1 || !fmap2 || !coords || !out) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  if (radius == 3 && C % 4 == 0) {
    dim3 grid((HW + kAltPix - 1) / kAltPix, S, B);
    hipLaunchKernelGGL(altcorr_r3_kernel, grid, dim3(256), 0, st, fmap1, fmap2, coords, out, S, H,
                       W, H2, W2, C);
  } else {
    dim3 grid((HW + 255) / 256, S, B);
    hipLaunchKernelGGL(altcorr_generic_kernel, grid, dim3(256), 0, st, fmap1, fmap2, coords, out,
                       S, H, W, H2, W2, C, radius);
  }
  return check_launch();
}
