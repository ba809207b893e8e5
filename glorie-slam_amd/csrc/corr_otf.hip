// Volume-free correlation lookup on the matrix cores ("on the fly", scope rows A2/A3) for gfx950.
//
// Same result contract as CorrBlock.__call__ (/root/reference/src/modules/droid_net/corr.py:43-53)
// and AltCorrBlock (corr.py:79-145): for every edge (i -> j), source pixel p and pyramid level l
// the bilinearly blended 7x7 window of <f_i[p]/4, pool_l(f_j/4)[q]> around coords[p] / 2^l --
// without ever materialising the 61 MB/edge correlation volume whose per-pixel planes make the
// windowed gather fetch 4.3x its useful bytes (profiles/r01_pmc_gathers.json).
//
// Structure.  A workgroup owns an 8 x 8 block of source pixels, each of its four waves a 4 x 4 sub-block
// (a 16 x 1 run of pixels has a ~23 x 8 bounding box at level 0, a 4 x 4 block ~11 x 11: a third fewer
// targets to multiply and to pull through L2); the wave's four 16-byte A fragments (128
// channels) stay in registers.  Per level the wave takes the bounding box of the 16 windows
// (smooth flow -> ~11 x 11 target pixels at level 0), evaluates the dense 16 x |bbox| block of dot
// products with v_mfma_f32_16x16x32_f16 (B fragments are 16-byte channel runs of the pooled,
// channel-last feature map, L2 resident), rounds to fp16 like the reference volume and parks it in
// LDS; every pixel then picks and blends its own window from LDS with the reference's fp16
// rounding sequence.  If the 16 windows do not share a compact bbox (bbox > 256 targets: flow
// discontinuities, random coords) the wave falls back to one 8x8 bbox per pixel -- 16x the MFMA
// work for that tile, still correct.  Outputs of the 4 levels are staged in LDS and written as
// 16-byte runs (8 pixels of a block row) per channel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int kOtfCap = 256;            // max targets of a shared bbox
constexpr int kOtfLdR = kOtfCap + 8;    // fp16 elements per source-pixel row of the result buffer
constexpr int kOtfPx = 64;              // source pixels per workgroup (4 waves x 16)
constexpr int kOtfLdO = kOtfPx + 8;

struct OtfLevels {
  const _Float16* f2[4];   // [frames][h_l*w_l][C] channel-last, pooled, pre-scaled by 1/4
  int h[4], w[4];
};

__device__ __forceinline__ _Float16 otf_blend4(_Float16 s00, _Float16 s01, _Float16 s10, _Float16 s11,
                                               _Float16 w00, _Float16 w01, _Float16 w10, _Float16 w11) {
#pragma clang fp contract(off)
  _Float16 acc = (_Float16)0.0f;
  _Float16 t;
  t = s00 * w00; acc = acc + t;
  t = s01 * w01; acc = acc + t;
  t = s10 * w10; acc = acc + t;
  t = s11 * w11; acc = acc + t;
  return acc;
}
__device__ __forceinline__ _Float16 otf_weight(float prod) {
  asm volatile("" : "+v"(prod));   // keep the fp32 rounding step (see corr.hip: weight_cast)
  return (_Float16)prod;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// grid (ceil(w/8) * ceil(h/8), N), 256 threads.  C = 128 channels.
__global__ __launch_bounds__(256) void corr_otf_kernel(
    const _Float16* __restrict__ f1, OtfLevels lv, int num_levels, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, _Float16* __restrict__ out,
    int HW, int out_channels) {
  constexpr int C = 128;
  __shared__ _Float16 Rbuf[4][16][kOtfLdR];
  __shared__ _Float16 obuf[196][kOtfLdO];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int col = lane & 15, kg = lane >> 4;
  const int n = blockIdx.y;
  const int W0 = lv.w[0], H0 = lv.h[0];
  const int nbx = (W0 + 7) >> 3;
  const int by = blockIdx.x / nbx, bx = blockIdx.x - by * nbx;
  // pixel `q` (0..15) of wave `v`: row 8 by + 4 (v >> 1) + (q >> 2), column 8 bx + 4 (v & 1) + (q & 3);
  // its slot in the staged output is (local row) * 8 + (local column)
  auto lslot = [](int v, int q) { return ((4 * (v >> 1) + (q >> 2)) << 3) + 4 * (v & 1) + (q & 3); };
  const int sy = 8 * by + 4 * (wv >> 1) + (col >> 2), sx = 8 * bx + 4 * (wv & 1) + (col & 3);
  const int fi = (int)ii[n], fj = (int)jj[n];

  // A fragments: this lane's source pixel (clamped into the map), channels 32*kk + 8*kg .. +7
  const int pa = min(sy, H0 - 1) * W0 + min(sx, W0 - 1);
  f16x8 afrag[4];
  {
    const f16x8* src = reinterpret_cast<const f16x8*>(f1 + ((size_t)fi * HW + pa) * C + kg * 8);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) afrag[kk] = src[kk * 4];
  }
  const float x0 = coords[((size_t)n * 2 + 0) * HW + pa];
  const float y0 = coords[((size_t)n * 2 + 1) * HW + pa];
  _Float16(*R)[kOtfLdR] = Rbuf[wv];

  float inv = 1.0f;
  for (int l = 0; l < num_levels; ++l) {
    const int hl = lv.h[l], wl = lv.w[l];
    const _Float16* f2 = lv.f2[l] + (size_t)fj * hl * wl * C;
    const float xs = x0 * inv, ys = y0 * inv;
    inv *= 0.5f;
    const float fx = floorf(xs), fy = floorf(ys);
    const float dx = xs - fx, dy = ys - fy;
    const int ix0 = static_cast<int>(fx) - 3, iy0 = static_cast<int>(fy) - 3;
    // shared bounding box of the 16 windows, clamped to the map (lanes with the same `col`
    // hold the same pixel, so a full-wave reduction is a reduction over the 16 pixels)
    int bx0 = max(wave_min_i(ix0), 0), bx1 = min(wave_max_i(ix0) + 7, wl - 1);
    int by0 = max(wave_min_i(iy0), 0), by1 = min(wave_max_i(iy0) + 7, hl - 1);
    const bool grouped = (bx1 >= bx0) && (by1 >= by0) && ((bx1 - bx0 + 1) * (by1 - by0 + 1) <= kOtfCap);
    const bool empty = (bx1 < bx0) || (by1 < by0);
    const int nsub = empty ? 0 : (grouped ? 1 : 16);
    const _Float16 w00 = otf_weight((1.0f - dx) * (1.0f - dy));
    const _Float16 w01 = otf_weight((1.0f - dx) * dy);
    const _Float16 w10 = otf_weight(dx * (1.0f - dy));
    const _Float16 w11 = otf_weight(dx * dy);

    if (empty) {   // all windows outside the map: the reference leaves zeros
      for (int o = lane; o < 16 * 49; o += 64) obuf[l * 49 + (o >> 4)][lslot(wv, o & 15)] = (_Float16)0.0f;
    }
    for (int sub = 0; sub < nsub; ++sub) {
      int sx0 = bx0, sx1 = bx1, sy0 = by0, sy1 = by1;
      if (!grouped) {  // bbox of pixel `sub` only
        const int px_ix0 = __shfl(ix0, sub, 64), px_iy0 = __shfl(iy0, sub, 64);
        sx0 = max(px_ix0, 0); sx1 = min(px_ix0 + 7, wl - 1);
        sy0 = max(px_iy0, 0); sy1 = min(px_iy0 + 7, hl - 1);
      }
      const int bw = sx1 - sx0 + 1, bh = sy1 - sy0 + 1;
      const int nb = (bw > 0 && bh > 0) ? bw * bh : 0;
      // ---- dense block: R[src][t] = fp16( <f1[src], f2[target t]> ) ----
      // software pipeline: the B fragments of tile t+1 are in flight while tile t is multiplied
      auto load_b = [&](int t0, f16x8 (&b)[4]) {
        const int t = t0 + col;
        const bool tv = t < nb;
        const int ty = sy0 + (tv ? t / bw : 0), tx = sx0 + (tv ? t % bw : 0);
        const f16x8* bsrc = reinterpret_cast<const f16x8*>(f2 + ((size_t)ty * wl + tx) * C + kg * 8);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b[kk] = bsrc[kk * 4];
      };
      f16x8 bcur[4], bnxt[4];
      if (nb > 0) load_b(0, bcur);
      for (int t0 = 0; t0 < nb; t0 += 16) {
        if (t0 + 16 < nb) load_b(t0 + 16, bnxt);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[kk], bcur[kk], acc, 0, 0, 0);
        const int t = t0 + col;
        if (t < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) R[kg * 4 + r][t] = (_Float16)acc[r];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bcur[kk] = bnxt[kk];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // ---- window extraction + bilinear blend (reference rounding sequence) ----
      // uniform trip count: the cross-lane reads below must see their source lanes (0..15) active
      for (int it = 0; it < 13; ++it) {
        const int o = it * 64 + lane;
        const int px = o & 15, ch = min(o >> 4, 48);
        const int i = ch / 7, j = ch - i * 7;
        const int pix0 = __shfl(ix0, px, 64), piy0 = __shfl(iy0, px, 64);
        const int wpk0 = __shfl((int)__builtin_bit_cast(unsigned short, w00) | ((int)__builtin_bit_cast(unsigned short, w01) << 16), px, 64);
        const int wpk1 = __shfl((int)__builtin_bit_cast(unsigned short, w10) | ((int)__builtin_bit_cast(unsigned short, w11) << 16), px, 64);
        const int x1 = pix0 + i, y1 = piy0 + j;
        auto fetch = [&](int xx, int yy) -> _Float16 {
          return (xx >= sx0 && xx <= sx1 && yy >= sy0 && yy <= sy1) ? R[px][(yy - sy0) * bw + (xx - sx0)]
                                                                   : (_Float16)0.0f;
        };
        const _Float16 s00 = fetch(x1, y1), s01 = fetch(x1, y1 + 1);
        const _Float16 s10 = fetch(x1 + 1, y1), s11 = fetch(x1 + 1, y1 + 1);
        const _Float16 pw00 = __builtin_bit_cast(_Float16, (unsigned short)(wpk0 & 0xffff));
        const _Float16 pw01 = __builtin_bit_cast(_Float16, (unsigned short)((unsigned)wpk0 >> 16));
        const _Float16 pw10 = __builtin_bit_cast(_Float16, (unsigned short)(wpk1 & 0xffff));
        const _Float16 pw11 = __builtin_bit_cast(_Float16, (unsigned short)((unsigned)wpk1 >> 16));
        const _Float16 v = otf_blend4(s00, s01, s10, s11, pw00, pw01, pw10, pw11);
        if (o < 16 * 49 && (grouped || px == sub)) obuf[l * 49 + ch][lslot(wv, px)] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  // write-out: per channel and block row 8 pixels = 16 bytes
  const int total_ch = num_levels * 49;
  const int rows = min(8, H0 - 8 * by), cols = min(8, W0 - 8 * bx);
  if (cols == 8 && (W0 & 7) == 0) {
    for (int idx = tid; idx < total_ch * 8; idx += 256) {
      const int ch = idx >> 3, ry = idx & 7;
      if (ry < rows) {
        const uint4 v = *reinterpret_cast<const uint4*>(&obuf[ch][ry * 8]);
        *reinterpret_cast<uint4*>(out + ((size_t)n * out_channels + ch) * HW + (size_t)(8 * by + ry) * W0 + 8 * bx) = v;
      }
    }
  } else {
    for (int idx = tid; idx < total_ch * kOtfPx; idx += 256) {
      const int ch = idx / kOtfPx, q = idx - ch * kOtfPx;
      const int ry = q >> 3, rx = q & 7;
      if (ry < rows && rx < cols)
        out[((size_t)n * out_channels + ch) * HW + (size_t)(8 * by + ry) * W0 + 8 * bx + rx] = obuf[ch][q];
    }
  }
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_corr_otf(const void* fmap1, const void* const* fmap2_levels, int num_levels,
                               const float* coords, const int64_t* ii, const int64_t* jj, void* out,
                               int N, int h, int w, int C, void* stream) {
  if (N < 0 || h < 0 || w < 0 || num_levels < 1 || num_levels > 4) return GLORIE_EINVAL;
  if (N == 0 || h * w == 0) return GLORIE_OK;
  if (!fmap1 || !fmap2_levels || !coords || !ii || !jj || !out) return GLORIE_EINVAL;
  if (C != 128) return GLORIE_EUNSUPPORTED;
  OtfLevels lv{};
  for (int l = 0; l < num_levels; ++l) {
    if (!fmap2_levels[l]) return GLORIE_EINVAL;
    lv.f2[l] = reinterpret_cast<const _Float16*>(fmap2_levels[l]);
    lv.h[l] = h >> l;
    lv.w[l] = w >> l;
  }
  dim3 grid(((w + 7) / 8) * ((h + 7) / 8), N);
  hipLaunchKernelGGL(corr_otf_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(fmap1), lv, num_levels, coords, ii, jj,
                     reinterpret_cast<_Float16*>(out), h * w, num_levels * 49);
  return check_launch();
}
