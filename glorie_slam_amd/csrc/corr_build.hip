// All-pairs correlation pyramid of an edge, built straight into the layout the lookup reads (scope row A1):
//   CorrBlock.__init__ / CorrBlock.corr   /root/reference/src/modules/droid_net/corr.py:26-41,67-76
//   corr[p][q] = fp16( sum_c fp16(f1[c][p] / 4) * fp16(f2[c][q] / 4) )     (autocast: fp16 GEMM, fp32 accumulation)
//   level l + 1 = avg_pool2d(level l, 2, 2) on fp16 tensors (fp32 average rounded to fp16)
// The reference (and round 1 of this repo) runs a library GEMM into a row-major [h1*w1][h2][w2] volume, three
// avg_pool2d launches, and this repo then re-tiled every level (pad + permute + copy: ~250 MB of traffic per edge).
// Here one launch per batch of new edges writes all four levels once, in the 4x8-tiled layout of
// corr_lookup_r3_tiled_kernel, into the SLOT of an arena (CorrArena): adding / removing edges never moves a
// volume again (factor_graph.py:126,161 copy all of them through boolean masks).
//
// Workgroup = 32 source pixels x 8 target rows (two tiled block rows of level 0 = one block row of level 1 =
// two rows of level 2 = one row of level 3), 4 waves.  D = A B with A = 16 target pixels (rows), B = 16 source
// pixels (columns): a lane ends up with 4 CONSECUTIVE targets of one source pixel = one 8-byte LDS store into
// R[pixel][target]; the block rows leave LDS as 16-byte pieces, 640 contiguous bytes per (pixel, block row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct BuildArgs {
  const _Float16* f;          // [F][HW][128] channels-last feature maps, already scaled by 1/4
  const int64_t* ii; const int64_t* jj;   // frame of the source / target map per new edge
  const int* slot;            // arena slot per new edge
  _Float16* lvl[4];           // arena levels: [capacity * HW][plane_l] tiled planes
  int h, w, num_levels;
};

__device__ __forceinline__ unsigned pk2(_Float16 a, _Float16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ _Float16 pool4(_Float16 a, _Float16 b, _Float16 c, _Float16 d) {
  // avg_pool2d on half: float accumulation over (0,0),(0,1),(1,0),(1,1), divided by 4, rounded to half
  const float s = (((float)a + (float)b) + (float)c) + (float)d;
  return (_Float16)(s * 0.25f);
}

// dynamic LDS: R [32][8 * W8] | L1 [32][4 * W81] | L2 [32][2 * W82] | L3 [32][W83]   (W8x = padded widths)
__global__ __launch_bounds__(256, 2) void corr_build_kernel(BuildArgs a) {
  constexpr int C = 128, PX = 32;
  extern __shared__ __attribute__((aligned(16))) _Float16 bsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int h = a.h, w = a.w, HW = h * w;
  const int e = blockIdx.z, grp = blockIdx.y, p0 = blockIdx.x * PX;
  const int fi = (int)a.ii[e], fj = (int)a.jj[e];
  const size_t sl = (size_t)a.slot[e];
  const int W8 = ((w + 7) >> 3) << 3;                 // padded width of a level-0 block row
  const int w1 = w >> 1, w2 = w >> 2, w3 = w >> 3, h1 = h >> 1, h2 = h >> 2, h3 = h >> 3;
  const int W81 = ((w1 + 7) >> 3) << 3, W82 = ((w2 + 7) >> 3) << 3, W83 = ((w3 + 7) >> 3) << 3;
  _Float16* R = bsm;
  _Float16* L1 = R + PX * 8 * W8;
  _Float16* L2 = L1 + PX * 4 * W81;
  _Float16* L3 = L2 + PX * 2 * W82;

  // B operand: source pixels p0 + 16 nt + col (clamped), channels 32 kk + 8 kg .. + 7
  f16x8 bfrag[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int p = min(p0 + nt * 16 + col, HW - 1);
    const f16x8* src = reinterpret_cast<const f16x8*>(a.f + ((size_t)fi * HW + p) * C + kg * 8);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bfrag[nt][kk] = src[kk * 4];
  }
  // ---- level 0: targets t = local row * W8 + x of rows 8 grp .. 8 grp + 7 ----
  const int ntile = (8 * W8) >> 4;
  const _Float16* f2 = a.f + (size_t)fj * HW * C;
  for (int tile = wv; tile < ntile; tile += 4) {
    const int t = tile * 16 + col;
    const int ly = t / W8, x = t - ly * W8, y = 8 * grp + ly;
    const bool ok = y < h && x < w;
    const f16x8* src = reinterpret_cast<const f16x8*>(f2 + ((size_t)(ok ? y * w + x : 0)) * C + kg * 8);
    f16x8 af[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) af[kk] = src[kk * 4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) if (!ok) af[kk] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk], bfrag[nt][kk], acc, 0, 0, 0);
      *reinterpret_cast<uint2*>(R + (nt * 16 + col) * (8 * W8) + tile * 16 + kg * 4) =
          make_uint2(pk2((_Float16)acc[0], (_Float16)acc[1]), pk2((_Float16)acc[2], (_Float16)acc[3]));
    }
  }
  __syncthreads();
  // ---- pooled levels in LDS (each from the fp16 values of the level below) ----
  if (a.num_levels > 1) {
    for (int idx = tid; idx < PX * 4 * W81; idx += 256) {
      const int px = idx / (4 * W81), r = idx - px * (4 * W81), y = r / W81, x = r - y * W81;
      const _Float16* s = R + px * (8 * W8) + (2 * y) * W8 + 2 * x;
      L1[idx] = (x < w1) ? pool4(s[0], s[1], s[W8], s[W8 + 1]) : (_Float16)0.0f;
    }
    __syncthreads();
  }
  if (a.num_levels > 2) {
    for (int idx = tid; idx < PX * 2 * W82; idx += 256) {
      const int px = idx / (2 * W82), r = idx - px * (2 * W82), y = r / W82, x = r - y * W82;
      const _Float16* s = L1 + px * (4 * W81) + (2 * y) * W81 + 2 * x;
      L2[idx] = (x < w2) ? pool4(s[0], s[1], s[W81], s[W81 + 1]) : (_Float16)0.0f;
    }
    __syncthreads();
  }
  if (a.num_levels > 3) {
    for (int idx = tid; idx < PX * W83; idx += 256) {
      const int px = idx / W83, x = idx - px * W83;
      const _Float16* s = L2 + px * (2 * W82) + 2 * x;
      L3[idx] = (x < w3) ? pool4(s[0], s[1], s[W82], s[W82 + 1]) : (_Float16)0.0f;
    }
    __syncthreads();
  }
  // ---- write-out: 16-byte pieces = one row of a 4x8 block; rows / blocks beyond the level's size are skipped ----
  auto flush = [&](const _Float16* src, int rows, int Wp, _Float16* dstl, int hl, int wl, int y0) {
    // src [PX][rows][Wp]; level plane: [ceil(hl/4)][ceil(wl/8)][4][8]; global rows y0 .. y0 + rows - 1
    const int nbx = (wl + 7) >> 3, nby = (hl + 3) >> 2;
    const size_t plane = (size_t)nbx * nby * 32;
    const int per_px = rows * nbx;
    for (int idx = tid; idx < PX * per_px; idx += 256) {
      const int px = idx / per_px, r = idx - px * per_px;
      const int ly = r / nbx, bx = r - ly * nbx, y = y0 + ly;
      if (p0 + px >= HW || y >= hl) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)px * rows * Wp + ly * Wp + bx * 8);
      _Float16* d = dstl + (sl * HW + p0 + px) * plane + ((size_t)(y >> 2) * nbx + bx) * 32 + (y & 3) * 8;
      *reinterpret_cast<uint4*>(d) = v;
    }
  };
  flush(R, 8, W8, a.lvl[0], h, w, 8 * grp);
  if (a.num_levels > 1) flush(L1, 4, W81, a.lvl[1], h1, w1, 4 * grp);
  if (a.num_levels > 2) flush(L2, 2, W82, a.lvl[2], h2, w2, 2 * grp);
  if (a.num_levels > 3) flush(L3, 1, W83, a.lvl[3], h3, w3, grp);
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_corr_build(const void* fmaps_cl, const int64_t* ii, const int64_t* jj, const int* slots,
                                 void* const* levels, int num_levels, int n_new, int h, int w, int C, void* stream) {
  if (n_new < 0 || h <= 0 || w <= 0 || num_levels < 1 || num_levels > 4) return GLORIE_EINVAL;
  if (n_new == 0) return GLORIE_OK;
  if (!fmaps_cl || !ii || !jj || !slots || !levels) return GLORIE_EINVAL;
  if (C != 128 || (w & 7)) return GLORIE_EUNSUPPORTED;      // widths that are not multiples of 8 keep the torch builder
  BuildArgs a{};
  a.f = reinterpret_cast<const _Float16*>(fmaps_cl);
  a.ii = ii; a.jj = jj; a.slot = slots; a.h = h; a.w = w; a.num_levels = num_levels;
  for (int l = 0; l < num_levels; ++l) {
    if (!levels[l]) return GLORIE_EINVAL;
    a.lvl[l] = reinterpret_cast<_Float16*>(levels[l]);
  }
  auto pad8 = [](int v) { return ((v + 7) >> 3) << 3; };
  const size_t lds = sizeof(_Float16) * 32 * (size_t)(8 * pad8(w) + 4 * pad8(w >> 1) + 2 * pad8(w >> 2) + pad8(w >> 3));
  if (lds > 80 * 1024) return GLORIE_EUNSUPPORTED;
  static PerDeviceOnce attr;
  if (attr.first()) {
    GLORIE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(corr_build_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)));
  }
  const dim3 grid((h * w + 31) / 32, (h + 7) / 8, n_new);
  hipLaunchKernelGGL(corr_build_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  return check_launch();
}
