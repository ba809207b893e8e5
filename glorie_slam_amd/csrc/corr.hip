// Correlation-volume lookup for gfx950 (rows A2 of the scope table).
//
// Replaces droid_backends.corr_index_forward
//   (/root/reference/src/lib/correlation_kernels.cu:19-70,126-155) and the 4-level loop +
//   torch.cat of CorrBlock.__call__ (/root/reference/src/modules/droid_net/corr.py:43-53).
//
// Work decomposition (wave64 native, not a 16x16 CUDA block):
//   one pixel of the source frame is served by 8 consecutive lanes; lane `row` (0..7) owns
//   window row y1 = floor(y0)-r+row and fetches its 8 contiguous x-samples with ONE
//   unaligned 16-byte load (global_load_dwordx4 on a 2-byte aligned address).  A wave thus
//   serves 8 pixels and issues one wide load per level instead of 64 scalar loads per pixel.
//   Row `row+1` is pulled from the neighbouring lane with DPP-style shuffles, after which
//   lane `row` (<7) produces the 7 outputs (i, j=row), i=0..6, entirely in registers and
//   stores them once -- no zero-fill pass and no read-modify-write of the output as in the
//   reference (4 RMW per output there).
//
// Numerics: for fp16 the reference's rounding sequence is reproduced exactly:
//   out = ((((+0 + h(s00*w00)) + h(s01*w01)) + h(s10*w10)) + h(s11*w11)), every product and
//   every sum rounded to fp16 (c10::Half operators), w** = h(float weight).  Out-of-bounds
//   samples are skipped in the reference; adding +0 instead is bit-identical because the
//   accumulator can never be -0.  fp contraction is disabled in that block.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/glorie_hip.h"
#include "common.hiph"

namespace glorie {

constexpr int kMaxLevels = 4;

struct CorrLevels {
  const void* vol[kMaxLevels];
  int h2[kMaxLevels];
  int w2[kMaxLevels];
};

template <typename T> struct Row8;
template <> struct __attribute__((packed, aligned(2))) Row8<_Float16> { _Float16 v[8]; };
template <> struct __attribute__((packed, aligned(4))) Row8<float> { float v[8]; };

// bilinear blend of the 2x2 neighbourhood in the reference's accumulation order
__device__ __forceinline__ _Float16 blend4(_Float16 s00, _Float16 s01, _Float16 s10, _Float16 s11,
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
// fp32: nvcc contracts `corr += s * w` into an FMA chain in the reference build
__device__ __forceinline__ float blend4(float s00, float s01, float s10, float s11,
                                        float w00, float w01, float w10, float w11) {
  float acc = 0.0f;
  acc = fmaf(s00, w00, acc);
  acc = fmaf(s01, w01, acc);
  acc = fmaf(s10, w10, acc);
  acc = fmaf(s11, w11, acc);
  return acc;
}

// float -> T weight.  The reference rounds the fp32 product to fp32 FIRST and then to fp16
// (`scalar_t(dx * dy)`); hipcc would otherwise fuse mul+cvt into v_fma_mixlo_f16 (a single
// rounding), which differs in rare near-tie cases.  The empty asm pins the fp32 value.
template <typename T>
__device__ __forceinline__ T weight_cast(float prod) {
  asm volatile("" : "+v"(prod));
  return (T)prod;
}

__device__ __forceinline__ _Float16 shfl_next(_Float16 v) {
  // move one fp16 from lane+1; done on the 32-bit container
  int x = (int)__builtin_bit_cast(unsigned short, v);
  x = __shfl_down(x, 1, 64);
  return __builtin_bit_cast(_Float16, (unsigned short)x);
}
__device__ __forceinline__ float shfl_next(float v) { return __shfl_down(v, 1, 64); }

// 256 threads = 32 pixels x 8 window rows.  grid = (ceil(HW/32), N)
template <typename T>
__global__ __launch_bounds__(256) void corr_lookup_r3_kernel(
    CorrLevels lv, int num_levels, int scale_coords,
    const float* __restrict__ coords, T* __restrict__ out,
    int HW, int out_channels) {
  constexpr int R = 3, RD = 7, WIN = 8;
  const int tid = threadIdx.x;
  const int row = tid & 7;
  const int p = blockIdx.x * 32 + (tid >> 3);
  const int n = blockIdx.y;
  const bool live = p < HW;
  const int pc = live ? p : HW - 1;

  const float x0 = coords[((size_t)n * 2 + 0) * HW + pc];
  const float y0 = coords[((size_t)n * 2 + 1) * HW + pc];

  float inv = 1.0f;
  for (int l = 0; l < num_levels; ++l) {
    const int h2 = lv.h2[l], w2 = lv.w2[l];
    const float xs = scale_coords ? x0 * inv : x0;
    const float ys = scale_coords ? y0 * inv : y0;
    inv *= 0.5f;
    const float fx = floorf(xs), fy = floorf(ys);
    const float dx = xs - fx, dy = ys - fy;
    const int ix0 = static_cast<int>(fx) - R;
    const int y1 = static_cast<int>(fy) - R + row;

    T s[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) s[i] = (T)0.0f;
    if (y1 >= 0 && y1 < h2) {
      const T* rowp = reinterpret_cast<const T*>(lv.vol[l]) +
                      ((size_t)n * HW + pc) * ((size_t)h2 * w2) + (size_t)y1 * w2;
      if (ix0 >= 0 && ix0 + WIN <= w2) {
        Row8<T> r;
        __builtin_memcpy(&r, rowp + ix0, sizeof(r));
#pragma unroll
        for (int i = 0; i < WIN; ++i) s[i] = r.v[i];
      } else {
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
          const int x1 = ix0 + i;
          if (x1 >= 0 && x1 < w2) s[i] = rowp[x1];
        }
      }
    }
    // window row j+1 lives in the next lane (same pixel for row < 7)
    T nx[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) nx[i] = shfl_next(s[i]);

    const T w00 = weight_cast<T>((1.0f - dx) * (1.0f - dy));
    const T w01 = weight_cast<T>((1.0f - dx) * dy);
    const T w10 = weight_cast<T>(dx * (1.0f - dy));
    const T w11 = weight_cast<T>(dx * dy);

    if (live && row < RD) {
      T* o = out + ((size_t)n * out_channels + (size_t)l * RD * RD + row) * HW + p;
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        const T v = blend4(s[i], nx[i], s[i + 1], nx[i + 1], w00, w01, w10, w11);
        o[(size_t)i * RD * HW] = v;
      }
    }
  }
}


// ------------------------------------------------------------------------------------
// v2: lane = (pixel group g of 8 consecutive pixels, window row r).  Per level a lane issues
// 8 independent 16-byte row loads (one per pixel of its group) before using any of them, and
// every output channel of the group is ONE 16-byte store; across the 8 groups of a wave that
// is a full 128-byte line per channel (the v1 kernel issues 2-byte stores: 8x more store
// instructions, 1.8x write amplification in WRITE_SIZE).  Row r+1 comes from lane+1 via a DPP
// row shift.  Requires HW % 8 == 0.  Arithmetic identical to v1 (bit-exact fp16).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned dpp_next(unsigned v) {
  // lane i <- lane i+1 inside a row of 16 lanes (row_shl:1); the last row lane is unused
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
}

template <typename T> struct Pack8;
template <> struct Pack8<_Float16> {
  typedef __attribute__((ext_vector_type(4))) unsigned vec;  // 8 halfs
  static __device__ __forceinline__ void next(const _Float16 (&s)[8], _Float16 (&n)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = __builtin_bit_cast(unsigned short, s[2 * i]);
      const unsigned hi = __builtin_bit_cast(unsigned short, s[2 * i + 1]);
      const unsigned v = dpp_next(lo | (hi << 16));
      n[2 * i] = __builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu));
      n[2 * i + 1] = __builtin_bit_cast(_Float16, (unsigned short)(v >> 16));
    }
  }
};
template <> struct Pack8<float> {
  static __device__ __forceinline__ void next(const float (&s)[8], float (&n)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) n[i] = __uint_as_float(dpp_next(__float_as_uint(s[i])));
  }
};

template <typename T>
struct __attribute__((aligned(16))) Out8 { T v[8]; };

template <typename T>
__global__ __launch_bounds__(256) void corr_lookup_r3_v2_kernel(
    CorrLevels lv, int num_levels, int scale_coords, const float* __restrict__ coords,
    T* __restrict__ out, int HW, int out_channels) {
  constexpr int R = 3, RD = 7, WIN = 8, PG = 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = lane & 7, grp = lane >> 3;
  const int n = blockIdx.y;
  const int pb = (blockIdx.x * 4 + wv) * 64 + grp * PG;   // first pixel of this lane's group
  const bool live = pb < HW;                              // HW % 8 == 0: groups are all-or-nothing
  const int pc = live ? pb : HW - PG;

  float x0[PG], y0[PG];
  {
    const float4* cx = reinterpret_cast<const float4*>(coords + ((size_t)n * 2 + 0) * HW + pc);
    const float4* cy = reinterpret_cast<const float4*>(coords + ((size_t)n * 2 + 1) * HW + pc);
    const float4 a = cx[0], b = cx[1], c = cy[0], d = cy[1];
    x0[0] = a.x; x0[1] = a.y; x0[2] = a.z; x0[3] = a.w; x0[4] = b.x; x0[5] = b.y; x0[6] = b.z; x0[7] = b.w;
    y0[0] = c.x; y0[1] = c.y; y0[2] = c.z; y0[3] = c.w; y0[4] = d.x; y0[5] = d.y; y0[6] = d.z; y0[7] = d.w;
  }
  float inv = 1.0f;
  for (int l = 0; l < num_levels; ++l) {
    const int h2 = lv.h2[l], w2 = lv.w2[l];
    const T* vol = reinterpret_cast<const T*>(lv.vol[l]);
    T s[PG][WIN];
    float dxs[PG], dys[PG];
    // phase 1: issue the 8 row loads of this lane
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const float xs = scale_coords ? x0[q] * inv : x0[q];
      const float ys = scale_coords ? y0[q] * inv : y0[q];
      const float fx = floorf(xs), fy = floorf(ys);
      dxs[q] = xs - fx;
      dys[q] = ys - fy;
      const int ix0 = static_cast<int>(fx) - R;
      const int y1 = static_cast<int>(fy) - R + row;
#pragma unroll
      for (int i = 0; i < WIN; ++i) s[q][i] = (T)0.0f;
      if (y1 >= 0 && y1 < h2) {
        const T* rowp = vol + ((size_t)n * HW + pc + q) * ((size_t)h2 * w2) + (size_t)y1 * w2;
        if (ix0 >= 0 && ix0 + WIN <= w2) {
          Row8<T> r;
          __builtin_memcpy(&r, rowp + ix0, sizeof(r));
#pragma unroll
          for (int i = 0; i < WIN; ++i) s[q][i] = r.v[i];
        } else {
#pragma unroll
          for (int i = 0; i < WIN; ++i) {
            const int x1 = ix0 + i;
            if (x1 >= 0 && x1 < w2) s[q][i] = rowp[x1];
          }
        }
      }
    }
    inv *= 0.5f;
    // phase 2: blend and store, one 16-byte store per output channel
    Out8<T> o[RD];
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      T nx[WIN];
      Pack8<T>::next(s[q], nx);
      const float dx = dxs[q], dy = dys[q];
      const T w00 = weight_cast<T>((1.0f - dx) * (1.0f - dy));
      const T w01 = weight_cast<T>((1.0f - dx) * dy);
      const T w10 = weight_cast<T>(dx * (1.0f - dy));
      const T w11 = weight_cast<T>(dx * dy);
#pragma unroll
      for (int i = 0; i < RD; ++i)
        o[i].v[q] = blend4(s[q][i], nx[i], s[q][i + 1], nx[i + 1], w00, w01, w10, w11);
    }
    if (live && row < RD) {
      T* op = out + ((size_t)n * out_channels + (size_t)l * RD * RD + row) * HW + pb;
#pragma unroll
      for (int i = 0; i < RD; ++i)
        *reinterpret_cast<Out8<T>*>(op + (size_t)i * RD * HW) = o[i];
    }
  }
}

// ------------------------------------------------------------------------------------
// v3: the v2 lane mapping on a TILED pyramid.  Each plane is stored as 4-row x 8-column blocks of
// 64 bytes ([ceil(h2/4)][ceil(w2/8)][4][8] halfs, zero padded): the 8x8 window of a pixel then touches
// 5.2 of the 64-byte HBM sectors on average instead of 8 rows x 1.2 sectors (the per-pixel planes
// share nothing between neighbouring pixels, so the over-fetch of a row-major plane is structural:
// 4.3x the useful bytes, profiles/r01_pmc_kernels.json).  A window row lives in two horizontally
// adjacent blocks; the lane issues one unaligned 16-byte load into each (at +sh and +sh-8 halfs of
// the block row, sh = window start mod 8; the bytes outside the block row are masked off, the
// padding columns hold zeros) and merges them with a bit-field insert -- no per-tap fallback path.
// Arithmetic identical to v1/v2 (bit-exact fp16).  The volume needs 16 bytes of readable slack on
// either side (a plane is >= 256 bytes: the Python side allocates one spare plane before and after).
// ------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// CL: the output is channels-last with the window padded to 8 x 8 per level - channel l * 64 + row * 8 + tap, rows / taps 7
// zero - i.e. one 512-byte row of 256 halfs per pixel.  The lane that owns window row `row` of a pixel then writes ONE 16-byte
// piece (its 7 taps + a zero), the 8 row lanes of a pixel group together one full 128-byte line per (pixel, level): the same
// store efficiency as the planar layout, and corr_encoder[0] can consume the map as a 1x1 implicit-GEMM convolution with
// permuted weight columns (no library GEMM over the planar layout, no separate bias / ReLU pass).
template <bool CL>
__global__ __launch_bounds__(256) void corr_lookup_r3_tiled_kernel(
    CorrLevels lv, int num_levels, const float* __restrict__ coords, _Float16* __restrict__ out,
    int HW, int out_channels, const int* __restrict__ slots, int coords_xy) {
  typedef _Float16 T;
  constexpr int R = 3, RD = 7, WIN = 8, PG = 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = lane & 7, grp = lane >> 3;
  const int n = blockIdx.y;
  // volume of edge n: its own position in the stack, or - for an arena (CorrArena) - the slot it was built into
  const size_t vn = slots ? (size_t)slots[n] : (size_t)n;
  const int pb = (blockIdx.x * 4 + wv) * 64 + grp * PG;   // first pixel of this lane's group
  const bool live = pb < HW;                              // HW % 8 == 0: groups are all-or-nothing
  const int pc = live ? pb : HW - PG;

  float x0[PG], y0[PG];
  if (coords_xy) {
    // [N][HW][2] as the reprojection writes it (no permute + copy in front of the lookup): 8 pixels = 64 contiguous bytes
    const float4* cp = reinterpret_cast<const float4*>(coords + ((size_t)n * HW + pc) * 2);
    const float4 a = cp[0], b = cp[1], c = cp[2], d = cp[3];
    x0[0] = a.x; y0[0] = a.y; x0[1] = a.z; y0[1] = a.w; x0[2] = b.x; y0[2] = b.y; x0[3] = b.z; y0[3] = b.w;
    x0[4] = c.x; y0[4] = c.y; x0[5] = c.z; y0[5] = c.w; x0[6] = d.x; y0[6] = d.y; x0[7] = d.z; y0[7] = d.w;
  } else {
    const float4* cx = reinterpret_cast<const float4*>(coords + ((size_t)n * 2 + 0) * HW + pc);
    const float4* cy = reinterpret_cast<const float4*>(coords + ((size_t)n * 2 + 1) * HW + pc);
    const float4 a = cx[0], b = cx[1], c = cy[0], d = cy[1];
    x0[0] = a.x; x0[1] = a.y; x0[2] = a.z; x0[3] = a.w; x0[4] = b.x; x0[5] = b.y; x0[6] = b.z; x0[7] = b.w;
    y0[0] = c.x; y0[1] = c.y; y0[2] = c.z; y0[3] = c.w; y0[4] = d.x; y0[5] = d.y; y0[6] = d.z; y0[7] = d.w;
  }
  float inv = 1.0f;
  for (int l = 0; l < num_levels; ++l) {
    const int h2 = lv.h2[l], w2 = lv.w2[l];
    const int nbx = (w2 + 7) >> 3, nby = (h2 + 3) >> 2;
    const size_t plane = (size_t)nbx * nby * 32;           // halfs per tiled plane
    const T* vol = reinterpret_cast<const T*>(lv.vol[l]);
    u32x4 la[PG], lb[PG];
    int shs[PG], oka[PG], okb[PG];
    float dxs[PG], dys[PG];
    // phase 1: issue the 16 loads of this lane (2 per pixel of the group)
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const float xs = x0[q] * inv, ys = y0[q] * inv;
      const float fx = floorf(xs), fy = floorf(ys);
      dxs[q] = xs - fx;
      dys[q] = ys - fy;
      const int ix0 = static_cast<int>(fx) - R;
      const int y1 = static_cast<int>(fy) - R + row;
      const bool yok = y1 >= 0 && y1 < h2;
      const int bx0 = ix0 >> 3, sh = ix0 & 7;               // floor division / modulo for negative starts too
      const bool va = yok && bx0 >= 0 && bx0 < nbx;
      const bool vb = yok && sh > 0 && bx0 + 1 >= 0 && bx0 + 1 < nbx;
      const int yc = yok ? y1 : 0;
      const T* base = vol + (vn * HW + pc + q) * plane + ((size_t)(yc >> 2) * nbx) * 32 + (yc & 3) * 8;
      const T* pa = base + (va ? bx0 * 32 + sh : 0);
      const T* pbk = base + (vb ? (bx0 + 1) * 32 + sh - 8 : 0);
      __builtin_memcpy(&la[q], pa, 16);
      __builtin_memcpy(&lb[q], pbk, 16);
      shs[q] = sh; oka[q] = va; okb[q] = vb;
    }
    inv *= 0.5f;
    // phase 2: merge the two block rows, blend and store, one 16-byte store per output channel
    Out8<T> o[RD];
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int cnt = 8 - shs[q];                         // taps served by block A: halfs [0, cnt)
      u32x4 sv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = cnt - 2 * k;
        const unsigned m = t >= 2 ? 0xffffffffu : (t == 1 ? 0x0000ffffu : 0u);
        const unsigned a = oka[q] ? la[q][k] : 0u, b = okb[q] ? lb[q][k] : 0u;
        sv[k] = (a & m) | (b & ~m);
      }
      T sq[WIN];
      __builtin_memcpy(sq, &sv, 16);
      T nx[WIN];
      Pack8<T>::next(sq, nx);
      const float dx = dxs[q], dy = dys[q];
      const T w00 = weight_cast<T>((1.0f - dx) * (1.0f - dy));
      const T w01 = weight_cast<T>((1.0f - dx) * dy);
      const T w10 = weight_cast<T>(dx * (1.0f - dy));
      const T w11 = weight_cast<T>(dx * dy);
#pragma unroll
      for (int i = 0; i < RD; ++i)
        o[i].v[q] = blend4(sq[i], nx[i], sq[i + 1], nx[i + 1], w00, w01, w10, w11);
    }
    if (CL) {
      if (live) {
        T* op = out + ((size_t)n * HW + pb) * out_channels + l * 64 + row * 8;
#pragma unroll
        for (int q = 0; q < PG; ++q) {
          Out8<T> px;
#pragma unroll
          for (int i = 0; i < RD; ++i) px.v[i] = row < RD ? o[i].v[q] : (T)0.0f;
          px.v[RD] = (T)0.0f;
          *reinterpret_cast<Out8<T>*>(op + (size_t)q * out_channels) = px;
        }
      }
    } else if (live && row < RD) {
      T* op = out + ((size_t)n * out_channels + (size_t)l * RD * RD + row) * HW + pb;
#pragma unroll
      for (int i = 0; i < RD; ++i)
        *reinterpret_cast<Out8<T>*>(op + (size_t)i * RD * HW) = o[i];
    }
  }
}

// generic-radius fallback: one thread per pixel, same arithmetic order.
template <typename T>
__global__ __launch_bounds__(256) void corr_lookup_generic_kernel(
    CorrLevels lv, int num_levels, int scale_coords, int radius,
    const float* __restrict__ coords, T* __restrict__ out, int HW, int out_channels) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= HW) return;
  const int rd = 2 * radius + 1;
  const float x0 = coords[((size_t)n * 2 + 0) * HW + p];
  const float y0 = coords[((size_t)n * 2 + 1) * HW + p];
  float inv = 1.0f;
  for (int l = 0; l < num_levels; ++l) {
    const int h2 = lv.h2[l], w2 = lv.w2[l];
    const float xs = scale_coords ? x0 * inv : x0;
    const float ys = scale_coords ? y0 * inv : y0;
    inv *= 0.5f;
    const float fx = floorf(xs), fy = floorf(ys);
    const float dx = xs - fx, dy = ys - fy;
    const int ix0 = static_cast<int>(fx) - radius;
    const int iy0 = static_cast<int>(fy) - radius;
    const T* plane = reinterpret_cast<const T*>(lv.vol[l]) + ((size_t)n * HW + p) * ((size_t)h2 * w2);
    const T w00 = weight_cast<T>((1.0f - dx) * (1.0f - dy));
    const T w01 = weight_cast<T>((1.0f - dx) * dy);
    const T w10 = weight_cast<T>(dx * (1.0f - dy));
    const T w11 = weight_cast<T>(dx * dy);
    auto fetch = [&](int i, int j) -> T {
      const int x1 = ix0 + i, y1 = iy0 + j;
      return (x1 >= 0 && x1 < w2 && y1 >= 0 && y1 < h2) ? plane[(size_t)y1 * w2 + x1] : (T)0.0f;
    };
    for (int i = 0; i < rd; ++i)
      for (int j = 0; j < rd; ++j) {
        const T v = blend4(fetch(i, j), fetch(i, j + 1), fetch(i + 1, j), fetch(i + 1, j + 1),
                           w00, w01, w10, w11);
        out[((size_t)n * out_channels + (size_t)l * rd * rd + (size_t)i * rd + j) * HW + p] = v;
      }
  }
}

template <typename T>
static int launch_lookup(const CorrLevels& lv, int L, int scale, const float* coords, void* out,
                         int N, int HW, int radius, hipStream_t st) {
  if (N == 0 || HW == 0) return GLORIE_OK;
  const int chans = L * (2 * radius + 1) * (2 * radius + 1);
  if (radius == 3 && HW % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(coords) & 15) == 0) {
    dim3 grid((HW + 255) / 256, N);
    hipLaunchKernelGGL(corr_lookup_r3_v2_kernel<T>, grid, dim3(256), 0, st, lv, L, scale, coords,
                       reinterpret_cast<T*>(out), HW, chans);
  } else if (radius == 3) {
    dim3 grid((HW + 31) / 32, N);
    hipLaunchKernelGGL(corr_lookup_r3_kernel<T>, grid, dim3(256), 0, st, lv, L, scale, coords,
                       reinterpret_cast<T*>(out), HW, chans);
  } else {
    dim3 grid((HW + 255) / 256, N);
    hipLaunchKernelGGL(corr_lookup_generic_kernel<T>, grid, dim3(256), 0, st, lv, L, scale, radius,
                       coords, reinterpret_cast<T*>(out), HW, chans);
  }
  return check_launch();
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_corr_index_fwd(const void* volume, const float* coords, void* out, int N,
                                     int h1, int w1, int h2, int w2, int radius, int dtype,
                                     void* stream) {
  if (N < 0 || h1 < 0 || w1 < 0 || h2 < 0 || w2 < 0 || radius < 0) return GLORIE_EINVAL;
  if (N == 0 || h1 * w1 == 0) return GLORIE_OK;
  if (!volume || !coords || !out) return GLORIE_EINVAL;
  CorrLevels lv{};
  lv.vol[0] = volume; lv.h2[0] = h2; lv.w2[0] = w2;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GLORIE_F16) return launch_lookup<_Float16>(lv, 1, 0, coords, out, N, h1 * w1, radius, st);
  if (dtype == GLORIE_F32) return launch_lookup<float>(lv, 1, 0, coords, out, N, h1 * w1, radius, st);
  return GLORIE_EUNSUPPORTED;
}

extern "C" int glorie_corr_lookup_pyramid(const void* const* volumes, int num_levels,
                                          const float* coords, void* out, int N, int h1, int w1,
                                          int h2, int w2, int radius, int dtype, void* stream) {
  if (num_levels < 1 || num_levels > kMaxLevels || N < 0 || h1 < 0 || w1 < 0 || radius < 0)
    return GLORIE_EINVAL;
  if (N == 0 || h1 * w1 == 0) return GLORIE_OK;
  if (!volumes || !coords || !out) return GLORIE_EINVAL;
  CorrLevels lv{};
  for (int l = 0; l < num_levels; ++l) {
    if (!volumes[l]) return GLORIE_EINVAL;
    lv.vol[l] = volumes[l];
    lv.h2[l] = h2 >> l;   // matches h2 // 2**l of corr.py:38
    lv.w2[l] = w2 >> l;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GLORIE_F16)
    return launch_lookup<_Float16>(lv, num_levels, 1, coords, out, N, h1 * w1, radius, st);
  if (dtype == GLORIE_F32)
    return launch_lookup<float>(lv, num_levels, 1, coords, out, N, h1 * w1, radius, st);
  return GLORIE_EUNSUPPORTED;
}

static int lookup_tiled(const void* const* volumes, int num_levels, const float* coords, void* out, int N, int h1,
                        int w1, int h2, int w2, const int* slots, void* stream, bool channels_last = false, int coords_xy = 0);

extern "C" int glorie_corr_lookup_pyramid_tiled(const void* const* volumes, int num_levels,
                                                const float* coords, void* out, int N, int h1, int w1,
                                                int h2, int w2, void* stream) {
  return lookup_tiled(volumes, num_levels, coords, out, N, h1, w1, h2, w2, nullptr, stream);
}

extern "C" int glorie_corr_lookup_arena(const void* const* levels, int num_levels, const int* slots,
                                        const float* coords, void* out, int N, int h1, int w1, int h2, int w2,
                                        void* stream) {
  if (N > 0 && !slots) return GLORIE_EINVAL;
  return lookup_tiled(levels, num_levels, coords, out, N, h1, w1, h2, w2, slots, stream);
}

extern "C" int glorie_corr_lookup_tiled_cl(const void* const* levels, int num_levels, const int* slots,
                                           const float* coords, int coords_xy, void* out, int N, int h1, int w1,
                                           int h2, int w2, void* stream) {
  if (num_levels != 4) return GLORIE_EUNSUPPORTED;             // the 256-channel row holds 4 levels of 8 x 8
  return lookup_tiled(levels, num_levels, coords, out, N, h1, w1, h2, w2, slots, stream, true, coords_xy != 0);
}

static int lookup_tiled(const void* const* volumes, int num_levels, const float* coords, void* out, int N, int h1,
                        int w1, int h2, int w2, const int* slots, void* stream, bool channels_last, int coords_xy) {
  if (num_levels < 1 || num_levels > kMaxLevels || N < 0 || h1 < 0 || w1 < 0) return GLORIE_EINVAL;
  const int HW = h1 * w1;
  if (N == 0 || HW == 0) return GLORIE_OK;
  if (!volumes || !coords || !out) return GLORIE_EINVAL;
  if (HW % 8 || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(coords) & 15))
    return GLORIE_EUNSUPPORTED;
  CorrLevels lv{};
  for (int l = 0; l < num_levels; ++l) {
    if (!volumes[l]) return GLORIE_EINVAL;
    lv.vol[l] = volumes[l];
    lv.h2[l] = h2 >> l;
    lv.w2[l] = w2 >> l;
  }
  dim3 grid((HW + 255) / 256, N);
  if (channels_last)
    hipLaunchKernelGGL(corr_lookup_r3_tiled_kernel<true>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       lv, num_levels, coords, reinterpret_cast<_Float16*>(out), HW, num_levels * 64, slots, coords_xy);
  else
    hipLaunchKernelGGL(corr_lookup_r3_tiled_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       lv, num_levels, coords, reinterpret_cast<_Float16*>(out), HW, num_levels * 49, slots, coords_xy);
  return check_launch();
}
