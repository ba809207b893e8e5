// Correlation pyramid in the DISPLACEMENT-MAJOR, source-tiled layout and its lookup (scope rows A1 + A2, round 3).
//
// Reference:
//   CorrBlock.__init__ / .corr            /root/reference/src/modules/droid_net/corr.py:26-41,67-76   (builder)
//   CorrBlock.__call__ -> corr_index_forward_kernel
//                                         /root/reference/src/modules/droid_net/corr.py:43-53,
//                                         /root/reference/src/lib/correlation_kernels.cu:19-70        (lookup)
//   UpdateModule.corr_encoder[0] (1x1, 196 -> 128, ReLU)   /root/reference/src/modules/droid_net/droid_net.py:73-77
//
// Why a new layout.  The reference (and rounds 1-2 of this repo) keep one (h>>l) x (w>>l) plane per SOURCE pixel.  The 8 x 8
// window a pixel reads shares no cache line with the window of its neighbour - their planes are different - so a lookup
// fetches ~3.5 128-byte lines per (pixel, level) for 128 useful bytes (profiles/r02_pmc_kernels.json: 311 MB read for 88.5 MB).
// Here the SAME values are stored so that neighbouring source pixels that look at the same displacement share a line:
//
//   D_l[slot][tile][dy][dx >> 1][lane][dx & 1]   fp16, one 256-byte line per (tile, dy, displacement-column PAIR): a lane's
//                                                two neighbouring displacement columns are one dword (round 4)
//     tile = 8 x 8 block of source pixels (row-major over ceil(h/8) x ceil(w/8)),  lane = (sy & 7) * 8 + (sx & 7)
//     dy   = (ty - (sy >> l) + ((h>>l) >> 1)) mod (h>>l)          (ty, tx) = target pixel of level l
//     dx   = (tx - (sx >> l) + ((w>>l) >> 1)) mod Wp,  Wp = (w>>l) rounded up to even (odd widths: one zero column)
//
// which is a bijection of every source pixel's plane (a cyclic shift by the pixel's own level-l position, centred so that
// zero flow sits mid-plane and the wrap is at +-half an image of displacement).  A wave owns one tile; for window row j
// lane s reads the FIVE dwords D_l[tile][by(s) + j][(bx(s) >> 1) + c][s] that hold its 8 window columns (the launch's cost is
// its number of gather instructions - 32.0 / 26.8 / 24.5 us with 256 / 128 / 0 per tile - so two taps travel per load: 160
// instead of 256 gathers, +12 % bytes for windows that start on an odd column): wherever the flow is locally constant all
// 64 lanes hit ONE line, and the union over the 8 x 8 window is ~(8 + spread)^2 / 2 lines per 64 pixels instead of 64 x 3.5.
// Same footprint (planes are padded to whole tiles: +6.7 % at 60 x 80), same values, so the arithmetic of the lookup
// (c10::Half products and sums, see corr.hip) is reproduced bit for bit.
//
// The lookup keeps a pixel's 4 x 7 x 7 outputs in its lane, so corr_encoder[0] runs as an MFMA epilogue in the same launch
// (out^T[128 ch x 64 px] = W[128 x 224] corr^T, K ordered l*56 + j*8 + i, tap 7 of every row zero): the lane's packed
// window rows ARE the B fragments of v_mfma_f32_32x32x16_f16 after one v_permlane32_swap per register pair, the weights sit
// in LDS, and the 196-channel map never exists in HBM (it can still be written, channels-last, for tests and for callers
// that want the reference's tensor).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>
#include "common.hiph"

namespace glorie {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct DmArgs {
  const _Float16* lvl[4];     // arena levels [capacity][ntiles][(h>>l) * Wp][64], Wp = even(w>>l)
  const int* slots;           // slot of edge n (NULL: slot0 + n)
  int slot0;
  const float* coords;        // [N][HW][2] (coords_xy) or [N][2][HW], UNscaled
  int coords_xy;
  int N, h, w, ntx, nty;
  _Float16* corr_cl;          // NULL or [N*HW][256] channels-last lookup (channel l*64 + j*8 + i, i <-> x)
  const _Float16* enc_w;      // NULL or [128][224] fp16, column l*56 + j*8 + i (zero for i == 7)
  const float* enc_b;         // [128]
  _Float16* enc_out;          // rows of enc_stride halfs per edge-pixel
  int enc_stride;
  unsigned enc_bytes;         // N * h * w * enc_stride * 2 (< 2^31: glorie_corr_dm_lookup splits larger calls)
};

constexpr unsigned kOobStore = 0xc0000000u;   // same for the encoder's output rows (one launch covers less than 2 GB of them)
constexpr unsigned kOob = 0x40000000u;   // byte offset far beyond any plane: the buffer load returns 0 without a memory access

__device__ __forceinline__ _Float16 to_half_rn(float prod) {
  // the reference rounds the fp32 weight product to fp32 first and then to fp16 (`scalar_t(dx * dy)`); the empty asm keeps hipcc
  // from fusing mul + cvt into one v_fma_mixlo_f16 (single rounding, differs in rare near-ties) - same as corr.hip
  asm volatile("" : "+v"(prod));
  return (_Float16)prod;
}
__device__ __forceinline__ h2 splat(_Float16 v) { return h2{v, v}; }
// Window columns i = 0..7 of a lane start at displaced column bx; the five dwords D[0..4] of a window row hold the displaced
// columns e .. e + 9, e = bx - (bx & 1).  A packed pair of taps is ONE v_perm_b32 of two neighbouring dwords with a per-lane
// selector (perm(D[k+1], D[k], sel): selector bytes 0-3 name D[k], 4-7 D[k+1], 0x0c yields zero):
//   aligned pair (2k, 2k+1):  bx even: D[k] as it is (0x03020100);  bx odd: (D[k].hi, D[k+1].lo) (0x05040302)
//   odd pair (2k+1, 2k+2):    bx even: (D[k].hi, D[k+1].lo);        bx odd: D[k+1] as it is (0x07060504)
// and a tap whose target column lies outside the level's map gets 0x0c0c for its half - zero padding costs no instruction.
__device__ __forceinline__ void dm_selectors(int px, unsigned ux0, unsigned wl, unsigned (&selp)[4], unsigned (&selq)[4]) {
  const unsigned lo_a = px ? 0x0302u : 0x0100u, hi_a = px ? 0x0504u : 0x0302u;      // aligned pair: halves (2k, 2k+1)
  const unsigned lo_o = px ? 0x0504u : 0x0302u, hi_o = px ? 0x0706u : 0x0504u;      // odd pair: halves (2k+1, 2k+2)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool v0 = ux0 + (unsigned)(2 * k) < wl, v1 = ux0 + (unsigned)(2 * k + 1) < wl, v2 = ux0 + (unsigned)(2 * k + 2) < wl;
    selp[k] = (v0 ? lo_a : 0x0c0cu) | ((v1 ? hi_a : 0x0c0cu) << 16);
    selq[k] = (v1 ? lo_o : 0x0c0cu) | (((k < 3 && v2) ? hi_o : 0x0c0cu) << 16);      // beyond the window: (tap 7, 0)
  }
}
__device__ __forceinline__ h2 perm2(unsigned d1, unsigned d0, unsigned sel) {
  return __builtin_bit_cast(h2, __builtin_amdgcn_perm(d1, d0, sel));
}

// one term of the reference's accumulation: products and sums individually rounded to fp16 (no contraction)
__device__ __forceinline__ h2 acc_term(h2 acc, h2 s, h2 w) {
#pragma clang fp contract(off)
  const h2 t = s * w;
  return acc + t;
}
__device__ __forceinline__ h2 acc_term0(h2 s, h2 w) {
#pragma clang fp contract(off)
  return s * w;
}

// ---- one level of one tile: gather the 8 x 8 window of every lane's pixel ------------------------------------------
template <int L>
__device__ __forceinline__ void dm_gather(const DmArgs& a, size_t slot_tile, int sy, int sx, int lane, float x0, float y0,
                                          unsigned (&raw)[8][5], unsigned (&selp)[4], unsigned (&selq)[4], float& fdx, float& fdy) {
  const int hl = a.h >> L, wl = a.w >> L, wp = (wl + 1) & ~1;
  const float inv = 1.0f / (float)(1 << L);
  const float xs = x0 * inv, ys = y0 * inv;
  const float fx = floorf(xs), fy = floorf(ys);
  fdx = xs - fx;
  fdy = ys - fy;
  const int ix0 = static_cast<int>(fx) - 3, iy0 = static_cast<int>(fy) - 3;
  int bx = ix0 - (sx >> L) + (wl >> 1);
  bx %= wp;
  bx = bx < 0 ? bx + wp : bx;
  const int by = iy0 - (sy >> L) + (hl >> 1);
  const int px = bx & 1;
  dm_selectors(px, (unsigned)ix0, (unsigned)wl, selp, selq);
  const _Float16* base = a.lvl[L] + slot_tile * ((size_t)hl * wp * 64);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, hl * wp * 128, 0x00020000);
  unsigned coff[5], roff[8];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    // dword c holds window columns 2c - px and 2c + 1 - px
    const int i0 = 2 * c - px, i1 = i0 + 1;
    const bool need = (i0 >= 0 && i0 < 8 && (unsigned)(ix0 + i0) < (unsigned)wl) ||
                      (i1 >= 0 && i1 < 8 && (unsigned)(ix0 + i1) < (unsigned)wl);
    const int d = ((bx >> 1) + c) % (wp >> 1);          // small levels wrap more than once (wp / 2 may be < 5)
    coff[c] = need ? (unsigned)d * 256u : kOob;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ty = iy0 + j;
    int d = (by + j) % hl;
    d = d < 0 ? d + hl : d;
    roff[j] = (ty >= 0 && ty < hl) ? (unsigned)(d * (wp >> 1)) * 256u + (unsigned)lane * 4u : kOob;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int c = 0; c < 5; ++c)
      raw[j][c] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(roff[j] + coff[c]), 0, 0);
}

// ---- bilinear blend of one level: 7 window rows of 8 packed taps (tap 7 zero) ---------------------------------------
// out(i, j) = (((0 + s(i,j) w00) + s(i,j+1) w01) + s(i+1,j) w10) + s(i+1,j+1) w11   (correlation_kernels.cu:52-64: the four
// contributions reach corr[i][j] in exactly this order), evaluated for taps (2k, 2k+1) at once on the packed fp16 pipes
__device__ __forceinline__ void dm_blend(const unsigned (&raw)[8][5], float fdx, float fdy, const unsigned (&selp)[4],
                                         const unsigned (&selq)[4], u32x4 (&rows)[7]) {
  const h2 w00 = splat(to_half_rn((1.0f - fdx) * (1.0f - fdy)));
  const h2 w01 = splat(to_half_rn((1.0f - fdx) * fdy));
  const h2 w10 = splat(to_half_rn(fdx * (1.0f - fdy)));
  const h2 w11 = splat(to_half_rn(fdx * fdy));
  h2 P[2][4], Q[2][4];            // window rows j, j + 1: pairs (2k, 2k+1) and (2k+1, 2k+2)
  auto pack_row = [&](int j, h2 (&p)[4], h2 (&q)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      p[k] = perm2(raw[j][k + 1], raw[j][k], selp[k]);
      q[k] = perm2(raw[j][k + 1], raw[j][k], selq[k]);
    }
  };
  pack_row(0, P[0], Q[0]);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    pack_row(j + 1, P[(j + 1) & 1], Q[(j + 1) & 1]);
    const h2(&p0)[4] = P[j & 1];
    const h2(&q0)[4] = Q[j & 1];
    const h2(&p1)[4] = P[(j + 1) & 1];
    const h2(&q1)[4] = Q[(j + 1) & 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h2 acc = h2{(_Float16)0.0f, (_Float16)0.0f};
      acc = acc_term(acc, p0[k], w00);
      acc = acc_term(acc, p1[k], w01);
      acc = acc_term(acc, q0[k], w10);
      acc = acc_term(acc, q1[k], w11);
      unsigned u = __builtin_bit_cast(unsigned, acc);
      if (k == 3) u &= 0xffffu;           // tap 7 is padding
      rows[j][k] = u;
    }
  }
}

template <int L>
__device__ __forceinline__ void dm_level(const DmArgs& a, size_t slot_tile, int sy, int sx, int lane, float x0, float y0,
                                         bool live, size_t row) {
  unsigned raw[8][5], selp[4], selq[4];
  float fdx, fdy;
  dm_gather<L>(a, slot_tile, sy, sx, lane, x0, y0, raw, selp, selq, fdx, fdy);
  u32x4 rows[7];
  dm_blend(raw, fdx, fdy, selp, selq, rows);
  if (live) {
    u32x4* op = reinterpret_cast<u32x4*>(a.corr_cl + row * 256 + L * 64);
#pragma unroll
    for (int j = 0; j < 7; ++j) op[j] = rows[j];
    op[7] = u32x4{0u, 0u, 0u, 0u};
  }
}

// plain lookup (channels-last 256-channel result): grid (ceil(ntiles / 4), N); 256 threads = 4 waves = 4 consecutive tiles
__global__ __launch_bounds__(256) void corr_dm_lookup_kernel(DmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = a.ntx * a.nty;
  const int n = blockIdx.y;
  const int tile = blockIdx.x * 4 + wv;
  if (tile >= ntiles) return;
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int sy = ty * 8 + (lane >> 3), sx = tx * 8 + (lane & 7);
  const bool live = sy < a.h && sx < a.w;
  const int HW = a.h * a.w;
  const int p = min(sy, a.h - 1) * a.w + min(sx, a.w - 1);
  float x0, y0;
  if (a.coords_xy) {
    const float2 c = *reinterpret_cast<const float2*>(a.coords + ((size_t)n * HW + p) * 2);
    x0 = c.x; y0 = c.y;
  } else {
    x0 = a.coords[((size_t)n * 2 + 0) * HW + p];
    y0 = a.coords[((size_t)n * 2 + 1) * HW + p];
  }
  if (!live) { x0 = -1.0e6f; y0 = -1.0e6f; }          // padding lanes: every tap out of range, no memory access
  const size_t slot = a.slots ? (size_t)a.slots[n] : (size_t)n;
  const size_t slot_tile = slot * (size_t)ntiles + (size_t)tile;
  const size_t row = (size_t)n * HW + p;
  dm_level<0>(a, slot_tile, sy, sx, lane, x0, y0, live, row);
  dm_level<1>(a, slot_tile, sy, sx, lane, x0, y0, live, row);
  dm_level<2>(a, slot_tile, sy, sx, lane, x0, y0, live, row);
  dm_level<3>(a, slot_tile, sy, sx, lane, x0, y0, live, row);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused lookup + encoder, software-pipelined (the form the update step launches).
//   * a wave may have 63 vector-memory loads in flight (vmcnt), one level is 64: the loads of level l+1 are issued ROW BY
//     ROW into the registers of the window rows of level l as the blend retires them, so ~56-64 loads stay outstanding
//     from the first instruction to the last blend without a second register set;
//   * K is ordered l*56 + j*8 + i (7 rows x 8 taps per level, tap 7 zero: 224 = 14 k-steps of 16, the packing of
//     glorie_corr_otf_encode): a k-step is two consecutive window rows, pairs straddle levels (row 6 of level 0 waits for
//     row 0 of level 1);
//   * the encoder weights are staged into LDS behind the first 64 loads.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kEnc2Lds = 232;   // halfs per weight row in LDS (224 + 8)

// Per-level gather state.  The window's displaced rows are consecutive modulo the plane, so the row's byte offset is carried
// incrementally (+ one row, reset at the plane's end: 3 operations) next to an unsigned counter for the row's validity (3
// more) instead of being derived row by row; the five column offsets and the eight pair selectors are fixed per level.
// Offsets of invalid rows / unneeded dwords are replaced by kOob (the load returns 0 without a memory access).
struct DmLevel {
  __amdgpu_buffer_rsrc_t rs;
  unsigned coff[5];
  unsigned selp[4], selq[4];
  unsigned r, uy;                 // running: byte offset of the next window row (+ the lane's dword), its target row as unsigned
  unsigned rstep, rend, lane4, hl;
  h2 w00, w01, w10, w11;
};

template <int L>
__device__ __forceinline__ void dm_setup(const DmArgs& a, size_t slot_tile, int sy, int sx, int lane, float x0, float y0,
                                         DmLevel& st) {
  const int hl = a.h >> L, wl = a.w >> L, wp = (wl + 1) & ~1, wp2 = wp >> 1;
  const float inv = 1.0f / (float)(1 << L);
  const float xs = x0 * inv, ys = y0 * inv;
  const float fx = floorf(xs), fy = floorf(ys);
  const float fdx = xs - fx, fdy = ys - fy;
  const int ix0 = static_cast<int>(fx) - 3, iy0 = static_cast<int>(fy) - 3;
  // |displacement| < 2 planes for every coordinate that has a valid tap at all; others only need SOME in-range value
  int bx = ix0 - (sx >> L) + (wl >> 1), by = iy0 - (sy >> L) + (hl >> 1);
  bx = bx < 0 ? bx + wp : (bx >= wp ? bx - wp : bx);
  by = by < 0 ? by + hl : (by >= hl ? by - hl : by);
  const int px = bx & 1;
  dm_selectors(px, (unsigned)ix0, (unsigned)wl, st.selp, st.selq);
  const _Float16* base = a.lvl[L] + slot_tile * ((size_t)hl * wp * 64);
  st.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, hl * wp * 128, 0x00020000);
  const unsigned cend = (unsigned)wp2 * 256u;
  unsigned c = (unsigned)(bx >> 1) * 256u;
  // dword cc holds window columns 2 cc - px and 2 cc + 1 - px: needed if either has its target inside the map
  const unsigned ux = (unsigned)(ix0 - px);
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) {
    const bool n0 = (cc > 0 || px == 0) && (cc < 4 || px == 1) && ux + (unsigned)(2 * cc) < (unsigned)wl;
    const bool n1 = cc < 4 && ux + (unsigned)(2 * cc + 1) < (unsigned)wl;
    st.coff[cc] = (n0 || n1) ? c : kOob;
    c += 256u;
    c = c == cend ? 0u : c;
  }
  st.lane4 = (unsigned)lane * 4u;
  st.rstep = cend;
  st.rend = (unsigned)hl * cend + st.lane4;
  st.r = (unsigned)by * cend + st.lane4;
  st.uy = (unsigned)iy0;
  st.hl = (unsigned)hl;
  st.w00 = splat(to_half_rn((1.0f - fdx) * (1.0f - fdy)));
  st.w01 = splat(to_half_rn((1.0f - fdx) * fdy));
  st.w10 = splat(to_half_rn(fdx * (1.0f - fdy)));
  st.w11 = splat(to_half_rn(fdx * fdy));
}

// the next window row (rows are requested in order, 0..7): five dwords = ten displaced columns
__device__ __forceinline__ void dm_load_row(DmLevel& st, unsigned (&row)[5]) {
  const unsigned roff = st.uy < st.hl ? st.r : kOob;
  st.uy += 1u;
  st.r += st.rstep;
  st.r = st.r == st.rend ? st.lane4 : st.r;
#pragma unroll
#ifdef EXP_DM_NO_GATHER
  for (int i = 0; i < 5; ++i) row[i] = (roff + st.coff[i]) & 0x3c003c00u;          // ablation: arithmetic only, no memory
#else
  for (int i = 0; i < 5; ++i) row[i] = __builtin_amdgcn_raw_buffer_load_b32(st.rs, (int)(roff + st.coff[i]), 0, 0);
#endif
}

// the 4 aligned tap pairs (2k, 2k+1) and the 4 odd pairs (2k+1, 2k+2) of a window row
__device__ __forceinline__ void dm_pack_row(const unsigned (&row)[5], const DmLevel& st, h2 (&p)[4], h2 (&q)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p[k] = perm2(row[k + 1], row[k], st.selp[k]);
    q[k] = perm2(row[k + 1], row[k], st.selq[k]);
  }
}

// EXACT: the reference's accumulator starts at +0 and channel 7 of every row is a zero.  Both only matter to a caller that
// reads the lookup itself (0 + -0 = +0; the padding tap's blend is a finite sum of tap-7 values): the encoder-only form
// starts from the first product and leaves the padding lane as it falls (its weight column is zero)
template <bool EXACT>
__device__ __forceinline__ u32x4 dm_blend_row(const DmLevel& st, const h2 (&p0)[4], const h2 (&q0)[4], const h2 (&p1)[4],
                                              const h2 (&q1)[4]) {
  u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h2 acc;
    if (EXACT) acc = acc_term(h2{(_Float16)0.0f, (_Float16)0.0f}, p0[k], st.w00);
    else acc = acc_term0(p0[k], st.w00);
    acc = acc_term(acc, p1[k], st.w01);
    acc = acc_term(acc, q0[k], st.w10);
    acc = acc_term(acc, q1[k], st.w11);
    unsigned u = __builtin_bit_cast(unsigned, acc);
    if (EXACT && k == 3) u &= 0xffffu;
    r[k] = u;
  }
  return r;
}

// k-step ks: rows (2 ks, 2 ks + 1) in global row order g = 7 l + j
__device__ __forceinline__ void dm_kstep(int ks, u32x4 lo, u32x4 hi, const _Float16* wlds, int lane, f32x16 (&acc)[4][2]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(lo[v], hi[v], false, false);
    lo[v] = sw[0];
    hi[v] = sw[1];
  }
  const f16x8 b0 = __builtin_bit_cast(f16x8, lo), b1 = __builtin_bit_cast(f16x8, hi);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const f16x8 af = *reinterpret_cast<const f16x8*>(wlds + (mb * 32 + (lane & 31)) * kEnc2Lds + ks * 16 + (lane >> 5) * 8);
    acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b0, acc[mb][0], 0, 0, 0);
    acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b1, acc[mb][1], 0, 0, 0);
  }
}

// one level of the tile; `nxt` is the state of level L + 1: its window rows are requested into the registers this level
// retires (nothing follows level 3)
template <int L, bool CORR>
__device__ __forceinline__ void dm_level_pipelined(const DmArgs& a, bool live, size_t row, const _Float16* wlds, int lane,
                                                   unsigned (&raw)[8][5], const DmLevel& cur, DmLevel& nxt,
                                                   u32x4& pending, f32x16 (&acc)[4][2]) {
  h2 P[2][4], Q[2][4];
  dm_pack_row(raw[0], cur, P[0], Q[0]);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    dm_pack_row(raw[j + 1], cur, P[(j + 1) & 1], Q[(j + 1) & 1]);
    const u32x4 out = dm_blend_row<CORR>(cur, P[j & 1], Q[j & 1], P[(j + 1) & 1], Q[(j + 1) & 1]);
    if (L < 3) {                                    // window row j is retired: its registers take row j of what follows
      // (fenced: left alone, the scheduler gathers the requests of several rows into one clump behind a vmcnt(0), which
      // drains the queue ten times per tile)
      __builtin_amdgcn_sched_barrier(0);
      dm_load_row(nxt, raw[j]);
      if (j == 6) dm_load_row(nxt, raw[7]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (CORR && live) *reinterpret_cast<u32x4*>(a.corr_cl + row * 256 + L * 64 + j * 8) = out;
    const int g = 7 * L + j;
#ifdef EXP_DM_NO_MFMA
    acc[0][0][g & 15] += __uint_as_float(out[0] ^ out[1] ^ out[2] ^ out[3]);      // ablation: keep the blend alive, no MFMA
#else
    if (g & 1) dm_kstep(g >> 1, pending, out, wlds, lane, acc);
    else pending = out;
#endif
  }
  if (CORR && live) *reinterpret_cast<u32x4*>(a.corr_cl + row * 256 + L * 64 + 56) = u32x4{0u, 0u, 0u, 0u};
}

// a work unit = (edge n, tile): only the wave-uniform part is kept across the pipeline, the lane's pixel is recomputed
struct DmUnit { int n, tile, slot_tile; };
struct DmPixel { int sy, sx, pix; bool live; };
__device__ __forceinline__ void dm_unit(const DmArgs& a, int unit, DmUnit& u) {
  const int ntiles = a.ntx * a.nty;
  u.n = unit / ntiles;
  u.tile = unit - u.n * ntiles;
  const int slot = a.slots ? a.slots[u.n] : a.slot0 + u.n;
  u.slot_tile = __builtin_amdgcn_readfirstlane(slot * ntiles + u.tile);
}
__device__ __forceinline__ DmPixel dm_pixel(const DmArgs& a, const DmUnit& u, int lane) {
  DmPixel p;
  const int ty = u.tile / a.ntx, tx = u.tile - ty * a.ntx;
  p.sy = ty * 8 + (lane >> 3);
  p.sx = tx * 8 + (lane & 7);
  p.live = p.sy < a.h && p.sx < a.w;
  p.pix = min(p.sy, a.h - 1) * a.w + min(p.sx, a.w - 1);
  return p;
}
__device__ __forceinline__ float2 dm_coords(const DmArgs& a, const DmUnit& u, const DmPixel& p, bool on) {
  const int HW = a.h * a.w;
  float2 c;
  if (a.coords_xy) c = *reinterpret_cast<const float2*>(a.coords + ((size_t)u.n * HW + p.pix) * 2);
  else c = make_float2(a.coords[((size_t)u.n * 2 + 0) * HW + p.pix], a.coords[((size_t)u.n * 2 + 1) * HW + p.pix]);
  if (!(on && p.live)) { c.x = -1.0e6f; c.y = -1.0e6f; }          // padding lanes: every tap out of range, no memory access
  return c;
}

#ifdef EXP_DM_TIMESTAMPS
// experiment (tools/exp_corr_timeline.py): shader-clock stamps of a wave's phases, written over the first 64 bytes of the
// output row of the tile's first pixel
#define DM_STAMP(k) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k]) : : "memory")
#else
#define DM_STAMP(k)
#endif

// One tile per wave, 4 waves (consecutive tiles) per workgroup.  A persistent form (2 workgroups per CU, waves walking
// their units with the next tile's level 0 prefetched under the epilogue) was measured and is not faster: the launch is
// bound by the per-SIMD instruction stream (~3200 vector instructions per tile at 2 waves per SIMD), not by how the tiles
// are dealt (45 vs 43 us at G8); a ticket counter is much slower (2048 waves hit one word at once: ~90 tickets/us).
template <bool CORR>
__global__ __launch_bounds__(256, 2) void corr_dm_encode_kernel(DmArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 wlds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int total = a.N * a.ntx * a.nty;
  const int HW = a.h * a.w;
  const int unit = blockIdx.x * 4 + wv;
  const bool has = unit < total;
#ifdef EXP_DM_TIMESTAMPS
  unsigned long long stamp[13];
#endif
  DM_STAMP(0);

  DmUnit u;
  dm_unit(a, has ? unit : total - 1, u);
  const DmPixel px = dm_pixel(a, u, lane);
  DM_STAMP(10);
  float2 c = dm_coords(a, u, px, has);
#ifdef EXP_DM_TIMESTAMPS
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(c.x), "+v"(c.y) : : "memory");
  DM_STAMP(11);
#endif
  unsigned raw[8][5];
  DmLevel cur;
  dm_setup<0>(a, (size_t)u.slot_tile, px.sy, px.sx, lane, c.x, c.y, cur);
#ifdef EXP_DM_TIMESTAMPS
  asm volatile("" : "+v"(cur.coff[7]), "+v"(cur.r) : : "memory");
  DM_STAMP(12);
#endif
#pragma unroll
  for (int j = 0; j < 8; ++j) dm_load_row(cur, raw[j]);
  DM_STAMP(1);
#ifdef EXP_DM_TIMESTAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // latency of the first 64 gathers, measured alone
  DM_STAMP(8);
#endif
  // encoder weights [128][224] -> LDS rows of kEnc2Lds halfs + the bias, behind the first 64 gathers (128 * 28 = 14 * 256
  // pieces of 16 bytes: all 14 loads of a thread are issued before the first LDS store).  Requesting them before the unit's
  // coordinates was measured and is slower: vmcnt retires in order, so the coordinates then wait for the weights as well.
  u32x4 wreg[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) wreg[k] = *reinterpret_cast<const u32x4*>(a.enc_w + (size_t)(threadIdx.x + 256 * k) * 8);
  const float bias_in = a.enc_b[threadIdx.x & 127];
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    const int idx = threadIdx.x + 256 * k;
    const int r = idx / 28, cc = idx - r * 28;
    *reinterpret_cast<u32x4*>(wlds + r * kEnc2Lds + cc * 8) = wreg[k];
  }
  reinterpret_cast<float*>(wlds + 128 * kEnc2Lds)[threadIdx.x] = bias_in;      // (twice: no branch between the gathers
                                                                                 //  and their use)
  __syncthreads();
  DM_STAMP(2);

  // the accumulators start at the bias: C/D of 32x32 puts channel 32 mb + 8 g + 4 (lane >> 5) + q in register 4 g + q.
  // (A wave past the last unit runs on with every tap out of range and nothing stored.)
  f32x16 acc[4][2];
  {
    const float* bl = reinterpret_cast<const float*>(wlds + 128 * kEnc2Lds) + 4 * (lane >> 5);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * mb + 8 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[mb][0][4 * g + q] = acc[mb][1][4 * g + q] = b[q];
      }
  }
  u32x4 pending = u32x4{0u, 0u, 0u, 0u};
  const bool live = px.live && has;
  const size_t row = (size_t)u.n * HW + px.pix;
  DmLevel nxt;
  dm_setup<1>(a, (size_t)u.slot_tile, px.sy, px.sx, lane, c.x, c.y, nxt);
  dm_level_pipelined<0, CORR>(a, live, row, wlds, lane, raw, cur, nxt, pending, acc);
  DM_STAMP(3);
  cur = nxt;
  dm_setup<2>(a, (size_t)u.slot_tile, px.sy, px.sx, lane, c.x, c.y, nxt);
  dm_level_pipelined<1, CORR>(a, live, row, wlds, lane, raw, cur, nxt, pending, acc);
  DM_STAMP(4);
  cur = nxt;
  dm_setup<3>(a, (size_t)u.slot_tile, px.sy, px.sx, lane, c.x, c.y, nxt);
  dm_level_pipelined<2, CORR>(a, live, row, wlds, lane, raw, cur, nxt, pending, acc);
  DM_STAMP(5);
  dm_level_pipelined<3, CORR>(a, live, row, wlds, lane, raw, nxt, nxt, pending, acc);
  DM_STAMP(6);

  // epilogue.  C/D of 32x32: a lane holds channel 32 mb + 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of pixel 32 nb + (lane & 31);
  // swapping the upper half of block 0 with the lower half of block 1 leaves every lane with ITS pixel: 8 consecutive
  // channels = 16 bytes.  Stored from there, one instruction would write 64 pieces of 16 bytes into 64 different lines
  // (measured: ~380 clocks per store instruction, 6 k clocks per tile); instead the 64 bytes a pixel owns of each channel
  // block go through a 4 KB LDS stage of the wave (16-byte slots XOR-swizzled by the pixel, conflict-free both ways) and
  // leave with 4 lanes per pixel: 16 contiguous 64-byte runs per store.
  _Float16* stage = wlds + (128 * kEnc2Lds + 512) + wv * 2048;
  const int wslot = lane * 4, wsw = (lane >> 1) & 3;                   // own pixel: slots 4 lane + (g ^ wsw)
  const int tyx = u.tile / a.ntx, txx = u.tile - tyx * a.ntx;
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)a.enc_out, 0, a.enc_bytes, 0x00020000);
  unsigned ooff[4];                                                    // byte offset of the run this lane stores in pass k
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int P = 16 * k + (lane >> 2);
    const int sy = tyx * 8 + (P >> 3), sx = txx * 8 + (P & 7);
    const unsigned off = ((unsigned)(u.n * HW + sy * a.w + sx) * (unsigned)a.enc_stride + (unsigned)(lane & 3) * 8u) * 2u;
    ooff[k] = (has && sy < a.h && sx < a.w) ? off : kOobStore;         // outside the map: dropped by the range check
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mb][0][4 * g + r]),
                                                         __float_as_uint(acc[mb][1][4 * g + r]), false, false);
        v[r] = __uint_as_float(sw[0]);
        v[4 + r] = __uint_as_float(sw[1]);
      }
      u32x4 o;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {                  // ReLU after the rounding: one packed max per two channels
        const h2 t = h2{(_Float16)v[2 * cc], (_Float16)v[2 * cc + 1]};
        o[cc] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(t, h2{(_Float16)0.0f, (_Float16)0.0f}));
      }
      *reinterpret_cast<u32x4*>(stage + (wslot + (g ^ wsw)) * 8) = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32x4 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int P = 16 * k + (lane >> 2);
      o[k] = *reinterpret_cast<const u32x4*>(stage + (P * 4 + ((lane & 3) ^ ((P >> 1) & 3))) * 8);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b128(o[k], ors, (int)(ooff[k] + 64u * mb), 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#ifdef EXP_DM_TIMESTAMPS
  DM_STAMP(9);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DM_STAMP(7);
  if (lane == 0 && live)
    for (int k = 0; k < 13; ++k) reinterpret_cast<unsigned long long*>(a.enc_out + row * (size_t)a.enc_stride)[k] = stamp[k];
#endif
}

// (Round 4, measured and removed: a warm-up kernel on a side stream that touched the lookup's lines - per tile and level the
// bounding box of the 64 windows, one dword per line - as soon as the reprojection had produced the coordinates.  In the steps
// the lookup runs on a cold pyramid: 50 us against 40 us when launched a second time, tools/exp_corr_warm.py.  The warm-up
// took the lookup from 48.4 to 44.5 us but the step from 969 to 940 it/s: its 89 MB of HBM reads and the fork / join compete
// with the flow encoder's convolutions for more than they save.)
// ---------------------------------------------------------------------------------------------------------------------
// Builder: all-pairs <f1/4, f2/4> (fp16 GEMM, fp32 accumulate, rounded to fp16) + three avg_pool2d levels, written in the
// displacement-major layout.  Workgroup = half a source tile (4 x 8 pixels) x 8 aligned target rows: level 0 on the matrix
// cores into LDS R[px][8 rows][W8], the pooled levels reduced in LDS from the fp16 values of the level below (what
// avg_pool2d of the fp16 volume computes), then every (pixel, target) value is emitted to its line: a thread gathers the 8
// pixels of one tile row that share (dy, dx) and writes their 16 bytes.
// ---------------------------------------------------------------------------------------------------------------------
struct DmBuildArgs {
  const _Float16* f;          // [F][HW][128] channels-last feature maps, already scaled by 1/4
  const int64_t* ii; const int64_t* jj;
  const int* slot;
  _Float16* lvl[4];
  int h, w, ntx, nty, num_levels;
};

__device__ __forceinline__ unsigned pk2h(_Float16 a, _Float16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ _Float16 pool4h(_Float16 a, _Float16 b, _Float16 c, _Float16 d) {
  // avg_pool2d on half: float accumulation over (0,0),(0,1),(1,0),(1,1), times 1/4, rounded to half
  const float s = (((float)a + (float)b) + (float)c) + (float)d;
  return (_Float16)(s * 0.25f);
}

__global__ __launch_bounds__(256, 2) void corr_dm_build_kernel(DmBuildArgs a) {
  constexpr int C = 128, PX = 32;
  extern __shared__ __attribute__((aligned(16))) _Float16 bsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int h = a.h, w = a.w, HW = h * w;
  const int e = blockIdx.z, grp = blockIdx.y;
  const int tile = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
  const int ntiles = a.ntx * a.nty;
  const int fi = (int)a.ii[e], fj = (int)a.jj[e];
  const size_t sl = (size_t)a.slot[e];
  const int W8 = ((w + 7) >> 3) << 3;
  const int w1 = w >> 1, w2 = w >> 2, w3 = w >> 3, h1 = h >> 1, h2 = h >> 2, h3 = h >> 3;
  const int W81 = ((w1 + 7) >> 3) << 3, W82 = ((w2 + 7) >> 3) << 3, W83 = ((w3 + 7) >> 3) << 3;
  const int RS = 8 * W8 + 4, L1S = 4 * W81 + 2, L2S = 2 * W82 + 2, L3S = W83 + 2;   // per-pixel strides (padded: bank spread)
  _Float16* R = bsm;
  _Float16* L1 = R + PX * RS;
  _Float16* L2 = L1 + PX * L1S;
  _Float16* L3 = L2 + PX * L2S;

  // B operand: the 32 source pixels of this half tile (row syh = k >> 3 of the half, column k & 7), clamped into the map
  f16x8 bfrag[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int k = nt * 16 + col;
    const int sy = min(tyi * 8 + half * 4 + (k >> 3), h - 1), sx = min(txi * 8 + (k & 7), w - 1);
    const f16x8* src = reinterpret_cast<const f16x8*>(a.f + ((size_t)fi * HW + sy * w + sx) * C + kg * 8);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bfrag[nt][kk] = src[kk * 4];
  }
  // ---- level 0: targets t = local row * W8 + x of rows 8 grp .. 8 grp + 7 ----
  const int ntile = (8 * W8) >> 4;
  const _Float16* f2 = a.f + (size_t)fj * HW * C;
  for (int t16 = wv; t16 < ntile; t16 += 4) {
    const int t = t16 * 16 + col;
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
      *reinterpret_cast<uint2*>(R + (nt * 16 + col) * RS + t16 * 16 + kg * 4) =
          make_uint2(pk2h((_Float16)acc[0], (_Float16)acc[1]), pk2h((_Float16)acc[2], (_Float16)acc[3]));
    }
  }
  __syncthreads();
  // ---- pooled levels in LDS (each from the fp16 values of the level below) ----
  if (a.num_levels > 1) {
    for (int idx = tid; idx < PX * 4 * W81; idx += 256) {
      const int px = idx / (4 * W81), r = idx - px * (4 * W81), y = r / W81, x = r - y * W81;
      const _Float16* s = R + px * RS + (2 * y) * W8 + 2 * x;
      L1[px * L1S + r] = (x < w1) ? pool4h(s[0], s[1], s[W8], s[W8 + 1]) : (_Float16)0.0f;
    }
    __syncthreads();
  }
  if (a.num_levels > 2) {
    for (int idx = tid; idx < PX * 2 * W82; idx += 256) {
      const int px = idx / (2 * W82), r = idx - px * (2 * W82), y = r / W82, x = r - y * W82;
      const _Float16* s = L1 + px * L1S + (2 * y) * W81 + 2 * x;
      L2[px * L2S + r] = (x < w2) ? pool4h(s[0], s[1], s[W81], s[W81 + 1]) : (_Float16)0.0f;
    }
    __syncthreads();
  }
  if (a.num_levels > 3) {
    for (int idx = tid; idx < PX * W83; idx += 256) {
      const int px = idx / W83, x = idx - px * W83;
      const _Float16* s = L2 + px * L2S + 2 * x;
      L3[px * L3S + x] = (x < w3) ? pool4h(s[0], s[1], s[W82], s[W82 + 1]) : (_Float16)0.0f;
    }
    __syncthreads();
  }
  // ---- emission: segment = (source row syh of the half tile, local target row r, displacement-column PAIR dx2) -> the 32
  // bytes of 8 lanes x 2 columns (the inverse of dx = (tx - sxl + cx) mod wp; a displacement that names the padding column
  // of an odd width stores a zero)
  auto emit = [&](const _Float16* src, int stride, int Wp, int lvl, int rows, int hl, int wl) {
    const int cy = hl >> 1, cx = wl >> 1;
    const int wp = (wl + 1) & ~1, wp2 = wp >> 1;
    const int y0 = (8 * grp) >> lvl;
    _Float16* dst = a.lvl[lvl] + (sl * ntiles + tile) * ((size_t)hl * wp * 64);
    const int nseg = 4 * rows * wp2;
    for (int idx = tid; idx < nseg; idx += 256) {
      const int dx2 = idx % wp2, t = idx / wp2, r = t % rows, syh = t / rows;
      const int tyl = y0 + r, sy = tyi * 8 + half * 4 + syh;
      if (tyl >= hl || sy >= h) continue;
      int dy = tyl - (sy >> lvl) + cy;
      dy = dy < 0 ? dy + hl : (dy >= hl ? dy - hl : dy);
      u32x4 o[2];
#pragma unroll
      for (int lx = 0; lx < 8; ++lx) {
        const int sxl = min(txi * 8 + lx, w - 1) >> lvl;
        _Float16 v[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          int txl = 2 * dx2 + sub - cx + sxl;
          txl = txl < 0 ? txl + wp : (txl >= wp ? txl - wp : txl);
          v[sub] = txl < wl ? src[(syh * 8 + lx) * stride + r * Wp + txl] : (_Float16)0.0f;
        }
        o[lx >> 2][lx & 3] = pk2h(v[0], v[1]);
      }
      u32x4* out = reinterpret_cast<u32x4*>(dst + (((size_t)dy * wp2 + dx2) * 64 + (half * 4 + syh) * 8) * 2);
      out[0] = o[0];
      out[1] = o[1];
    }
  };
  emit(R, RS, W8, 0, 8, h, w);
  if (a.num_levels > 1) emit(L1, L1S, W81, 1, 4, h1, w1);
  if (a.num_levels > 2) emit(L2, L2S, W82, 2, 2, h2, w2);
  if (a.num_levels > 3) emit(L3, L3S, W83, 3, 1, h3, w3);
}

}  // namespace glorie

using namespace glorie;

extern "C" long glorie_corr_dm_level_halfs(int h, int w, int level) {
  if (h <= 0 || w <= 0 || level < 0 || level > 3) return -1;
  const long ntiles = (long)((h + 7) / 8) * ((w + 7) / 8);
  return ntiles * (long)(h >> level) * (long)(((w >> level) + 1) & ~1) * 64;
}

extern "C" int glorie_corr_dm_build(const void* fmaps_cl, const int64_t* ii, const int64_t* jj, const int* slots,
                                    void* const* levels, int num_levels, int n_new, int h, int w, int C, void* stream) {
  if (n_new < 0 || h <= 0 || w <= 0 || num_levels < 1 || num_levels > 4) return GLORIE_EINVAL;
  if (n_new == 0) return GLORIE_OK;
  if (!fmaps_cl || !ii || !jj || !slots || !levels) return GLORIE_EINVAL;
  if (C != 128 || (h >> (num_levels - 1)) < 1 || (w >> (num_levels - 1)) < 1) return GLORIE_EUNSUPPORTED;
  DmBuildArgs a{};
  a.f = reinterpret_cast<const _Float16*>(fmaps_cl);
  a.ii = ii; a.jj = jj; a.slot = slots; a.h = h; a.w = w; a.num_levels = num_levels;
  a.ntx = (w + 7) / 8; a.nty = (h + 7) / 8;
  for (int l = 0; l < num_levels; ++l) {
    if (!levels[l]) return GLORIE_EINVAL;
    a.lvl[l] = reinterpret_cast<_Float16*>(levels[l]);
  }
  auto pad8 = [](int v) { return ((v + 7) >> 3) << 3; };
  const size_t lds = sizeof(_Float16) * 32 *
                     (size_t)((8 * pad8(w) + 4) + (4 * pad8(w >> 1) + 2) + (2 * pad8(w >> 2) + 2) + (pad8(w >> 3) + 2));
  if (lds > 80 * 1024) return GLORIE_EUNSUPPORTED;
  static PerDeviceOnce attr;
  if (attr.first()) {
    GLORIE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(corr_dm_build_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)));
  }
  const dim3 grid(a.ntx * a.nty * 2, (h + 7) / 8, n_new);
  hipLaunchKernelGGL(corr_dm_build_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  return check_launch();
}

extern "C" int glorie_corr_dm_lookup(const void* const* levels, const int* slots, const float* coords, int coords_xy, int N,
                                     int h, int w, void* corr_cl, const void* enc_w, const float* enc_b, void* enc_out,
                                     int enc_stride, void* stream) {
  if (N < 0 || h <= 0 || w <= 0) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!levels || !coords || (!corr_cl && !enc_out)) return GLORIE_EINVAL;
  if ((h >> 3) < 1 || (w >> 3) < 1) return GLORIE_EUNSUPPORTED;
  const bool enc = enc_out != nullptr;
  if (enc && (!enc_w || !enc_b || enc_stride < 128 || (enc_stride & 7) || (reinterpret_cast<uintptr_t>(enc_out) & 15) ||
              (reinterpret_cast<uintptr_t>(enc_w) & 15)))
    return GLORIE_EINVAL;
  if (corr_cl && (reinterpret_cast<uintptr_t>(corr_cl) & 15)) return GLORIE_EINVAL;
  if (coords_xy && (reinterpret_cast<uintptr_t>(coords) & 7)) return GLORIE_EINVAL;
  DmArgs a{};
  for (int l = 0; l < 4; ++l) {
    if (!levels[l]) return GLORIE_EINVAL;
    a.lvl[l] = reinterpret_cast<const _Float16*>(levels[l]);
  }
  a.slots = slots; a.coords = coords; a.coords_xy = coords_xy; a.N = N; a.h = h; a.w = w;
  a.ntx = (w + 7) / 8; a.nty = (h + 7) / 8;
  a.corr_cl = reinterpret_cast<_Float16*>(corr_cl);
  a.enc_w = reinterpret_cast<const _Float16*>(enc_w); a.enc_b = enc_b;
  a.enc_out = reinterpret_cast<_Float16*>(enc_out); a.enc_stride = enc_stride;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long units = (long)N * a.ntx * a.nty;
  if (units > 0x7fffffffL / 4) return GLORIE_EUNSUPPORTED;
  if (!enc) {
    hipLaunchKernelGGL(corr_dm_lookup_kernel, dim3((a.ntx * a.nty + 3) / 4, N), dim3(256), 0, st, a);
    return check_launch();
  }
  const size_t lds = sizeof(_Float16) * 128 * kEnc2Lds + 256 * sizeof(float) + 4 * 4096;   // weights, bias, store stages
  static PerDeviceOnce enc_attr;
  if (enc_attr.first()) {
    GLORIE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(corr_dm_encode_kernel<false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)));
    GLORIE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(corr_dm_encode_kernel<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)));
  }
  // the output rows are stored through a buffer descriptor (32-bit offsets, out-of-map lanes dropped by its range check):
  // a call whose rows span 2 GB or more is issued in runs of edges
  const size_t edge_bytes = (size_t)h * w * enc_stride * sizeof(_Float16);
  if (edge_bytes >= 0x80000000ull) return GLORIE_EUNSUPPORTED;
  int per = (int)std::min<size_t>((size_t)N, std::max<size_t>(1, 0x7fffffffull / edge_bytes));
  if (const char* ce = getenv("GLORIE_CORR_DM_CHUNK")) per = std::max(1, std::min(per, atoi(ce)));   // tests: force the split
  for (int n0 = 0; n0 < N; n0 += per) {
    DmArgs c = a;
    c.N = std::min(per, N - n0);
    c.slots = slots ? slots + n0 : nullptr;
    c.slot0 = n0;
    c.coords = coords + (size_t)n0 * h * w * 2;
    if (corr_cl) c.corr_cl = a.corr_cl + (size_t)n0 * h * w * 256;
    c.enc_out = a.enc_out + (size_t)n0 * h * w * enc_stride;
    c.enc_bytes = (unsigned)(edge_bytes * c.N);
    const unsigned blocks = (unsigned)(((long)c.N * a.ntx * a.nty + 3) / 4);
    if (corr_cl) hipLaunchKernelGGL((corr_dm_encode_kernel<true>), dim3(blocks), dim3(256), lds, st, c);
    else hipLaunchKernelGGL((corr_dm_encode_kernel<false>), dim3(blocks), dim3(256), lds, st, c);
  }
  return check_launch();
}
