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
#include <stdlib.h>
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

// =====================================================================================================
// 8x8 source tiles sharing ONE target box per level ("otf8").
//
// Round 1's 4x4-per-wave kernel (removed in round 4) pulled every target row through L2 once per wave: ~6 KB of features per
// source pixel (1.55 GB of L2 traffic per G8 lookup).  Here a workgroup owns an 8 x 8 block of source pixels
// and ALL of it shares the bounding box of its 64 windows (smooth flow: 16 x 16 targets at level 0, 12 x 12,
// 10 x 10, 9 x 9 above: 581 target rows = 2.3 KB per source pixel).  The four waves split the box's 16-target
// tiles; a wave loads a tile's rows once (A operand, straight from L2 into registers, the next tile in
// flight) and multiplies it with all four 16-pixel source tiles, whose fragments (B operand, 64 VGPRs) stay
// in registers for the whole kernel: 16 MFMAs per 4 KB of loads.  D comes out as 4 consecutive targets of one
// pixel per lane = one ds_write_b64 into R[pixel][target] (fp16: the value a materialised volume would hold).
//   Targets outside the map are zero rows of A, so R holds exact zeros there and the window extraction needs no
// bounds logic: lane = (pixel, window row) reads 8 consecutive halfs of R (5 dwords + a funnel shift), takes
// row + 1 from the next lane (DPP) and produces its 7 outputs with the reference's fp16 rounding sequence.
//   If the 64 windows do not fit one box of <= kCap8 targets (flow discontinuities, random coordinates) the level
// falls back to the four 4 x 4 quadrants, and a quadrant that still does not fit to single pixels - always
// correct, fast where the flow is smooth.
//   Outputs of all levels are staged in LDS as T[pixel][level * 56 + row * 8 + i] (i = 7: zero pad): that is the
// K order of the optional fused corr_encoder[0] (1x1 convolution 196 -> 128 + bias + ReLU, droid_net.py:73-74) whose
// weights are packed to match, so the 196-channel map never goes to HBM: relu(W T + b) is written channels-last.
// =====================================================================================================
constexpr int kCap8 = 320;              // max targets of a shared box (16 x 20, 17 x 18, ...)
constexpr int kLdR8 = kCap8 + 4;        // halfs per pixel row of R: 162 dwords = 2 mod 32 -> conflict-free b64 writes
constexpr int kEncK = 224;              // 4 levels x 7 rows x 8 (7 taps + pad): K of the fused encoder
constexpr int kLdT8 = kEncK + 8;        // halfs per pixel row of T: 116 dwords -> conflict-free b128 reads

__device__ __forceinline__ unsigned otf_dpp_next(unsigned v) {
  // lane i <- lane i+1 inside a row of 16 lanes (row_shl:1)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
}
__device__ __forceinline__ _Float16 h_lo(unsigned v) { return __builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); }
__device__ __forceinline__ _Float16 h_hi(unsigned v) { return __builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }
__device__ __forceinline__ unsigned h_pack(_Float16 a, _Float16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// wave64 min / max without LDS: four row_shr steps reduce every row of 16 lanes into its last lane, row_bcast:15
// and row_bcast:31 carry the row results along (lane 63 ends up with all 64), v_readlane broadcasts.  The
// __shfl_xor butterflies of the 4x4 kernel above are ds_bpermute round trips: 24 dependent ones per level were
// 3.7k cycles of pure latency per workgroup here.
template <bool IS_MIN>
__device__ __forceinline__ int wave_reduce_dpp(int v) {
  const int ident = IS_MIN ? 0x7fffffff : (int)0x80000000;
  auto op = [](int a, int b) { return IS_MIN ? min(a, b) : max(a, b); };
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() carries a fence, for which hipcc waits vmcnt(0) in
// front of s_barrier: that drains the target rows prefetched for the NEXT level right where they were issued
// (cdna_hip_programming.md, "Pipelining across barriers").  Global loads are consumed by the wave that issued them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct OtfEnc {
  const _Float16* w;     // [128][kEncK] packed: w[o][l*56 + j*8 + i] = W[o][l*49 + i*7 + j], i == 7 -> 0
  const float* b;        // [128]
  _Float16* out;         // channels-last rows: out[(n*HW + p) * stride + o]
  int stride;
  int dbg;               // ablation switches (tools/bench_corr.py): 1 no MFMA phase, 2 no extraction, 4 no A loads
  unsigned long long* stamps;   // dbg & 32: s_memtime checkpoints of one workgroup's waves [4][32]
};

#define OTF_STAMP(k) do { if ((enc.dbg & 32) && blockIdx.x == 40 && blockIdx.y == 20 && lane == 0) enc.stamps[wv * 32 + (k)] = __builtin_readcyclecounter(); } while (0)

template <bool WRITE_CORR, bool ENCODE>
__global__ __launch_bounds__(256, 2) void corr_otf8_kernel(
    const _Float16* __restrict__ f1, OtfLevels lv, int num_levels, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, _Float16* __restrict__ out,
    int HW, int out_channels, OtfEnc enc) {
  constexpr int C = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  _Float16* R = reinterpret_cast<_Float16*>(smem8);                       // [64][kLdR8] (+ 8 halfs of read slack)
  _Float16* T = reinterpret_cast<_Float16*>(smem8) + 64 * kLdR8 + 8;       // [64][kLdT8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform: scalar branches on tile counts
  const int col = lane & 15, kg = lane >> 4;
  const int W0 = lv.w[0], H0 = lv.h[0];
  const int nbx = (W0 + 7) >> 3, nby = (H0 + 7) >> 3;
  // Blocks are dealt to the 8 XCDs round-robin (block b -> XCD b % 8, an observed placement used for speed
  // only).  Re-numbered so that every XCD walks ONE contiguous run of (edge, tile) pairs: the workgroups that
  // are resident on an XCD at a time then read the pyramid of one or two target frames (1.6 MB each) through
  // that XCD's 4 MB L2, instead of all XCDs streaming all 13 MB of feature maps.  Bijective for any grid size.
  int n, by, bx;
  {
    const int tiles = nbx * nby, total = tiles * (int)gridDim.y;
    const int b = (int)blockIdx.y * tiles + (int)blockIdx.x;
    const int q = total >> 3, rmd = total & 7, xcd = b & 7, k = b >> 3;
    const int nb_ = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + k;
    n = nb_ / tiles;
    const int t = nb_ - n * tiles;
    by = t / nbx;
    bx = t - by * nbx;
  }
  const int fi = (int)ii[n], fj = (int)jj[n];
  // pixel p (0..63) of the block: row 8 by + (p >> 3), column 8 bx + (p & 7); clamped into the map for loads
  auto pix_of = [&](int p) { return min(8 * by + (p >> 3), H0 - 1) * W0 + min(8 * bx + (p & 7), W0 - 1); };

  // B operand: the four 16-pixel source tiles, channels 32 kk + 8 kg .. + 7 of pixel 16 nt + col.  The 64 rows
  // (16 KB) are fetched ONCE per workgroup as full 256-byte lines (thread t: row t >> 2, four 16-byte chunks
  // 4 i + (t & 3)) into the R area and every wave reads all of its fragments from there - fragment-shaped loads
  // by all four waves pulled 64 KB through the texture path for the same 16 KB
  f16x8 bfrag[4][4];
  {
    const int r = tid >> 2, c4 = tid & 3;
    const f16x8* src = reinterpret_cast<const f16x8*>(f1 + ((size_t)fi * HW + pix_of(r)) * C);
    f16x8 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = src[4 * i + c4];
    // staging image: row r at r * 272 bytes (17 chunks: conflict-free b128 reads of one chunk column of 16 rows)
    f16x8* st = reinterpret_cast<f16x8*>(smem8);
#pragma unroll
    for (int i = 0; i < 4; ++i) st[r * 17 + 4 * i + c4] = v[i];
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) bfrag[nt][kk] = st[(nt * 16 + col) * 17 + kk * 4 + kg];
    __syncthreads();                     // R is written by the first MFMA phase
  }
  // coordinates: of pixel `lane` (for the boxes: every wave sees all 64 pixels) and of the two pixels whose
  // window rows this lane extracts (round r: pixel 32 r + 8 wv + (lane >> 3), row lane & 7)
  const float* cx = coords + ((size_t)n * 2 + 0) * HW;
  const float* cy = coords + ((size_t)n * 2 + 1) * HW;
  const float bxc = cx[pix_of(lane)], byc = cy[pix_of(lane)];
  const int row = lane & 7;
  int epx[2];
  float exc[2], eyc[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    epx[r] = 32 * r + 8 * wv + (lane >> 3);
    exc[r] = cx[pix_of(epx[r])];
    eyc[r] = cy[pix_of(epx[r])];
  }
  if (enc.dbg & 8) {                 // ablation: prologue only
    if (bfrag[0][0][0] == (_Float16)123.0f && bxc == 77.0f && exc[0] + exc[1] + eyc[0] + eyc[1] == 1.0f) out[0] = bfrag[3][3][7] + bfrag[1][2][1] + bfrag[2][0][0];
    return;
  }

  OTF_STAMP(0);
  // ---- boxes of all levels up front (wave-uniform scalars; the 16 reductions are independent of each other) ----
  auto origin = [&](float c, int l) { return static_cast<int>(floorf(fminf(fmaxf(c * (1.0f / (float)(1 << l)), -1.0e6f), 1.0e6f))) - 3; };
  int wx0_[4], wy0_[4], wbw_[4], wnb_[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int ox = origin(bxc, l), oy = origin(byc, l);
    const int ax0 = wave_reduce_dpp<true>(ox), ax1 = wave_reduce_dpp<false>(ox) + 7;
    const int ay0 = wave_reduce_dpp<true>(oy), ay1 = wave_reduce_dpp<false>(oy) + 7;
    const long long area = (long long)(ax1 - ax0 + 1) * (ay1 - ay0 + 1);
    wx0_[l] = ax0; wy0_[l] = ay0; wbw_[l] = ax1 - ax0 + 1;
    wnb_[l] = (l < num_levels && area <= kCap8) ? (int)area : 0;       // 0: no shared box at this level
  }
  // uniform selects instead of dynamically indexed arrays (those would live in scratch)
  auto sel = [](int l, const int (&v)[4]) { return l == 0 ? v[0] : (l == 1 ? v[1] : (l == 2 ? v[2] : v[3])); };
  OTF_STAMP(1);

  // A operand: tiles wv, wv + 4, ... of 16 target rows each (kCap8 / 16 = 20 tiles at most = 5 per wave).  The loads
  // of the first THREE tiles of a box are issued back to back - the rows come from the Infinity Cache / HBM the first
  // time an edge's workgroups touch them (~2k cycles; with one tile in flight per wave the matrix cores idled 80 %
  // of the phase) - the fourth and fifth tile of a large box follow into buffers 0 and 1 as those are multiplied
  // (a fourth buffer spills: 64 VGPRs of B fragments + the extraction's temporaries are live next to them).
  f16x8 abuf[3][4];
  unsigned okm = 0;                      // bit k: this lane's target row of the tile in buffer k is inside the map
  // (rows outside the map must be ZERO rows of A.  The mask is applied when the tile is multiplied: a select on the
  //  loaded value at issue time makes the compiler wait for every load right where it is issued.
  //  Tried: loading the tile as full 256-byte lines and turning it into fragments through a wave-private LDS image -
  //  four row addresses per lane instead of one spill the kernel, 108 -> 139 us.)
  auto load_tile = [&](int l, int gx0, int gy0, int bw, int nb, int k, int buf, f16x8 (&dst)[4]) {
    const int hl = lv.h[l], wl = lv.w[l];
    const _Float16* f2 = lv.f2[l] + (size_t)fj * hl * wl * C;
    const int t = (wv + 4 * k) * 16 + col;
    const int ty_ = (int)(((float)t + 0.5f) * (1.0f / (float)bw));
    const int ty = gy0 + ty_, tx = gx0 + (t - ty_ * bw);
    const bool ok = t < nb && tx >= 0 && tx < wl && ty >= 0 && ty < hl;
    okm = (okm & ~(1u << buf)) | ((ok ? 1u : 0u) << buf);
    const f16x8* src = reinterpret_cast<const f16x8*>(f2 + ((size_t)(ok ? ty * wl + tx : 0)) * C + kg * 8);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dst[kk] = src[kk * 4];
  };
  auto issue = [&](int l, int gx0, int gy0, int bw, int nb) {
    // unconditional: a tile beyond the box loads row 0 of the map (masked like a row outside the map) - straight-line
    // code keeps the three tiles' loads in flight together (hipcc waits vmcnt(0) at the joins of per-tile branches)
#pragma unroll
    for (int k = 0; k < 3; ++k) load_tile(l, gx0, gy0, bw, nb, k, k, abuf[k]);
  };
  auto mul_tile = [&](int tile, int buf, const f16x8 (&a)[4]) {
    const bool ok = (okm >> buf) & 1u;
    f16x8 am[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) am[kk] = ok ? a[kk] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[kk], bfrag[nt][kk], acc, 0, 0, 0);
      // lane: targets tile*16 + 4 kg .. + 3 of pixel 16 nt + col
      const unsigned lo = h_pack((_Float16)acc[0], (_Float16)acc[1]), hi = h_pack((_Float16)acc[2], (_Float16)acc[3]);
      *reinterpret_cast<uint2*>(R + (nt * 16 + col) * kLdR8 + tile * 16 + kg * 4) = make_uint2(lo, hi);
    }
  };
  // dense block R[pixel][t] = fp16(<f1[pixel], f2[target t]>) for this wave's tiles
  auto mfma_tiles = [&](int l, int gx0, int gy0, int bw, int nb) {
    if (enc.dbg & 1) return;
    // three tiles were issued ahead; tiles 3 and 4 of a large box follow into buffers 0 and 1 as those free up
    const bool has3 = (wv + 12) * 16 < nb, has4 = (wv + 16) * 16 < nb;
    if (wv * 16 < nb) mul_tile(wv, 0, abuf[0]);
    if (has3) load_tile(l, gx0, gy0, bw, nb, 3, 0, abuf[0]);
    if ((wv + 4) * 16 < nb) mul_tile(wv + 4, 1, abuf[1]);
    if (has4) load_tile(l, gx0, gy0, bw, nb, 4, 1, abuf[1]);
    if ((wv + 8) * 16 < nb) mul_tile(wv + 8, 2, abuf[2]);
    if (has3) mul_tile(wv + 12, 0, abuf[0]);
    if (has4) mul_tile(wv + 16, 1, abuf[1]);
  };
  // window extraction + bilinear blend (reference rounding sequence) of the member pixels of a box
  auto extract = [&](int l, int gx0, int gy0, int bw, int mode, int qd, int single_px) {
    const float inv = 1.0f / (float)(1 << l);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (enc.dbg & 2) break;
      const int px = epx[r];
      const float xs = fminf(fmaxf(exc[r] * inv, -1.0e6f), 1.0e6f), ys = fminf(fmaxf(eyc[r] * inv, -1.0e6f), 1.0e6f);
      const float fx = floorf(xs), fy = floorf(ys);
      const float dx = xs - fx, dy = ys - fy;
      const int eix = static_cast<int>(fx) - 3, eiy = static_cast<int>(fy) - 3;
      const _Float16 w00 = otf_weight((1.0f - dx) * (1.0f - dy)), w01 = otf_weight((1.0f - dx) * dy);
      const _Float16 w10 = otf_weight(dx * (1.0f - dy)), w11 = otf_weight(dx * dy);
      const bool member = mode == 0 || (mode == 1 ? ((((px >> 5) & 1) == (qd >> 1)) && (((px >> 2) & 1) == (qd & 1)))
                                                  : (px == single_px));
      // first half of this lane's window row inside R[px]; non-members read a harmless in-range spot
      const int e = member ? ((eiy + row - gy0) * bw + (eix - gx0)) : 0;
      const unsigned* rw = reinterpret_cast<const unsigned*>(R + px * kLdR8) + (e >> 1);
      unsigned d0 = rw[0], d1 = rw[1], d2 = rw[2], d3 = rw[3], d4 = rw[4];
      if (e & 1) {
        d0 = __builtin_amdgcn_alignbit(d1, d0, 16); d1 = __builtin_amdgcn_alignbit(d2, d1, 16);
        d2 = __builtin_amdgcn_alignbit(d3, d2, 16); d3 = __builtin_amdgcn_alignbit(d4, d3, 16);
      }
      const unsigned n0 = otf_dpp_next(d0), n1 = otf_dpp_next(d1), n2 = otf_dpp_next(d2), n3 = otf_dpp_next(d3);
      const _Float16 sv[8] = {h_lo(d0), h_hi(d0), h_lo(d1), h_hi(d1), h_lo(d2), h_hi(d2), h_lo(d3), h_hi(d3)};
      const _Float16 nx[8] = {h_lo(n0), h_hi(n0), h_lo(n1), h_hi(n1), h_lo(n2), h_hi(n2), h_lo(n3), h_hi(n3)};
      _Float16 o[8];
#pragma unroll
      for (int i = 0; i < 7; ++i) o[i] = otf_blend4(sv[i], nx[i], sv[i + 1], nx[i + 1], w00, w01, w10, w11);
      o[7] = (_Float16)0.0f;
      if (member && row < 7) {
        uint4 pk = make_uint4(h_pack(o[0], o[1]), h_pack(o[2], o[3]), h_pack(o[4], o[5]), h_pack(o[6], o[7]));
        *reinterpret_cast<uint4*>(T + px * kLdT8 + l * 56 + row * 8) = pk;
      }
    }
  };

  bool pre = false;                       // abuf holds the tiles of the whole-block box of the next level to run
  if (wnb_[0] > 0 && !(enc.dbg & 4)) { issue(0, wx0_[0], wy0_[0], wbw_[0], wnb_[0]); pre = true; }
  for (int l = 0; l < num_levels; ++l) {
    const int bx0_ = sel(l, wx0_), by0_ = sel(l, wy0_), bw_ = sel(l, wbw_), nb_ = sel(l, wnb_);
    if (nb_ > 0) {
      OTF_STAMP(2 + l * 5);
      if (!pre && !(enc.dbg & 4)) issue(l, bx0_, by0_, bw_, nb_);
      mfma_tiles(l, bx0_, by0_, bw_, nb_);
      pre = false;
      OTF_STAMP(3 + l * 5);
      // the rows of the next level travel while this level's windows are extracted
      if (l + 1 < num_levels && !(enc.dbg & 4)) {
        const int nn = sel(l + 1, wnb_);
        if (nn > 0) {
          issue(l + 1, sel(l + 1, wx0_), sel(l + 1, wy0_), sel(l + 1, wbw_), nn);
          pre = true;
        }
      }
      lds_barrier();
      OTF_STAMP(4 + l * 5);
      extract(l, bx0_, by0_, bw_, 0, 0, -1);
      OTF_STAMP(5 + l * 5);
      lds_barrier();
      OTF_STAMP(6 + l * 5);
    } else {
      // the 64 windows do not share a box of <= kCap8 targets: quadrants, then single pixels
      const int ox = origin(bxc, l), oy = origin(byc, l);
      for (int qd = 0; qd < 4; ++qd) {
        const bool mine = (((lane >> 5) & 1) == (qd >> 1)) && (((lane >> 2) & 1) == (qd & 1));   // pixel `lane` in quadrant qd
        const int qx0 = wave_reduce_dpp<true>(mine ? ox : 0x7fffffff), qx1 = wave_reduce_dpp<false>(mine ? ox : (int)0x80000000) + 7;
        const int qy0 = wave_reduce_dpp<true>(mine ? oy : 0x7fffffff), qy1 = wave_reduce_dpp<false>(mine ? oy : (int)0x80000000) + 7;
        const bool qfit = (long long)(qx1 - qx0 + 1) * (qy1 - qy0 + 1) <= kCap8;
        const int nsingle = qfit ? 1 : 16;
        for (int sg = 0; sg < nsingle; ++sg) {
          int gx0 = qx0, gy0 = qy0, bw = qx1 - qx0 + 1, nb = bw * (qy1 - qy0 + 1);
          int single_px = -1;
          if (!qfit) {   // pixel sg of the quadrant: block row 4 (qd >> 1) + (sg >> 2), column 4 (qd & 1) + (sg & 3)
            single_px = ((4 * (qd >> 1) + (sg >> 2)) << 3) + 4 * (qd & 1) + (sg & 3);
            gx0 = __shfl(ox, single_px, 64);
            gy0 = __shfl(oy, single_px, 64);
            bw = 8; nb = 64;
          }
          issue(l, gx0, gy0, bw, nb);
          mfma_tiles(l, gx0, gy0, bw, nb);
          lds_barrier();
          extract(l, gx0, gy0, bw, qfit ? 1 : 2, qd, single_px);
          lds_barrier();
        }
      }
    }
  }
  // (the last extraction ended with a barrier: T is complete)
  if (enc.dbg & 16) return;          // ablation: no output phase
  OTF_STAMP(22);
  const int rows = min(8, H0 - 8 * by), cols = min(8, W0 - 8 * bx);
  if (WRITE_CORR) {
    // out[n][l*49 + i*7 + j][y][x]: per (channel, block row) 8 pixels = 16 bytes
    const int total_ch = num_levels * 49;
    for (int idx = tid; idx < total_ch * 8; idx += 256) {
      const int ch = idx >> 3, ry = idx & 7;
      if (ry >= rows) continue;
      const int l = ch / 49, rem = ch - l * 49, i = rem / 7, j = rem - i * 7;
      const _Float16* src = T + (ry * 8) * kLdT8 + l * 56 + j * 8 + i;
      _Float16 v[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) v[x] = src[x * kLdT8];
      _Float16* dst = out + ((size_t)n * out_channels + ch) * HW + (size_t)(8 * by + ry) * W0 + 8 * bx;
      if (cols == 8 && (W0 & 7) == 0) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(h_pack(v[0], v[1]), h_pack(v[2], v[3]), h_pack(v[4], v[5]), h_pack(v[6], v[7]));
      } else {
        for (int x = 0; x < cols; ++x) dst[x] = v[x];
      }
    }
  }
  if (ENCODE) {
    // relu(W T + b): A = packed weights (16 output channels x 32 k per fragment, from L2), B = T rows (ds_read_b128);
    // wave wv owns output channels 32 wv .. + 31 for all 64 pixels
    f32x4 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (num_levels * 56 + 31) / 32;
    const f16x8* wsrc0 = reinterpret_cast<const f16x8*>(enc.w + (size_t)(32 * wv + col) * kEncK + kg * 8);
    const f16x8* wsrc1 = reinterpret_cast<const f16x8*>(enc.w + (size_t)(32 * wv + 16 + col) * kEncK + kg * 8);
    f16x8 a0 = wsrc0[0], a1 = wsrc1[0];
    for (int kk = 0; kk < nk; ++kk) {
      f16x8 a0n = a0, a1n = a1;
      if (kk + 1 < nk) { a0n = wsrc0[(kk + 1) * 4]; a1n = wsrc1[(kk + 1) * 4]; }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f16x8 b = *reinterpret_cast<const f16x8*>(T + (nt * 16 + col) * kLdT8 + kk * 32 + kg * 8);
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, acc[1][nt], 0, 0, 0);
      }
      a0 = a0n; a1 = a1n;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int o0 = 32 * wv + 16 * mt + 4 * kg;            // this lane: output channels o0 .. o0 + 3 of pixel 16 nt + col
      const float4 bias = *reinterpret_cast<const float4*>(enc.b + o0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int p = nt * 16 + col, ry = p >> 3, rx = p & 7;
        if (ry < rows && rx < cols) {
          const f32x4 v = acc[mt][nt];
          const _Float16 h0 = (_Float16)fmaxf(v[0] + bias.x, 0.f), h1 = (_Float16)fmaxf(v[1] + bias.y, 0.f);
          const _Float16 h2 = (_Float16)fmaxf(v[2] + bias.z, 0.f), h3 = (_Float16)fmaxf(v[3] + bias.w, 0.f);
          _Float16* dst = enc.out + ((size_t)n * HW + (size_t)(8 * by + ry) * W0 + 8 * bx + rx) * enc.stride + o0;
          *reinterpret_cast<uint2*>(dst) = make_uint2(h_pack(h0, h1), h_pack(h2, h3));
        }
      }
    }
  }
  OTF_STAMP(23);
}

}  // namespace glorie

using namespace glorie;

static int otf_levels(const void* const* fmap2_levels, int num_levels, int h, int w, OtfLevels& lv) {
  for (int l = 0; l < num_levels; ++l) {
    if (!fmap2_levels[l]) return GLORIE_EINVAL;
    lv.f2[l] = reinterpret_cast<const _Float16*>(fmap2_levels[l]);
    lv.h[l] = h >> l;
    lv.w[l] = w >> l;
  }
  return GLORIE_OK;
}

constexpr size_t kOtf8Lds = sizeof(_Float16) * (64 * kLdR8 + 8 + 64 * kLdT8);

template <bool WC, bool EN>
static int launch_otf8(const void* fmap1, const OtfLevels& lv, int num_levels, const float* coords, const int64_t* ii,
                       const int64_t* jj, void* out, int N, int h, int w, const OtfEnc& enc, hipStream_t st) {
  static PerDeviceOnce attr_set;
  if (attr_set.first()) {
    GLORIE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_otf8_kernel<WC, EN>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOtf8Lds)));
  }
  dim3 grid(((w + 7) / 8) * ((h + 7) / 8), N);
  hipLaunchKernelGGL((corr_otf8_kernel<WC, EN>), grid, dim3(256), kOtf8Lds, st, reinterpret_cast<const _Float16*>(fmap1),
                     lv, num_levels, coords, ii, jj, reinterpret_cast<_Float16*>(out), h * w, num_levels * 49, enc);
  return check_launch();
}

extern "C" int glorie_corr_otf(const void* fmap1, const void* const* fmap2_levels, int num_levels,
                               const float* coords, const int64_t* ii, const int64_t* jj, void* out,
                               int N, int h, int w, int C, void* stream) {
  if (N < 0 || h < 0 || w < 0 || num_levels < 1 || num_levels > 4) return GLORIE_EINVAL;
  if (N == 0 || h * w == 0) return GLORIE_OK;
  if (!fmap1 || !fmap2_levels || !coords || !ii || !jj || !out) return GLORIE_EINVAL;
  if (C != 128) return GLORIE_EUNSUPPORTED;
  OtfLevels lv{};
  GLORIE_TRY(otf_levels(fmap2_levels, num_levels, h, w, lv));
  OtfEnc e0{};
#ifdef EXP_OTF_DBG                         // instrumentation builds only (tools/otf_timeline.py)
  e0.dbg = getenv("GLORIE_OTF_DBG") ? atoi(getenv("GLORIE_OTF_DBG")) : 0;
  e0.stamps = getenv("GLORIE_OTF_STAMPS") ? (unsigned long long*)strtoull(getenv("GLORIE_OTF_STAMPS"), nullptr, 0) : nullptr;
  if (!e0.stamps) e0.dbg &= ~32;
#endif
  return launch_otf8<true, false>(fmap1, lv, num_levels, coords, ii, jj, out, N, h, w, e0, (hipStream_t)stream);
}

extern "C" int glorie_corr_otf_encode(const void* fmap1, const void* const* fmap2_levels, int num_levels,
                                      const float* coords, const int64_t* ii, const int64_t* jj, void* corr_out,
                                      int N, int h, int w, int C, const void* enc_w, const float* enc_b,
                                      void* enc_out, int enc_stride, void* stream) {
  if (N < 0 || h < 0 || w < 0 || num_levels != 4) return GLORIE_EINVAL;
  if (N == 0 || h * w == 0) return GLORIE_OK;
  if (!fmap1 || !fmap2_levels || !coords || !ii || !jj || !enc_w || !enc_b || !enc_out) return GLORIE_EINVAL;
  if (C != 128 || enc_stride < 128 || (enc_stride & 3) || (reinterpret_cast<uintptr_t>(enc_out) & 7)) return GLORIE_EUNSUPPORTED;
  OtfLevels lv{};
  GLORIE_TRY(otf_levels(fmap2_levels, num_levels, h, w, lv));
  OtfEnc enc{reinterpret_cast<const _Float16*>(enc_w), enc_b, reinterpret_cast<_Float16*>(enc_out), enc_stride, 0, nullptr};
#ifdef EXP_OTF_DBG
  enc.dbg = getenv("GLORIE_OTF_DBG") ? atoi(getenv("GLORIE_OTF_DBG")) : 0;
  enc.stamps = getenv("GLORIE_OTF_STAMPS") ? (unsigned long long*)strtoull(getenv("GLORIE_OTF_STAMPS"), nullptr, 0) : nullptr;
  if (!enc.stamps) enc.dbg &= ~32;
#endif
  if (corr_out)
    return launch_otf8<true, true>(fmap1, lv, num_levels, coords, ii, jj, corr_out, N, h, w, enc, (hipStream_t)stream);
  return launch_otf8<false, true>(fmap1, lv, num_levels, coords, ii, jj, nullptr, N, h, w, enc, (hipStream_t)stream);
}
