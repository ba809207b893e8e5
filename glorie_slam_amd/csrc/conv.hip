// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores with fused epilogues: the wide
// convolutions of the update operator (scope row A4; /root/reference/src/modules/droid_net/
// droid_net.py:69-139, gru.py:5-34), 75 % of a BA-update iteration.
//
//   out[p][n] = epilogue( sum_{tap d} sum_{c} x[p + off(d)][c] * w[d][n][c] )        fp16 in, fp32 accumulate
//
// * Operands are channels-last fp16 rows ([pixel][C]).  The input may be two channel segments
//   with their own base pointer and row stride (segment A: ca channels, segment B: cb): the GRU
//   input [net | inp, corr, flow] is never concatenated, and r*net can live in its own buffer
//   while other workgroups still read net as halo rows (a fused r*net epilogue that overwrote a
//   net slice in place would race with them).
// * Workgroup tile = 128 output channels x 128 (or 256) pixels, 4 waves as 2 x 2, each 64 channels
//   x 64 (128) pixels = 4 x 4 (4 x 8) accumulator blocks of v_mfma_f32_16x16x32_f16.  The MFMA "A"
//   operand is the WEIGHT tile, "B" the pixel tile, so a lane ends up with 4 consecutive output
//   channels of one pixel (8-byte epilogue loads/stores) instead of 4 pixels of one channel.
// * K loop = 64-channel chunks x taps (taps innermost: the 9 shifted reads of a chunk hit L2).
//   Both tiles are staged with buffer_load_dwordx4 ... lds (HBM/L2 -> LDS without touching VGPRs)
//   into ONE LDS stage (32 KB), 3 workgroups per CU.  Per step: barrier (tile landed) -> all 16
//   fragment reads -> barrier (every wave holds its fragments: the stage is free) -> DMA of the
//   next tile into the same stage -> 32 MFMAs from registers while it streams in.  Occupancy plus
//   this register-level double buffering measured faster than two or three LDS stages with fewer
//   resident workgroups (ST = 2 is kept as a template variant; the other measurements are listed at
//   the dispatch below).  The LDS image is
//   lane-linear as the DMA requires; the 16-byte slot of a row is XOR-swizzled on the SOURCE
//   address and on the fragment read (conflict-free b128 reads).
// * Zero padding: a lane whose row is outside the map for the current tap sets bit 31 of its
//   buffer offset; the hardware range check then writes zeros into LDS (tools/probes/
//   buffer_lds_probe.hip pins that behaviour) -- no branch and no zero page in the K loop.
// * Epilogues: bias + activation; the GRU z/r gates (sigmoid, r * net); the GRU blend
//   (1 - z) * net + z * tanh(.).  The per-edge global-context terms come from glorie_gru_glo_terms.
// * Consecutive workgroup ids are remapped so that one XCD (private L2) owns a contiguous range of
//   pixel tiles: the 9 taps, the halo rows and the output-channel tiles of a pixel range hit L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { EPI_BIAS_ACT = 0, EPI_GRU_ZR = 1, EPI_GRU_Q = 2, EPI_GLO = 3, EPI_HEADS = 4, EPI_UPSAMPLE = 5 };
enum { CACT_NONE = 0, CACT_RELU = 1, CACT_SIGMOID = 2 };

struct ConvArgs {
  const _Float16* xa; int xa_stride; int cha;   // segment A: cha 64-channel chunks
  const _Float16* xb; int xb_stride; int chb;   // segment B
  const _Float16* w;                            // [taps][npad][C] halfs (+ 64 halfs of padding)
  int taps, npad, nout;
  long P; int H, W, HW;
  _Float16* out; int out_stride;
  _Float16* out2; int out2_stride;
  const float* terms; int terms_stride; int act;
  const _Float16* net; int net_stride;
  const _Float16* z; int z_stride;
  const _Float16* pre; int pre_stride;          // per-pixel term added before the gate non-linearity (or null)
  const int* pre_map;                           // map (edge) -> map of `pre` it reads (null: its own)
  long pbeg;                                    // first pixel of this launch (a layer may be split into two launches)
  int pair;                                     // weight rows packed so that a lane owns 8 consecutive channels (conv_epilogue_tile)
  unsigned long long* stamps;                   // EXP_CONV_STAMPS builds: s_memtime checkpoints (tools/conv_timeline.py)
  int pp_full, pp_nbr;                          // conv_pp_kernel: workgroups with full (256-pixel) tiles, 16-pixel blocks per wave of the rest
  // EPI_HEADS: the first tap_groups 128-channel tiles feed the tap GEMM of a 3x3 head instead of being stored
  const f16x8* tap_w; float* tap_out; int tap_groups, tap_ncols;
  // EPI_UPSAMPLE: convex upsampling of the disparity maps with the tile's logits as the mask
  const float* up_disps; const int64_t* up_ix; float* up_out; int up_f32;
};

constexpr int kTileN = 128;       // output channels per workgroup
constexpr long kConvPpwMinPixels = 512L * 512;   // conv_ppw_kernel is the automatic choice from two rounds of 512-pixel tiles on (below: equal or slower)

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale / v_rcp / four FMAs / v_div_fmas / v_div_fixup per
// element): the gates are rounded to fp16 right after, and the epilogues evaluate 128 of these per lane
__device__ __forceinline__ float csigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ctanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// pixel of `pre` that output pixel p reads: maps that share their context features share one map of the term
__device__ __forceinline__ long pre_pixel(const ConvArgs& a, long p) {
  if (!a.pre_map) return p;
  const int e = (int)(p / a.HW);
  return (long)a.pre_map[e] * a.HW + (p - (long)e * a.HW);
}

// fused epilogue for the 4 consecutive output channels n..n+3 of pixel p (map e): bias + activation, the GRU gates
// (z = sigmoid, r * net) or the GRU blend
template <int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x4 v, long p, int e, int n) {
  f16x4 o;
  if (EPI == EPI_BIAS_ACT) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.terms) b = *reinterpret_cast<const float4*>(a.terms + n);
    float f[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (a.act == CACT_RELU) f[k] = fmaxf(f[k], 0.0f);
      else if (a.act == CACT_SIGMOID) f[k] = csigmoid(f[k]);
      o[k] = (_Float16)f[k];
    }
    *reinterpret_cast<f16x4*>(a.out + p * a.out_stride + n) = o;
  } else if (EPI == EPI_GRU_ZR) {
    // channels 0..127: z = sigmoid(.) ; 128..255: r -> r * net          (gru.py:28-30)
    const float4 g = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + n);
    const float s[4] = {csigmoid(v[0] + g.x), csigmoid(v[1] + g.y), csigmoid(v[2] + g.z),
                        csigmoid(v[3] + g.w)};
    if (n < 128) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (_Float16)s[k];
      *reinterpret_cast<f16x4*>(a.out + p * a.out_stride + n) = o;
    } else {
      const f16x4 nv = *reinterpret_cast<const f16x4*>(a.net + p * a.net_stride + (n - 128));
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (_Float16)(s[k] * (float)nv[k]);
      *reinterpret_cast<f16x4*>(a.out2 + p * a.out2_stride + (n - 128)) = o;
    }
  } else {
    // net' = (1 - z) * net + z * tanh(.)                                (gru.py:31-33)
    const float4 g = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + n);
    const f16x4 nv = *reinterpret_cast<const f16x4*>(a.net + p * a.net_stride + n);
    const f16x4 zv = *reinterpret_cast<const f16x4*>(a.z + p * a.z_stride + n);
    const float qv[4] = {ctanh(v[0] + g.x), ctanh(v[1] + g.y), ctanh(v[2] + g.z), ctanh(v[3] + g.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float zz = (float)zv[k];
      o[k] = (_Float16)((1.0f - zz) * (float)nv[k] + zz * qv[k]);
    }
    *reinterpret_cast<f16x4*>(a.out + p * a.out_stride + n) = o;
  }
}

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// Epilogue of a wave's MB x NB accumulator blocks without a branch: the lane owns channels nbase + 16*mi .. +3 of pixels
// pbase + 16*ni.  Every load is issued with a clamped (always valid) address and every store goes through a buffer
// descriptor with bit 31 of its offset set for pixels / channels past the end (the range check drops it) - a guarded
// load + store per block is a chain of MB * NB dependent memory round trips at the tail of every workgroup.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4e;

// The same epilogue for PAIRED weight rows (round 4; ConvArgs.pair, update_ops.pack_conv_igemm(pair=True)).  The ablation
// of the z|r gate launch (tools/bench_gate_epilogue.py: 263 us, of which loads 26, stores 21, sigmoid 15 - 211 us without the
// three) showed the epilogue's cost to be its vector-memory INSTRUCTIONS: a lane owns 4 channels per 16-row MFMA block, so
// every load / store moves 8 bytes per lane and the wave needs MB * NB of each.  With the rows of every 32-channel group
// packed as  row 16 blk + r  <->  channel 8 (r / 4) + 4 blk + r % 4,  the lane's 4 rows of blocks 2b and 2b + 1 are 8
// CONSECUTIVE channels: 16-byte loads and stores, half as many.  The per-map gate terms (float, identical for every pixel
// of a map) are loaded once per wave instead of once per pixel block and re-loaded only where a tile straddles two maps.
// Same products, same sums, same rounding: the outputs are bit-identical to the unpaired kernel's.
template <int EPI, int MB, int NB>
__device__ __forceinline__ void conv_epilogue_pair(const ConvArgs& a, f32x4 (&acc)[MB][NB], long pbase, int nwave) {
  static_assert(MB % 2 == 0, "pairs of 16-row blocks");
  constexpr int PB = MB / 2;
  const int kg = (threadIdx.x & 63) >> 4;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_GRU_ZR ? a.out2 : a.out), 0,
                                                                      0x7fffffff, 0x00020000);
  const bool upper = EPI == EPI_GRU_ZR && nwave >= 128;          // wave-uniform: this wave holds r (channels 128..255)
  const bool late = EPI != EPI_BIAS_ACT && a.pre;                // paired launches always add the context term here
  int nch[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) nch[b] = min(nwave + 32 * b + 8 * kg, a.nout - 8);
  float4 g[PB][2];
  int e_cur = -1;
  if (EPI == EPI_BIAS_ACT) {
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      g[b][0] = a.terms ? *reinterpret_cast<const float4*>(a.terms + nch[b]) : make_float4(0.f, 0.f, 0.f, 0.f);
      g[b][1] = a.terms ? *reinterpret_cast<const float4*>(a.terms + nch[b] + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    const long p = pbase + ni * 16;
    const bool okp = p < a.P;
    const long pc = okp ? p : a.P - 1;
    const int e = (int)(pc / a.HW);
    if (EPI != EPI_BIAS_ACT && __builtin_amdgcn_ballot_w64(e != e_cur) != 0ull) {      // first block, or the tile crosses into the next map
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        g[b][0] = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + nch[b]);
        g[b][1] = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + nch[b] + 4);
      }
      e_cur = e;
    }
    const long pp = late ? pre_pixel(a, pc) : 0;
    f16x8 nv[PB], zv[PB], pl[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if (late) pl[b] = *reinterpret_cast<const f16x8*>(a.pre + pp * a.pre_stride + nch[b]);
      if (EPI == EPI_GRU_Q) {
        nv[b] = *reinterpret_cast<const f16x8*>(a.net + pc * a.net_stride + nch[b]);
        zv[b] = *reinterpret_cast<const f16x8*>(a.z + pc * a.z_stride + nch[b]);
      } else if (upper) {
        nv[b] = *reinterpret_cast<const f16x8*>(a.net + pc * a.net_stride + (nch[b] - 128));
      }
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const int n = nwave + 32 * b + 8 * kg;
      f16x8 o;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        f32x4 v = acc[2 * b + hb][ni];
        const float4 gg = g[b][hb];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int s = 4 * hb + k;
          float f = v[k];
          if (late) f += (float)pl[b][s];
          f += (k == 0 ? gg.x : k == 1 ? gg.y : k == 2 ? gg.z : gg.w);
          if (EPI == EPI_BIAS_ACT) {
            if (a.act == CACT_RELU) f = fmaxf(f, 0.0f);
            else if (a.act == CACT_SIGMOID) f = csigmoid(f);
            o[s] = (_Float16)f;
          } else if (EPI == EPI_GRU_ZR) {
            o[s] = (_Float16)(upper ? csigmoid(f) * (float)nv[b][s] : csigmoid(f));
          } else {
            const float zz = (float)zv[b][s];
            o[s] = (_Float16)((1.0f - zz) * (float)nv[b][s] + zz * ctanh(f));
          }
        }
      }
      const bool ok = okp && n + 8 <= a.nout;
      if (upper) {
        const unsigned vo = ok ? (unsigned)((pc * a.out2_stride + (n - 128)) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, o), rO2, vo, 0, 0);
      } else {
        const unsigned vo = ok ? (unsigned)((pc * a.out_stride + n) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, o), rO, vo, 0, 0);
      }
    }
  }
}

template <int EPI, int MB, int NB>
__device__ __forceinline__ void conv_epilogue_tile(const ConvArgs& a, f32x4 (&acc)[MB][NB], long pbase, int nbase,
                                                   int out_ch0 = 0) {
  if constexpr (MB % 2 == 0 && (EPI == EPI_BIAS_ACT || EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q)) {
    if (a.pair) {                                                  // launch-uniform
      conv_epilogue_pair<EPI, MB, NB>(a, acc, pbase, nbase - 4 * (int)((threadIdx.x & 63) >> 4));
      return;
    }
  }
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_GRU_ZR ? a.out2 : a.out), 0,
                                                                      0x7fffffff, 0x00020000);
  const bool upper = EPI == EPI_GRU_ZR && nbase >= 128;          // wave-uniform: this wave holds r (channels 128..255)
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    const long p = pbase + ni * 16;
    const bool okp = p < a.P;
    const long pc = okp ? p : a.P - 1;
    const int e = (int)(pc / a.HW);
    float4 g[MB];
    f16x4 nv[MB], zv[MB], pl[MB];
    const bool late = EPI != EPI_BIAS_ACT && a.pre;                     // launch-uniform
    const long pp = late ? pre_pixel(a, pc) : 0;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const int nc = min(nbase + mi * 16, a.nout - 4);
#ifdef EXP_EPI_NO_LOADS
      // ablation (tools/exp_conv_wreg.py): what do the epilogue's scattered 8-byte loads cost?
      pl[mi] = f16x4{1, 1, 1, 1}; nv[mi] = f16x4{1, 1, 1, 1}; zv[mi] = f16x4{1, 1, 1, 1};
      g[mi] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
#endif
      if (late) pl[mi] = *reinterpret_cast<const f16x4*>(a.pre + pp * a.pre_stride + nc);
      if (EPI == EPI_BIAS_ACT) {
        g[mi] = a.terms ? *reinterpret_cast<const float4*>(a.terms + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        g[mi] = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + nc);
        if (EPI == EPI_GRU_Q) {
          nv[mi] = *reinterpret_cast<const f16x4*>(a.net + pc * a.net_stride + nc);
          zv[mi] = *reinterpret_cast<const f16x4*>(a.z + pc * a.z_stride + nc);
        } else if (upper) {
          nv[mi] = *reinterpret_cast<const f16x4*>(a.net + pc * a.net_stride + (nc - 128));
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const int n = nbase + mi * 16;
      f32x4 v = acc[mi][ni];
      if (late) { v[0] += (float)pl[mi][0]; v[1] += (float)pl[mi][1]; v[2] += (float)pl[mi][2]; v[3] += (float)pl[mi][3]; }
      const float f[4] = {v[0] + g[mi].x, v[1] + g[mi].y, v[2] + g[mi].z, v[3] + g[mi].w};
      f16x4 o;
      if (EPI == EPI_BIAS_ACT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = f[k];
          if (a.act == CACT_RELU) t = fmaxf(t, 0.0f);
          else if (a.act == CACT_SIGMOID) t = csigmoid(t);
          o[k] = (_Float16)t;
        }
      } else if (EPI == EPI_GRU_ZR) {
        // channels 0..127: z = sigmoid(.) ; 128..255: r -> r * net          (gru.py:28-30)
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (_Float16)(upper ? csigmoid(f[k]) * (float)nv[mi][k] : csigmoid(f[k]));
      } else {
        // net' = (1 - z) * net + z * tanh(.)                                (gru.py:31-33)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zz = (float)zv[mi][k];
          o[k] = (_Float16)((1.0f - zz) * (float)nv[mi][k] + zz * ctanh(f[k]));
        }
      }
#ifdef EXP_EPI_NO_SIGMOID
      if (EPI != EPI_BIAS_ACT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (_Float16)f[k];
      }
#endif
#ifdef EXP_EPI_NO_STORE
      if (o[0] != (_Float16)12345.0f) continue;
#endif
      const bool ok = okp && n < a.nout;
      if (upper) {
        const unsigned vo = ok ? (unsigned)((pc * a.out2_stride + (n - 128)) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rO2, vo, 0, 0);
      } else {
        const unsigned vo = ok ? (unsigned)((pc * a.out_stride + (n - out_ch0)) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rO, vo, 0, 0);
      }
    }
  }
}

// EPI_HEADS: the tile's 128 channels are the hidden layer of one 3x3 head (droid_net.py:85-93: conv3x3 -> ReLU -> conv3x3
// 128 -> K).  The hidden activations are never stored: relu(acc + bias) as fp16 IS the MFMA B operand of the head's tap
// GEMM  taps[d*K + j][pixel] = < w2[j][:, d], hidden[:, pixel] >  (the K order of a GEMM is free: k-slot (kg, s) of chunk c
// is channel wm*64 + (2c + s/4)*16 + kg*4 + s%4, the weight fragments are packed to match, update_ops.
// pack_head_taps).  Each channel half (wave pair) does 16 MFMAs, the halves are summed through LDS and the float tap rows
// are written as planes [group*9K + row][pixel]; glorie_conv_stencil finishes the head.  Saves the store and
// the re-read of 2 x 128 channels per pixel (88 + 88 MB per iteration at 36x60x80) and the tap kernel's launch.
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue_heads(const ConvArgs& a, f32x4 (&acc)[MB][NB], long p0, int n0, int wm,
                                                    int wn, int kg, int col, int lane, char* smem) {
  static_assert(MB == 4 && NB == 4, "128 x 128 tile");
  const int grp = n0 >> 7;
  const f16x8* wp = a.tap_w + (size_t)(grp * 2 + wm) * 4 * 64 + lane;          // [group][half][chunk*2 + row block][64]
  f16x8 wf[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) wf[c][rb] = wp[(c * 2 + rb) * 64];
  float4 b[MB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) b[mi] = *reinterpret_cast<const float4*>(a.terms + n0 + wm * 64 + mi * 16 + kg * 4);
  f32x4 t[2][NB];
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    f16x8 hf[2];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const f32x4 v = acc[mi][ni];
      const int h = (mi & 1) * 4;
      hf[mi >> 1][h + 0] = (_Float16)fmaxf(v[0] + b[mi].x, 0.0f);
      hf[mi >> 1][h + 1] = (_Float16)fmaxf(v[1] + b[mi].y, 0.0f);
      hf[mi >> 1][h + 2] = (_Float16)fmaxf(v[2] + b[mi].z, 0.0f);
      hf[mi >> 1][h + 3] = (_Float16)fmaxf(v[3] + b[mi].w, 0.0f);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) s = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][rb], hf[c], s, 0, 0, 0);
      t[rb][ni] = s;
    }
  }
  // t[rb][ni][r] = tap row rb*16 + kg*4 + r of pixel wn*64 + ni*16 + col, summed over this wave's 64 channels
  float* red = reinterpret_cast<float*>(smem);          // [pixel wave][8 blocks][4][64 lanes]
  __syncthreads();                                      // the stage is no longer read
  if (wm == 1) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ni = 0; ni < NB; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wn * 8 + rb * 4 + ni) * 4 + r) * 64 + lane] = t[rb][ni][r];
  }
  __syncthreads();
  if (wm == 0) {
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
      const long p = p0 + wn * 64 + ni * 16 + col;
      if (p >= a.P) continue;
      float* dst = a.tap_out + (size_t)(grp * a.tap_ncols) * a.P + p;          // tap planes [groups * ncols][P]
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int n = rb * 16 + kg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = t[rb][ni][r] + red[((wn * 8 + rb * 4 + ni) * 4 + r) * 64 + lane];
          if (n + r < a.tap_ncols) dst[(size_t)(n + r) * a.P] = v;
        }
      }
    }
  }
}

// EPI_GLO: the global-context reduction of the ConvGRU (gru.py:25-26) as the epilogue of its 1x1 convolution w: nothing
// is stored per pixel; the workgroup reduces  sigmoid(acc + bias[n]) * net[p][n]  over the pixels of its tile that belong
// to its map (tiles are laid per map, see the kernel) and writes one partial sum per channel to out = float
// [map][tile of the map][128]; glorie_gru_glo_from_tiles sums the tiles of a map in a fixed order.  Replaces a 44 MB store +
// two 44 MB reads of the intermediate map.
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue_glo(const ConvArgs& a, f32x4 (&acc)[MB][NB], long p0, long pbase, int wm,
                                                  int wn, int kg, int col, char* smem, int tile) {
  static_assert(MB == 4, "128-channel tile");
  const long pend = (p0 / a.HW + 1) * a.HW;             // end of this tile's map
  float s0[MB][4];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int k = 0; k < 4; ++k) s0[mi][k] = 0.0f;
  const int nb = wm * 64 + kg * 4;
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    const long p = pbase + ni * 16;
    const bool okp = p < pend;
    const long pc = okp ? p : pend - 1;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const int n = nb + mi * 16;
      const float4 b = *reinterpret_cast<const float4*>(a.terms + n);
      const f16x4 nv = *reinterpret_cast<const f16x4*>(a.net + pc * a.net_stride + n);
      const f32x4 v = acc[mi][ni];
      s0[mi][0] += okp ? csigmoid(v[0] + b.x) * (float)nv[0] : 0.0f;
      s0[mi][1] += okp ? csigmoid(v[1] + b.y) * (float)nv[1] : 0.0f;
      s0[mi][2] += okp ? csigmoid(v[2] + b.z) * (float)nv[2] : 0.0f;
      s0[mi][3] += okp ? csigmoid(v[3] + b.w) * (float)nv[3] : 0.0f;
    }
  }
  // sum over the 16 pixel lanes of the block column (lanes that share kg), then over the two pixel waves through LDS
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) s0[mi][k] += __shfl_xor(s0[mi][k], off, 64);
  float* red = reinterpret_cast<float*>(smem);          // [pixel wave][128]
  __syncthreads();                                      // the stage is no longer read
  if (col == 0) {
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[wn * 128 + nb + mi * 16 + k] = s0[mi][k];
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < 128) reinterpret_cast<float*>(a.out)[(size_t)tile * 128 + tid] = red[tid] + red[128 + tid];
}

// EPI_UPSAMPLE: GraphAgg's upmask convolution (droid_net.py:46-48,62-64: 1x1, 128 -> 576 = 9 taps x 8 x 8 sub-pixels) with the
// convex upsampling of the disparity map (droid_net.py:9-23, depth_video.py:140-144) as its epilogue: the 44 MB of fp16 logits
// are never written and read back.  The weight rows are packed as channel' = a*128 + b*16 + tap (sub-row a, sub-column b,
// taps padded 9 -> 16 with zero rows; update_ops.pack_upmask_conv), so the workgroup of channel tile a holds, for its 128
// pixels, all 9 taps of the 8 sub-pixels of sub-row a.  The logits (+ bias) go to LDS as fp16 - the rounding the stored map
// had - and 128 x 2 threads each finish 4 sub-pixels of one pixel with the arithmetic of cvx_upsample_nhwc_kernel, in its
// order: the same bits as the two-launch form.
__device__ __forceinline__ float round_prob(float v, bool f32) { return f32 ? v : (float)(_Float16)v; }

template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue_upsample(const ConvArgs& a, f32x4 (&acc)[MB][NB], long p0, int n0, int wm,
                                                       int wn, int kg, int col, char* smem) {
  static_assert(MB == 4 && NB == 4, "128 x 128 tile");
  _Float16* T = reinterpret_cast<_Float16*>(smem);      // [128 pixels][8 chunks of 16 halfs], chunk b at b ^ (pixel & 7)
  float4 bias[MB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) bias[mi] = *reinterpret_cast<const float4*>(a.terms + n0 + wm * 64 + mi * 16 + kg * 4);
  __syncthreads();                                      // the stage is no longer read
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    const int pl = wn * 64 + ni * 16 + col;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const int b = wm * 4 + mi;
      const f32x4 v = acc[mi][ni];
      const f16x4 h = {(_Float16)(v[0] + bias[mi].x), (_Float16)(v[1] + bias[mi].y), (_Float16)(v[2] + bias[mi].z),
                       (_Float16)(v[3] + bias[mi].w)};
      *reinterpret_cast<f16x4*>(T + pl * 128 + ((b ^ (pl & 7)) * 16 + kg * 4)) = h;
    }
  }
  __syncthreads();
  const int tid = threadIdx.x;
  const int pl = tid & 127, hb = tid >> 7;              // pixel of the tile, sub-columns 4*hb .. 4*hb + 3
  const long p = p0 + pl;
  if (p >= a.P) return;
  const int asub = n0 >> 7;
  const int m = (int)(p / a.HW);
  const int q = (int)(p - (long)m * a.HW);
  const int y = q / a.W, x = q - y * a.W;
  const int frame = (int)a.up_ix[m];
  const float* dmap = a.up_disps + (size_t)frame * a.HW;
  float nb[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    nb[k] = dmap[in ? yy * a.W + xx : q];
    if (!in) nb[k] = 0.0f;
  }
  float outv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = hb * 4 + j;
    const f16x8* src = reinterpret_cast<const f16x8*>(T + pl * 128 + (b ^ (pl & 7)) * 16);
    const f16x8 lo = src[0], hi = src[1];
    float v[9];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)lo[k];
    v[8] = (float)hi[0];
    float mx = v[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) mx = fmaxf(mx, v[k]);
    float e[9], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { e[k] = expf(v[k] - mx); sum += e[k]; }
    float o = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k)                          // product and sum rounded separately (no contraction), like the
      o = __fadd_rn(o, __fmul_rn(round_prob(e[k] / sum, a.up_f32 != 0), nb[k]));   // stand-alone kernel and torch
    outv[j] = o;
  }
  float* dst = a.up_out + (size_t)frame * 64 * a.HW + (size_t)(8 * y + asub) * (8 * a.W) + 8 * x + hb * 4;
  *reinterpret_cast<float4*>(dst) = make_float4(outv[0], outv[1], outv[2], outv[3]);
}

// NB = 16-pixel blocks per wave, BK = channels per K step (32 | 64), NW = waves (2 channel halves x
// NW/2 pixel groups: pixel tile = NW/2 * 16*NB), ST = LDS stages (1: single stage + register-resident
// fragments, the shipped form; 2: classic double buffering with one __syncthreads per step)
template <int EPI, int NB, int BK, int NW, int ST, int MB = 4>
__global__ __launch_bounds__(64 * NW, (ST == 1 && MB * NB <= 16) ? 3 : 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int TN = 32 * MB;               // output channels per workgroup (MB 16-channel blocks per wave)
  static_assert(ST == 1 || ST == 2, "LDS stages");
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource types are device-only)
  constexpr int PT = (NW / 2) * 16 * NB;      // pixels per workgroup
  constexpr int RB = BK * 2;                  // bytes per staged row
  constexpr int SL = RB / 16;                 // 16-byte slots per row
  constexpr int RPI = 64 / SL;                // rows per wave-wide DMA instruction
  constexpr int XI = PT / RPI / NW;           // DMA instructions per wave per step: pixel tile
  constexpr int WI = TN / RPI / NW;           //                                      weight tile
  constexpr int XBYTES = PT * RB, WBYTES = TN * RB;
  constexpr int KK = BK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ST x (pixel tile, weight tile), one array
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int wm = wv & 1, wn = wv >> 1;

  // 16-byte slot swizzle key of an LDS row: ds_read_b128 of 16 consecutive rows is conflict-free for
  // the hardware's lane groups ({0-3,12-15,20-27}, ...) with these keys (brute-force checked)
  auto key = [](int row) { return RB == 128 ? (row & 7) : ((row >> 1) & 3); };

  // XCD-aware (bijective) remap of the workgroup id, then (pixel tile, output-channel tile)
  const int nwg = gridDim.x, ntn = (a.nout + TN - 1) / TN;
  const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  const int pt = lid / ntn, nt = lid - pt * ntn;
  // EPI_GLO tiles every map separately (ceil(HW / PT) tiles per map, the last one ragged): the grouping of a map's pixel
  // sums then does not depend on where the map sits in the batch (the result is invariant under edge permutations)
  const int tpm = (a.HW + PT - 1) / PT;
  const long p0 = EPI == EPI_GLO ? (long)(pt / tpm) * a.HW + (long)(pt % tpm) * PT : a.pbeg + (long)pt * PT;
  const int n0 = nt * TN;

  const int cpc = 64 / BK;                    // K steps per 64-channel chunk
  const int nsteps_tap = (a.cha + a.chb) * cpc;
  const int C = (a.cha + a.chb) * 64;
  const int T = a.taps * nsteps_tap;

  // Staging through buffer descriptors: address = base + SGPR offset (tap shift, channel chunk:
  // wave-uniform, changes per step) + VGPR offset (row, swizzled slot: per lane, loop-invariant).
  // A lane whose row falls outside the map for this tap gets bit 31 set in its VGPR offset: the
  // hardware range check fails and the DMA writes ZEROS into LDS -- zero padding costs three VALU
  // instructions per load and no branch.  The descriptor base sits `back` rows before the tensor
  // so that the SGPR offset of the (-1,-1) tap is not negative.
  const int back = a.taps == 9 ? a.W + 1 : 0;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xa - (long)back * a.xa_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xb - (long)back * a.xb_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);

  // staging roles: DMA instruction i of wave wv fills rows (i*4 + wv)*RPI .. +RPI-1; lane -> (row, slot)
  const int srow = lane / SL, slot = lane % SL;
  // rows of consecutive instructions are NW * RPI apart and share the swizzle key (it depends on the row modulo RPI
  // only), so one VGPR offset per operand serves all of them: instruction i adds i * NW * RPI rows to the SGPR offset
  const int row0 = wv * RPI + srow;
  const int sw0 = (slot ^ key(row0)) << 3;           // swizzled 16-byte slot, in halfs
  const unsigned voffA0 = (unsigned)(((p0 + row0) * a.xa_stride + sw0) * 2);
  const unsigned voffB0 = (unsigned)(((p0 + row0) * a.xb_stride + sw0) * 2);
  const unsigned woff0 = (unsigned)(((size_t)row0 * C + sw0) * 2);
  int vmask[XI];                                     // rows past the end have no valid tap: their loads are dropped
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const long p = p0 + (i * NW + wv) * RPI + srow;
    int m = 0;
    if (p < a.P) {
      const int pi_ = (int)p, xw = pi_ % a.W, yh = (pi_ / a.W) % a.H;      // p < P < 2^31 (the 31-bit offset check of the launch)
      if (a.taps == 9) {
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          const int dy = d / 3 - 1, dx = d % 3 - 1;
          if ((unsigned)(yh + dy) < (unsigned)a.H && (unsigned)(xw + dx) < (unsigned)a.W) m |= 1 << d;
        }
      } else {
        m = 1;
      }
    }
    vmask[i] = m;
  }

  auto stage = [&](int t, int buf) {
    // K order: 64-channel chunk outermost, the taps inside it -- the 9 shifted reads of a chunk of
    // the workgroup's pixel rows follow each other, so all but the first hit L2 (tap-major order
    // has a reuse distance of the whole [pixels x C] panel of every resident workgroup: > L2)
    const int per_chunk = a.taps * cpc;
    const int ch = t / per_chunk, rem = t - ch * per_chunk;
    const int d = rem / cpc, sub = rem - d * cpc;        // tap, BK-wide part of the 64-channel chunk
    const int shift = (a.taps == 9 ? (d / 3 - 1) * a.W + (d % 3 - 1) : 0) + back;
    const bool segA = ch < a.cha;
    const int xs = segA ? a.xa_stride : a.xb_stride;
    const unsigned xsoff = (unsigned)((shift * xs + (segA ? ch : ch - a.cha) * 64 + sub * BK) * 2);
    const unsigned wsoff = (unsigned)((((size_t)d * a.npad + n0) * C + ch * 64 + sub * BK) * 2);
    char* lx = smem + buf * (XBYTES + WBYTES);
    char* lw = lx + XBYTES;
    const unsigned xstep = (unsigned)(NW * RPI * xs * 2), wstep = (unsigned)(NW * RPI * C * 2);
#ifdef EXP_NO_PIXEL_DMA
    if (d == 0)                                  // ablation: the pixel tile is staged once per chunk (wrong results)
#endif
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const unsigned inv = ~((unsigned)vmask[i] >> d);
      const unsigned vo = (inv << 31) | (segA ? voffA0 : voffB0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(segA ? rA : rB,
          (__attribute__((address_space(3))) void*)(lx + (i * NW + wv) * RPI * RB), 16, vo, xsoff + i * xstep, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW,
          (__attribute__((address_space(3))) void*)(lw + (i * NW + wv) * RPI * RB), 16, woff0, wsoff + i * wstep, 0, 0);
  };

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = 16*blk + col, logical slot kk*4 + kg, swizzle key(row) = key(col)
  int foff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) foff[kk] = col * RB + (((kk * 4 + kg) ^ key(col)) << 4);
  const int wbase = XBYTES + wm * (16 * MB) * RB, xbase_l = wn * (16 * NB) * RB;

#ifdef EXP_CONV_STAMPS
  // experiment (tools/conv_timeline.sh): shader-clock stamps of K-tiles 10-13 of every 61st workgroup, [16][NW][4][8]
  const int swg = lid / 61;
#define CV_STAMP(k) do { if (a.stamps && lane == 0 && lid % 61 == 0 && swg < 16 && t >= 10 && t < 14) { unsigned long long ts_;   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");                                              \
    a.stamps[((swg * NW + wv) * 4 + (t - 10)) * 8 + (k)] = ts_; } } while (0)
#else
#define CV_STAMP(k)
#endif
  stage(0, 0);
  int cur = 0;                                 // LDS stage holding tile t
  for (int t = 0; t < T; ++t) {
    CV_STAMP(0);
    // every DMA piece of tile t must have landed before anybody reads it.  The compiler's own wait in front of the barrier
    // counts register-spill traffic into vmcnt: a halo-staged experiment of the 256-channel variant (9 spilled VGPRs) got
    // `s_waitcnt vmcnt(5)` here, left its last weight pieces in flight and produced non-repeatable upper channels.  None
    // of the shipped variants spills, but the wait is explicit now.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_STAMP(1);
    __syncthreads();                           // tile t landed (vmcnt(0) + barrier); ST = 2: the other stage is free
    CV_STAMP(2);
    const char* base = smem + cur * (XBYTES + WBYTES);
    // all fragment reads of the step go out first (one exposed LDS latency per step, not per kk), the
    // DMA of the next tile is issued in their shadow, then the MFMAs run back to back
    f16x8 wf[KK][MB], xf[KK][NB];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
        wf[kk][mi] = *reinterpret_cast<const f16x8*>(base + wbase + mi * 16 * RB + foff[kk]);
#pragma unroll
      for (int ni = 0; ni < NB; ++ni)
        xf[kk][ni] = *reinterpret_cast<const f16x8*>(base + xbase_l + ni * 16 * RB + foff[kk]);
    }
    if (ST == 1) {
      // single LDS stage, 3 workgroups per CU.  Every fragment of the step is in registers now, so
      // once all waves got theirs the buffer is free: tile t+1 streams into it under the MFMAs.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CV_STAMP(3);
      __builtin_amdgcn_s_barrier();
      CV_STAMP(4);
      if (t + 1 < T) stage(t + 1, 0);
      CV_STAMP(5);
    } else {
      if (t + 1 < T) stage(t + 1, cur ^ 1);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][mi], xf[kk][ni], acc[mi][ni], 0, 0, 0);
    CV_STAMP(6);
    if (ST == 2) cur ^= 1;
  }

  // ---- epilogue: lane owns channels n0 + wm*16*MB + mi*16 + kg*4 .. +3 of pixel p0 + wn*16*NB + ni*16 + col ----
  if constexpr (EPI == EPI_GLO) {
    if constexpr (MB == 4 && NB == 4 && NW == 4 && ST == 1)
      conv_epilogue_glo<MB, NB>(a, acc, p0, p0 + wn * (16 * NB) + col, wm, wn, kg, col, smem, pt);
  } else if constexpr (EPI == EPI_UPSAMPLE) {
    if constexpr (MB == 4 && NB == 4 && NW == 4 && ST == 1)
      conv_epilogue_upsample<MB, NB>(a, acc, p0, n0, wm, wn, kg, col, smem);
  } else if constexpr (EPI == EPI_HEADS) {
    if constexpr (MB == 4 && NB == 4 && NW == 4 && ST == 1) {
      if (nt < a.tap_groups)                                                    // workgroup-uniform
        conv_epilogue_heads<MB, NB>(a, acc, p0, n0, wm, wn, kg, col, lane, smem);
      else
        conv_epilogue_tile<EPI_BIAS_ACT, MB, NB>(a, acc, p0 + wn * (16 * NB) + col, n0 + wm * (16 * MB) + kg * 4,
                                                 a.tap_groups * 128);
    }
  } else {
    conv_epilogue_tile<EPI, MB, NB>(a, acc, p0 + wn * (16 * NB) + col, n0 + wm * (16 * MB) + kg * 4);
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// conv_igemm_kernel for 3x3 layers with the PIXEL tile shared by the nine taps of a 64-channel chunk.
//
// The nine taps of a chunk read the same pixel rows shifted by (dy W + dx): instead of staging 128 rows per tap, the
// workgroup stages the 128 + 2 (W + 1) rows its taps touch ONCE per chunk (a haloed tile: 290 rows at W = 80, 37 KB) and
// every tap reads its fragments from there at its own row offset - 181 KB of DMA per chunk instead of 288 KB for a
// 128-channel tile.  tools/exp_conv.sh measured the ceiling of that first (pixel tile staged at tap 0 only, wrong results):
// 128-channel layers -13...-27 %, the z|r gates -5 %.  The real kernel keeps a fraction of it, because the haloed tile arrives
// as one burst of 37 pieces at every chunk boundary with nothing to compute beside it (a second tile buffer would cost the
// third resident workgroup): q gate 170.7 -> 164.4 us, heads 160.7 -> 156.4 us, 448 -> 128 -3...5 %, the 128 -> 128 layer
// unchanged; +0.9 % on the BA-update step in an interleaved A/B (950-956 -> 960-962 it/s).  The 256-channel tile (448 -> 256
// 324 -> 363 us: 256 VGPRs with spills, the weights are two thirds of its staging anyway) and the 64-channel tile (43.7 ->
// 47.8 us) lose and stay on conv_igemm_kernel; the dispatch takes this kernel for the 128-channel tile only.
// Everything else is conv_igemm_kernel<EPI, 4, 64, 4, 1, MB>: same tile mapping, K order (chunk outermost, taps inside),
// single LDS stage for the weights (wait, barrier, fragment reads, barrier, DMA of the next tile, MFMAs), same MFMA sequence
// and epilogues - results are bit-identical (tests/test_gpu_update_op.py).  What differs:
//   * zero padding moves from the DMA (bit 31 of a lane's offset) to the fragment: a lane whose pixel has no neighbour for
//     this tap replaces what it read by zeros (32 selects per K-tile, placed behind the barrier that frees the stage);
//     rows of the halo outside [0, P) are zero-filled by the DMA's range check as before;
//   * a fragment row starts anywhere, so its swizzle key (row & 7) is computed per tap (28 vector instructions per K-tile);
//     ds_read_b128 of 16 consecutive rows is conflict-free for every start row with that key (brute-force checked against
//     the lane groups of MI355X_MICROARCH.md);
//   * the halo tile is re-staged after the barrier that follows the fragment reads of a chunk's last tap - the same point
//     where the weights of the next tile are staged.  (Re-staging it in three parts as its rows go dead - the first W rows
//     behind tap 2, the next W behind tap 5, the rest at the boundary - was measured: the step LOSES 2.6 %, 931-935 -> 907-910
//     it/s; K-tiles that carry 14 pieces instead of 4 cost more than the one burst saves.)
// LDS: weights 32 MB rows + 128 + 2 W + 2 pixel rows (the last DMA piece overlaps its predecessor so that the tile takes no
// more than that: 53.5 KB at W = 80, three workgroups per CU; with whole 8-row pieces it was two and 4 % slower than the
// plain kernel); up to W = 83 for the 128-channel tile.
// ------------------------------------------------------------------------------------------------------------------
template <int EPI, int MB>
__global__ __launch_bounds__(256, MB <= 4 ? 3 : 2) void conv_halo_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NB = 4, NW = 4, TN = 32 * MB, PT = 128, RB = 128, SL = 8, RPI = 8, KK = 2;
  constexpr int WI = TN / RPI / NW;                    // weight DMA instructions per wave per K-tile
  constexpr int WBYTES = TN * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [weights][halo pixel rows]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int wm = wv & 1, wn = wv >> 1;
  const int halo = a.W + 1;                            // rows in front of (and behind) the tile
  const int hrows = PT + 2 * halo;
  const int npieces = (hrows + RPI - 1) / RPI;         // DMA pieces of 8 rows
  char* const lx = smem + WBYTES;

  const int nwg = gridDim.x, ntn = (a.nout + TN - 1) / TN;
  const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  const int pt = lid / ntn, nt = lid - pt * ntn;
  const int tpm = (a.HW + PT - 1) / PT;
  const long p0 = EPI == EPI_GLO ? (long)(pt / tpm) * a.HW + (long)(pt % tpm) * PT : a.pbeg + (long)pt * PT;
  const int n0 = nt * TN;
  const int nchunks = a.cha + a.chb;
  const int C = nchunks * 64;
  const int T = 9 * nchunks;

  // descriptors `halo` rows in front of the tensors: halo row h of the tile is pixel p0 - halo + h
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xa - (long)halo * a.xa_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xb - (long)halo * a.xb_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
  const int srow = lane / SL, slot = lane % SL;
  const int row0 = wv * RPI + srow;
  const int sw0 = (slot ^ (row0 & 7)) << 3;            // swizzled 16-byte slot, in halfs (key = row & 7 = srow)
  const unsigned voffA0 = (unsigned)(((p0 + row0) * a.xa_stride + sw0) * 2);
  const unsigned voffB0 = (unsigned)(((p0 + row0) * a.xb_stride + sw0) * 2);
  const unsigned woff0 = (unsigned)(((size_t)row0 * C + sw0) * 2);
  constexpr int XIMAX = 12;                            // pieces per wave: ceil(ceil((128 + 2 W + 2) / 8) / 4), W <= 127
  unsigned hmask = 0;                                  // bit i: the row of piece i * 4 + wv this lane stages lies inside [0, P)
#pragma unroll
  for (int i = 0; i < XIMAX; ++i) {
    const long p = p0 - halo + (long)((i * NW + wv) * RPI + srow);
    hmask |= (p >= 0 && p < a.P) ? (1u << i) : 0u;
  }
  auto stage_pixels = [&](int ch) {
    const bool segA = ch < a.cha;
    const int xs = segA ? a.xa_stride : a.xb_stride;
    const unsigned xsoff = (unsigned)(((segA ? ch : ch - a.cha) * 64) * 2);
    const unsigned xstep = (unsigned)(NW * RPI * xs * 2);
    unsigned hm = hmask;
    asm volatile("" : "+v"(hm));                       // keeps the twelve masked offsets from being hoisted out of the K loop
#pragma unroll
    for (int i = 0; i < XIMAX; ++i) {
      const int piece = i * NW + wv;                   // wave-uniform
      if (piece < npieces - 1 || (piece == npieces - 1 && (hrows & 7) == 0)) {
        const unsigned inv = ~(hm >> i);
        const unsigned vo = (inv << 31) | (segA ? voffA0 : voffB0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(segA ? rA : rB,
            (__attribute__((address_space(3))) void*)(lx + piece * RPI * RB), 16, vo, xsoff + i * xstep, 0, 0);
      } else if (piece == npieces - 1) {
        // the last piece ends with the tile's last row (it re-writes up to 7 rows of its predecessor with the same bytes):
        // the tile then takes exactly 128 + 2 W + 2 rows of LDS, which is what lets three workgroups share a CU at W = 80
        const int rowL = hrows - RPI + srow;
        const long pL = p0 - halo + rowL;
        const unsigned invL = (pL >= 0 && pL < a.P) ? 0u : 0x80000000u;
        const unsigned voL = invL | (unsigned)(((p0 + rowL) * xs + ((slot ^ (rowL & 7)) << 3)) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(segA ? rA : rB,
            (__attribute__((address_space(3))) void*)(lx + (hrows - RPI) * RB), 16, voL, xsoff, 0, 0);
      }
    }
  };
  auto stage_weights = [&](int ch, int d) {
    const unsigned wsoff = (unsigned)((((size_t)d * a.npad + n0) * C + ch * 64) * 2);
    const unsigned wstep = (unsigned)(NW * RPI * C * 2);
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW,
          (__attribute__((address_space(3))) void*)(smem + (i * NW + wv) * RPI * RB), 16, woff0, wsoff + i * wstep, 0, 0);
  };

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // tap validity of this lane's four pixels (blocks ni): bit 9 ni + d
  unsigned long long pmask = 0ull;
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    const long p = p0 + wn * (16 * NB) + ni * 16 + col;
    if (p < a.P) {
      const int pi_ = (int)p, xw = pi_ % a.W, yh = (pi_ / a.W) % a.H;      // p < P < 2^31 (the 31-bit offset check of the launch)
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        const int dy = d / 3 - 1, dx = d % 3 - 1;
        if ((unsigned)(yh + dy) < (unsigned)a.H && (unsigned)(xw + dx) < (unsigned)a.W) pmask |= 1ull << (9 * ni + d);
      }
    }
  }
  int woffr[KK];                                       // weight fragments: row = 16 blk + col, slot kk * 4 + kg, key = col & 7
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) woffr[kk] = col * RB + (((kk * 4 + kg) ^ (col & 7)) << 4);
  const int wbase = wm * (16 * MB) * RB;
  const int rb0 = wn * (16 * NB) + col + halo;         // halo row of this lane's pixel of block 0 for the centre tap

  stage_pixels(0);
  stage_weights(0, 0);
  int ch = 0, d = 0;                                   // chunk and tap of tile t
  for (int t = 0; t < T; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // tile t (and at d == 0 the chunk's pixel rows) landed
    const int shift = (d / 3 - 1) * a.W + (d % 3 - 1);
    f16x8 wf[KK][MB], xf[KK][NB];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
        wf[kk][mi] = *reinterpret_cast<const f16x8*>(smem + wbase + mi * 16 * RB + woffr[kk]);
    {
      // the four pixel blocks of a lane are 16 rows apart: same swizzle key, one address + immediate offsets
      const int rr = rb0 + shift;
      const unsigned o0 = (unsigned)WBYTES + (unsigned)rr * RB + (unsigned)((kg ^ (rr & 7)) << 4);
      const char* b0 = smem + o0;
      const char* b1 = smem + (o0 ^ 64u);
#pragma unroll
      for (int ni = 0; ni < NB; ++ni) {
        xf[0][ni] = *reinterpret_cast<const f16x8*>(b0 + ni * 16 * RB);
        xf[1][ni] = *reinterpret_cast<const f16x8*>(b1 + ni * 16 * RB);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave holds its fragments
    {
      // zero padding: a pixel without a neighbour for this tap contributes a zero fragment
      const unsigned pm = (unsigned)(pmask >> d);      // bit 9 ni
      const f16x8 zf = {};
#pragma unroll
      for (int ni = 0; ni < NB; ++ni) {
        const bool ok = (pm >> (9 * ni)) & 1u;
        xf[0][ni] = ok ? xf[0][ni] : zf;
        xf[1][ni] = ok ? xf[1][ni] : zf;
      }
    }
    int ch1 = ch, d1 = d + 1;
    if (d1 == 9) { d1 = 0; ch1 = ch + 1; }
    if (t + 1 < T) {
      if (d1 == 0) stage_pixels(ch1);                  // the chunk's last tap has been read: its pixel rows are free
      stage_weights(ch1, d1);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][mi], xf[kk][ni], acc[mi][ni], 0, 0, 0);
    ch = ch1;
    d = d1;
  }

  if constexpr (EPI == EPI_GLO) {
    if constexpr (MB == 4) conv_epilogue_glo<MB, NB>(a, acc, p0, p0 + wn * (16 * NB) + col, wm, wn, kg, col, smem, pt);
  } else if constexpr (EPI == EPI_UPSAMPLE) {
    if constexpr (MB == 4) conv_epilogue_upsample<MB, NB>(a, acc, p0, n0, wm, wn, kg, col, smem);
  } else if constexpr (EPI == EPI_HEADS) {
    if constexpr (MB == 4) {
      if (nt < a.tap_groups)                                                    // workgroup-uniform
        conv_epilogue_heads<MB, NB>(a, acc, p0, n0, wm, wn, kg, col, lane, smem);
      else
        conv_epilogue_tile<EPI_BIAS_ACT, MB, NB>(a, acc, p0 + wn * (16 * NB) + col, n0 + wm * (16 * MB) + kg * 4,
                                                 a.tap_groups * 128);
    }
  } else {
    conv_epilogue_tile<EPI, MB, NB>(a, acc, p0 + wn * (16 * NB) + col, n0 + wm * (16 * MB) + kg * 4);
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// conv_pp_kernel: 256 channels x 256 pixels per workgroup, EIGHT waves in two groups that run HALF A K-TILE APART
// ("ping-pong", round 5).  conv_igemm_kernel leaves the overlap of one workgroup's fragment reads / DMA issue with the
// other resident workgroup's MFMAs to chance (tools/conv_timeline.sh: a K-tile of the z|r launch takes 3852 cycles per
// workgroup, two resident, for 2 x 1024 cycles of MFMA per SIMD - the DMA issue alone 980 cycles, because all four waves
// push their 12 pieces into the CU's one address unit at the same moment).  Here the overlap is built in:
//   * wave w and wave w + 4 share a SIMD and belong to different groups; group 1 executes ONE extra barrier before its
//     loop, so between any two workgroup barriers one group is in its MEM phase (24 ds_read_b128 of tile t, lgkmcnt(0))
//     and the other in its MFMA phase (64 MFMAs at raised priority) - the matrix pipe of every SIMD always has a wave
//     whose operands are already in registers;
//   * per-wave tile 128 channels x 64 pixels (MB = 8, NB = 4: the fragment reads per MFMA of the 256 x 128 tile) but
//     half the DMA bytes per MFMA (64 KB per K-tile for 256 x 256 outputs); two LDS stages (128 KB), one workgroup per CU;
//   * the CU's one address unit takes ~17 cycles per 1 KB piece: the 64 pieces of a K-tile are half of its MFMA time and must
//     not bunch up in one phase.  LDS = two pixel stages + THREE weight stages (160 KB): a pixel stage frees at an even
//     barrier - group 0 refills it first thing in its MEM phase (8 pieces per wave); a weight stage has a period more of
//     slack - group 1 refills it behind the fragment reads of ITS MEM phase (8 pieces per wave, waited for with vmcnt(8) one
//     K-tile later): 32 pieces per period, none beside MFMAs.  (Measured on the way, tools/conv_pp_timeline.sh: half of the
//     pieces between group 1's MFMAs - every piece stalls that wave's MFMA issue ~90 cycles, 1930 instead of 1204 cycles for
//     the phase, 3760 per K-tile; all 64 pieces by group 0 with two full stages - its MEM phase 1792 cycles, 3392 per K-tile.)
//   * the DMA pieces are inline asm (m0 written in the same statement): hipcc drains every LDS-DMA it KNOWS of with
//     vmcnt(0) before the next ds_read that might alias it, which would pin group 1's pieces to one phase of flight.
// K order, MFMA sequence per accumulator, zero padding (bit 31 of the lane's offset) and the epilogues are those of
// conv_igemm_kernel<EPI, 4, 64, 4, 1, 8>: the outputs are bit-identical (tests/test_gpu_update_op.py).
// The 128-CHANNEL layers (q gate, heads) were given the same schedule over a haloed pixel tile in two LDS buffers + three
// weight stages (128 x 256 tile, wave tile 64 x 64, 157 KB; bit-identical on four shapes) and LOST to the haloed 128 x 128
// tile at three workgroups per CU: q gate 151 vs 144 us, 128 -> 384 182 vs 157 us.  A phase then holds 32 MFMAs (~750-860
// cycles to issue) against the same two barriers, the 16 fragment reads + zero-padding selects + pieces of the MEM phase take as
// long (810-850), and a K-tile costs 2440 cycles for half the MACs of the 3144-cycle tile above
// (profiles/r05_conv_pph_timeline.txt); 64-MFMA phases need 512-pixel tiles or two taps per K-tile, neither fits 160 KB of LDS
// with a double-buffered halo.  Removed.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16_asm(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
  // every SGPR operand is SALU-produced at the call sites (checked in the ISA: no v_readfirstlane feeds them, which would need
  // five wait states in front of the VMEM instruction); s_nop 0: the M0 write -> LDS-DMA hazard
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :: "v"(voff), "s"(r), "s"(soff), "s"(lds) : "memory");
}
// a pixel piece with its zero-padding bit formed INSIDE the statement: offset = voff | (~(mask >> shift) << 31).  As separate
// C++ expressions hipcc forms the offsets of all of a tile's pieces ahead of the first DMA (eight more live VGPRs in a loop
// that sits at the 256-register limit; same speed in an interleaved A/B of two builds, 228-230 us for the z|r gate launch)
__device__ __forceinline__ void dma16_masked_asm(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned mask, unsigned shift,
                                                 unsigned soff, unsigned lds) {
  unsigned tmp;
  asm volatile("v_lshrrev_b32 %0, %5, %6\n\tv_not_b32 %0, %0\n\tv_lshl_or_b32 %0, %0, 31, %1\n\t"
               "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
               : "=&v"(tmp) : "v"(voff), "s"(r), "s"(soff), "s"(lds), "s"(shift), "v"(mask) : "memory");
}
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory");        \
                          __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_WAIT_LGKM_BARRIER() do { __builtin_amdgcn_sched_barrier(0);                                        \
                                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           \
                                    __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_WAIT_VM_BARRIER() do { __builtin_amdgcn_sched_barrier(0);                                          \
                                  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");               \
                                  __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_WAIT_ALL_BARRIER() do { __builtin_amdgcn_sched_barrier(0);                                         \
                                   asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
                                   __builtin_amdgcn_sched_barrier(0); } while (0)

// (Round 6: the same wave tile on v_mfma_f32_32x32x16_f16 - 32 MFMAs per K-tile, MFMA phase 5-7 % shorter - was built and
// removed: the 32 x 32 result layout doubles the epilogue's address work and the launch is 7-12 % slower,
// profiles/r06_conv_m32.txt.)
template <int EPI, int NB>
__device__ __forceinline__ void conv_pp_tile(const ConvArgs& a, const long p0, const int n0, const int lid, char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MB = 8, NWS = 4, TN = 256, PT = 64 * NB, RB = 128, SL = 8, RPI = 8, KK = 2;
  constexpr int XI = PT / RPI / NWS, WI = TN / RPI / NWS;      // 8 pixel pieces per wave of group 0, 8 weight pieces per wave of group 1
  constexpr int XBYTES = PT * RB, WBYTES = TN * RB;
  constexpr int WBASE = 2 * 256 * RB;                           // LDS: pixel tiles 0 1 | weight tiles 0 1 2  (160 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wv & 3;                                        // staging slot inside the group
  const int col = lane & 15, kg = lane >> 4;
  const int wm = wv & 1, wn = wv >> 1;
  const int grp = wv >> 2;                                      // waves w and w + 4 share a SIMD

  const int nchunks = a.cha + a.chb;
  const int C = nchunks * 64;
  const int T = a.taps * nchunks;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  int foff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) foff[kk] = col * RB + (((kk * 4 + kg) ^ (col & 7)) << 4);
  const int wfrag = WBASE + wm * (16 * MB) * RB, xfrag = wn * (16 * NB) * RB;
  f16x8 wf[KK][MB], xf[KK][NB];

  // MEM phase = the wave's 24 fragment reads and its 8 DMA pieces.  Inside one wave the two do not mix: a piece between
  // reads stalls the in-order issue on the address unit's queue and the reads behind it (own work 1568 / 1722 cycles per MEM
  // phase against 1404 with the pieces first and 1164 with the reads first, tools/conv_pp_timeline.sh).  Across waves they
  // do: staging slots 0 / 1 read first, slots 2 / 3 issue their pieces first, so the LDS pipe and the address unit both have
  // work during the whole phase.
  auto read_frags = [&](int xb, int wb) {
    const char* bx = smem + xb * (256 * RB) + xfrag;
    const char* bw = smem + wb * WBYTES + wfrag;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) wf[kk][mi] = *reinterpret_cast<const f16x8*>(bw + mi * 16 * RB + foff[kk]);
#pragma unroll
      for (int ni = 0; ni < NB; ++ni) xf[kk][ni] = *reinterpret_cast<const f16x8*>(bx + ni * 16 * RB + foff[kk]);
    }
  };
  const bool pieces_first = w4 >= 2;
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);      // (no measurable effect here: 225.5 us with, 227.4 without, 225.8 with the MEM phase raised instead)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][mi], xf[kk][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

#ifdef EXP_CONV_STAMPS
  // experiment (tools/conv_pp_timeline.py): shader-clock stamps of K-tiles 10-13 of every 41st workgroup, [16][8][4][4]
  const int swg = lid / 41;
#define PP_STAMP(k) do { if (a.stamps && lane == 0 && lid >= 0 && lid % 41 == 0 && swg < 16 && t >= 10 && t < 14) { unsigned long long ts_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");                                              \
    a.stamps[((swg * NW + wv) * 4 + (t - 10)) * 4 + (k)] = ts_; } } while (0)
#else
#define PP_STAMP(k)
#endif
  const int srow = lane / SL, slot = lane % SL;
  const int row0 = w4 * RPI + srow;                             // piece i of staging slot w4: rows (i * NWS + w4) * RPI .. + RPI - 1
  const int sw0 = (slot ^ (row0 & 7)) << 3;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (unsigned)(w4 * RPI * RB);
  // K-tile t: 64-channel chunk t / taps (outermost), tap t % taps - the K order of conv_igemm_kernel

  // barrier k = the k-th workgroup barrier; group 0: MEM(t) between barriers 2t and 2t + 1, MFMA(t) between 2t + 1 and
  // 2t + 2; group 1 one barrier later (MEM(t) in the odd period 2t + 1).
  //   pixel tile t + 1 -> stage (t + 1) % 2, free behind barrier 2t (group 1 read tile t - 1 in period 2t - 1), due at barrier
  //     2t + 2: group 0, first thing in its MEM(t) phase, vmcnt(0) at the end of its MFMA(t) phase;
  //   weight tile t + 2 -> stage (t + 2) % 3, free behind barrier 2t (ditto), due at barrier 2t + 4: group 1 in its MEM(t)
  //     phase (period 2t + 1) behind its fragment reads; it waits for tile t + 1 there with vmcnt(8) - three periods of flight.
  if (grp == 0) {
    const int back = a.taps == 9 ? a.W + 1 : 0;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.xa - (long)back * a.xa_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.xb - (long)back * a.xb_stride), 0, 0x7fffffff, 0x00020000);
    const unsigned voffA0 = (unsigned)(((p0 + row0) * a.xa_stride + sw0) * 2);
    const unsigned voffB0 = (unsigned)(((p0 + row0) * a.xb_stride + sw0) * 2);
    unsigned vmask[(XI + 2) / 3];                        // 9 tap bits per piece, three pieces per register
#pragma unroll
    for (int i = 0; i < (XI + 2) / 3; ++i) vmask[i] = 0;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const long p = p0 + (i * NWS + w4) * RPI + srow;
      unsigned m = 0;
      if (p < a.P) {
        const int pi_ = (int)p, xw = pi_ % a.W, yh = (pi_ / a.W) % a.H;
        if (a.taps == 9) {
#pragma unroll
          for (int d = 0; d < 9; ++d) {
            const int dy = d / 3 - 1, dx = d % 3 - 1;
            if ((unsigned)(yh + dy) < (unsigned)a.H && (unsigned)(xw + dx) < (unsigned)a.W) m |= 1u << d;
          }
        } else {
          m = 1;
        }
      }
      vmask[i / 3] |= m << (9 * (i % 3));
    }
    struct PixTile { unsigned xsoff, xstep, l0, d; bool segA; };
    auto pix_tile = [&](int t, int buf) {
      PixTile pt_;
      const int ch = t / a.taps, d = t - ch * a.taps;
      const int shift = (a.taps == 9 ? (d / 3 - 1) * a.W + (d % 3 - 1) : 0) + back;
      pt_.segA = ch < a.cha;
      const int xs = pt_.segA ? a.xa_stride : a.xb_stride;
      pt_.xsoff = (unsigned)((shift * xs + (pt_.segA ? ch : ch - a.cha) * 64) * 2);
      pt_.xstep = (unsigned)(NWS * RPI * xs * 2);
      pt_.l0 = lds0 + buf * (256 * RB);
      pt_.d = (unsigned)d;
      return pt_;
    };
    auto pix_piece = [&](const PixTile& pt_, int i) {
      dma16_masked_asm(pt_.segA ? rA : rB, pt_.segA ? voffA0 : voffB0, vmask[i / 3], pt_.d + 9 * (i % 3),
                       pt_.xsoff + i * pt_.xstep, pt_.l0 + i * (NWS * RPI * RB));
    };
    {
      const PixTile p0_ = pix_tile(0, 0);
#pragma unroll
      for (int i = 0; i < XI; ++i) pix_piece(p0_, i);
    }
    for (int t = 0; t < T; ++t) {
      PP_WAIT_VM_BARRIER();                               // barrier 2t: pixel tile t landed (own pieces), weight tile t: group 1
      PP_STAMP(0);
#ifdef EXP_PP_NO_PIECES
      const bool more = false;                            // ablation (wrong results): no DMA in the K loop
#else
      const bool more = t + 1 < T;
#endif
      const PixTile nx = pix_tile(more ? t + 1 : 0, (t + 1) & 1);
      if (pieces_first) {
        if (more) {
#pragma unroll
          for (int i = 0; i < XI; ++i) pix_piece(nx, i);
        }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(t & 1, t % 3);
      } else {
        read_frags(t & 1, t % 3);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
          for (int i = 0; i < XI; ++i) pix_piece(nx, i);
        }
      }
      PP_STAMP(3);
      PP_WAIT_LGKM_BARRIER();                             // barrier 2t + 1
      PP_STAMP(1);
      mfmas();
      PP_STAMP(2);
    }
    PP_BARRIER();                                         // barrier 2T (group 1's last one)
  } else {
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
    const unsigned woff0 = (unsigned)(((size_t)row0 * C + sw0) * 2);
    const unsigned wstep = (unsigned)(NWS * RPI * C * 2);
    auto w_piece = [&](int t, int buf, int i) {
      const int ch = t / a.taps, d = t - ch * a.taps;
      const unsigned wsoff = (unsigned)((((size_t)d * a.npad + n0) * C + ch * 64) * 2);
      dma16_asm(rW, woff0, wsoff + i * wstep, lds0 + WBASE + buf * WBYTES + i * (NWS * RPI * RB));
    };
#pragma unroll
    for (int i = 0; i < WI; ++i) w_piece(0, 0, i);
    if (1 < T) {
#pragma unroll
      for (int i = 0; i < WI; ++i) w_piece(1, 1, i);
    }
    PP_WAIT_VM_BARRIER();                                 // barrier 0: weight tiles 0 and 1 landed
    int wb = 0;                                           // t % 3
    for (int t = 0; t < T; ++t) {
      PP_BARRIER();                                       // barrier 2t + 1
      PP_STAMP(0);
#ifdef EXP_PP_NO_PIECES
      const bool more = false;
#else
      const bool more = t + 2 < T;
#endif
      const int wnext = wb == 0 ? 2 : wb - 1;             // (t + 2) % 3
      if (pieces_first) {
        if (more) {
#pragma unroll
          for (int i = 0; i < WI; ++i) w_piece(t + 2, wnext, i);
        }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(t & 1, wb);
      } else {
        read_frags(t & 1, wb);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
          for (int i = 0; i < WI; ++i) w_piece(t + 2, wnext, i);
        }
      }
      PP_STAMP(3);
      __builtin_amdgcn_sched_barrier(0);
      if (more) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier 2t + 2: weight tile t + 1 landed
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(1);
      mfmas();
      PP_STAMP(2);
      wb = wb == 2 ? 0 : wb + 1;
    }
  }
  conv_epilogue_tile<EPI, MB, NB>(a, acc, p0 + wn * (16 * NB) + col, n0 + wm * (16 * MB) + kg * 4);
#endif
}

// Tiles: `pp_full` workgroups with 256-pixel tiles (a whole number of rounds over the chip's CUs), then the rest of the map in
// tiles of 64 * pp_nbr pixels - one partial round of SMALLER tiles instead of a mostly empty round of full ones (G8: 675
// tiles on 256 CUs = 3 rounds of which the last is 64 % full; 512 full tiles + 218 tiles of 192 pixels = 2.75 rounds).
// Both ranges are spread over the XCDs by the same bijective remap (pp_full is a multiple of 8).
template <int EPI>
__global__ __launch_bounds__(512, 1) void conv_pp_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntn = a.nout / 256;
  const int bid = blockIdx.x, nfull = a.pp_full;
  const bool full = bid < nfull;
  const int base = full ? 0 : nfull, nwg = full ? nfull : (int)gridDim.x - nfull, j = bid - base;
  const int xcd = j & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (j >> 3);
  const int pt = lid / ntn, nt = lid - pt * ntn;
  const int n0 = nt * 256;
  if (full) {
    conv_pp_tile<EPI, 4>(a, (long)pt * 256, n0, lid, smem);
  } else {
    const long p0 = (long)(nfull / ntn) * 256 + (long)pt * (64 * a.pp_nbr);
    if (a.pp_nbr == 3) conv_pp_tile<EPI, 3>(a, p0, n0, -1, smem);
    else if (a.pp_nbr == 2) conv_pp_tile<EPI, 2>(a, p0, n0, -1, smem);
    else conv_pp_tile<EPI, 1>(a, p0, n0, -1, smem);
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// conv_ppw_kernel (round 6): the ping-pong schedule of conv_pp_kernel for the 128-CHANNEL layers (q gate 320 -> 128, the heads'
// 128 -> 384 as three channel tiles, corr_encoder[1] / flow layers 128 -> 128): 128 channels x 512 pixels per workgroup.
// The round-5 attempt (a haloed 128 x 256 tile, wave tile 64 x 64) lost because a phase held only 32 MFMAs against the same
// two barriers.  Here the WAVE tile stays 128 channels x 64 pixels (MB = 8, NB = 4: 64 MFMAs per phase, 24 fragment reads),
// the eight waves sit side by side along the pixels, and - because group g (waves 4g .. 4g + 3) reads ONLY its own 256
// pixel rows - the staging needs no cross-group hand-over for the pixels at all:
//   LDS (160 KB exactly): pixel buffers [group][2] x 32 KB, two weight stages x 16 KB;
//   group g, in its MEM phase of K-tile t: DMA of ITS pixel half of tile t + 1 into its other buffer (read last in its own
//     MEM phase of tile t - 1; 8 pieces per wave), waited for with vmcnt(0) at the end of its MFMA phase - two periods of flight;
//   group 0 additionally stages the weight tile t + 1 (4 pieces per wave) into stage (t + 1) & 1, whose last reader (group 1,
//     tile t - 1, period 2t - 1) retired its reads in front of barrier 2t; group 1 reads it three barriers later.
// K order, MFMA sequence per accumulator, zero padding and epilogues are those of conv_igemm_kernel<EPI, 4, 64, 4, 1, 4>:
// bit-identical outputs (tests/test_gpu_update_op.py::test_conv_wide_pingpong_tiles_are_bit_identical).  The heads' tap GEMM
// needs no LDS here: a wave holds all 128 hidden channels of its 64 pixels (the two 64-channel halves are summed in registers
// in the order the 128 x 128 tile sums them through LDS).
// ------------------------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void conv_epilogue_heads_w(const ConvArgs& a, f32x4 (&acc)[8][NB], long pwave, int n0, int kg,
                                                      int col, int lane) {
  const int grp = n0 >> 7;
  const f16x8* wp = a.tap_w + (size_t)grp * 2 * 4 * 64 + lane;                  // [group][half][chunk*2 + row block][64]
  f16x8 wf[2][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) wf[h][c][rb] = wp[((h * 2 + c) * 2 + rb) * 64];
  float4 b[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) b[mi] = *reinterpret_cast<const float4*>(a.terms + n0 + mi * 16 + kg * 4);
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    f16x8 hf[4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const f32x4 v = acc[mi][ni];
      const int h = (mi & 1) * 4;
      hf[mi >> 1][h + 0] = (_Float16)fmaxf(v[0] + b[mi].x, 0.0f);
      hf[mi >> 1][h + 1] = (_Float16)fmaxf(v[1] + b[mi].y, 0.0f);
      hf[mi >> 1][h + 2] = (_Float16)fmaxf(v[2] + b[mi].z, 0.0f);
      hf[mi >> 1][h + 3] = (_Float16)fmaxf(v[3] + b[mi].w, 0.0f);
    }
    const long p = pwave + ni * 16 + col;
    float* dst = a.tap_out + (size_t)(grp * a.tap_ncols) * a.P + (p < a.P ? p : a.P - 1);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][c][rb], hf[c], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][c][rb], hf[2 + c], s1, 0, 0, 0);
      }
      const int n = rb * 16 + kg * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (p < a.P && n + r < a.tap_ncols) dst[(size_t)(n + r) * a.P] = s0[r] + s1[r];
    }
  }
}

template <int EPI, int NB>
__device__ __forceinline__ void conv_ppw_tile(const ConvArgs& a, const long p0, const int n0, const int nt, char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MB = 8, NWS = 4, RB = 128, SL = 8, RPI = 8, KK = 2;
  constexpr int PG = 64 * NB;                                   // pixel rows of one group's half of the tile
  constexpr int XI = PG / RPI / NWS, WI = 128 / RPI / NWS;      // 2 NB pixel pieces per wave; 4 weight pieces per wave of group 0
  constexpr int XBUF = 256 * RB;                                // one pixel buffer, [group][buffer]
  constexpr int WBASE = 4 * XBUF, WBYTES = 128 * RB;            // weight stages 0 1 behind the four pixel buffers (160 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wv & 3;                                        // staging slot inside the group = pixel quarter of its half
  const int col = lane & 15, kg = lane >> 4;
  const int grp = wv >> 2;                                      // waves w and w + 4 share a SIMD

  const int nchunks = a.cha + a.chb;
  const int C = nchunks * 64;
  const int T = a.taps * nchunks;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  int foff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) foff[kk] = col * RB + (((kk * 4 + kg) ^ (col & 7)) << 4);
  const int xfrag = grp * 2 * XBUF + w4 * (16 * NB) * RB;
  f16x8 wf[KK][MB], xf[KK][NB];

  auto read_frags = [&](int buf) {                              // pixel buffer and weight stage of tile t: both t & 1
    const char* bx = smem + xfrag + buf * XBUF;
    const char* bw = smem + WBASE + buf * WBYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) wf[kk][mi] = *reinterpret_cast<const f16x8*>(bw + mi * 16 * RB + foff[kk]);
#pragma unroll
      for (int ni = 0; ni < NB; ++ni) xf[kk][ni] = *reinterpret_cast<const f16x8*>(bx + ni * 16 * RB + foff[kk]);
    }
  };
  const bool pieces_first = w4 >= 2;                            // (conv_pp_tile: the LDS pipe and the address unit both stay busy)
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][mi], xf[kk][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int srow = lane / SL, slot = lane % SL;
  const int row0 = w4 * RPI + srow;                             // piece i of staging slot w4: rows (i * NWS + w4) * RPI .. + RPI - 1
  const int sw0 = (slot ^ (row0 & 7)) << 3;
  const unsigned ldsb = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned lds0 = ldsb + (unsigned)(grp * 2 * XBUF + w4 * RPI * RB);
  const long ph = p0 + (long)grp * PG;                          // first pixel of this group's half

  // ---- pixel staging of this wave's group (both groups) ----
  const int back = a.taps == 9 ? a.W + 1 : 0;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xa - (long)back * a.xa_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xb - (long)back * a.xb_stride), 0, 0x7fffffff, 0x00020000);
  const unsigned voffA0 = (unsigned)(((ph + row0) * a.xa_stride + sw0) * 2);
  const unsigned voffB0 = (unsigned)(((ph + row0) * a.xb_stride + sw0) * 2);
  unsigned vmask[(XI + 2) / 3];                                 // 9 tap bits per piece, three pieces per register
#pragma unroll
  for (int i = 0; i < (XI + 2) / 3; ++i) vmask[i] = 0;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const long p = ph + (i * NWS + w4) * RPI + srow;
    unsigned m = 0;
    if (p < a.P) {
      const int pi_ = (int)p, xw = pi_ % a.W, yh = (pi_ / a.W) % a.H;
      if (a.taps == 9) {
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          const int dy = d / 3 - 1, dx = d % 3 - 1;
          if ((unsigned)(yh + dy) < (unsigned)a.H && (unsigned)(xw + dx) < (unsigned)a.W) m |= 1u << d;
        }
      } else {
        m = 1;
      }
    }
    vmask[i / 3] |= m << (9 * (i % 3));
  }
  struct PixTile { unsigned xsoff, xstep, l0, d; bool segA; };
  auto pix_tile = [&](int t, int buf) {
    PixTile pt_;
    const int ch = t / a.taps, d = t - ch * a.taps;
    const int shift = (a.taps == 9 ? (d / 3 - 1) * a.W + (d % 3 - 1) : 0) + back;
    pt_.segA = ch < a.cha;
    const int xs = pt_.segA ? a.xa_stride : a.xb_stride;
    pt_.xsoff = (unsigned)((shift * xs + (pt_.segA ? ch : ch - a.cha) * 64) * 2);
    pt_.xstep = (unsigned)(NWS * RPI * xs * 2);
    pt_.l0 = lds0 + buf * XBUF;
    pt_.d = (unsigned)d;
    return pt_;
  };
  auto pix_piece = [&](const PixTile& pt_, int i) {
    dma16_masked_asm(pt_.segA ? rA : rB, pt_.segA ? voffA0 : voffB0, vmask[i / 3], pt_.d + 9 * (i % 3),
                     pt_.xsoff + i * pt_.xstep, pt_.l0 + i * (NWS * RPI * RB));
  };
  // ---- weight staging (group 0) ----
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
  const unsigned woff0 = (unsigned)(((size_t)row0 * C + sw0) * 2);
  const unsigned wstep = (unsigned)(NWS * RPI * C * 2);
  const unsigned ldsw0 = ldsb + (unsigned)(WBASE + w4 * RPI * RB);
  auto w_piece = [&](int t, int buf, int i) {
    const int ch = t / a.taps, d = t - ch * a.taps;
    const unsigned wsoff = (unsigned)((((size_t)d * a.npad + n0) * C + ch * 64) * 2);
    dma16_asm(rW, woff0, wsoff + i * wstep, ldsw0 + buf * WBYTES + i * (NWS * RPI * RB));
  };

  // barrier k = the k-th workgroup barrier; group 0: MEM(t) between barriers 2t and 2t + 1, MFMA(t) between 2t + 1 and 2t + 2;
  // group 1 one barrier later.  Every wave waits vmcnt(0) in front of the barrier that opens its MEM phase (its own pieces of
  // the tile it is about to read - and, for group 0, the weight tile both groups are about to read - have landed).
  {
    const PixTile p0_ = pix_tile(0, 0);
#pragma unroll
    for (int i = 0; i < XI; ++i) pix_piece(p0_, i);
  }
  if (grp == 0) {
#pragma unroll
    for (int i = 0; i < WI; ++i) w_piece(0, 0, i);
  } else {
    PP_WAIT_VM_BARRIER();                                 // barrier 0
  }
  for (int t = 0; t < T; ++t) {
    PP_WAIT_VM_BARRIER();                                 // group 0: barrier 2t; group 1: barrier 2t + 1
    const bool more = t + 1 < T;
    const PixTile nx = pix_tile(more ? t + 1 : 0, (t + 1) & 1);
    if (pieces_first) {
      if (more) {
#pragma unroll
        for (int i = 0; i < XI; ++i) pix_piece(nx, i);
        if (grp == 0) {
#pragma unroll
          for (int i = 0; i < WI; ++i) w_piece(t + 1, (t + 1) & 1, i);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags(t & 1);
    } else {
      read_frags(t & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int i = 0; i < XI; ++i) pix_piece(nx, i);
        if (grp == 0) {
#pragma unroll
          for (int i = 0; i < WI; ++i) w_piece(t + 1, (t + 1) & 1, i);
        }
      }
    }
    PP_WAIT_LGKM_BARRIER();                               // group 0: barrier 2t + 1; group 1: barrier 2t + 2
    mfmas();
  }
  if (grp == 0) PP_BARRIER();                             // barrier 2T (group 1's last one)

  const long pwave = p0 + (long)wv * (16 * NB);
  if constexpr (EPI == EPI_HEADS) {
    if (nt < a.tap_groups)                                                      // workgroup-uniform
      conv_epilogue_heads_w<NB>(a, acc, pwave, n0, kg, col, lane);
    else
      conv_epilogue_tile<EPI_BIAS_ACT, MB, NB>(a, acc, pwave + col, n0 + kg * 4, a.tap_groups * 128);
  } else {
    conv_epilogue_tile<EPI, MB, NB>(a, acc, pwave + col, n0 + kg * 4);
  }
#endif
}

// Tiles as in conv_pp_kernel: `pp_full` workgroups with full (512-pixel) tiles, then the rest of the map in tiles of
// 128 * pp_nbr pixels; both ranges spread over the XCDs by the same bijective remap.
template <int EPI>
__global__ __launch_bounds__(512, 1) void conv_ppw_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntn = a.nout / 128;
  const int bid = blockIdx.x, nfull = a.pp_full;
  const bool full = bid < nfull;
  const int base = full ? 0 : nfull, nwg = full ? nfull : (int)gridDim.x - nfull, j = bid - base;
  const int xcd = j & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (j >> 3);
  const int pt = lid / ntn, nt = lid - pt * ntn;
  const int n0 = nt * 128;
  if (full) {
    conv_ppw_tile<EPI, 4>(a, (long)pt * 512, n0, nt, smem);
  } else {
    const long p0 = (long)(nfull / ntn) * 512 + (long)pt * (128 * a.pp_nbr);
    if (a.pp_nbr == 3) conv_ppw_tile<EPI, 3>(a, p0, n0, nt, smem);
    else if (a.pp_nbr == 2) conv_ppw_tile<EPI, 2>(a, p0, n0, nt, smem);
    else conv_ppw_tile<EPI, 1>(a, p0, n0, nt, smem);
  }
#endif
}

template <int EPI, int NB, int BK, int NW, int ST, int MB>
static void launch_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr int RB = BK * 2, PT = (NW / 2) * 16 * NB;
  constexpr size_t stage_bytes = (size_t)ST * (PT * RB + 32 * MB * RB);
  constexpr size_t lds = stage_bytes;
  static PerDeviceOnce attr;            // > 64 KB of dynamic LDS needs the opt-in once per kernel
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<EPI, NB, BK, NW, ST, MB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_igemm_kernel<EPI, NB, BK, NW, ST, MB>), grid, dim3(64 * NW), lds, st, a);
}

template <int EPI, int MB>
static void launch_halo_one(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  static size_t granted[kMaxDevices] = {};    // more dynamic LDS than 64 KB needs the opt-in (per kernel, device and size)
  size_t& g = granted[current_device()];
  if (lds > 64 * 1024 && lds > g) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<EPI, MB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    g = lds;
  }
  hipLaunchKernelGGL((conv_halo_kernel<EPI, MB>), grid, dim3(256), lds, st, a);
}
// LDS of conv_halo_kernel: weight stage + haloed pixel rows, and never less than the
// [pixel][channel] tile its seeding prologue / epilogues lay out
static size_t halo_lds_bytes(int W, int MB) {
  const size_t need = (size_t)32 * MB * 128 + (size_t)(128 + 2 * (W + 1)) * 128;
  const size_t seed = (size_t)128 * 32 * MB * 2;
  return need > seed ? need : seed;
}
template <int EPI>
static void launch_pp_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds = 5 * 256 * 128;                     // two pixel stages + three weight stages
  static PerDeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pp_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  }
  hipLaunchKernelGGL((conv_pp_kernel<EPI>), grid, dim3(512), lds, st, a);
}
// 256-channel x 256-pixel ping-pong tiles (conv_pp_kernel): layers whose output channels are a multiple of 256
static int launch_conv_pp(const ConvArgs& a_, int epilogue, hipStream_t st) {
  if ((a_.nout & 255) || a_.pbeg != 0) return GLORIE_EUNSUPPORTED;
  ConvArgs a = a_;
  const int ncu = device_cus();
  const long ntn = a.nout / 256;
  const long pt_all = (a.P + 255) / 256;
  // whole rounds of full tiles (one workgroup per CU, `ntn` channel tiles per pixel tile); what is left goes into one partial
  // round of smaller tiles when that makes the round shorter
  long full_pt = pt_all * ntn / ncu * ncu / ntn;
  full_pt -= full_pt % 8;                                   // (the XCD remap of both ranges wants multiples of 8)
  if (full_pt < 0) full_pt = 0;
  long rest_px = a.P - full_pt * 256;
  int nbr = 4;
  long rest_pt = 0;
  if (rest_px > 0) {
    const long per_cu = (rest_px * ntn + ncu - 1) / ncu;    // pixels per CU if the rest is spread evenly
    nbr = (int)((per_cu + 63) / 64);
    if (nbr > 4 || full_pt == 0) nbr = 4;
    if (nbr < 1) nbr = 1;
    rest_pt = (rest_px + 64 * nbr - 1) / (64 * nbr);
  }
  if (nbr == 4) { full_pt = pt_all; rest_pt = 0; }
  const long nwg = (full_pt + rest_pt) * ntn;
  if (nwg <= 0) return GLORIE_OK;
  if (nwg > 0x7fffffffL) return GLORIE_EINVAL;
  a.pp_full = (int)(full_pt * ntn);
  a.pp_nbr = nbr;
  const dim3 grid((unsigned)nwg);
  switch (epilogue) {
    case EPI_BIAS_ACT: launch_pp_one<EPI_BIAS_ACT>(a, grid, st); break;
    case EPI_GRU_ZR: launch_pp_one<EPI_GRU_ZR>(a, grid, st); break;
    default: return GLORIE_EUNSUPPORTED;
  }
  return check_launch();
}

template <int EPI>
static void launch_ppw_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds = 4 * 256 * 128 + 2 * 128 * 128;     // four pixel buffers + two weight stages = 160 KB
  static PerDeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ppw_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  }
  hipLaunchKernelGGL((conv_ppw_kernel<EPI>), grid, dim3(512), lds, st, a);
}
// 128-channel x 512-pixel ping-pong tiles (conv_ppw_kernel): layers whose output channels are a multiple of 128
static int launch_conv_ppw(const ConvArgs& a_, int epilogue, hipStream_t st) {
  if ((a_.nout & 127) || a_.pbeg != 0) return GLORIE_EUNSUPPORTED;
  ConvArgs a = a_;
  const int ncu = device_cus();
  const long ntn = a.nout / 128;
  const long pt_all = (a.P + 511) / 512;
  // whole rounds of full tiles, the rest in one partial round of smaller tiles when that makes the round shorter (launch_conv_pp)
  long full_pt = pt_all * ntn / ncu * ncu / ntn;
  full_pt -= full_pt % 8;
  if (full_pt < 0) full_pt = 0;
  long rest_px = a.P - full_pt * 512;
  int nbr = 4;
  long rest_pt = 0;
  if (rest_px > 0) {
    const long per_cu = (rest_px * ntn + ncu - 1) / ncu;
    nbr = (int)((per_cu + 127) / 128);
    if (nbr > 4 || full_pt == 0) nbr = 4;
    if (nbr < 1) nbr = 1;
    rest_pt = (rest_px + 128 * nbr - 1) / (128 * nbr);
  }
  if (nbr == 4) { full_pt = pt_all; rest_pt = 0; }
  const long nwg = (full_pt + rest_pt) * ntn;
  if (nwg <= 0) return GLORIE_OK;
  if (nwg > 0x7fffffffL) return GLORIE_EINVAL;
  a.pp_full = (int)(full_pt * ntn);
  a.pp_nbr = nbr;
  const dim3 grid((unsigned)nwg);
  switch (epilogue) {
    case EPI_BIAS_ACT: launch_ppw_one<EPI_BIAS_ACT>(a, grid, st); break;
    case EPI_GRU_Q: launch_ppw_one<EPI_GRU_Q>(a, grid, st); break;
    case EPI_HEADS: launch_ppw_one<EPI_HEADS>(a, grid, st); break;
    default: return GLORIE_EUNSUPPORTED;
  }
  return check_launch();
}

template <int MB>
static int launch_conv_halo(const ConvArgs& a, int epilogue, hipStream_t st) {
  constexpr int PT = 128, TN = 32 * MB;
  if (a.pair && (MB % 2)) return GLORIE_EINVAL;        // paired weight rows need pairs of 16-row blocks (conv_epilogue_pair)
  const long ptiles = (a.P - a.pbeg + PT - 1) / PT;
  if (ptiles <= 0) return GLORIE_OK;
  const long nwg = ptiles * ((a.nout + TN - 1) / TN);
  if (nwg > 0x7fffffffL) return GLORIE_EINVAL;
  const dim3 grid((unsigned)nwg);
  const size_t lds = halo_lds_bytes(a.W, MB);
  switch (epilogue) {
    case EPI_BIAS_ACT: launch_halo_one<EPI_BIAS_ACT, MB>(a, grid, lds, st); break;
    case EPI_GRU_ZR: launch_halo_one<EPI_GRU_ZR, MB>(a, grid, lds, st); break;
    case EPI_GRU_Q:
      if constexpr (MB == 4) launch_halo_one<EPI_GRU_Q, MB>(a, grid, lds, st);
      else return GLORIE_EUNSUPPORTED;
      break;
    case EPI_HEADS:
      if constexpr (MB == 4) launch_halo_one<EPI_HEADS, MB>(a, grid, lds, st);
      else return GLORIE_EUNSUPPORTED;
      break;
    default: return GLORIE_EUNSUPPORTED;
  }
  return check_launch();
}

template <int NB, int BK, int NW, int ST, int MB = 4>
static int launch_conv(const ConvArgs& a, int epilogue, hipStream_t st, long max_ptiles = -1) {
  constexpr int PT = (NW / 2) * 16 * NB;
  if (a.pair && (MB % 2)) return GLORIE_EINVAL;        // paired weight rows need pairs of 16-row blocks (conv_epilogue_pair)
  long ptiles = (a.P - a.pbeg + PT - 1) / PT;
  if (epilogue == EPI_GLO) ptiles = (a.P / a.HW) * ((a.HW + PT - 1) / PT);      // tiles laid per map
  if (max_ptiles >= 0) ptiles = ptiles < max_ptiles ? ptiles : max_ptiles;
  if (ptiles <= 0) return GLORIE_OK;
  const long nwg = ptiles * ((a.nout + 32 * MB - 1) / (32 * MB));
  if (nwg > 0x7fffffffL) return GLORIE_EINVAL;
  const dim3 grid((unsigned)nwg);
  switch (epilogue) {
    case EPI_BIAS_ACT: launch_one<EPI_BIAS_ACT, NB, BK, NW, ST, MB>(a, grid, st); break;
    case EPI_GRU_ZR: launch_one<EPI_GRU_ZR, NB, BK, NW, ST, MB>(a, grid, st); break;
    case EPI_GLO:
      if constexpr (MB == 4 && NB == 4 && NW == 4 && ST == 1) launch_one<EPI_GLO, NB, BK, NW, ST, MB>(a, grid, st);
      else return GLORIE_EUNSUPPORTED;
      break;
    case EPI_HEADS:
      if constexpr (MB == 4 && NB == 4 && NW == 4 && ST == 1) launch_one<EPI_HEADS, NB, BK, NW, ST, MB>(a, grid, st);
      else return GLORIE_EUNSUPPORTED;
      break;
    case EPI_UPSAMPLE:
      if constexpr (MB == 4 && NB == 4 && NW == 4 && ST == 1) launch_one<EPI_UPSAMPLE, NB, BK, NW, ST, MB>(a, grid, st);
      else return GLORIE_EUNSUPPORTED;
      break;
    default: launch_one<EPI_GRU_Q, NB, BK, NW, ST, MB>(a, grid, st); break;
  }
  return check_launch();
}

}  // namespace glorie

using namespace glorie;

static int conv_igemm_impl(const void* xa, int xa_stride, int ca, const void* xb, int xb_stride,
                           int cb, const void* w_packed, int taps, int nout, int epilogue,
                           const float* terms, int terms_stride, int act, const void* net,
                           int net_stride, const void* z, int z_stride, void* out, int out_stride,
                           void* out2, int out2_stride, const void* pre, int pre_stride, const int* pre_map,
                           int N, int H, int W, void* stream, const void* tap_w, float* tap_out, int tap_groups,
                           int tap_ncols, const float* up_disps = nullptr, const int64_t* up_ix = nullptr,
                           float* up_out = nullptr, int up_f32 = 0) {
  if (N < 0 || H <= 0 || W <= 0 || (taps != 1 && taps != 9) || nout <= 0 || (nout & 3)) return GLORIE_EINVAL;
  if (ca < 0 || cb < 0 || (ca % 64) || (cb % 64) || ca + cb == 0) return GLORIE_EINVAL;
  const int pair = (epilogue >> 8) & 1;                 // GLORIE_CONV_PAIR16: the weights come from a paired packing
  const int policy = (epilogue >> 12) & 15;             // GLORIE_CONV_POLICY_*: tile choice forced by the caller (tests, bench_conv)
  epilogue &= 0xff;
  if (policy > 7) return GLORIE_EINVAL;
  if (epilogue < 0 || epilogue > 5) return GLORIE_EINVAL;
  if (pair && (epilogue > EPI_GRU_Q || (nout & 31))) return GLORIE_EINVAL;
  if (epilogue == EPI_UPSAMPLE && (!up_disps || !up_ix || !up_out || !terms || nout != 1024 || taps != 1 || pre))
    return GLORIE_EINVAL;
  if (epilogue == EPI_HEADS && (!tap_w || !tap_out || !terms || tap_groups < 1 || tap_ncols < 1 || tap_ncols > 32 ||
                                nout < 128 * tap_groups || (nout & 127) || act != CACT_RELU || pre))
    return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;               // nothing to do: pointers of empty maps may be null
  if ((ca && (!xa || (xa_stride & 7))) || (cb && (!xb || (xb_stride & 7)))) return GLORIE_EINVAL;
  if (!w_packed || !out || (out_stride & 3)) return GLORIE_EINVAL;
  if (epilogue == EPI_GRU_ZR && (nout != 256 || !terms || !net || !out2 || (terms_stride & 3))) return GLORIE_EINVAL;
  if (epilogue == EPI_GRU_Q && (nout != 128 || !terms || !net || !z || (terms_stride & 3))) return GLORIE_EINVAL;
  if (pre && (epilogue == EPI_BIAS_ACT || epilogue == EPI_GLO || (pre_stride & 3))) return GLORIE_EINVAL;
  if (epilogue == EPI_GLO && (nout != 128 || taps != 1 || !terms || !net)) return GLORIE_EINVAL;
  ConvArgs a;
  a.xa = reinterpret_cast<const _Float16*>(xa); a.xa_stride = xa_stride; a.cha = ca / 64;
  a.xb = reinterpret_cast<const _Float16*>(xb); a.xb_stride = xb_stride; a.chb = cb / 64;
  a.w = reinterpret_cast<const _Float16*>(w_packed);
  a.taps = taps; a.nout = nout; a.npad = (nout + kTileN - 1) / kTileN * kTileN;
  a.P = (long)N * H * W; a.H = H; a.W = W; a.HW = H * W;
  a.out = reinterpret_cast<_Float16*>(out); a.out_stride = out_stride;
  a.out2 = reinterpret_cast<_Float16*>(out2); a.out2_stride = out2_stride;
  a.terms = terms; a.terms_stride = terms_stride; a.act = act;
  a.net = reinterpret_cast<const _Float16*>(net); a.net_stride = net_stride;
  a.z = reinterpret_cast<const _Float16*>(z); a.z_stride = z_stride;
  a.pre = reinterpret_cast<const _Float16*>(pre); a.pre_stride = pre_stride; a.pre_map = pre ? pre_map : nullptr;
  a.pbeg = 0;
  a.tap_w = reinterpret_cast<const f16x8*>(tap_w); a.tap_out = tap_out; a.tap_groups = tap_groups;
  a.tap_ncols = tap_ncols;
  a.up_disps = up_disps; a.up_ix = up_ix; a.up_out = up_out; a.up_f32 = up_f32;
  // the context term joins in the epilogue (16-byte pieces with paired weights; rounds 2-3 seeded the accumulators with it
  // through an LDS image of the [pixel][channel] tile, which the paired channel order made obsolete)
  a.pair = pair;
  a.pp_full = 0; a.pp_nbr = 4;
#ifdef EXP_CONV_STAMPS
  a.stamps = getenv("GLORIE_CONV_STAMPS") ? (unsigned long long*)strtoull(getenv("GLORIE_CONV_STAMPS"), nullptr, 0) : nullptr;
#else
  a.stamps = nullptr;
#endif
  // buffer-descriptor addressing: 31-bit byte offsets per input segment and for the weights
  const long lim = 0x7fffffffL;
  if (((a.P + W + 2) * (long)xa_stride + 64) * 2 > lim || ((a.P + W + 2) * (long)xb_stride + 64) * 2 > lim ||
      ((long)taps * a.npad * (ca + cb) + 64) * 2 > lim || (pre && (a.P * (long)pre_stride + 512) * 2 > lim) ||
      (a.P * (long)out_stride + nout) * 2 > lim || (out2 && (a.P * (long)out2_stride + 128) * 2 > lim))
    return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // Measured on the update operator's layers at 36x60x80 (tools/bench_conv.py): the single-stage
  // 128x128 tile at 3 workgroups per CU (1050 TFLOP/s on the 448->256 layer) beats every variant with
  // more LDS stages and fewer resident workgroups: 2 stages 930, 2 stages + register-resident fragments
  // (loads two steps ahead, ST = 4) 915, 8 waves with a 3-stage ring and counted vmcnt 830, 256-pixel
  // tiles 850, 32-channel steps at 4 workgroups per CU 824; the same wave tile on v_mfma_f32_32x32x16_f16
  // (16 instead of 32 MFMAs per step) 945.
  // 3x3 layers on the 128-channel tile: the pixel tile is shared by the nine taps of a chunk (conv_halo_kernel) while its haloed
  // tile leaves three workgroups per CU (image width <= 83); small launches keep their 64-pixel tiles, the 256- and 64-channel
  // tiles their per-tap staging (measured slower with the shared tile).  Policy 5 (nohalo): per-tap staging everywhere.
  if (policy == 6) return launch_conv_pp(a, epilogue, st);     // GLORIE_CONV_POLICY_PP: forced by the caller
  if (policy == 7) return launch_conv_ppw(a, epilogue, st);    // GLORIE_CONV_POLICY_PPW
  {
    // 128-channel layers on the 128 x 512 ping-pong tile (conv_ppw_kernel): 3x3 layers whose output channels are a multiple of
    // 128 (but not of 256: those take conv_pp_kernel) once the map fills the chip with 512-pixel tiles.  GLORIE_CONV_PPW = 0 / 1
    // switches it off / forces it wherever the kernel applies (A/B runs, tests of the heads entry point)
    const char* ppw = getenv("GLORIE_CONV_PPW");
    const bool fits = (nout & 127) == 0 && (epilogue == EPI_BIAS_ACT || epilogue == EPI_GRU_Q || epilogue == EPI_HEADS);
    const bool want = ppw ? ppw[0] == '1' : (taps == 9 && (nout == 128 || epilogue == EPI_HEADS) && epilogue != EPI_BIAS_ACT &&
                                             a.P >= kConvPpwMinPixels);
    if (policy == 0 && fits && want && !(ppw && ppw[0] == '0')) return launch_conv_ppw(a, epilogue, st);
  }
  {
    if (policy == 0 && taps == 9 && a.pbeg == 0) {
      const long tiles128 = (a.P + 127) / 128 * ((nout + 127) / 128);
      if (nout > 64 && (nout & 255) != 0 && W <= 83 && (epilogue <= EPI_GRU_Q || epilogue == EPI_HEADS) &&
          !(epilogue == EPI_BIAS_ACT && tiles128 <= 384))
        return launch_conv_halo<4>(a, epilogue, st);
    }
  }
  if (epilogue == EPI_HEADS || epilogue == EPI_UPSAMPLE) return launch_conv<4, 64, 4, 1>(a, epilogue, st);
  // layers with <= 64 output channels (flow_encoder[2]) use a 64-channel tile instead of padding to 128
  if (nout <= 64 && epilogue == EPI_BIAS_ACT) return launch_conv<4, 64, 4, 1, 2>(a, epilogue, st);
  // Tile choice.  Default: 128 x 128 (3 workgroups per CU); layers whose output channels are a multiple of 256 (the z|r
  // gates) take 256 channels x 128 pixels - 4 waves of 128 x 64, 2 workgroups per CU, 25 % fewer staged bytes per MFMA:
  // 448 -> 256 at 36x60x80 345 -> 315 us, the fused gate launch 277 -> 265 us stand-alone.
  // Measured and NOT used: 192 channels x 128 pixels for the 384-channel head convolution (155 -> 167 us);
  // 128 channels x 256 pixels for the 128-channel layers (q gate 174 -> 176 us); 64-pixel tiles
  // (4 per CU; z|r 329 us); whole rounds on 128-pixel tiles + the remainder as a second launch of 64-pixel tiles
  // (292 us: a 64-pixel workgroup costs 0.8 of a 128-pixel one and workgroups do not run in lockstep rounds, so the
  // partly filled last round is cheaper than a round model says); two LDS stages with one barrier per K-tile and the next
  // tile's DMA issued in the shadow of the fragment reads (round 3: 448 -> 256 at 32-channel K-tiles 320 -> 406 us, 448 -> 128
  // 186 -> 197 us at 64-channel, 248 us at 32-channel K-tiles); the single stage with 32-channel K-tiles at FOUR workgroups
  // per CU (113-128 VGPRs: 448 -> 128 186 -> 271 us).  The policy bits keep the variants reachable for tests and
  // tools/bench_conv.py.
  const int ntn = (nout + 127) / 128;
  const long slots128 = 3L * 256;
  const long full_rounds = ((a.P + 127) / 128 * ntn) / slots128;
  const long pt_a = full_rounds * slots128 / ntn;               // pixel tiles of the whole rounds
  // policy (bits 12-15 of `epilogue`, include/glorie_hip.h): 0 auto, 1 = 128 x 128, 2 = 64-pixel tiles, 3 = whole rounds +
  // remainder, 4 = 128 x 256, 5 = auto without the haloed tile
  const char t0c = policy == 1 ? '1' : policy == 2 ? '6' : policy == 3 ? 's' : policy == 4 ? 'w' : 0;
  // z|r gates and other 3x3 layers with a multiple of 256 output channels, once the map fills the chip at least once with
  // 256-pixel tiles: the ping-pong kernel (z|r gate launch at G8 251 -> 234 us in an interleaved A/B, tools/bench_conv.py)
  if (policy == 0 && (nout & 255) == 0 && taps == 9 && epilogue <= EPI_GRU_ZR && a.P >= 256L * 256)
    return launch_conv_pp(a, epilogue, st);
  if (t0c == 0 && (nout & 255) == 0) return launch_conv<4, 64, 4, 1, 8>(a, epilogue, st);
  // small launches (GraphAgg's convolutions run on the 8 keyframe maps, 300 pixel tiles): 128-pixel tiles would leave most
  // of the 768 workgroup slots empty and every CU with one latency-bound workgroup - 64-pixel tiles double the workgroups
  if (t0c == 0 && epilogue == EPI_BIAS_ACT && (a.P + 127) / 128 * ntn <= 384) return launch_conv<2, 64, 4, 1>(a, epilogue, st);
  // 1x1 layers have two to four K steps: prologue and epilogue dominate and more, smaller workgroups hide them better
  // (corr_encoder[0] 256 -> 128 at 36x60x80: 33.7 -> 31.8 us)
  if (t0c == 0 && epilogue == EPI_BIAS_ACT && taps == 1) return launch_conv<2, 64, 4, 1>(a, epilogue, st);
  if (t0c == 'w') return launch_conv<8, 64, 4, 1, 4>(a, epilogue, st);
  if (t0c == '6') return launch_conv<2, 64, 4, 1>(a, epilogue, st);
  if (t0c == 's' && full_rounds >= 1) {
    const int rc_a = launch_conv<4, 64, 4, 1>(a, epilogue, st, pt_a);
    if (rc_a != GLORIE_OK) return rc_a;
    a.pbeg = pt_a * 128;
    return a.pbeg < a.P ? launch_conv<2, 64, 4, 1>(a, epilogue, st) : GLORIE_OK;
  }
  return launch_conv<4, 64, 4, 1>(a, epilogue, st);
}

extern "C" int glorie_conv_igemm(const void* xa, int xa_stride, int ca, const void* xb, int xb_stride,
                                 int cb, const void* w_packed, int taps, int nout, int epilogue,
                                 const float* terms, int terms_stride, int act, const void* net,
                                 int net_stride, const void* z, int z_stride, void* out, int out_stride,
                                 void* out2, int out2_stride, const void* pre, int pre_stride, const int* pre_map,
                                 int N, int H, int W, void* stream) {
  if ((epilogue & 0xff) == EPI_HEADS || (epilogue & 0xff) == EPI_UPSAMPLE) return GLORIE_EINVAL;   // have their own entry points
  return conv_igemm_impl(xa, xa_stride, ca, xb, xb_stride, cb, w_packed, taps, nout, epilogue, terms, terms_stride, act,
                         net, net_stride, z, z_stride, out, out_stride, out2, out2_stride, pre, pre_stride, pre_map, N, H,
                         W, stream, nullptr, nullptr, 0, 0);
}

extern "C" int glorie_conv_igemm_heads(const void* x, int x_stride, int c, const void* w_packed, int taps, int nout,
                                       const float* bias, const void* tap_w, int groups, int K, float* tap_out,
                                       void* out, int out_stride, int N, int H, int W, void* stream) {
  if (K < 1 || K > 3 || groups < 1 || groups > 4) return GLORIE_EINVAL;
  if (nout > 128 * groups && !out) return GLORIE_EINVAL;
  // `out` holds channels 128*groups .. nout-1 only; with no such channels any non-null pointer passes the checks
  void* o = out ? out : (void*)tap_out;
  return conv_igemm_impl(x, x_stride, c, nullptr, 0, 0, w_packed, taps, nout, EPI_HEADS, bias, 0, CACT_RELU, nullptr, 0,
                         nullptr, 0, o, out ? out_stride : 4, nullptr, 0, nullptr, 0, nullptr, N, H, W, stream, tap_w,
                         tap_out, groups, 9 * K);
}

extern "C" int glorie_conv_upsample(const void* x, int x_stride, int c, const void* w_packed, const float* bias,
                                    const float* disps, const int64_t* ix, float* disps_up, int softmax_f32, int M,
                                    int H, int W, void* stream) {
  if (!disps || !ix || !disps_up || !bias) return GLORIE_EINVAL;
  // no per-pixel output: `out` only has to pass the pointer checks
  return conv_igemm_impl(x, x_stride, c, nullptr, 0, 0, w_packed, 1, 1024, EPI_UPSAMPLE, bias, 0, CACT_NONE, nullptr, 0,
                         nullptr, 0, disps_up, 4, nullptr, 0, nullptr, 0, nullptr, M, H, W, stream, nullptr, nullptr, 0,
                         0, disps, ix, disps_up, softmax_f32);
}
