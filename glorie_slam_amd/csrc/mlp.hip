// Fused neural-point decoders on the matrix cores (scope row R3) for gfx950.
//
// Replaces the chains of small nn.Linear launches of
//   MLP_geometry.forward   /root/reference/src/modules/conv_onet/models/decoder.py:175-225
//   MLP_color.get_feature_at_pos (per-neighbour F_theta)              decoder.py:340-389, 228-243
//   MLP_color.forward                                                 decoder.py:391-433
// with three kernels that keep every activation in registers:
//   geo  : Fourier(93, sin) -> 5 x (Linear(32) ReLU + fc_c(c)) with the skip at layer 2 -> occ
//   nb   : per neighbour [sin,cos](rel B)(20) ++ col_feat(32) -> Linear(128) softplus -> IDW sum
//          -> Linear(32).  The second layer is linear, so it is applied ONCE to the weighted
//          sum:  sum_k w_k (W2 y_k + b2) = W2 (sum_k w_k y_k) + b2 sum_k w_k   (8x fewer MACs).
//   col  : [sin,cos](p B)(40) ++ [sin,cos](v B)(40) -> 5 x (Linear(128) softplus + fc_c(c)),
//          skip at layer 2 -> sigmoid rgb
//
// All GEMMs are exact fp32 on v_mfma_f32_16x16x4_f32 (the reference runs fp32; no xf32 on gfx950), in the
// transposed form  H'^T = W^T H^T : the 16 MFMA rows are output channels (A = weights from LDS), the 16
// columns are a wave's 16 samples (B = activations).  The accumulator fragment of one layer is then
// directly the B fragment of the next (see mlp_col_v3_kernel), so no activation ever touches LDS.  A
// workgroup = 8 waves = 128 samples; the skip connection is two GEMMs on the split weight (embedding
// rows / hidden rows) - the concatenated activation is never built.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NT>
__device__ __forceinline__ void zero(f32x4 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- packed parameters (device pointers into one buffer, see point_ops.pack_decoders) ----
struct GeoParams {
  const float* B;          // [3][96]   Fourier matrix (93 cols, zero padded)
  const float* W0;         // [96][32]
  const float* W1;         // [32][32]
  const float* W2;         // [32][32]
  const float* W3e;        // [96][32]  embedding rows of the skip layer
  const float* W3h;        // [32][32]  hidden rows of the skip layer
  const float* W4;         // [32][32]
  const float* Wout;       // [32][16]  (col 0 real)
  const float* Fc;         // [5][32][32]
  const float* bias;       // [5][32]   pts_linears biases
  const float* fcb;        // [5][32]   fc_c biases
  const float* bout;       // [4]       output bias (element 0)
};

struct NbParams {
  const float* B;          // [3][10]
  const float* W1;         // [52][128]
  const float* b1;         // [128]
  const float* W2;         // [128][32]
  const float* b2;         // [32]
};

struct ColParams {
  const float* Bp;         // [3][20]
  const float* Bv;         // [3][20]
  const float* W0;         // [80][128]
  const float* W1;         // [128][128]
  const float* W2;         // [128][128]
  const float* W3e;        // [80][128]
  const float* W3h;        // [128][128]
  const float* W4;         // [128][128]
  const float* Wout;       // [128][16] (cols 0..2 real)
  const float* Fc;         // [5][32][128]
  const float* bias;       // [5][128]
  const float* fcb;        // [5][128]
  const float* bout;       // [4]       output bias (rgb)
};

// ====================================================================================
// 128 samples / 8 waves per workgroup.  The colour weights (442 KB) are streamed through LDS in 32-row
// chunks (double buffered, shared by the 8 waves); the geometry and per-neighbour weights are resident.
// ====================================================================================
constexpr int kTM2 = 128;

// torch.nn.Softplus(beta=100, threshold=20) on the bare base-2 hardware transcendentals (v_exp_f32 /
// v_log_f32, 1 ulp each), in the overflow-free form
//   softplus(x) = max(x, 0) + (ln 2 / 100) log2(1 + 2^(-100 log2(e) |x|))
// which needs no threshold branch: for 100 x > 20 the second term is log2(1 + < 2^-28) = 0 in fp32, i.e.
// exactly x like torch's threshold.  On this chip VALU work does not hide behind fp32 MFMAs of the same
// SIMD (tools/probes/mfma_valu_overlap.hip: the two add up), and expf / logf cost 3.7x these two instructions.
__device__ __forceinline__ float softplus100_fast(float x) {
#ifdef EXP_MLP_NO_SOFTPLUS
  return fmaxf(x, 0.0f);                                    // ablation (tools/exp_mlp_ablate.sh): what do the transcendentals cost?
#endif
  const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * __builtin_fabsf(x));
  return fmaf(0.006931471805599453f, __builtin_amdgcn_logf(1.0f + e), fmaxf(x, 0.0f));
}
// sin / cos of 2 pi rev on v_sin_f32 / v_cos_f32 (argument in revolutions, reduced with v_fract_f32)
__device__ __forceinline__ float sin_rev(float rev) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev)); }
__device__ __forceinline__ float cos_rev(float rev) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(rev)); }

struct ChunkRegs { float4 a, b; };

// thread t of 512 moves 8 floats: row = t >> 4 (0..31), cols (t & 15) * 8 .. + 7
__device__ __forceinline__ ChunkRegs chunk_load(const float* __restrict__ Wall, int chunk) {
  const int t = threadIdx.x;
  const float4* src = reinterpret_cast<const float4*>(Wall + ((size_t)chunk * 32 + (t >> 4)) * 128 + (t & 15) * 8);
  ChunkRegs r;
  r.a = src[0];
  r.b = src[1];
  return r;
}
// ---- geometry decoder, transposed formulation (see the colour decoder below for the idea) ----------------
// All weights of the network (480 K-rows of 32 outputs, 61 KB, + the 32 x 16 output layer) are staged once per
// workgroup as the LDS image built by point_ops.pack_decoders: row k holds output 16 to + r at 2 r + to, so
// one ds_read_b64 per lane fetches the A operands of the two 16-channel output blocks.  Embedding (96, sin),
// feature c (32) and hidden state (32) live in registers as B fragments; no barrier after the staging.
constexpr int kGeoRows = 480;
constexpr int kGeoImage = kGeoRows * 32 + 32 * 16;   // floats

template <int NT, int NB>
__device__ __forceinline__ void mma_geo(f32x4 (&acc)[2], const f32x4 (&b)[NB], const float* Wrows) {
  const int lane = threadIdx.x & 63;
  const float* wp = Wrows + (lane >> 4) * 4 * 32 + (lane & 15) * 2;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float2 w = *reinterpret_cast<const float2*>(wp + (16 * t + rr) * 32);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b[t][rr], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b[t][rr], acc[1], 0, 0, 0);
    }
}

__global__ __launch_bounds__(512, 4) void mlp_geo_v3_kernel(GeoParams P, const float* __restrict__ image,
                                                            const float* __restrict__ pts,
                                                            const float* __restrict__ c_geo,
                                                            const float* __restrict__ geo_feats,
                                                            const int64_t* __restrict__ I,
                                                            const float* __restrict__ wts,
                                                            const uint8_t* __restrict__ has, int Q,
                                                            float* __restrict__ raw) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int qs = blockIdx.x * kTM2 + wv * 16 + r;
  const int q = min(qs, Q - 1);
  for (int idx = tid; idx < kGeoImage / 4; idx += 512)
    reinterpret_cast<float4*>(smem)[idx] = reinterpret_cast<const float4*>(image)[idx];
  // B fragments: channel 16t + 4g + rr of sample r
  f32x4 e[6], c[2], h[2], acc[2];
  {
    const float x = pts[(size_t)q * 3 + 0], y = pts[(size_t)q * 3 + 1], z = pts[(size_t)q * 3 + 2];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const float4 b0 = *reinterpret_cast<const float4*>(P.B + 16 * t + 4 * g);
      const float4 b1 = *reinterpret_cast<const float4*>(P.B + 96 + 16 * t + 4 * g);
      const float4 b2 = *reinterpret_cast<const float4*>(P.B + 192 + 16 * t + 4 * g);
      e[t][0] = sin_rev(fmaf(z, b2.x, fmaf(y, b1.x, x * b0.x)));
      e[t][1] = sin_rev(fmaf(z, b2.y, fmaf(y, b1.y, x * b0.y)));
      e[t][2] = sin_rev(fmaf(z, b2.z, fmaf(y, b1.z, x * b0.z)));
      e[t][3] = sin_rev(fmaf(z, b2.w, fmaf(y, b1.w, x * b0.w)));
    }
    // columns 93..95 of B are zero padding (sin 0 = 0) and so are the matching weight rows
    if (geo_feats) {
      // IDW interpolation of the neighbours' features right here (decoder.py:130-173; same summation order
      // as idw_gather_kernel): the loads travel with the weight image, c_geo never exists in HBM
      c[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      c[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool on = has[q] != 0;
      // the rows of GEO_NB neighbours are requested together and unconditionally (index clamped, zero-weight rows dropped
      // by a select - the same bits): a load under `if (wk != 0)` is one exposed L2 round trip per neighbour at the head
      // of every workgroup
#ifndef GLORIE_GEO_NB
#define GLORIE_GEO_NB 2
#endif
      constexpr int GEO_NB = GLORIE_GEO_NB;
#pragma unroll
      for (int k0 = 0; k0 < 8; k0 += GEO_NB) {
        float wk[GEO_NB];
        float4 v[GEO_NB][2];
#pragma unroll
        for (int k = 0; k < GEO_NB; ++k) {
          wk[k] = wts[(size_t)q * 8 + k0 + k];
          const long ik = max(I[(size_t)q * 8 + k0 + k], 0L);
#pragma unroll
          for (int t = 0; t < 2; ++t)
            v[k][t] = *reinterpret_cast<const float4*>(geo_feats + (size_t)ik * 32 + 16 * t + 4 * g);
        }
#pragma unroll
        for (int k = 0; k < GEO_NB; ++k) {
          const bool use = on && wk[k] != 0.0f;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            c[t][0] = use ? c[t][0] + wk[k] * v[k][t].x : c[t][0];
            c[t][1] = use ? c[t][1] + wk[k] * v[k][t].y : c[t][1];
            c[t][2] = use ? c[t][2] + wk[k] * v[k][t].z : c[t][2];
            c[t][3] = use ? c[t][3] + wk[k] * v[k][t].w : c[t][3];
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(c_geo + (size_t)q * 32 + 16 * t + 4 * g);
        c[t][0] = v.x; c[t][1] = v.y; c[t][2] = v.z; c[t][3] = v.w;
      }
    }
  }
  __syncthreads();
  auto act = [&](int li) {   // ReLU(acc + bias) + fc_c bias
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(P.bias + li * 32 + 16 * t + 4 * g);
      const float4 fb = *reinterpret_cast<const float4*>(P.fcb + li * 32 + 16 * t + 4 * g);
      acc[t][0] = fmaxf(acc[t][0] + bb.x, 0.0f) + fb.x;
      acc[t][1] = fmaxf(acc[t][1] + bb.y, 0.0f) + fb.y;
      acc[t][2] = fmaxf(acc[t][2] + bb.z, 0.0f) + fb.z;
      acc[t][3] = fmaxf(acc[t][3] + bb.w, 0.0f) + fb.w;
    }
  };
  auto next_layer = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) { h[t] = acc[t]; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
  const float* W = smem;   // row offsets: W0 0 | Fc0 96 | W1 128 | Fc1 160 | W2 192 | Fc2 224 | W3e 256 | W3h 352
                           //              | Fc3 384 | W4 416 | Fc4 448 | Wout 480 (16 columns, natural order)
  zero<2>(acc);
  mma_geo<6, 6>(acc, e, W);               act(0); mma_geo<2, 2>(acc, c, W + 96 * 32);
  next_layer();
  mma_geo<2, 2>(acc, h, W + 128 * 32);    act(1); mma_geo<2, 2>(acc, c, W + 160 * 32);
  next_layer();
  mma_geo<2, 2>(acc, h, W + 192 * 32);    act(2); mma_geo<2, 2>(acc, c, W + 224 * 32);
  next_layer();
  mma_geo<6, 6>(acc, e, W + 256 * 32);
  mma_geo<2, 2>(acc, h, W + 352 * 32);    act(3); mma_geo<2, 2>(acc, c, W + 384 * 32);
  next_layer();
  mma_geo<2, 2>(acc, h, W + 416 * 32);    act(4); mma_geo<2, 2>(acc, c, W + 448 * 32);
  // output layer 32 -> 1 (16 padded columns): lanes with g == 0 get channel 0 of their sample in o[0]
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* wp = W + kGeoRows * 32 + (4 * g) * 16 + r;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 16], acc[t][rr], o, 0, 0, 0);
  }
  if (g == 0 && qs < Q) raw[(size_t)qs * 4 + 3] = has[qs] ? o[0] + P.bout[0] : -100.0f;  // Renderer.py:206-207
}

// ---- colour decoder, transposed formulation ------------------------------------------------
// Every layer is evaluated as  H'^T = W^T . H^T : the MFMA's 16 "rows" are output channels (A = weights
// from the LDS chunk), its 16 "columns" are the wave's 16 samples (B = activations).  The D fragment
// of lane (r = lane & 15, g = lane >> 4) then holds channels 16t + 4g + rr (t = 0..7, rr = 0..3) of
// sample r - and the B operand of the next layer wants, for k-slot g, *some* channel of sample r.  The
// order in which the reduction runs over channels is free, so k-step (t, rr) simply takes channel
// 16t + 4g + rr from slot g: the accumulator registers of one layer are the B operands of the next,
// no transposition through LDS, no activation buffer at all.  The embedding and the colour feature
// use the same channel -> (step, slot) assignment.  LDS only holds the double-buffered weight chunk
// (33 KB), so two workgroups share a CU and one's barriers hide behind the other's MFMAs.
constexpr int kLdw3 = 132;                  // chunk row stride in LDS (floats)
constexpr int kChunkFloats3 = 32 * kLdw3;

__device__ __forceinline__ void chunk_store3(float* Wb, const ChunkRegs& r) {
  const int t = threadIdx.x;
  float4* dst = reinterpret_cast<float4*>(Wb + (t >> 4) * kLdw3 + (t & 15) * 8);
  dst[0] = r.a;
  dst[1] = r.b;
}

// acc[to] += W[16 tt + 4g + rr][16 to + r] * b[T0 + tt][rr]  for tt < NT, rr < 4 (rows relative to the chunk).
// The columns of a chunk row are stored permuted - column 16 to + r at (to >> 2) * 64 + 4 r + (to & 3) - so
// the eight A operands of a k-step are two ds_read_b128 (16 lanes read 256 contiguous bytes).
template <int T0, int NT, int NB>
__device__ __forceinline__ void mma_t(f32x4 (&acc)[8], const f32x4 (&b)[NB], const float* Wb) {
  const int lane = threadIdx.x & 63;
  const float* wp = Wb + (lane >> 4) * 4 * kLdw3 + (lane & 15) * 4;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float bv = b[T0 + tt][rr];
      const float4 w0 = *reinterpret_cast<const float4*>(wp + (16 * tt + rr) * kLdw3);
      const float4 w1 = *reinterpret_cast<const float4*>(wp + (16 * tt + rr) * kLdw3 + 64);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, bv, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, bv, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, bv, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, bv, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, bv, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, bv, acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, bv, acc[7], 0, 0, 0);
    }
}

__global__ __launch_bounds__(512, 4) void mlp_col_v3_kernel(ColParams P, const float* __restrict__ Wall,
                                                            const float* __restrict__ pts,
                                                            const float* __restrict__ views,
                                                            const float* __restrict__ c_col, int Q,
                                                            float* __restrict__ raw) {
  extern __shared__ float smem[];
  float* Wbuf = smem;                         // [2][32][132]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * kTM2;
  const int qs = q0 + wv * 16 + r;            // this lane's sample (all four k-slots of a column share it)
  const int q = min(qs, Q - 1);

  // ---- B fragments: the embedding e (80 channels) stays in registers; the colour feature c (32
  // channels) is re-read from L2 before each of its five uses (8 registers short of 4 waves / SIMD)
  f32x4 e[5], c[2];
  auto load_c = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(c_col + (size_t)q * 32 + 16 * t + 4 * g);
      c[t][0] = v.x; c[t][1] = v.y; c[t][2] = v.z; c[t][3] = v.w;
    }
  };
  {
    // phases in revolutions: the 2 pi of the reference's embedding is the period of v_sin / v_cos
    const float px = pts[(size_t)q * 3 + 0], py = pts[(size_t)q * 3 + 1], pz = pts[(size_t)q * 3 + 2];
    float vx = views[(size_t)q * 3 + 0], vy = views[(size_t)q * 3 + 1], vz = views[(size_t)q * 3 + 2];
    const float nrm = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
    vx = vx / nrm; vy = vy / nrm; vz = vz / nrm;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int f = 16 * t + 4 * g + rr;   // feature index 0..79: [sin p | cos p | sin v | cos v] x 20
        const int blk = f / 20, ff = f - blk * 20;
        const float* Bm = blk < 2 ? P.Bp : P.Bv;
        const float x = blk < 2 ? px : vx, y = blk < 2 ? py : vy, z = blk < 2 ? pz : vz;
        const float a = fmaf(z, Bm[40 + ff], fmaf(y, Bm[20 + ff], x * Bm[ff]));
        e[t][rr] = (blk & 1) ? cos_rev(a) : sin_rev(a);
      }
  }

  f32x4 acc[8], h[8];
  constexpr int NC = 27;
  ChunkRegs nxt = chunk_load(Wall, 0);
  chunk_store3(Wbuf, nxt);
  __syncthreads();

  auto act = [&](int li) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(P.bias + li * 128 + 16 * t + 4 * g);
      const float4 fb = *reinterpret_cast<const float4*>(P.fcb + li * 128 + 16 * t + 4 * g);
      acc[t][0] = softplus100_fast(acc[t][0] + bb.x) + fb.x;
      acc[t][1] = softplus100_fast(acc[t][1] + bb.y) + fb.y;
      acc[t][2] = softplus100_fast(acc[t][2] + bb.z) + fb.z;
      acc[t][3] = softplus100_fast(acc[t][3] + bb.w) + fb.w;
    }
  };
  auto next_layer = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t) { h[t] = acc[t]; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };

#define GL_CHUNK(cidx, BODY)                                        \
  {                                                                 \
    if ((cidx) + 1 < NC) nxt = chunk_load(Wall, (cidx) + 1);        \
    const float* Wb = Wbuf + ((cidx) & 1) * kChunkFloats3;          \
    BODY;                                                           \
    if ((cidx) + 1 < NC) chunk_store3(Wbuf + (((cidx) + 1) & 1) * kChunkFloats3, nxt); \
    __syncthreads();                                                \
  }

  // layer 0: W0 (80 rows -> chunks 0..2), Fc0 (chunk 3)
  zero<8>(acc);
  GL_CHUNK(0, (mma_t<0, 2, 5>(acc, e, Wb)))
  GL_CHUNK(1, (mma_t<2, 2, 5>(acc, e, Wb)))
  GL_CHUNK(2, (mma_t<4, 1, 5>(acc, e, Wb)))
  load_c();
  act(0);
  GL_CHUNK(3, (mma_t<0, 2, 2>(acc, c, Wb)))
  // layer 1
  next_layer();
  GL_CHUNK(4, (mma_t<0, 2, 8>(acc, h, Wb)))
  GL_CHUNK(5, (mma_t<2, 2, 8>(acc, h, Wb)))
  GL_CHUNK(6, (mma_t<4, 2, 8>(acc, h, Wb)))
  GL_CHUNK(7, (mma_t<6, 2, 8>(acc, h, Wb)))
  load_c();
  act(1);
  GL_CHUNK(8, (mma_t<0, 2, 2>(acc, c, Wb)))
  // layer 2
  next_layer();
  GL_CHUNK(9, (mma_t<0, 2, 8>(acc, h, Wb)))
  GL_CHUNK(10, (mma_t<2, 2, 8>(acc, h, Wb)))
  GL_CHUNK(11, (mma_t<4, 2, 8>(acc, h, Wb)))
  GL_CHUNK(12, (mma_t<6, 2, 8>(acc, h, Wb)))
  load_c();
  act(2);
  GL_CHUNK(13, (mma_t<0, 2, 2>(acc, c, Wb)))
  // layer 3 (skip): W3e on the embedding, W3h on the hidden state
  next_layer();
  GL_CHUNK(14, (mma_t<0, 2, 5>(acc, e, Wb)))
  GL_CHUNK(15, (mma_t<2, 2, 5>(acc, e, Wb)))
  GL_CHUNK(16, (mma_t<4, 1, 5>(acc, e, Wb)))
  GL_CHUNK(17, (mma_t<0, 2, 8>(acc, h, Wb)))
  GL_CHUNK(18, (mma_t<2, 2, 8>(acc, h, Wb)))
  GL_CHUNK(19, (mma_t<4, 2, 8>(acc, h, Wb)))
  GL_CHUNK(20, (mma_t<6, 2, 8>(acc, h, Wb)))
  load_c();
  act(3);
  GL_CHUNK(21, (mma_t<0, 2, 2>(acc, c, Wb)))
  // layer 4
  next_layer();
  GL_CHUNK(22, (mma_t<0, 2, 8>(acc, h, Wb)))
  GL_CHUNK(23, (mma_t<2, 2, 8>(acc, h, Wb)))
  GL_CHUNK(24, (mma_t<4, 2, 8>(acc, h, Wb)))
  GL_CHUNK(25, (mma_t<6, 2, 8>(acc, h, Wb)))
  load_c();
  act(4);
  GL_CHUNK(26, (mma_t<0, 2, 2>(acc, c, Wb)))
#undef GL_CHUNK
  // output layer 128 -> 3 (A straight from global: Wout is [128][16], 3 real columns); lanes with g == 0
  // end up with channels 0..3 of their sample
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* wp = P.Wout + (4 * g) * 16 + r;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 16], acc[t][rr], o, 0, 0, 0);
  }
  if (g == 0 && qs < Q) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      raw[(size_t)qs * 4 + ch] = 1.0f / (1.0f + __expf(-(o[ch] + P.bout[ch])));
  }
}

// ---- colour decoder on the fp16 matrix cores with fp32 accuracy (3-term split) -------------------------------------
// The fp32 MFMA (16x16x4, 64 flops per cycle and SIMD) is the slowest matrix path of the chip; the fp16 one (16x16x32) is
// 16x faster.  Every weight and activation is split as x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 mantissa bits) and
// a product is accumulated as hi*hi + hi*lo + lo*hi in fp32 - the dropped lo*lo term is 2^-22 relative, the fp16 products
// are exact in the fp32 accumulator - so a 32-row chunk costs 3 x 8 MFMAs of 16 cycles instead of 64 of 32.  The
// transposed formulation carries over: the accumulator blocks (2t, 2t+1) of a layer, split and packed, are the eight
// k-slots of a lane's B operand for chunk t of the next layer (slot s <-> chunk row 16 (s >> 2) + 4 g + (s & 3), the same
// row order as the fp32 chunks); the weights are packed once per chunk as A fragments
// [hi|lo][out block 8][lane 64][8 halfs] (point_ops.pack_decoders) and streamed through LDS as they are (linear copy,
// conflict-free ds_read_b128).  Softplus, biases and the 128 -> 3 output layer stay fp32 as in mlp_col_v3_kernel.
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
constexpr int kChunkFloats16 = 4096;          // 2 x 8 x 64 x 8 halfs

__device__ __forceinline__ ChunkRegs chunk_load16(const float* __restrict__ W16, int chunk) {
  const float4* src = reinterpret_cast<const float4*>(W16 + (size_t)chunk * kChunkFloats16 + threadIdx.x * 8);
  ChunkRegs r;
  r.a = src[0];
  r.b = src[1];
  return r;
}
__device__ __forceinline__ void chunk_store16(float* Wb, const ChunkRegs& r) {
  float4* dst = reinterpret_cast<float4*>(Wb + threadIdx.x * 8);
  dst[0] = r.a;
  dst[1] = r.b;
}
// split two accumulator blocks (channels 4g + rr of block 0 -> slots 0..3, of block 1 -> slots 4..7) into hi / lo halfs
__device__ __forceinline__ void split2(const f32x4 a, const f32x4 b, h16x8& hi, h16x8& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    hi[s] = (_Float16)v[s];
#ifdef EXP_MLP_NO_SPLIT
    lo[s] = (_Float16)0.0f;                                 // ablation: the cost of forming the low halves
#else
    lo[s] = (_Float16)(v[s] - (float)hi[s]);
#endif
  }
}
// Range guard of the split kernels.  The split holds |x| <= 65504 (fp16 max).  A larger operand becomes hi = inf,
// lo = x - inf = -inf, and its products a.hi * inf + a.hi * (-inf) are NaN in EVERY output channel it feeds; softplus, the
// (NaN-keeping) ReLU below, the biases, the next split and the fp32 output layers all pass a NaN on, so an overflow anywhere
// in a network surfaces as a non-finite value at its output.  The kernels therefore test only what they are about to store
// (a handful of values per lane - tracking max |x| at every split costs ~100 registers' worth of spills in the colour
// kernel) and raise the caller's flag; the host then repeats the call on the exact-fp32 MFMA kernels.  (Small magnitudes are
// safe: below 2^-3 the low half falls into fp16 subnormals and the split's error becomes ABSOLUTE, 2^-25 per operand.)
__device__ __forceinline__ bool not_finite(float x) { return !(__builtin_fabsf(x) <= 3.0e38f); }
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.0f ? 0.0f : x; }
__device__ __forceinline__ void report_range(bool bad, int* __restrict__ range_flag) {
  if (range_flag && __builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(range_flag, 1);
}


// ---- colour decoder, round 3: NB column blocks (16 NB samples) per wave ------------------------------------------------
// Round 2's colour kernel (mlp_col_v4_kernel, removed in round 4) gave a wave 16 samples: every A fragment (weights) read from LDS feeds 3 MFMAs, and the 442 KB weight
// stream is re-staged for every 128 samples - LDS reads (1 KB per 3 x 16 MFMA cycles and wave: 1024 LDS cycles against 768
// matrix cycles per chunk and CU) and the L2 -> LDS stream (2.1 GB per 614k-sample batch) bound it, not the matrix pipe
// (MfmaUtil 41 %).  Here a wave owns NB = 2 blocks of 16 samples: an A fragment feeds 6 MFMAs, a workgroup of WAVES waves
// covers 32 WAVES samples per staged chunk, and the second block's vector work (softplus, splits) overlaps the first one's
// MFMAs inside the same wave.  Same packed weights, same arithmetic per sample (bit-identical results to v4).
template <int THREADS>
__device__ __forceinline__ void chunk_copy16(const float* __restrict__ W16, int chunk, float4 (&regs)[4096 / 4 / THREADS]) {
  const float4* src = reinterpret_cast<const float4*>(W16 + (size_t)chunk * kChunkFloats16);
#pragma unroll
  for (int i = 0; i < 4096 / 4 / THREADS; ++i) regs[i] = src[threadIdx.x + i * THREADS];
}
template <int THREADS>
__device__ __forceinline__ void chunk_put16(float* Wb, const float4 (&regs)[4096 / 4 / THREADS]) {
  float4* dst = reinterpret_cast<float4*>(Wb);
#pragma unroll
  for (int i = 0; i < 4096 / 4 / THREADS; ++i) dst[threadIdx.x + i * THREADS] = regs[i];
}

template <int NB>
__device__ __forceinline__ void mma_h3n(f32x4 (&acc)[NB][8], const h16x8 (&bhi)[NB], const h16x8 (&blo)[NB], const float* Wb) {
  const int lane = threadIdx.x & 63;
  const h16x8* wp = reinterpret_cast<const h16x8*>(Wb) + lane;
#pragma unroll
  for (int to = 0; to < 8; ++to) {
    const h16x8 ahi = wp[to * 64], alo = wp[(8 + to) * 64];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      acc[nb][to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi[nb], acc[nb][to], 0, 0, 0);
      acc[nb][to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo[nb], acc[nb][to], 0, 0, 0);
      acc[nb][to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi[nb], acc[nb][to], 0, 0, 0);
    }
  }
}

// Round 6, measured and not kept: three waves per SIMD instead of two (236 -> 168 registers: the embedding formed where it is
// consumed, the colour feature re-read before each use, the weight chunk L2 -> LDS by DMA instead of through 16 VGPRs, layer 3
// with its hidden chunks first) - 0.858-0.860 ms for the three decoder kernels against 0.838-0.853 ms, frame 5.49 / 5.63 against
// 5.47 / 5.61 ms in interleaved runs (tools/time_mlp.py): like every occupancy change tried on these kernels, no gain.  (The
// first build also failed the fixture test - a wrong chunk order in the re-ordered layer - and was not debugged further.)
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64, NB == 1 ? 4 : 2)
void mlp_col_v5_kernel(ColParams P, const float* __restrict__ W16, const float* __restrict__ pts,
                       const float* __restrict__ views, const float* __restrict__ c_col, int Q, float* __restrict__ raw,
                       int* __restrict__ range_flag) {
  bool bad = false;
  constexpr int THREADS = WAVES * 64;
  extern __shared__ float smem[];
  float* Wbuf = smem;                         // [2][4096]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * (WAVES * 16 * NB);
  int qs[NB], q[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    qs[nb] = q0 + (wv * NB + nb) * 16 + r;    // this lane's sample of block nb (all four k-slots of a column share it)
    q[nb] = min(qs[nb], Q - 1);
  }
  h16x8 ehi[3][NB], elo[3][NB], chi[NB], clo[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const size_t qq = (size_t)q[nb];
    const float px = pts[qq * 3 + 0], py = pts[qq * 3 + 1], pz = pts[qq * 3 + 2];
    float vx = views[qq * 3 + 0], vy = views[qq * 3 + 1], vz = views[qq * 3 + 2];
    const float nrm = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
    vx = vx / nrm; vy = vy / nrm; vz = vz / nrm;
    f32x4 e[6];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int f = 16 * t + 4 * g + rr;   // feature index 0..79: [sin p | cos p | sin v | cos v] x 20
        const int blk = f / 20, ff = f - blk * 20;
        const float* Bm = blk < 2 ? P.Bp : P.Bv;
        const float x = blk < 2 ? px : vx, y = blk < 2 ? py : vy, z = blk < 2 ? pz : vz;
        const float a = fmaf(z, Bm[40 + ff], fmaf(y, Bm[20 + ff], x * Bm[ff]));
        e[t][rr] = (blk & 1) ? cos_rev(a) : sin_rev(a);
      }
    e[5] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) split2(e[2 * c], e[2 * c + 1], ehi[c][nb], elo[c][nb]);
    const float4 v0 = *reinterpret_cast<const float4*>(c_col + qq * 32 + 4 * g);
    const float4 v1 = *reinterpret_cast<const float4*>(c_col + qq * 32 + 16 + 4 * g);
    split2(f32x4{v0.x, v0.y, v0.z, v0.w}, f32x4{v1.x, v1.y, v1.z, v1.w}, chi[nb], clo[nb]);
  }

  f32x4 acc[NB][8];
  h16x8 hhi[4][NB], hlo[4][NB];
  constexpr int NC = 27;
  float4 nxt[4096 / 4 / THREADS];
  chunk_copy16<THREADS>(W16, 0, nxt);
  chunk_put16<THREADS>(Wbuf, nxt);
  __syncthreads();

  auto act = [&](int li) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(P.bias + li * 128 + 16 * t + 4 * g);
      const float4 fb = *reinterpret_cast<const float4*>(P.fcb + li * 128 + 16 * t + 4 * g);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb][t][0] = softplus100_fast(acc[nb][t][0] + bb.x) + fb.x;
        acc[nb][t][1] = softplus100_fast(acc[nb][t][1] + bb.y) + fb.y;
        acc[nb][t][2] = softplus100_fast(acc[nb][t][2] + bb.z) + fb.z;
        acc[nb][t][3] = softplus100_fast(acc[nb][t][3] + bb.w) + fb.w;
      }
    }
  };
  auto next_layer = [&]() {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) split2(acc[nb][2 * c], acc[nb][2 * c + 1], hhi[c][nb], hlo[c][nb]);
      zero<8>(acc[nb]);
    }
  };

#define GL_CHUNK(cidx, BHI, BLO)                                    \
  {                                                                 \
    if ((cidx) + 1 < NC) chunk_copy16<THREADS>(W16, (cidx) + 1, nxt); \
    mma_h3n<NB>(acc, BHI, BLO, Wbuf + ((cidx) & 1) * kChunkFloats16); \
    if ((cidx) + 1 < NC) chunk_put16<THREADS>(Wbuf + (((cidx) + 1) & 1) * kChunkFloats16, nxt); \
    __syncthreads();                                                \
  }

#pragma unroll
  for (int nb = 0; nb < NB; ++nb) zero<8>(acc[nb]);
  GL_CHUNK(0, ehi[0], elo[0])
  GL_CHUNK(1, ehi[1], elo[1])
  GL_CHUNK(2, ehi[2], elo[2])
  act(0);
  GL_CHUNK(3, chi, clo)
  next_layer();
  GL_CHUNK(4, hhi[0], hlo[0])
  GL_CHUNK(5, hhi[1], hlo[1])
  GL_CHUNK(6, hhi[2], hlo[2])
  GL_CHUNK(7, hhi[3], hlo[3])
  act(1);
  GL_CHUNK(8, chi, clo)
  next_layer();
  GL_CHUNK(9, hhi[0], hlo[0])
  GL_CHUNK(10, hhi[1], hlo[1])
  GL_CHUNK(11, hhi[2], hlo[2])
  GL_CHUNK(12, hhi[3], hlo[3])
  act(2);
  GL_CHUNK(13, chi, clo)
  next_layer();                               // layer 3 (skip): W3e on the embedding, W3h on the hidden state
  GL_CHUNK(14, ehi[0], elo[0])
  GL_CHUNK(15, ehi[1], elo[1])
  GL_CHUNK(16, ehi[2], elo[2])
  GL_CHUNK(17, hhi[0], hlo[0])
  GL_CHUNK(18, hhi[1], hlo[1])
  GL_CHUNK(19, hhi[2], hlo[2])
  GL_CHUNK(20, hhi[3], hlo[3])
  act(3);
  GL_CHUNK(21, chi, clo)
  next_layer();
  GL_CHUNK(22, hhi[0], hlo[0])
  GL_CHUNK(23, hhi[1], hlo[1])
  GL_CHUNK(24, hhi[2], hlo[2])
  GL_CHUNK(25, hhi[3], hlo[3])
  act(4);
  GL_CHUNK(26, chi, clo)
#undef GL_CHUNK
  // output layer 128 -> 3 in fp32 as in mlp_col_v3_kernel
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wp = P.Wout + (4 * g) * 16 + r;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 16], acc[nb][t][rr], o, 0, 0, 0);
    if (g == 0 && qs[nb] < Q) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        bad = bad | not_finite(o[ch]);
        raw[(size_t)qs[nb] * 4 + ch] = 1.0f / (1.0f + __expf(-(o[ch] + P.bout[ch])));
      }
    }
  }
  report_range(bad, range_flag);
}

// per-neighbour F_theta, transposed formulation (see mlp_col_v3_kernel): W1 (52 x 128) resident in LDS with
// the permuted column order, the 52 input channels of neighbour k built directly as B fragments - lane
// (r, g) owns sample r and supplies, for k-slot g, colour-feature channels 16t + 4g + rr (two 16-byte loads
// of the neighbour's feature row) and embedding features 4s + g (s = 0..4; sin for feature < 10, cos
// otherwise).  No staging buffer and no barrier inside the neighbour loop; the weighted sum of the eight
// softplus outputs stays in the accumulator layout, which is again the B layout of the second layer.
__global__ __launch_bounds__(512, 4) void mlp_nb_v3_kernel(NbParams P, const float* __restrict__ pts,
                                                           const float* __restrict__ cloud,
                                                           const float* __restrict__ col_feats,
                                                           const int64_t* __restrict__ I,
                                                           const float* __restrict__ wts,
                                                           const uint8_t* __restrict__ has, int Q,
                                                           float* __restrict__ c_col) {
  extern __shared__ float smem[];
  float* W1s = smem;                          // [52][132], columns permuted like the colour chunks
  float* b1s = W1s + 52 * kLdw3;              // [128]
  float* wbuf = b1s + 128;                    // [128][8] IDW weights (0 for an absent neighbour)
  int* ibuf = reinterpret_cast<int*>(wbuf + kTM2 * 8);  // [128][8] neighbour ids (0 for an absent one)
  float* bs = reinterpret_cast<float*>(ibuf + kTM2 * 8);  // [20][4] B[:, f mod 10] (revolutions per metre)
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * kTM2;
  if (tid < 80) {
    const int f = tid >> 2, d = tid & 3;
    bs[tid] = d < 3 ? P.B[d * 10 + (f < 10 ? f : f - 10)] : 0.0f;
  }
  for (int idx = tid; idx < 52 * 128; idx += 512) {
    const int col = idx & 127, to = col >> 4, rc = col & 15;
    W1s[(idx >> 7) * kLdw3 + (to >> 2) * 64 + rc * 4 + (to & 3)] = P.W1[idx];
  }
  if (tid < 128) b1s[tid] = P.b1[tid];
  for (int idx = tid; idx < kTM2 * 8; idx += 512) {
    const int row = idx >> 3;
    const int q = min(q0 + row, Q - 1);
    const int ii = (int)I[(size_t)q * 8 + (idx & 7)];
    wbuf[idx] = (q0 + row < Q && ii >= 0) ? wts[(size_t)q * 8 + (idx & 7)] : 0.0f;
    ibuf[idx] = ii < 0 ? 0 : ii;
  }
  __syncthreads();
  const int srow = wv * 16 + r;               // this lane's sample row inside the workgroup
  const int qs = q0 + srow;
  const int q = min(qs, Q - 1);
  const float qx = pts[(size_t)q * 3 + 0], qy = pts[(size_t)q * 3 + 1], qz = pts[(size_t)q * 3 + 2];
  // embedding features of this lane: f = 4s + g; frequency column (f mod 10), sin for f < 10.  The
  // frequency vectors sit in LDS as [20][4] (15 registers short otherwise); phases in revolutions.
  const float* bsl = bs + 4 * g;
  f32x4 ysum[8];
  zero<8>(ysum);
  float sw = 0.0f;
  const float* wpe = W1s + g * kLdw3 + r * 4;               // embedding rows 4s + g
  const float* wpc = W1s + (20 + 4 * g) * kLdw3 + r * 4;    // colour-feature rows 20 + 16t + 4g + rr
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    const int pt = ibuf[srow * 8 + k];
    const float w = wbuf[srow * 8 + k];
    const float4 c0 = *reinterpret_cast<const float4*>(col_feats + (size_t)pt * 32 + 4 * g);
    const float4 c1 = *reinterpret_cast<const float4*>(col_feats + (size_t)pt * 32 + 16 + 4 * g);
    const float rx = cloud[(size_t)pt * 3 + 0] - qx, ry = cloud[(size_t)pt * 3 + 1] - qy,
                rz = cloud[(size_t)pt * 3 + 2] - qz;
    f32x4 acc[8];
    zero<8>(acc);
    // 13 k-steps, software pipelined by hand: the two 16-byte weight reads of step i + 1 are issued before
    // the eight MFMAs of step i; the scheduling barrier keeps the compiler from hoisting all 26 reads
    // (104 registers) to the top.  Embedding steps first: they only need the neighbour position, the
    // colour-feature loads (a full L2 round trip behind the index) complete meanwhile.
#define NB_LOAD(WP) w0n = *reinterpret_cast<const float4*>(WP); w1n = *reinterpret_cast<const float4*>((WP) + 64);
#define NB_STEP(BV, NEXT)                                                                \
    {                                                                                    \
      const float4 w0 = w0n, w1 = w1n;                                                   \
      const float bv = (BV);                                                             \
      NEXT                                                                               \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, bv, acc[0], 0, 0, 0);          \
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, bv, acc[1], 0, 0, 0);          \
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, bv, acc[2], 0, 0, 0);          \
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, bv, acc[3], 0, 0, 0);          \
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, bv, acc[4], 0, 0, 0);          \
      acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, bv, acc[5], 0, 0, 0);          \
      acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, bv, acc[6], 0, 0, 0);          \
      acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, bv, acc[7], 0, 0, 0);          \
      __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    float4 w0n, w1n;
    NB_LOAD(wpe)
#pragma unroll
    for (int sidx = 0; sidx < 5; ++sidx) {
      const float4 bf = *reinterpret_cast<const float4*>(bsl + 16 * sidx);
      const float a = fmaf(rz, bf.z, fmaf(ry, bf.y, rx * bf.x));
      const float ev = (4 * sidx + g >= 10) ? cos_rev(a) : sin_rev(a);
      if (sidx < 4) NB_STEP(ev, NB_LOAD(wpe + 4 * (sidx + 1) * kLdw3))
      else NB_STEP(ev, NB_LOAD(wpc))
    }
    NB_STEP(c0.x, NB_LOAD(wpc + 1 * kLdw3))
    NB_STEP(c0.y, NB_LOAD(wpc + 2 * kLdw3))
    NB_STEP(c0.z, NB_LOAD(wpc + 3 * kLdw3))
    NB_STEP(c0.w, NB_LOAD(wpc + 16 * kLdw3))
    NB_STEP(c1.x, NB_LOAD(wpc + 17 * kLdw3))
    NB_STEP(c1.y, NB_LOAD(wpc + 18 * kLdw3))
    NB_STEP(c1.z, NB_LOAD(wpc + 19 * kLdw3))
    NB_STEP(c1.w, )
#undef NB_STEP
#undef NB_LOAD
    sw += w;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(b1s + 16 * t + 4 * g);
      ysum[t][0] += w * softplus100_fast(acc[t][0] + bb.x);
      ysum[t][1] += w * softplus100_fast(acc[t][1] + bb.y);
      ysum[t][2] += w * softplus100_fast(acc[t][2] + bb.z);
      ysum[t][3] += w * softplus100_fast(acc[t][3] + bb.w);
    }
  }
  // second layer 128 -> 32 (A = W2 straight from global / L2, 16 KB shared by everybody)
  f32x4 o[2];
  zero<2>(o);
  {
    const float* wp = P.W2 + (4 * g) * 32 + r;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 32], ysum[t][rr], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 32 + 16], ysum[t][rr], o[1], 0, 0, 0);
      }
  }
  if (qs < Q) {
    const bool h = has[qs] != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 b2 = *reinterpret_cast<const float4*>(P.b2 + 16 * t + 4 * g);
      float4 v;
      v.x = h ? o[t][0] + b2.x * sw : 0.0f;
      v.y = h ? o[t][1] + b2.y * sw : 0.0f;
      v.z = h ? o[t][2] + b2.z * sw : 0.0f;
      v.w = h ? o[t][3] + b2.w * sw : 0.0f;
      *reinterpret_cast<float4*>(c_col + (size_t)qs * 32 + 16 * t + 4 * g) = v;
    }
  }
}

// geometry decoder on the fp16 matrix cores with the 3-term split (see mlp_col_v5_kernel): the 480 K-rows are 15 chunks
// of 32, packed by point_ops.pack_decoders as A fragments [chunk][hi|lo][out block 2][lane 64][8 halfs] (1024 floats per
// chunk) and staged once per workgroup; the 32 -> 1 output layer stays fp32.  The 96-channel embedding is evaluated where
// it is consumed (layer 0 and again at the skip layer) instead of being kept: 24 v_sin per sample against 24 registers
// that the 128-register budget of four waves per SIMD does not have.
constexpr int kGeoFrag = 15 * 1024;            // floats

__device__ __forceinline__ void mma_g3(f32x4 (&acc)[2], const h16x8 bhi, const h16x8 blo, const float* chunk) {
  const h16x8* wp = reinterpret_cast<const h16x8*>(chunk) + (threadIdx.x & 63);
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    const h16x8 ahi = wp[to * 64], alo = wp[(2 + to) * 64];
    acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[to], 0, 0, 0);
    acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, acc[to], 0, 0, 0);
    acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, acc[to], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// GO (gather only): the measurement instantiation behind stage_flags bit 2 of glorie_render_mlp - the kernel's R2 phase
// alone (ids, weights, the 8 feature rows and their interpolation) with the decoder removed, so that bench.py can time the
// feature pull AS THE PRODUCT PERFORMS IT (inside this kernel, not in a stand-alone gather launch) in its own process.
template <bool GO>
__global__ __launch_bounds__(512, 4) void mlp_geo_v4_kernel(GeoParams P, const float* __restrict__ frags,
                                                            const float* __restrict__ wout,
                                                            const float* __restrict__ pts,
                                                            const float* __restrict__ c_geo,
                                                            const float* __restrict__ geo_feats,
                                                            const int64_t* __restrict__ I,
                                                            const float* __restrict__ wts,
                                                            const uint8_t* __restrict__ has, int Q,
                                                            float* __restrict__ raw, int* __restrict__ range_flag) {
  bool bad = false;
  extern __shared__ float smem[];              // [15][1024] fragments | [32][16] output layer
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  if constexpr (!GO) {                        // (the weights are no part of the feature pull)
#pragma unroll 2
    for (int idx = tid; idx < kGeoFrag / 4; idx += 512)
      reinterpret_cast<float4*>(smem)[idx] = reinterpret_cast<const float4*>(frags)[idx];
    if (tid < 128) reinterpret_cast<float4*>(smem + kGeoFrag)[tid] = reinterpret_cast<const float4*>(wout)[tid];
  }
  __syncthreads();
  // persistent over the sample blocks: the 61 KB of fragments are staged once per workgroup (a launch used to stage them
  // for every 128 samples, 4800 times per 614k-sample batch, with the first MFMA of a workgroup waiting behind it)
  const int nblk = (Q + kTM2 - 1) / kTM2;
  // per-sample inputs of a block: position, neighbour ids, IDW weights.  They are requested one block AHEAD, so a block's
  // feature gather (which needs the ids) starts at once instead of behind a second dependent memory round trip
  struct Meta { float x, y, z; int id[8]; float wk[8]; bool on; };
  auto load_meta = [&](int blk, Meta& m) {
    const int qq = min(blk * kTM2 + wv * 16 + r, Q - 1);
    m.x = pts[(size_t)qq * 3 + 0]; m.y = pts[(size_t)qq * 3 + 1]; m.z = pts[(size_t)qq * 3 + 2];
    m.on = has[qq] != 0;
    if (geo_feats) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        m.id[k] = (int)max(I[(size_t)qq * 8 + k], 0L);
        m.wk[k] = wts[(size_t)qq * 8 + k];
      }
    }
  };
  Meta cur;
  if ((int)blockIdx.x < nblk) load_meta(blockIdx.x, cur);
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  const int qs = blk * kTM2 + wv * 16 + r;
  const int q = min(qs, Q - 1);
  h16x8 chi, clo, hhi, hlo;
  f32x4 acc[2];
  const float x = cur.x, y = cur.y, z = cur.z;
  const bool on = cur.on;
  Meta nxt = cur;
  {
    f32x4 c[2];
    if (geo_feats) {
      // IDW interpolation of the neighbours' features (decoder.py:130-173), as in mlp_geo_v3_kernel
      c[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      c[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int GEO_NB = GLORIE_GEO_NB;
#pragma unroll
      for (int k0 = 0; k0 < 8; k0 += GEO_NB) {
        float4 v[GEO_NB][2];
#pragma unroll
        for (int k = 0; k < GEO_NB; ++k) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
            v[k][t] = *reinterpret_cast<const float4*>(geo_feats + (size_t)cur.id[k0 + k] * 32 + 16 * t + 4 * g);
        }
        if (k0 == 0 && blk + (int)gridDim.x < nblk) load_meta(blk + gridDim.x, nxt);   // behind the first gathers
#pragma unroll
        for (int k = 0; k < GEO_NB; ++k) {
          const float wk = cur.wk[k0 + k];
          const bool use = on && wk != 0.0f;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            c[t][0] = use ? c[t][0] + wk * v[k][t].x : c[t][0];
            c[t][1] = use ? c[t][1] + wk * v[k][t].y : c[t][1];
            c[t][2] = use ? c[t][2] + wk * v[k][t].z : c[t][2];
            c[t][3] = use ? c[t][3] + wk * v[k][t].w : c[t][3];
          }
        }
      }
    } else {
      if (blk + (int)gridDim.x < nblk) load_meta(blk + gridDim.x, nxt);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(c_geo + (size_t)q * 32 + 16 * t + 4 * g);
        c[t][0] = v.x; c[t][1] = v.y; c[t][2] = v.z; c[t][3] = v.w;
      }
    }
    split2(c[0], c[1], chi, clo);
    if constexpr (GO) {
      // the interpolated feature's sum stands in for the occupancy
      if (g == 0 && qs < Q) raw[(size_t)qs * 4 + 3] = (c[0][0] + c[0][1]) + (c[1][2] + c[1][3]);
      cur = nxt;
      continue;
    }
  }
  auto act = [&](int li) {   // ReLU(acc + bias) + fc_c bias
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(P.bias + li * 32 + 16 * t + 4 * g);
      const float4 fb = *reinterpret_cast<const float4*>(P.fcb + li * 32 + 16 * t + 4 * g);
      acc[t][0] = relu_keep_nan(acc[t][0] + bb.x) + fb.x;
      acc[t][1] = relu_keep_nan(acc[t][1] + bb.y) + fb.y;
      acc[t][2] = relu_keep_nan(acc[t][2] + bb.z) + fb.z;
      acc[t][3] = relu_keep_nan(acc[t][3] + bb.w) + fb.w;
    }
  };
  auto next_layer = [&]() {
    split2(acc[0], acc[1], hhi, hlo);
    zero<2>(acc);
  };
  auto ch = [&](int c) { return smem + c * 1024; };
  // embedding blocks (2cc, 2cc + 1) -> chunk `chunk`: evaluated, split, multiplied, forgotten
  auto embed = [&](int cc, int chunk) {
    f32x4 e[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * cc + u;
      const float4 b0 = *reinterpret_cast<const float4*>(P.B + 16 * t + 4 * g);
      const float4 b1 = *reinterpret_cast<const float4*>(P.B + 96 + 16 * t + 4 * g);
      const float4 b2 = *reinterpret_cast<const float4*>(P.B + 192 + 16 * t + 4 * g);
      e[u][0] = sin_rev(fmaf(z, b2.x, fmaf(y, b1.x, x * b0.x)));
      e[u][1] = sin_rev(fmaf(z, b2.y, fmaf(y, b1.y, x * b0.y)));
      e[u][2] = sin_rev(fmaf(z, b2.z, fmaf(y, b1.z, x * b0.z)));
      e[u][3] = sin_rev(fmaf(z, b2.w, fmaf(y, b1.w, x * b0.w)));
    }
    h16x8 ehi, elo;
    split2(e[0], e[1], ehi, elo);
    mma_g3(acc, ehi, elo, ch(chunk));
  };
  // chunks: W0 0-2 | Fc0 3 | W1 4 | Fc1 5 | W2 6 | Fc2 7 | W3e 8-10 | W3h 11 | Fc3 12 | W4 13 | Fc4 14
  zero<2>(acc);
  embed(0, 0); embed(1, 1); embed(2, 2);
  act(0); mma_g3(acc, chi, clo, ch(3));
  next_layer();
  mma_g3(acc, hhi, hlo, ch(4));  act(1); mma_g3(acc, chi, clo, ch(5));
  next_layer();
  mma_g3(acc, hhi, hlo, ch(6));  act(2); mma_g3(acc, chi, clo, ch(7));
  next_layer();
  embed(0, 8); embed(1, 9); embed(2, 10);
  mma_g3(acc, hhi, hlo, ch(11)); act(3); mma_g3(acc, chi, clo, ch(12));
  next_layer();
  mma_g3(acc, hhi, hlo, ch(13)); act(4); mma_g3(acc, chi, clo, ch(14));
  // output layer 32 -> 1 (16 padded columns) in fp32: lanes with g == 0 get channel 0 of their sample in o[0]
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* wp = smem + kGeoFrag + (4 * g) * 16 + r;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 16], acc[t][rr], o, 0, 0, 0);
  }
  if (g == 0 && qs < Q) {
    bad = bad | (on && not_finite(o[0]));
    raw[(size_t)qs * 4 + 3] = on ? o[0] + P.bout[0] : -100.0f;  // Renderer.py:206-207
  }
  cur = nxt;
  }
  report_range(bad, range_flag);
}

// per-neighbour F_theta on the fp16 matrix cores with the 3-term split (see mlp_col_v5_kernel).  The 52 input channels of a
// neighbour are two 32-slot chunks: chunk 0 = the 20 embedding features (lane slot s < 5 <-> feature 4s + g, slots 5..7
// zero), chunk 1 = the 32 colour-feature channels (slot s <-> channel 16 (s >> 2) + 4 g + (s & 3), the two 16-byte loads of
// the feature row).  W1 is packed split, as A fragments [chunk][hi|lo][out block][lane][8] (point_ops.pack_decoders), and copied
// to LDS once per workgroup.
template <bool GO>
__global__ __launch_bounds__(512, 4) void mlp_nb_v4_kernel(NbParams P, const float* __restrict__ W1frag,
                                                           const float* __restrict__ pts,
                                                           const float* __restrict__ cloud,
                                                           const float* __restrict__ col_feats,
                                                           const int64_t* __restrict__ I,
                                                           const float* __restrict__ wts,
                                                           const uint8_t* __restrict__ has, int Q,
                                                           float* __restrict__ c_col, int* __restrict__ range_flag) {
  bool bad = false;
  extern __shared__ float smem[];
  h16x8* W1f = reinterpret_cast<h16x8*>(smem);   // [2 chunks][2 hi|lo][8][64] fragments of 8 halfs = 32 KB
  float* b1s = smem + 8192;                   // [128]
  float* wbuf = b1s + 128;                    // [128][8] IDW weights (0 for an absent neighbour)
  int* ibuf = reinterpret_cast<int*>(wbuf + kTM2 * 8);  // [128][8] neighbour ids (0 for an absent one)
  float* bs = reinterpret_cast<float*>(ibuf + kTM2 * 8);  // [20][4] B[:, f mod 10] (revolutions per metre)
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * kTM2;
  if (tid < 80) {
    const int f = tid >> 2, d = tid & 3;
    bs[tid] = d < 3 ? P.B[d * 10 + (f < 10 ? f : f - 10)] : 0.0f;
  }
  if constexpr (!GO) {                                        // (the weights are no part of the feature pull)
    for (int idx = tid; idx < 8192 / 4; idx += 512)           // the split fragments as packed (point_ops.pack_decoders)
      reinterpret_cast<float4*>(smem)[idx] = reinterpret_cast<const float4*>(W1frag)[idx];
    if (tid < 128) b1s[tid] = P.b1[tid];
  }
  for (int idx = tid; idx < kTM2 * 8; idx += 512) {
    const int row = idx >> 3;
    const int q = min(q0 + row, Q - 1);
    const int ii = (int)I[(size_t)q * 8 + (idx & 7)];
    wbuf[idx] = (q0 + row < Q && ii >= 0) ? wts[(size_t)q * 8 + (idx & 7)] : 0.0f;
    ibuf[idx] = ii < 0 ? 0 : ii;
  }
  __syncthreads();
  const int srow = wv * 16 + r;               // this lane's sample row inside the workgroup
  const int qs = q0 + srow;
  const int q = min(qs, Q - 1);
  const float qx = pts[(size_t)q * 3 + 0], qy = pts[(size_t)q * 3 + 1], qz = pts[(size_t)q * 3 + 2];
  const float* bsl = bs + 4 * g;
  f32x4 ysum[8];
  zero<8>(ysum);
  float sw = 0.0f;
  const h16x8* wl = W1f + lane;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
#ifdef EXP_NB_NO_GATHER
    const int pt = (q0 + srow) & 0xffff;                    // ablation: consecutive rows instead of the neighbours'
#else
    const int pt = ibuf[srow * 8 + k];
#endif
    const float w = wbuf[srow * 8 + k];
    const float4 c0 = *reinterpret_cast<const float4*>(col_feats + (size_t)pt * 32 + 4 * g);
    const float4 c1 = *reinterpret_cast<const float4*>(col_feats + (size_t)pt * 32 + 16 + 4 * g);
    const float rx = cloud[(size_t)pt * 3 + 0] - qx, ry = cloud[(size_t)pt * 3 + 1] - qy,
                rz = cloud[(size_t)pt * 3 + 2] - qz;
    if constexpr (GO) {
      // measurement instantiation: the neighbour's feature row and position, weighted - no embedding, no layers
      ysum[0][0] += w * (c0.x + c1.x + rx); ysum[0][1] += w * (c0.y + c1.y + ry);
      ysum[0][2] += w * (c0.z + c1.z + rz); ysum[0][3] += w * (c0.w + c1.w);
      sw += w;
      continue;
    }
    f32x4 acc[8];
    zero<8>(acc);
    // chunk 0: embedding features 4s + g (sin for feature < 10), phases in revolutions
    {
      float ev[8];
#pragma unroll
      for (int sidx = 0; sidx < 5; ++sidx) {
        const float4 bf = *reinterpret_cast<const float4*>(bsl + 16 * sidx);
        const float a = fmaf(rz, bf.z, fmaf(ry, bf.y, rx * bf.x));
#ifdef EXP_NB_NO_SINCOS
        ev[sidx] = a;
#else
        ev[sidx] = (4 * sidx + g >= 10) ? cos_rev(a) : sin_rev(a);
#endif
      }
      ev[5] = ev[6] = ev[7] = 0.0f;
      h16x8 bhi, blo;
      split2(f32x4{ev[0], ev[1], ev[2], ev[3]}, f32x4{ev[4], ev[5], ev[6], ev[7]}, bhi, blo);
#pragma unroll
      for (int to = 0; to < 8; ++to) {
#ifdef EXP_NB_NO_LDSW
        const h16x8 ahi = bhi, alo = blo;                     // ablation: no fragment reads
#else
        const h16x8 ahi = wl[to * 64], alo = wl[(8 + to) * 64];
#endif
#ifdef EXP_NB_NO_MFMA
        acc[to][0] += (float)ahi[0] * (float)bhi[to & 7] + (float)alo[1] * (float)blo[to & 7];      // ablation: no matrix work
#else
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, acc[to], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);     // keeps the compiler from hoisting all 32 fragment reads (128 registers)
      }
    }
    // chunk 1: the neighbour's colour feature (a full L2 round trip behind the index: issued above, used here)
    {
      h16x8 bhi, blo;
      split2(f32x4{c0.x, c0.y, c0.z, c0.w}, f32x4{c1.x, c1.y, c1.z, c1.w}, bhi, blo);
#pragma unroll
      for (int to = 0; to < 8; ++to) {
#ifdef EXP_NB_NO_LDSW
        const h16x8 ahi = bhi, alo = blo;
#else
        const h16x8 ahi = wl[(16 + to) * 64], alo = wl[(24 + to) * 64];
#endif
#ifdef EXP_NB_NO_MFMA
        acc[to][1] += (float)ahi[0] * (float)bhi[to & 7] + (float)alo[1] * (float)blo[to & 7];
#else
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, acc[to], 0, 0, 0);
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, acc[to], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);     // keeps the compiler from hoisting all 32 fragment reads (128 registers)
      }
    }
    sw += w;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(b1s + 16 * t + 4 * g);
      ysum[t][0] += w * softplus100_fast(acc[t][0] + bb.x);
      ysum[t][1] += w * softplus100_fast(acc[t][1] + bb.y);
      ysum[t][2] += w * softplus100_fast(acc[t][2] + bb.z);
      ysum[t][3] += w * softplus100_fast(acc[t][3] + bb.w);
    }
  }
  // second layer 128 -> 32 in fp32 as in mlp_nb_v3_kernel (A = W2 straight from global / L2)
  f32x4 o[2];
  zero<2>(o);
  {
    const float* wp = P.W2 + (4 * g) * 32 + r;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 32], ysum[t][rr], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * t + rr) * 32 + 16], ysum[t][rr], o[1], 0, 0, 0);
      }
  }
  if (qs < Q) {
    const bool h = has[qs] != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 b2 = *reinterpret_cast<const float4*>(P.b2 + 16 * t + 4 * g);
      float4 v;
      v.x = h ? o[t][0] + b2.x * sw : 0.0f;
      v.y = h ? o[t][1] + b2.y * sw : 0.0f;
      v.z = h ? o[t][2] + b2.z * sw : 0.0f;
      v.w = h ? o[t][3] + b2.w * sw : 0.0f;
      bad = bad | not_finite(v.x + v.y + v.z + v.w);
      *reinterpret_cast<float4*>(c_col + (size_t)qs * 32 + 16 * t + 4 * g) = v;
    }
  }
  report_range(bad, range_flag);
}

// Round 3, measured and not kept for the per-neighbour kernel (614,400 samples, v4 = 394 us): two column blocks per wave in
// lockstep 415-463 us; two accumulator sets with the MFMAs of neighbour k+1 issued between the softplus instructions of
// neighbour k (software pipeline inside the wave) 495 us at 2 waves per SIMD, 624 us at 3 (spills) - four resident waves per
// SIMD alternating their phases already overlap the matrix pipe with the vector ALU better than one wave can by itself;
// the kernel sits on its vector floor (8 x 128 softplus per sample, two quarter-rate transcendentals each: ~340 us).  A
// degree-6 polynomial for log2(1 + e) on packed fp32 FMAs instead of v_log_f32 was slower too (405 us: v_pk_fma_f32 beside
// MFMAs is no faster than two v_fma_f32, MI355X_MICROARCH.md), as was the colour kernel with it (450 vs 427 us).
//
// Round 4: what the kernel's 397 us are made of (tools/exp_mlp_ablate.sh, builds with -DEXP_NB_* / -DEXP_MLP_*; every line is
// the kernel with ONE thing removed): fragment reads from LDS 268 (-129), MFMAs 303 (-94), softplus 304 (-91), neighbour
// gather made sequential 366 (-31), low halves of the split 375 (-20), sin / cos 394; without MFMAs, fragment reads and
// softplus 170.  The transcendentals are NOT the floor the round-3 note took them for (v_exp_f32 / v_log_f32 issue at
// ~5/3 of a plain VALU slot on this part, MI355X_MICROARCH.md; a packed-fp16 polynomial for the correction term needs
// degree 8 on [0, 8] and its monomial coefficients ~8^k / k! cancel catastrophically in fp16) - the A operand is: the same
// 32 KB of W1 fragments are read from LDS in front of every MFMA, 1 KB per read, 1024 clocks of the CU's one LDS pipe per
// 816 clocks of MFMA on its four SIMDs.  Keeping the 16 high fragments in 64 VGPRs (a third of the reads, 198 VGPRs, 2
// waves per SIMD, a wave walking 16-sample blocks with the next neighbour's row prefetched; bit-identical) measured 432 us:
// what the reads save, the lost occupancy costs (4 -> 2 waves per SIMD hide the gather and interleave MFMA with VALU
// worse).  Not kept.  The untried middle followed at the end of the round: two 16-sample blocks per wave with the 128 output
// features walked in two halves (2 x 4 accumulator tiles + 2 x 8 sum tiles: 162 VGPRs, no spill, 3 waves per SIMD, 4-wave
// workgroups; every fragment read feeds six MFMAs; bit-identical) - and a 640x480 frame takes 5.42-5.51 ms with it against
// 5.42-5.56 ms without, the single-batch 1/8 share 0.894-0.902 against 0.891-0.918 ms (interleaved builds, same box): halving
// the fragment reads buys exactly what the fourth wave per SIMD was worth.  Not kept either.
}  // namespace glorie

using namespace glorie;

// offsets (in floats) of every section inside the packed parameter buffer; the layout is
// produced by glorie_slam_amd.point_ops.pack_decoders and mirrored here.
namespace {
struct Cursor {
  const float* p;
  const float* take(size_t n) { const float* r = p; p += n; return r; }
};
}  // namespace

extern "C" size_t glorie_decoder_pack_floats(void) {
  size_t geo = 3 * 96 + 96 * 32 + 32 * 32 * 2 + 96 * 32 + 32 * 32 * 2 + 32 * 16 + 5 * 32 * 32 + 5 * 32 * 2 + 4;
  size_t nb = 3 * 10 + 2 + 52 * 128 + 128 + 128 * 32 + 32;
  size_t col = 3 * 20 * 2 + 80 * 128 + 128 * 128 * 2 + 80 * 128 + 128 * 128 * 2 + 128 * 16 + 5 * 32 * 128 +
               5 * 128 * 2 + 4;
  return geo + nb + col + (size_t)27 * 32 * 128 + (size_t)(480 * 32 + 32 * 16) + (size_t)27 * kChunkFloats16 + (size_t)8192 + (size_t)kGeoFrag;
}

extern "C" int glorie_render_mlp(const float* packed, const float* pts, const float* views,
                                 const float* cloud_pos, const float* col_feats,
                                 const float* c_geo, const float* geo_feats, const int64_t* I,
                                 const float* weights, const uint8_t* has, int Q, float* c_col_scratch,
                                 float* raw, int stage_flags, int* range_flag, void* stream) {
  const int stage_color = stage_flags & 1;
  const bool force_f32 = (stage_flags & 2) != 0;
  const bool gather_only = (stage_flags & 4) != 0;        // measurement: the R2 phases of the geometry / per-neighbour kernels alone
  if (Q < 0) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!packed || !pts || !has || !raw) return GLORIE_EINVAL;
  if (!c_geo && !(geo_feats && I && weights)) return GLORIE_EINVAL;   // interpolated feature, or what it takes
  if (stage_color && (!views || !cloud_pos || !col_feats || !I || !weights || !c_col_scratch))
    return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Cursor c{packed};
  GeoParams g;
  g.B = c.take(3 * 96); g.W0 = c.take(96 * 32); g.W1 = c.take(32 * 32); g.W2 = c.take(32 * 32);
  g.W3e = c.take(96 * 32); g.W3h = c.take(32 * 32); g.W4 = c.take(32 * 32); g.Wout = c.take(32 * 16);
  g.Fc = c.take(5 * 32 * 32); g.bias = c.take(5 * 32); g.fcb = c.take(5 * 32);
  g.bout = c.take(4);
  NbParams n;
  n.B = c.take(3 * 10); c.take(2); n.W1 = c.take(52 * 128); n.b1 = c.take(128); n.W2 = c.take(128 * 32);
  n.b2 = c.take(32);
  ColParams k;
  k.Bp = c.take(3 * 20); k.Bv = c.take(3 * 20); k.W0 = c.take(80 * 128); k.W1 = c.take(128 * 128);
  k.W2 = c.take(128 * 128); k.W3e = c.take(80 * 128); k.W3h = c.take(128 * 128); k.W4 = c.take(128 * 128);
  k.Wout = c.take(128 * 16); k.Fc = c.take(5 * 32 * 128); k.bias = c.take(5 * 128); k.fcb = c.take(5 * 128);
  k.bout = c.take(4);
  const float* col_chunks = c.take((size_t)27 * 32 * 128);
  const float* geo_image = c.take((size_t)kGeoImage);
  const float* col_chunks16 = c.take((size_t)27 * kChunkFloats16);
  const float* nb_frags = c.take((size_t)8192);
  const float* geo_frags = c.take((size_t)kGeoFrag);
  const int blocks2 = (Q + kTM2 - 1) / kTM2;
  const size_t geo_lds = sizeof(float) * kGeoImage;
  static PerDeviceOnce geo_attr;
  if (geo_attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_geo_v3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo_lds);
  }
  const char* f32env = getenv("GLORIE_MLP_F32");
  const bool geo_f32 = force_f32 || (f32env && f32env[0] == '1');
  const size_t geo4_lds = sizeof(float) * (kGeoFrag + 32 * 16);
  static PerDeviceOnce geo4_attr;
  if (geo4_attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_geo_v4_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo4_lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_geo_v4_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo4_lds);
  }
  if (gather_only) {
    if (force_f32 || !(geo_feats && I && weights)) return GLORIE_EINVAL;
    hipLaunchKernelGGL(mlp_geo_v4_kernel<true>, dim3(blocks2 < 512 ? blocks2 : 512), dim3(512), geo4_lds, st, g, geo_frags,
                       geo_image + kGeoRows * 32, pts, c_geo, geo_feats, I, weights, has, Q, raw, range_flag);
    if (stage_color) {
      const size_t nb4_lds = sizeof(float) * (8192 + 128 + kTM2 * 8 + 80) + sizeof(int) * kTM2 * 8;
      static PerDeviceOnce attr_go;
      if (attr_go.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_nb_v4_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb4_lds);
      }
      hipLaunchKernelGGL(mlp_nb_v4_kernel<true>, dim3(blocks2), dim3(512), nb4_lds, st, n, nb_frags, pts, cloud_pos, col_feats,
                         I, weights, has, Q, c_col_scratch, range_flag);
    }
    return check_launch();
  }
  if (geo_f32)
    hipLaunchKernelGGL(mlp_geo_v3_kernel, dim3(blocks2), dim3(512), geo_lds, st, g, geo_image, pts, c_geo,
                       c_geo ? nullptr : geo_feats, I, weights, has, Q, raw);
  else
    hipLaunchKernelGGL(mlp_geo_v4_kernel<false>, dim3(blocks2 < 512 ? blocks2 : 512), dim3(512), geo4_lds, st, g, geo_frags,
                       geo_image + kGeoRows * 32, pts, c_geo, c_geo ? nullptr : geo_feats, I, weights, has, Q, raw,
                       range_flag);
  if (stage_color) {
    const size_t nb_lds = sizeof(float) * (52 * kLdw3 + 128 + kTM2 * 8 + 80) + sizeof(int) * kTM2 * 8;
    const size_t col_lds = sizeof(float) * 2 * kChunkFloats3;
    static PerDeviceOnce attr;
    if (attr.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_nb_v3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_col_v3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
    }
    // per-neighbour and colour decoders: fp16 matrix cores with the 3-term split (fp32 accuracy); GLORIE_MLP_F32=1 keeps
    // the fp32 MFMA kernels
    const char* f32 = force_f32 ? "1" : getenv("GLORIE_MLP_F32");
    const size_t nb4_lds = sizeof(float) * (8192 + 128 + kTM2 * 8 + 80) + sizeof(int) * kTM2 * 8;
    static PerDeviceOnce attr4;
    if (attr4.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_nb_v4_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb4_lds);
    }
    // colour decoder: 32 samples per wave, 4 waves per workgroup (measured per 614k-sample batch against 16 samples per wave
    // and against 8-wave workgroups: 427 vs 495 / 485 us; the other two forms were removed in round 4)
    if (f32 && f32[0] == '1')
      hipLaunchKernelGGL(mlp_nb_v3_kernel, dim3(blocks2), dim3(512), nb_lds, st, n, pts, cloud_pos, col_feats,
                         I, weights, has, Q, c_col_scratch);
    else
      hipLaunchKernelGGL(mlp_nb_v4_kernel<false>, dim3(blocks2), dim3(512), nb4_lds, st, n, nb_frags, pts, cloud_pos, col_feats,
                         I, weights, has, Q, c_col_scratch, range_flag);
    const size_t col16_lds = sizeof(float) * 2 * kChunkFloats16;
    if (f32 && f32[0] == '1')
      hipLaunchKernelGGL(mlp_col_v3_kernel, dim3(blocks2), dim3(512), col_lds, st, k, col_chunks, pts, views,
                         c_col_scratch, Q, raw);
    else
      hipLaunchKernelGGL((mlp_col_v5_kernel<2, 4>), dim3((Q + 127) / 128), dim3(256), col16_lds, st, k, col_chunks16, pts,
                         views, c_col_scratch, Q, raw, range_flag);
  }
  return check_launch();
}
