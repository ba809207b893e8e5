// Fused neural-point decoders on the matrix cores (scope row R3) for gfx950.
//
// Replaces the chains of small nn.Linear launches of
//   MLP_geometry.forward   /root/reference/src/modules/conv_onet/models/decoder.py:175-225
//   MLP_color.get_feature_at_pos (per-neighbour F_theta)              decoder.py:340-389, 228-243
//   MLP_color.forward                                                 decoder.py:391-433
// with three kernels that keep every activation in LDS/registers:
//   geo  : Fourier(93, sin) -> 5 x (Linear(32) ReLU + fc_c(c)) with the skip at layer 2 -> occ
//   nb   : per neighbour [sin,cos](rel B)(20) ++ col_feat(32) -> Linear(128) softplus -> IDW sum
//          -> Linear(32).  The second layer is linear, so it is applied ONCE to the weighted
//          sum:  sum_k w_k (W2 y_k + b2) = W2 (sum_k w_k y_k) + b2 sum_k w_k   (8x fewer MACs).
//   col  : [sin,cos](p B)(40) ++ [sin,cos](v B)(40) -> 5 x (Linear(128) softplus + fc_c(c)),
//          skip at layer 2 -> sigmoid rgb
//
// All GEMMs are exact fp32 on v_mfma_f32_16x16x4_f32 (the reference runs fp32; no xf32 on
// gfx950).  A workgroup = 4 waves = 64 samples; wave w owns rows 16w..16w+15 for the whole
// network, so activations never cross waves: A operands come from the wave's own LDS rows
// (leading dimension = 2 mod 32 -> conflict-free ds_read_b32 for the (row = lane&15,
// k = lane>>4) fragment), B operands (weights, K-major [K][N], L1/L2 resident) are read
// straight from global memory as coalesced 64-byte rows.  The skip connection is two GEMMs
// on the split weight (embedding rows / hidden rows) -- the concatenated activation is never
// built.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kTM = 64;            // samples per workgroup
constexpr float kTwoPi = 6.283185307179586f;

// acc[t] += A[16 rows x K] * W[K x (16*NT cols starting at col0)]
// A_lds points at row 0 of this wave's 16-row tile; W is K-major with leading dimension ldw.
// K must be a multiple of 4 (packed weights / staged activations are zero padded).
template <int NT>
__device__ __forceinline__ void gemm16(f32x4 (&acc)[NT], const float* __restrict__ A_lds, int lda,
                                       int K, const float* __restrict__ W, int ldw, int col0) {
  const int lane = threadIdx.x & 63;
  const float* ap = A_lds + (lane & 15) * lda + (lane >> 4);
  const float* wp = W + (size_t)(lane >> 4) * ldw + col0 + (lane & 15);
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 4) {
    const float a = ap[k0];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float b = wp[(size_t)k0 * ldw + 16 * t];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ float softplus100(float x) {
  // torch.nn.Softplus(beta=100, threshold=20): x if beta*x > 20
  const float bx = 100.0f * x;
  return bx > 20.0f ? x : log1pf(expf(bx)) * 0.01f;
}

// C/D fragment of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
template <int NT, typename F>
__device__ __forceinline__ void for_each_out(f32x4 (&acc)[NT], F f) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = f(acc[t][r], (lane >> 4) * 4 + r, 16 * t + (lane & 15));
}

template <int NT>
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[NT], float* __restrict__ H_lds, int ldh) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) H_lds[((lane >> 4) * 4 + r) * ldh + 16 * t + (lane & 15)] = acc[t][r];
}

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

// ------------------------------------------------------------------------------------
// geometry decoder
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_geo_kernel(GeoParams P, const float* __restrict__ pts,
                                                      const float* __restrict__ c_geo,
                                                      const uint8_t* __restrict__ has, int Q,
                                                      float* __restrict__ raw) {
  constexpr int LDE = 98, LDH = 34;
  __shared__ float emb[kTM * LDE];
  __shared__ float hbuf[kTM * LDH];
  __shared__ float cbuf[kTM * LDH];
  const int tid = threadIdx.x, wv = tid >> 6;
  const int q0 = blockIdx.x * kTM;
  // stage: Fourier embedding (sin only) and the interpolated feature c
  for (int idx = tid; idx < kTM * 96; idx += 256) {
    const int r = idx / 96, f = idx - r * 96;
    const int q = min(q0 + r, Q - 1);
    float v = 0.0f;
    if (f < 93) {
      const float x = kTwoPi * pts[(size_t)q * 3 + 0], y = kTwoPi * pts[(size_t)q * 3 + 1],
                  z = kTwoPi * pts[(size_t)q * 3 + 2];
      v = sinf(fmaf(z, P.B[2 * 96 + f], fmaf(y, P.B[96 + f], x * P.B[f])));
    }
    emb[r * LDE + f] = v;
  }
  for (int idx = tid; idx < kTM * 32; idx += 256) {
    const int r = idx >> 5, f = idx & 31;
    const int q = min(q0 + r, Q - 1);
    cbuf[r * LDH + f] = c_geo[(size_t)q * 32 + f];
  }
  __syncthreads();
  const float* E = emb + wv * 16 * LDE;
  float* H = hbuf + wv * 16 * LDH;
  const float* C = cbuf + wv * 16 * LDH;
  f32x4 acc[2];
  auto layer_tail = [&](int li) {
    for_each_out<2>(acc, [&](float v, int, int col) { return fmaxf(v + P.bias[li * 32 + col], 0.0f) + P.fcb[li * 32 + col]; });
    gemm16<2>(acc, C, LDH, 32, P.Fc + li * 32 * 32, 32, 0);
    __syncthreads();
    store_tile<2>(acc, H, LDH);
    __syncthreads();
  };
  zero<2>(acc); gemm16<2>(acc, E, LDE, 96, P.W0, 32, 0); layer_tail(0);
  zero<2>(acc); gemm16<2>(acc, H, LDH, 32, P.W1, 32, 0); layer_tail(1);
  zero<2>(acc); gemm16<2>(acc, H, LDH, 32, P.W2, 32, 0); layer_tail(2);
  zero<2>(acc); gemm16<2>(acc, E, LDE, 96, P.W3e, 32, 0); gemm16<2>(acc, H, LDH, 32, P.W3h, 32, 0); layer_tail(3);
  zero<2>(acc); gemm16<2>(acc, H, LDH, 32, P.W4, 32, 0); layer_tail(4);
  f32x4 o[1];
  zero<1>(o);
  gemm16<1>(o, H, LDH, 32, P.Wout, 16, 0);
  const int lane = tid & 63;
  if ((lane & 15) == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + wv * 16 + (lane >> 4) * 4 + r;
      if (q < Q) raw[(size_t)q * 4 + 3] = has[q] ? o[0][r] + P.bout[0] : -100.0f;  // Renderer.py:206-207
    }
  }
}

// ------------------------------------------------------------------------------------
// per-neighbour colour features (F_theta) + IDW sum
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_nb_kernel(NbParams P, const float* __restrict__ pts,
                                                     const float* __restrict__ cloud,
                                                     const float* __restrict__ col_feats,
                                                     const int64_t* __restrict__ I,
                                                     const float* __restrict__ wts,
                                                     const uint8_t* __restrict__ has, int Q,
                                                     float* __restrict__ c_col) {
  constexpr int LDX = 66, LDY = 130;  // 52 -> pad, 128 -> pad
  __shared__ float xbuf[kTM * LDX];
  __shared__ float ybuf[kTM * LDY];
  __shared__ float wbuf[kTM * 8];
  __shared__ int ibuf[kTM * 8];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int q0 = blockIdx.x * kTM;
  for (int idx = tid; idx < kTM * 8; idx += 256) {
    const int r = idx >> 3;
    const int q = min(q0 + r, Q - 1);
    const int ii = (int)I[(size_t)q * 8 + (idx & 7)];
    const float w = (q0 + r < Q && ii >= 0) ? wts[(size_t)q * 8 + (idx & 7)] : 0.0f;
    wbuf[idx] = w;
    ibuf[idx] = ii < 0 ? 0 : ii;
  }
  __syncthreads();
  f32x4 ysum[8];
  zero<8>(ysum);
  const float* X = xbuf + wv * 16 * LDX;
  for (int k = 0; k < 8; ++k) {
    // stage x = [sin(rel B) (10), cos(rel B) (10), col_feat (32)]; 4 threads per sample row
    {
      const int r = tid >> 2, part = tid & 3;
      const int q = min(q0 + r, Q - 1);
      const int pt = ibuf[r * 8 + k];
      const float rx = kTwoPi * (cloud[(size_t)pt * 3 + 0] - pts[(size_t)q * 3 + 0]);
      const float ry = kTwoPi * (cloud[(size_t)pt * 3 + 1] - pts[(size_t)q * 3 + 1]);
      const float rz = kTwoPi * (cloud[(size_t)pt * 3 + 2] - pts[(size_t)q * 3 + 2]);
      for (int f = part; f < 10; f += 4) {
        const float a = fmaf(rz, P.B[20 + f], fmaf(ry, P.B[10 + f], rx * P.B[f]));
        float s, c;
        sincosf(a, &s, &c);
        xbuf[r * LDX + f] = s;
        xbuf[r * LDX + 10 + f] = c;
      }
      const float4* src = reinterpret_cast<const float4*>(col_feats + (size_t)pt * 32 + part * 8);
      const float4 v0 = src[0], v1 = src[1];
      float* dst = xbuf + r * LDX + 20 + part * 8;
      dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w;
      dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
    }
    __syncthreads();
    f32x4 acc[8];
    zero<8>(acc);
    gemm16<8>(acc, X, LDX, 52, P.W1, 128, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wv * 16 + (lane >> 4) * 4 + r;
        const float w = wbuf[row * 8 + k];
        ysum[t][r] += w * softplus100(acc[t][r] + P.b1[16 * t + (lane & 15)]);
      }
    __syncthreads();
  }
  float* Y = ybuf + wv * 16 * LDY;
  store_tile<8>(ysum, Y, LDY);
  __syncthreads();
  f32x4 o[2];
  zero<2>(o);
  gemm16<2>(o, Y, LDY, 128, P.W2, 32, 0);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wv * 16 + (lane >> 4) * 4 + r;
      const int q = q0 + row;
      if (q < Q) {
        float sw = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sw += wbuf[row * 8 + k];
        const int col = 16 * t + (lane & 15);
        c_col[(size_t)q * 32 + col] = has[q] ? o[t][r] + P.b2[col] * sw : 0.0f;
      }
    }
}

// ------------------------------------------------------------------------------------
// colour decoder
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_col_kernel(ColParams P, const float* __restrict__ pts,
                                                      const float* __restrict__ views,
                                                      const float* __restrict__ c_col, int Q,
                                                      float* __restrict__ raw) {
  constexpr int LDE = 82, LDH = 130, LDC = 34;
  __shared__ float emb[kTM * LDE];
  __shared__ float hbuf[kTM * LDH];
  __shared__ float cbuf[kTM * LDC];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int q0 = blockIdx.x * kTM;
  // embedding: [sin(pB) 20 | cos(pB) 20 | sin(vB) 20 | cos(vB) 20], v = normalised view direction
  for (int idx = tid; idx < kTM * 40; idx += 256) {
    const int r = idx / 40, f = idx - r * 40;
    const int q = min(q0 + r, Q - 1);
    float x, y, z;
    const float* Bm;
    int base;
    if (f < 20) {
      x = pts[(size_t)q * 3 + 0]; y = pts[(size_t)q * 3 + 1]; z = pts[(size_t)q * 3 + 2];
      Bm = P.Bp; base = 0;
    } else {
      x = views[(size_t)q * 3 + 0]; y = views[(size_t)q * 3 + 1]; z = views[(size_t)q * 3 + 2];
      const float nrm = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);  // F.normalize(p=2, eps=1e-12)
      x /= nrm; y /= nrm; z /= nrm;
      Bm = P.Bv; base = 40;
    }
    const int ff = f % 20;
    const float a = fmaf(kTwoPi * z, Bm[40 + ff], fmaf(kTwoPi * y, Bm[20 + ff], (kTwoPi * x) * Bm[ff]));
    float s, c;
    sincosf(a, &s, &c);
    emb[r * LDE + base + ff] = s;
    emb[r * LDE + base + 20 + ff] = c;
  }
  for (int idx = tid; idx < kTM * 32; idx += 256) {
    const int r = idx >> 5, f = idx & 31;
    const int q = min(q0 + r, Q - 1);
    cbuf[r * LDC + f] = c_col[(size_t)q * 32 + f];
  }
  __syncthreads();
  const float* E = emb + wv * 16 * LDE;
  float* H = hbuf + wv * 16 * LDH;
  const float* C = cbuf + wv * 16 * LDC;
  f32x4 acc[8];
  auto layer_tail = [&](int li) {
    for_each_out<8>(acc, [&](float v, int, int col) { return softplus100(v + P.bias[li * 128 + col]) + P.fcb[li * 128 + col]; });
    gemm16<8>(acc, C, LDC, 32, P.Fc + li * 32 * 128, 128, 0);
    __syncthreads();
    store_tile<8>(acc, H, LDH);
    __syncthreads();
  };
  zero<8>(acc); gemm16<8>(acc, E, LDE, 80, P.W0, 128, 0); layer_tail(0);
  zero<8>(acc); gemm16<8>(acc, H, LDH, 128, P.W1, 128, 0); layer_tail(1);
  zero<8>(acc); gemm16<8>(acc, H, LDH, 128, P.W2, 128, 0); layer_tail(2);
  zero<8>(acc); gemm16<8>(acc, E, LDE, 80, P.W3e, 128, 0); gemm16<8>(acc, H, LDH, 128, P.W3h, 128, 0); layer_tail(3);
  zero<8>(acc); gemm16<8>(acc, H, LDH, 128, P.W4, 128, 0); layer_tail(4);
  f32x4 o[1];
  zero<1>(o);
  gemm16<1>(o, H, LDH, 128, P.Wout, 16, 0);
  const int col = lane & 15;
  if (col < 3) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + wv * 16 + (lane >> 4) * 4 + r;
      if (q < Q) raw[(size_t)q * 4 + col] = 1.0f / (1.0f + expf(-(o[0][r] + P.bout[col])));
    }
  }
}

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
  return geo + nb + col;
}

extern "C" int glorie_render_mlp(const float* packed, const float* pts, const float* views,
                                 const float* cloud_pos, const float* col_feats,
                                 const float* c_geo, const int64_t* I, const float* weights,
                                 const uint8_t* has, int Q, float* c_col_scratch, float* raw,
                                 int stage_color, void* stream) {
  if (Q < 0) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!packed || !pts || !c_geo || !has || !raw) return GLORIE_EINVAL;
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
  const int blocks = (Q + kTM - 1) / kTM;
  hipLaunchKernelGGL(mlp_geo_kernel, dim3(blocks), dim3(256), 0, st, g, pts, c_geo, has, Q, raw);
  if (stage_color) {
    hipLaunchKernelGGL(mlp_nb_kernel, dim3(blocks), dim3(256), 0, st, n, pts, cloud_pos, col_feats, I,
                       weights, has, Q, c_col_scratch);
    hipLaunchKernelGGL(mlp_col_kernel, dim3(blocks), dim3(256), 0, st, k, pts, views, c_col_scratch, Q, raw);
  }
  return check_launch();
}
