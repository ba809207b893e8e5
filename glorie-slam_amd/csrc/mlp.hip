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


// ====================================================================================
// v2 kernels: 128 samples / 8 waves per workgroup, weights streamed through LDS in 32-row
// chunks (double buffered, shared by the 8 waves), embedding / feature A-fragments kept in
// registers, fast softplus.  Same arithmetic (exact fp32 MFMA); only the transcendental
// approximations of softplus differ (v_exp/v_log, |err| < 1e-6).
// ====================================================================================
constexpr int kTM2 = 128;
constexpr int kLdw = 144;                 // chunk row stride: (k*144 + col) % 32 is conflict free
constexpr int kChunkFloats = 32 * kLdw;   // one 32 x 128 weight chunk in LDS
constexpr int kLdh = 130;

__device__ __forceinline__ float softplus100_fast(float x) {
  const float t = 100.0f * x;
  return t > 20.0f ? x : 0.01f * __logf(1.0f + __expf(t));
}

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
__device__ __forceinline__ void chunk_store(float* Wb, const ChunkRegs& r) {
  const int t = threadIdx.x;
  float4* dst = reinterpret_cast<float4*>(Wb + (t >> 4) * kLdw + (t & 15) * 8);
  dst[0] = r.a;
  dst[1] = r.b;
}

// acc[t] += a_k * W[k][16t + lane&15] for NS k-steps, A from a register fragment
template <int A0, int NS, int NREG>
__device__ __forceinline__ void mma_regs(f32x4 (&acc)[8], const float (&a)[NREG], const float* Wb) {
  const int lane = threadIdx.x & 63;
  const float* wp = Wb + (lane >> 4) * kLdw + (lane & 15);
#pragma unroll
  for (int sidx = 0; sidx < NS; ++sidx) {
    const float av = a[A0 + sidx];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wp[sidx * 4 * kLdw + 16 * t], acc[t], 0, 0, 0);
  }
}
// A from this wave's rows of the activation buffer (8 k-steps starting at column h0)
__device__ __forceinline__ void mma_lds(f32x4 (&acc)[8], const float* Hrow, int h0, const float* Wb) {
  const int lane = threadIdx.x & 63;
  const float* wp = Wb + (lane >> 4) * kLdw + (lane & 15);
  const float* ap = Hrow + h0 + (lane >> 4);
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx) {
    const float av = ap[4 * sidx];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wp[sidx * 4 * kLdw + 16 * t], acc[t], 0, 0, 0);
  }
}

__global__ __launch_bounds__(512) void mlp_col_v2_kernel(ColParams P, const float* __restrict__ Wall,
                                                         const float* __restrict__ pts,
                                                         const float* __restrict__ views,
                                                         const float* __restrict__ c_col, int Q,
                                                         float* __restrict__ raw) {
  extern __shared__ float smem[];
  float* Wbuf = smem;                         // [2][32][144]
  float* H = smem + 2 * kChunkFloats;         // [128][130]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * kTM2;
  const int q = min(q0 + wv * 16 + r, Q - 1);
  float* Hrow = H + (wv * 16 + r) * kLdh;     // A-operand row of this lane
  float* Hw = H + (wv * 16) * kLdh;           // this wave's 16 rows (C/D stores)

  // ---- A fragments held in registers: embedding e[20] (k = 4j + g) and feature c[8]
  float e[20], c[8];
  {
    const float px = kTwoPi * pts[(size_t)q * 3 + 0], py = kTwoPi * pts[(size_t)q * 3 + 1],
                pz = kTwoPi * pts[(size_t)q * 3 + 2];
    float vx = views[(size_t)q * 3 + 0], vy = views[(size_t)q * 3 + 1], vz = views[(size_t)q * 3 + 2];
    const float nrm = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
    vx = kTwoPi * (vx / nrm); vy = kTwoPi * (vy / nrm); vz = kTwoPi * (vz / nrm);
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int f = 4 * j + g;             // feature index 0..79
      const int blk = f / 20, ff = f - blk * 20;
      const float* Bm = blk < 2 ? P.Bp : P.Bv;
      const float x = blk < 2 ? px : vx, y = blk < 2 ? py : vy, z = blk < 2 ? pz : vz;
      const float a = fmaf(z, Bm[40 + ff], fmaf(y, Bm[20 + ff], x * Bm[ff]));
      e[j] = (blk & 1) ? cosf(a) : sinf(a);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = c_col[(size_t)q * 32 + 4 * j + g];
  }

  f32x4 acc[8];
  constexpr int NC = 27;
  ChunkRegs nxt = chunk_load(Wall, 0);
  chunk_store(Wbuf, nxt);
  __syncthreads();

  auto act = [&](int li) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float bb = P.bias[li * 128 + 16 * t + (lane & 15)];
      const float fb = P.fcb[li * 128 + 16 * t + (lane & 15)];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[t][rr] = softplus100_fast(acc[t][rr] + bb) + fb;
    }
  };
  auto store_h = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Hw[((lane >> 4) * 4 + rr) * kLdh + 16 * t + (lane & 15)] = acc[t][rr];
  };

#define GL_CHUNK(cidx, BODY)                                        \
  {                                                                 \
    if ((cidx) + 1 < NC) nxt = chunk_load(Wall, (cidx) + 1);        \
    const float* Wb = Wbuf + ((cidx) & 1) * kChunkFloats;           \
    BODY;                                                           \
    if ((cidx) + 1 < NC) chunk_store(Wbuf + (((cidx) + 1) & 1) * kChunkFloats, nxt); \
    __syncthreads();                                                \
  }

  // layer 0: W0 (80 rows -> chunks 0..2), Fc0 (chunk 3)
  zero<8>(acc);
  GL_CHUNK(0, (mma_regs<0, 8, 20>(acc, e, Wb)))
  GL_CHUNK(1, (mma_regs<8, 8, 20>(acc, e, Wb)))
  GL_CHUNK(2, (mma_regs<16, 4, 20>(acc, e, Wb)))
  act(0);
  GL_CHUNK(3, (mma_regs<0, 8, 8>(acc, c, Wb), store_h()))
  // layer 1
  zero<8>(acc);
  GL_CHUNK(4, (mma_lds(acc, Hrow, 0, Wb)))
  GL_CHUNK(5, (mma_lds(acc, Hrow, 32, Wb)))
  GL_CHUNK(6, (mma_lds(acc, Hrow, 64, Wb)))
  GL_CHUNK(7, (mma_lds(acc, Hrow, 96, Wb)))
  act(1);
  GL_CHUNK(8, (mma_regs<0, 8, 8>(acc, c, Wb), store_h()))
  // layer 2
  zero<8>(acc);
  GL_CHUNK(9, (mma_lds(acc, Hrow, 0, Wb)))
  GL_CHUNK(10, (mma_lds(acc, Hrow, 32, Wb)))
  GL_CHUNK(11, (mma_lds(acc, Hrow, 64, Wb)))
  GL_CHUNK(12, (mma_lds(acc, Hrow, 96, Wb)))
  act(2);
  GL_CHUNK(13, (mma_regs<0, 8, 8>(acc, c, Wb), store_h()))
  // layer 3 (skip): W3e on the embedding, W3h on the hidden state
  zero<8>(acc);
  GL_CHUNK(14, (mma_regs<0, 8, 20>(acc, e, Wb)))
  GL_CHUNK(15, (mma_regs<8, 8, 20>(acc, e, Wb)))
  GL_CHUNK(16, (mma_regs<16, 4, 20>(acc, e, Wb)))
  GL_CHUNK(17, (mma_lds(acc, Hrow, 0, Wb)))
  GL_CHUNK(18, (mma_lds(acc, Hrow, 32, Wb)))
  GL_CHUNK(19, (mma_lds(acc, Hrow, 64, Wb)))
  GL_CHUNK(20, (mma_lds(acc, Hrow, 96, Wb)))
  act(3);
  GL_CHUNK(21, (mma_regs<0, 8, 8>(acc, c, Wb), store_h()))
  // layer 4
  zero<8>(acc);
  GL_CHUNK(22, (mma_lds(acc, Hrow, 0, Wb)))
  GL_CHUNK(23, (mma_lds(acc, Hrow, 32, Wb)))
  GL_CHUNK(24, (mma_lds(acc, Hrow, 64, Wb)))
  GL_CHUNK(25, (mma_lds(acc, Hrow, 96, Wb)))
  act(4);
  GL_CHUNK(26, (mma_regs<0, 8, 8>(acc, c, Wb), store_h()))
#undef GL_CHUNK
  // output layer 128 -> 3 (B straight from global: 16 columns, 3 real)
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* ap = Hrow + g;
    const float* wp = P.Wout + g * 16 + r;
#pragma unroll 8
    for (int sidx = 0; sidx < 32; ++sidx)
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * sidx], wp[sidx * 64], o, 0, 0, 0);
  }
  if (r < 3) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int qq = q0 + wv * 16 + g * 4 + rr;
      if (qq < Q) raw[(size_t)qq * 4 + r] = 1.0f / (1.0f + __expf(-(o[rr] + P.bout[r])));
    }
  }
}

// per-neighbour F_theta, v2: W1 (52 x 128) resident in LDS for the whole workgroup
__global__ __launch_bounds__(512, 2) void mlp_nb_v2_kernel(NbParams P, const float* __restrict__ pts,
                                                        const float* __restrict__ cloud,
                                                        const float* __restrict__ col_feats,
                                                        const int64_t* __restrict__ I,
                                                        const float* __restrict__ wts,
                                                        const uint8_t* __restrict__ has, int Q,
                                                        float* __restrict__ c_col) {
  constexpr int LDX = 66;
  extern __shared__ float smem[];
  // 72 KB of LDS -> 2 workgroups (16 waves) per CU: the gather latency of one workgroup's staging phase
  // is covered by the other's MFMAs.  The [128][130] buffer of the second layer reuses the whole region
  // once the neighbour loop is over (the IDW weight sums it would overwrite are taken before).
  float* W1s = smem;                          // [52][144]
  float* X = W1s + 52 * kLdw;                 // [128][66]
  float* wbuf = X + kTM2 * LDX;               // [128][8]
  int* ibuf = reinterpret_cast<int*>(wbuf + kTM2 * 8);  // [128][8]
  float* Y = smem;                            // [128][130], aliases W1s | X | wbuf | ibuf
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * kTM2;
  for (int idx = tid; idx < 52 * 128; idx += 512) W1s[(idx >> 7) * kLdw + (idx & 127)] = P.W1[idx];
  for (int idx = tid; idx < kTM2 * 8; idx += 512) {
    const int row = idx >> 3;
    const int q = min(q0 + row, Q - 1);
    const int ii = (int)I[(size_t)q * 8 + (idx & 7)];
    wbuf[idx] = (q0 + row < Q && ii >= 0) ? wts[(size_t)q * 8 + (idx & 7)] : 0.0f;
    ibuf[idx] = ii < 0 ? 0 : ii;
  }
  __syncthreads();
  f32x4 ysum[8];
  zero<8>(ysum);
  const float* Xrow = X + (wv * 16 + r) * LDX + g;
  float b1[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) b1[t] = P.b1[16 * t + (lane & 15)];
  for (int k = 0; k < 8; ++k) {
    {  // stage x = [sin(rel B) 10 | cos(rel B) 10 | col_feat 32]; 4 threads per sample row
      const int row = tid >> 2, part = tid & 3;
      const int q = min(q0 + row, Q - 1);
      const int pt = ibuf[row * 8 + k];
      const float rx = kTwoPi * (cloud[(size_t)pt * 3 + 0] - pts[(size_t)q * 3 + 0]);
      const float ry = kTwoPi * (cloud[(size_t)pt * 3 + 1] - pts[(size_t)q * 3 + 1]);
      const float rz = kTwoPi * (cloud[(size_t)pt * 3 + 2] - pts[(size_t)q * 3 + 2]);
      for (int f = part; f < 10; f += 4) {
        const float a = fmaf(rz, P.B[20 + f], fmaf(ry, P.B[10 + f], rx * P.B[f]));
        float sn, cs;
        sincosf(a, &sn, &cs);
        X[row * LDX + f] = sn;
        X[row * LDX + 10 + f] = cs;
      }
      const float4* src = reinterpret_cast<const float4*>(col_feats + (size_t)pt * 32 + part * 8);
      const float4 v0 = src[0], v1 = src[1];
      float* dst = X + row * LDX + 20 + part * 8;
      dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w;
      dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
    }
    __syncthreads();
    f32x4 acc[8];
    zero<8>(acc);
    {
      const float* wp = W1s + g * kLdw + (lane & 15);
#pragma unroll
      for (int sidx = 0; sidx < 13; ++sidx) {
        const float av = Xrow[4 * sidx];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wp[sidx * 4 * kLdw + 16 * t], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float w = wbuf[(wv * 16 + g * 4 + rr) * 8 + k];
#pragma unroll
      for (int t = 0; t < 8; ++t) ysum[t][rr] += w * softplus100_fast(acc[t][rr] + b1[t]);
    }
    __syncthreads();
  }
  float sw[4];                                // sum of the IDW weights of this lane's 4 rows
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    sw[rr] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sw[rr] += wbuf[(wv * 16 + g * 4 + rr) * 8 + k];
  }
  __syncthreads();                            // everybody is done with W1s / X / wbuf: Y may overwrite them
  float* Yw = Y + (wv * 16) * kLdh;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Yw[(g * 4 + rr) * kLdh + 16 * t + (lane & 15)] = ysum[t][rr];
  __syncthreads();
  f32x4 o[2];
  zero<2>(o);
  gemm16<2>(o, Yw, kLdh, 128, P.W2, 32, 0);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = wv * 16 + g * 4 + rr;
      const int q = q0 + row;
      if (q < Q) {
        const int col = 16 * t + (lane & 15);
        c_col[(size_t)q * 32 + col] = has[q] ? o[t][rr] + P.b2[col] * sw[rr] : 0.0f;
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
  return geo + nb + col + (size_t)27 * 32 * 128;
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
  const float* col_chunks = c.take((size_t)27 * 32 * 128);
  const int blocks = (Q + kTM - 1) / kTM;
  hipLaunchKernelGGL(mlp_geo_kernel, dim3(blocks), dim3(256), 0, st, g, pts, c_geo, has, Q, raw);
  if (stage_color) {
    const int blocks2 = (Q + kTM2 - 1) / kTM2;
    const size_t nb_work = sizeof(float) * (52 * kLdw + kTM2 * 66 + kTM2 * 8) + sizeof(int) * kTM2 * 8;
    const size_t nb_y = sizeof(float) * kTM2 * kLdh;
    const size_t nb_lds = nb_work > nb_y ? nb_work : nb_y;
    const size_t col_lds = sizeof(float) * (2 * kChunkFloats + kTM2 * kLdh);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_nb_v2_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_col_v2_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
      attr = true;
    }
    hipLaunchKernelGGL(mlp_nb_v2_kernel, dim3(blocks2), dim3(512), nb_lds, st, n, pts, cloud_pos, col_feats,
                       I, weights, has, Q, c_col_scratch);
    hipLaunchKernelGGL(mlp_col_v2_kernel, dim3(blocks2), dim3(512), col_lds, st, k, col_chunks, pts, views,
                       c_col_scratch, Q, raw);
  }
  return check_launch();
}
