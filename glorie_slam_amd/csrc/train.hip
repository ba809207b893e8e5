// Training path of the neural-point renderer (scope row N1): forward with saved activations and the
// hand-written backward of
//   MLP_geometry.forward / MLP_color.forward / get_feature_at_pos / MLP_col_neighbor
//       /root/reference/src/modules/conv_onet/models/decoder.py:130-225, 228-243, 340-433
//   raw2outputs_nerf_color                         /root/reference/src/utils/common.py:261-299
// which `loss.backward()` of the mapper walks through autograd
//   /root/reference/src/mapper.py:390-515 (optimizer_update_one_step), :511-512
// plus the Adam update of the feature rows / decoder parameters (torch.optim.Adam semantics, mapper.py:612-624).
//
// The inference kernels of csrc/mlp.hip keep every activation on chip and therefore cannot be differentiated;
// a mapping iteration renders only 5000 rays (50k samples, 400k neighbour rows), so here every layer is its
// own launch with activations in HBM (16 KB per sample, ~1 GB of traffic per iteration = 0.2 ms at HBM speed):
//   mm_rows_kernel    out[q][j] = epilogue( sum_r in[q][r] * W(r, j) )      forward layers and input gradients
//   mm_wgrad_kernel   dW[n][k] += sum_q dY[q][n] X[q][k],  db[n] += sum_q dY[q][n]     (fp32 atomics)
// both on v_mfma_f32_16x16x4_f32 (exact fp32: training stays in the reference's precision), one wave = 16 rows
// (or 16 output rows of dW) x all columns, operands straight from L1/L2 (the matrices have <= 208 columns);
// and a handful of element-wise / gather kernels (Fourier features with learnable B, IDW gather / scatter,
// per-neighbour rows, compositing).  The launch sequence lives in glorie_render_train_fwd / _bwd so that one
// C call = one pass (no Python per layer).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { TACT_NONE = 0, TACT_RELU = 1, TACT_SOFTPLUS = 2, TACT_SIGMOID = 3 };
constexpr float kBeta = 100.0f;       // nn.Softplus(beta=100), decoder.py:108,312

// The transcendental parts run on the hardware's v_exp_f32 / v_log_f32 (base 2), like the inference kernels of
// csrc/mlp.hip: libm's expf / log1pf are ~60 VALU instructions per element and the 128-wide softplus layers of the
// per-neighbour network apply them to 51 M elements per pass.
__device__ __forceinline__ float t_act(float v, int act) {
  if (act == TACT_RELU) return fmaxf(v, 0.0f);
  if (act == TACT_SOFTPLUS) {         // softplus(beta = 100) = max(x, 0) + ln(1 + exp(-beta |x|)) / beta (exactly x beyond beta x = 20 in fp32)
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * __builtin_fabsf(v));
    return fmaf(0.006931471805599453f, __builtin_amdgcn_logf(1.0f + e), fmaxf(v, 0.0f));
  }
  if (act == TACT_SIGMOID) return 1.0f / (1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  return v;
}
// derivative of the activation expressed through its OUTPUT y (only outputs are saved)
__device__ __forceinline__ float t_dact(float y, int act) {
  if (act == TACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (act == TACT_SOFTPLUS) return 1.0f - __builtin_amdgcn_exp2f(-144.26950408889634f * y);   // sigmoid(beta z) = 1 - exp(-beta softplus(z))
  if (act == TACT_SIGMOID) return y * (1.0f - y);
  return 1.0f;
}

// ---------------------------------------------------------------------------------------------------------
// out[q][j] = [accumulate: out[q][j] +] act( sum_r in[q][r] W(r, j) + bias[j] ) * dact(saved[q][j]) + res[q][j]
//   TRANS:  W(r, j) = W[j * ldw + r]   (forward: nn.Linear weight [J][R])
//   !TRANS: W(r, j) = W[r * ldw + j]   (input gradient: dX = dY W, W [R][J])
// one wave = 16 rows x J columns (J <= 208: 13 tiles of 16), reduction in chunks of 16 (4 MFMA k-steps)
// ---------------------------------------------------------------------------------------------------------
struct MmArgs {
  const float* in; int ldi;
  const float* W; int ldw;
  const float* bias;
  const float* res; int ldr;
  const float* saved; int lds; int dact;
  float* out; int ldo;
  int Q, R, J, act, accumulate;
};

constexpr int kMaxJT = 13;
constexpr int kWsLd = 212;            // floats per staged W row: 4 * 212 = 16 (mod 32) -> conflict-free operand reads

// Workgroup = 4 waves = 64 rows x all J columns.  Per chunk of 16 reduction indices the W slice [16][J] is staged
// ONCE in LDS for the four waves (two buffers: one barrier per chunk); a wave reads its B operands from there
// (ds_read_b32, conflict-free) and its A operands as one float4 per lane (reduction index 4 g + s of the chunk
// goes to MFMA k-step s on both operands: the order of a sum is free).
template <bool TRANS>
__global__ __launch_bounds__(256) void mm_rows_kernel(MmArgs a) {
  __shared__ float Ws[2][16 * kWsLd];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long q0 = ((long)blockIdx.x * 4 + wv) * 16;
  const int njt = (a.J + 15) >> 4;
  f32x4 acc[kMaxJT];
#pragma unroll
  for (int t = 0; t < kMaxJT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long qa = q0 + i;
  const bool qok = qa < a.Q;
  const float* xrow = a.in + (qok ? qa : 0) * (long)a.ldi;
  const bool vec = ((a.ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.in) & 15) == 0);
  const int nchunk = (a.R + 15) >> 4;
  auto stage = [&](int c, int buf) {
    // Ws[buf][rr][j] = W(r0 + rr, j), rr = 4 g' + s' in the chunk's own order (plain row order: the A side permutes)
    const int r0 = c * 16;
    for (int e = tid; e < 16 * a.J; e += 256) {
      int rr, j;
      if (TRANS) { j = e >> 4; rr = e & 15; }          // W[j][r0 + rr]: 16 consecutive floats per row j
      else { rr = e / a.J; j = e - rr * a.J; }          // W[r0 + rr][j]: whole rows
      const int r = r0 + rr;
      const float v = r < a.R ? (TRANS ? a.W[(long)j * a.ldw + r] : a.W[(long)r * a.ldw + j]) : 0.0f;
      Ws[buf][rr * kWsLd + j] = v;
    }
  };
  // the A operands of chunk c + 1 are requested before the MFMAs of chunk c (one exposed global round trip per 16
  // reduction indices otherwise: as long as the 32-52 MFMAs of a chunk themselves)
  auto load_a = [&](int c, float (&out)[4]) {
    const int r0 = c * 16;
    if (vec && r0 + 16 <= a.R) {
      const float4 v = *reinterpret_cast<const float4*>(xrow + r0 + 4 * g);
      out[0] = qok ? v.x : 0.f; out[1] = qok ? v.y : 0.f; out[2] = qok ? v.z : 0.f; out[3] = qok ? v.w : 0.f;
    } else {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int r = r0 + 4 * g + s_;
        out[s_] = (qok && r < a.R) ? xrow[r] : 0.0f;
      }
    }
  };
  float an[4];
  load_a(0, an);
  stage(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    float av[4] = {an[0], an[1], an[2], an[3]};
    if (c + 1 < nchunk) {
      load_a(c + 1, an);
      stage(c + 1, buf ^ 1);
    }
    if (q0 < a.Q) {
#pragma unroll
      for (int t = 0; t < kMaxJT; ++t) {
        if (t < njt) {
          const float* wp = &Ws[buf][(4 * g) * kWsLd + t * 16 + i];
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], wp[s_ * kWsLd], acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (q0 >= a.Q) return;
  // lane: rows q0 + 4 g + rr, column t * 16 + i
#pragma unroll
  for (int t = 0; t < kMaxJT; ++t) {
    if (t >= njt) continue;
    const int j = t * 16 + i;
    if (j >= a.J) continue;
    const float b = a.bias ? a.bias[j] : 0.0f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const long q = q0 + 4 * g + rr;
      if (q >= a.Q) continue;
      float v = t_act(acc[t][rr] + b, a.act);
      if (a.saved) v *= t_dact(a.saved[q * a.lds + j], a.dact);
      if (a.res) v += a.res[q * a.ldr + j];
      float* o = a.out + q * a.ldo + j;
      if (a.accumulate) v += *o;
      *o = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// dW[n][k] += sum_q dY[q][n] X[q][k]   (k < K),   db[n] += sum_q dY[q][n]   (the "k == K" column of ones)
// A workgroup owns a contiguous range of rows q and walks it in chunks of 32 rows: dY[32][N] and X[32][K (+1)] are
// staged in LDS with whole-row (coalesced) loads - every element leaves HBM once - and the (n-tile, k-tile)
// pairs of the [N][K+1] result are dealt round-robin to the four waves (<= 28 accumulator tiles per wave).
// The partial result of the range is added to dW with fp32 atomics (one per element and workgroup).
// ---------------------------------------------------------------------------------------------------------
struct WgArgs {
  const float* dY; int ldy;
  const float* X; int ldx;
  float* dW; int ldw;
  float* db;
  long Q; int N, K; long rows_per_wg;
  int Ns, Ks;                 // LDS row strides (== 16 mod 32: conflict-free operand reads)
  int vec_y, vec_x;           // rows can be staged in 16-byte pieces (N resp. K, the leading dimension and the base 16-byte aligned)
};

// The reduction runs over the SAMPLES (50 k - 400 k rows), the result is tiny: a workgroup owns a contiguous range of rows
// and walks it four rows (= one MFMA k-step) at a time.  No LDS and no barrier: lane (i, g) takes its A operand
// dY[q0 + g][16 nt + i] and its B operand X[q0 + g][16 kt + i] straight from global memory (64 contiguous bytes per row and
// tile), every wave loads all NNT row tiles of dY and the column tiles kt = wave, wave + 4, ... (KT of them) of X, and the
// operands of the next DEPTH steps are in flight while a step is multiplied (a register ring, refilled slot by slot).
// (The first version staged 32-row chunks in LDS between two barriers: one exposed memory round trip per chunk, an order of
// magnitude above the fp32-MFMA time of these layers.)  Instantiated per layer shape so that the small layers (32 x 32:
// 8 accumulator registers) keep 8 steps in flight at full occupancy and the 128 x 209 layer (128 accumulator registers) two.
// The partial result of the range is added to dW with fp32 atomics (one per element and workgroup).
template <int NNT, int KT, int DEPTH>
__global__ __launch_bounds__(256) void mm_wgrad_kernel(WgArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const long qa = (long)blockIdx.x * a.rows_per_wg;
  const long qb = qa + a.rows_per_wg < a.Q ? qa + a.rows_per_wg : a.Q;
  const int KE = a.K + (a.db ? 1 : 0);
  const int nnt = (a.N + 15) >> 4, nkt = (KE + 15) >> 4;
  f32x4 acc[KT][NNT];
#pragma unroll
  for (int j = 0; j < KT; ++j)
#pragma unroll
    for (int t = 0; t < NNT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane column state: which columns exist, which one is the column of ones (the bias gradient)
  bool yok[NNT], xok[KT], xone[KT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) yok[t] = t < nnt && t * 16 + i < a.N;
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int col = (wv + 4 * j) * 16 + i;
    xok[j] = wv + 4 * j < nkt && col < a.K;
    xone[j] = wv + 4 * j < nkt && col == a.K && a.db;
  }
  auto load = [&](long q0, float (&yf)[NNT], float (&xf)[KT]) {
    const long q = q0 + g;
    const bool rok = q < qb;
    const float* yr = a.dY + (rok ? q : qa) * (long)a.ldy + i;
    const float* xr = a.X + (rok ? q : qa) * (long)a.ldx + wv * 16 + i;
#pragma unroll
    for (int t = 0; t < NNT; ++t) yf[t] = (rok && yok[t]) ? yr[t * 16] : 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) xf[j] = (rok && xok[j]) ? xr[j * 64] : ((rok && xone[j]) ? 1.0f : 0.0f);
  };
  float yf[DEPTH][NNT], xf[DEPTH][KT];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(qa + 4 * d, yf[d], xf[d]);
  for (long q0 = qa; q0 < qb; q0 += 4 * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[d][t], xf[d][j], acc[j][t], 0, 0, 0);
      load(q0 + 4 * (DEPTH + d), yf[d], xf[d]);
    }
  }
  // lane: rows n = nt * 16 + 4 g + rr of dW, column kt * 16 + i
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int kt = wv + 4 * j;
    if (kt >= nkt) continue;
    const int k = kt * 16 + i;
#pragma unroll
    for (int t = 0; t < NNT; ++t) {
      if (t >= nnt) continue;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int n = t * 16 + 4 * g + rr;
        if (n >= a.N) continue;
        const float v = acc[j][t][rr];
        if (k < a.K) atomicAdd(a.dW + (long)n * a.ldw + k, v);
        else if (k == a.K && a.db) atomicAdd(a.db + n, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fourier features  v = (2 pi x) B,  out = sin(v) [| cos(v)]      (decoder.py:8-37)
// x [Q,3] (row stride ldx; optional L2 normalisation of x: F.normalize(views), decoder.py:404), B [3][M]
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_x3(const float* x, long q, int ldx, int normalize, float (&v)[3]) {
  v[0] = x[q * ldx + 0]; v[1] = x[q * ldx + 1]; v[2] = x[q * ldx + 2];
  if (normalize) {
    const float nrm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] /= nrm; v[1] /= nrm; v[2] /= nrm;
  }
}
__device__ __forceinline__ float fourier_phase(const float (&x)[3], const float* B, int M, int m) {
  const float tp = 6.283185307179586f;
  return (tp * x[0]) * B[m] + (tp * x[1]) * B[M + m] + (tp * x[2]) * B[2 * M + m];
}

__global__ __launch_bounds__(256) void fourier_fwd_kernel(const float* __restrict__ x, int ldx, int normalize,
                                                          const float* __restrict__ B, int M, int concat, long Q,
                                                          float* __restrict__ out, int ldo) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Q * M) return;
  const long q = idx / M;
  const int m = (int)(idx - q * M);
  float xv[3];
  load_x3(x, q, ldx, normalize, xv);
  const float v = fourier_phase(xv, B, M, m);
  out[q * ldo + m] = sinf(v);
  if (concat) out[q * ldo + M + m] = cosf(v);
}

// dB[d][m] += sum_q 2 pi x[q][d] (g_sin[q][m] cos(v) - g_cos[q][m] sin(v));  grid (ceil(M/..), q-chunks)
__global__ __launch_bounds__(256) void fourier_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ B,
                                                          int M, int concat, long Q, const float* __restrict__ gout,
                                                          int ldg, float* __restrict__ dB, int rows_per_block) {
  // thread = (m, q-lane): 256 threads cover min(M, 32) columns x 8.. row lanes; simple strided loops
  const int m = threadIdx.x % 32, ql = threadIdx.x / 32;          // 32 columns x 8 row lanes
  const long qa = (long)blockIdx.x * rows_per_block;
  const long qb = qa + rows_per_block < Q ? qa + rows_per_block : Q;
  const float tp = 6.283185307179586f;
  for (int m0 = 0; m0 < M; m0 += 32) {
    const int mm = m0 + m;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (mm < M) {
      for (long q = qa + ql; q < qb; q += 8) {
        float xv[3];
        load_x3(x, q, ldx, 0, xv);
        const float v = fourier_phase(xv, B, M, mm);
        float gv = gout[q * ldg + mm] * cosf(v);
        if (concat) gv -= gout[q * ldg + M + mm] * sinf(v);
        s0 += tp * xv[0] * gv; s1 += tp * xv[1] * gv; s2 += tp * xv[2] * gv;
      }
    }
    // reduce the 8 row lanes through LDS
    __shared__ float red[3][8][32];
    red[0][ql][m] = s0; red[1][ql][m] = s1; red[2][ql][m] = s2;
    __syncthreads();
    if (ql < 3 && mm < M) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[ql][k][m];
      atomicAdd(dB + ql * M + mm, s);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// inverse-distance interpolation of the feature table:  c[q] = has ? sum_k w[q][k] feats[I[q][k]] : 0
// and its transpose  dfeats[I[q][k]] += w[q][k] dc[q]     (32 lanes = one 128-byte feature row)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void idw_fwd_kernel(const float* __restrict__ feats, const int64_t* __restrict__ I,
                                                      const float* __restrict__ w, const uint8_t* __restrict__ has,
                                                      long Q, float* __restrict__ c, int ldc) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long q = t >> 5;
  const int ch = (int)(t & 31);
  if (q >= Q) return;
  float acc = 0.0f;
  if (has[q]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float wk = w[q * 8 + k];
      const long ik = I[q * 8 + k];
      if (wk != 0.0f && ik >= 0) acc += wk * feats[ik * 32 + ch];
    }
  }
  c[q * ldc + ch] = acc;
}

__global__ __launch_bounds__(256) void idw_bwd_kernel(const float* __restrict__ dc, int ldc, const int64_t* __restrict__ I,
                                                      const float* __restrict__ w, const uint8_t* __restrict__ has,
                                                      long Q, float* __restrict__ dfeats) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long q = t >> 5;
  const int ch = (int)(t & 31);
  if (q >= Q || !has[q]) return;
  const float g = dc[q * ldc + ch];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float wk = w[q * 8 + k];
    const long ik = I[q * 8 + k];
    if (wk != 0.0f && ik >= 0) atomicAdd(dfeats + ik * 32 + ch, wk * g);
  }
}

// ---------------------------------------------------------------------------------------------------------
// per-neighbour rows of the colour decoder (decoder.py:361-383): row (q, k) = [sin(v) | cos(v) | col_feats[I]],
// v = 2 pi (cloud_pos[I] - p) B_rel;  64 threads per row group: thread = (row, column 0..51)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nb_rows_fwd_kernel(const float* __restrict__ pts, const float* __restrict__ cloud,
                                                          const float* __restrict__ feats, const int64_t* __restrict__ I,
                                                          const float* __restrict__ Brel, long Q, float* __restrict__ X,
                                                          int ldx) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = t >> 6;                       // (q, k)
  const int col = (int)(t & 63);
  if (row >= Q * 8 || col >= 52) return;
  const long q = row >> 3;
  long ik = I[row];
  if (ik < 0) ik = 0;                            // clamp(min=0) of the reference; its weight is zero
  float v;
  if (col < 20) {
    const int m = col < 10 ? col : col - 10;
    float rel[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) rel[d] = cloud[ik * 3 + d] - pts[q * 3 + d];
    const float ph = fourier_phase(rel, Brel, 10, m);
    v = col < 10 ? sinf(ph) : cosf(ph);
  } else {
    v = feats[ik * 32 + (col - 20)];
  }
  X[row * ldx + col] = v;
}

// dX [8Q,52] -> dcol_feats[I] (columns 20..51, atomics) and dB_rel (columns 0..19).  A workgroup walks many rows
// (4 per step) and adds its 30 partial sums of dB_rel ONCE: one atomic per 4 rows on the same 30 addresses
// serialised the whole launch (1.2 ms for 400k rows).
__global__ __launch_bounds__(256) void nb_rows_bwd_kernel(const float* __restrict__ dX, int ldx, const float* __restrict__ pts,
                                                          const float* __restrict__ cloud, const int64_t* __restrict__ I,
                                                          const float* __restrict__ Brel, long Q,
                                                          float* __restrict__ dfeats, float* __restrict__ dBrel,
                                                          long rows_per_block) {
  __shared__ float red[4][30];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long ra = (long)blockIdx.x * rows_per_block;
  const long rb = ra + rows_per_block < Q * 8 ? ra + rows_per_block : Q * 8;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const float tp = 6.283185307179586f;
  for (long row = ra + rl; row < rb; row += 4) {
    if (col >= 52) continue;
    const long q = row >> 3;
    long ik = I[row];
    if (ik < 0) ik = 0;       // feats[I.clamp(min=0)]: the weight of a missing neighbour is zero, so is its gradient row
    if (col >= 20) {
      const float g = dX[row * ldx + col];
      if (g != 0.0f) atomicAdd(dfeats + ik * 32 + (col - 20), g);
    } else if (col < 10) {
      float rel[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) rel[d] = cloud[ik * 3 + d] - pts[q * 3 + d];
      const float ph = fourier_phase(rel, Brel, 10, col);
      const float gv = dX[row * ldx + col] * cosf(ph) - dX[row * ldx + 10 + col] * sinf(ph);
      s0 += tp * rel[0] * gv; s1 += tp * rel[1] * gv; s2 += tp * rel[2] * gv;
    }
  }
  if (col < 10) { red[rl][col] = s0; red[rl][10 + col] = s1; red[rl][20 + col] = s2; }
  __syncthreads();
  if (threadIdx.x < 30) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (v != 0.0f) atomicAdd(dBrel + threadIdx.x, v);
  }
}

// c[q][ch] = has ? sum_k w[q][k] F[(q,k)][ch] : 0        and        dF[(q,k)][ch] = has ? w[q][k] dc[q][ch] : 0
__global__ __launch_bounds__(256) void wsum_fwd_kernel(const float* __restrict__ F, int ldf, const float* __restrict__ w,
                                                       const uint8_t* __restrict__ has, long Q, float* __restrict__ c,
                                                       int ldc) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long q = t >> 5;
  const int ch = (int)(t & 31);
  if (q >= Q) return;
  float acc = 0.0f;
  if (has[q]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += w[q * 8 + k] * F[(q * 8 + k) * ldf + ch];
  }
  c[q * ldc + ch] = acc;
}
__global__ __launch_bounds__(256) void wsum_bwd_kernel(const float* __restrict__ dc, int ldc, const float* __restrict__ w,
                                                       const uint8_t* __restrict__ has, long Q, float* __restrict__ dF,
                                                       int ldf) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long q = t >> 5;
  const int ch = (int)(t & 31);
  if (q >= Q) return;
  const float g = has[q] ? dc[q * ldc + ch] : 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) dF[(q * 8 + k) * ldf + ch] = w[q * 8 + k] * g;
}

// raw[q] = (rgb, has ? occ : -100)      (Renderer.py:206-207)
__global__ __launch_bounds__(256) void raw_pack_kernel(const float* __restrict__ rgb, int ldrgb, const float* __restrict__ occ,
                                                       const uint8_t* __restrict__ has, long Q, int color,
                                                       float* __restrict__ raw) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  float4 v = make_float4(0.f, 0.f, 0.f, has[q] ? occ[q] : -100.0f);
  if (color) { v.x = rgb[q * ldrgb]; v.y = rgb[q * ldrgb + 1]; v.z = rgb[q * ldrgb + 2]; }
  reinterpret_cast<float4*>(raw)[q] = v;
}
// d_raw [Q,4] -> dz[q][0..2] = g_rgb * rgb (1 - rgb) (sigmoid of the output layer), dz[q][3] = g_occ
__global__ __launch_bounds__(256) void raw_unpack_bwd_kernel(const float* __restrict__ draw, const float* __restrict__ rgb,
                                                             int ldrgb, const uint8_t* __restrict__ has, long Q, int color,
                                                             float* __restrict__ dz) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  const float4 g = reinterpret_cast<const float4*>(draw)[q];
  // the occupancy of a sample without neighbours is overwritten with -100 under no_grad (Renderer.py:206-207):
  // autograd does not see that assignment, the gradient computed at -100 still flows into the geometry decoder.
  // It matters: a ray whose samples ALL lack neighbours has weights ~4.5e-5 that are normalised by their sum.
  float4 o = make_float4(0.f, 0.f, 0.f, g.w);
  if (color) {
    const float y0 = rgb[q * ldrgb], y1 = rgb[q * ldrgb + 1], y2 = rgb[q * ldrgb + 2];
    o.x = g.x * y0 * (1.0f - y0); o.y = g.y * y1 * (1.0f - y1); o.z = g.z * y2 * (1.0f - y2);
  }
  reinterpret_cast<float4*>(dz)[q] = o;
}

// ---------------------------------------------------------------------------------------------------------
// compositing backward (common.py:261-299): one thread per ray
//   alpha_s = sigmoid(coef occ_s), T_s = prod_{j<s} (1 - alpha_j + 1e-10), w_s = alpha_s T_s, W = sum w + 1e-10
//   rgb = sum w c / W, depth = sum w z / W
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxS = 32;
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z_vals,
                                                            int R, int S, float coef, const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_rgb, float* __restrict__ draw) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float alpha[kMaxS], wgt[kMaxS];
  float T = 1.0f, W = 0.0f, cr = 0.f, cg = 0.f, cb = 0.f, dz = 0.f;
  for (int s = 0; s < S; ++s) {
    const float4 v = reinterpret_cast<const float4*>(raw)[(size_t)r * S + s];
    alpha[s] = 1.0f / (1.0f + expf(-coef * v.w));
    wgt[s] = alpha[s] * T;
    T *= (1.0f - alpha[s] + 1e-10f);
    W += wgt[s];
    cr += wgt[s] * v.x; cg += wgt[s] * v.y; cb += wgt[s] * v.z;
    dz += wgt[s] * z_vals[(size_t)r * S + s];
  }
  const float den = W + 1e-10f;
  const float mr = cr / den, mg = cg / den, mb = cb / den, md = dz / den;
  const float gd = g_depth ? g_depth[r] : 0.0f;
  const float gr = g_rgb ? g_rgb[(size_t)r * 3] : 0.f, gg = g_rgb ? g_rgb[(size_t)r * 3 + 1] : 0.f,
              gb = g_rgb ? g_rgb[(size_t)r * 3 + 2] : 0.f;
  float suffix = 0.0f;                 // sum_{s > j} g_w[s] w_s
  for (int j = S - 1; j >= 0; --j) {
    const float4 v = reinterpret_cast<const float4*>(raw)[(size_t)r * S + j];
    const float gw = (gd * (z_vals[(size_t)r * S + j] - md) + gr * (v.x - mr) + gg * (v.y - mg) + gb * (v.z - mb)) / den;
    // T_j = w_j / alpha_j would divide by ~0 for alpha -> 0: rebuild T_j as prod below instead
    float Tj = 1.0f;
    for (int k = 0; k < j; ++k) Tj *= (1.0f - alpha[k] + 1e-10f);
    const float ga = gw * Tj - suffix / (1.0f - alpha[j] + 1e-10f);
    float4 o;
    o.x = gr * wgt[j] / den; o.y = gg * wgt[j] / den; o.z = gb * wgt[j] / den;
    o.w = ga * coef * alpha[j] * (1.0f - alpha[j]);
    reinterpret_cast<float4*>(draw)[(size_t)r * S + j] = o;
    suffix += gw * wgt[j];
  }
}

// torch.optim.Adam (amsgrad = False, weight_decay = 0, maximize = False), one element per thread
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, const uint8_t* __restrict__ row_mask,
                                                   int row_len) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (row_mask && !row_mask[i / row_len]) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * mi / denom;
}

// the same with the step count in DEVICE memory (a mapping iteration recorded into a hipGraph: a count passed by value would
// be baked into the recording): the bias corrections are formed per thread from *step_dev
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                       const int* __restrict__ step_dev, const uint8_t* __restrict__ row_mask,
                                                       int row_len) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (row_mask && !row_mask[i / row_len]) return;
  const float step = (float)*step_dev;
  const float bc1 = 1.0f - powf(b1, step), bc2_sqrt = sqrtf(1.0f - powf(b2, step));
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * mi / denom;
}
__global__ void counter_add_kernel(int* ctr, int delta) { *ctr += delta; }

// the same update for MANY small tensors in one launch (the 52 decoder tensors of a mapping iteration: 52 launches of a few
// hundred elements each otherwise).  Table entry = 10 x 8 bytes: p, g, m, v, n, (lr, b1), (b2, eps), 3 x pad - nothing in it
// changes from step to step, so the host re-sends it only when a pointer moved;
// block (x, y) updates elements [1024 x, 1024 x + 1024) of tensor y.
struct AdamEntry { float* p; const float* g; float* m; float* v; long n; float lr, b1, b2, eps; long pad[3]; };
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamEntry* __restrict__ table, int step, const char* grad_base,
                                                         const int* __restrict__ step_dev) {
  if (step_dev) step = *step_dev;
  AdamEntry e = table[blockIdx.y];
  if (grad_base) e.g = reinterpret_cast<const float*>(grad_base + reinterpret_cast<size_t>(e.g));   // g = byte offset
  const long base = (long)blockIdx.x * 1024;
  if (base >= e.n) return;
  const float bc1 = 1.0f - powf(e.b1, (float)step);          // as glorie_adam_step forms them on the host
  const float bc2s = sqrtf(1.0f - powf(e.b2, (float)step));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long i = base + r * 256 + threadIdx.x;
    if (i >= e.n) break;
    const float gi = e.g[i];
    const float mi = e.b1 * e.m[i] + (1.0f - e.b1) * gi;
    const float vi = e.b2 * e.v[i] + (1.0f - e.b2) * gi * gi;
    e.m[i] = mi; e.v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + e.eps;
    e.p[i] -= (e.lr / bc1) * mi / denom;
  }
}

// dz[q][j] = dy[q][j] * act'(through the saved output y[q][j])
__global__ __launch_bounds__(256) void dact_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy,
                                                   int act, long Q, int J, float* __restrict__ dz, int ldz) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Q * J) return;
  const long q = idx / J;
  const int j = (int)(idx - q * J);
  dz[q * ldz + j] = dy[q * ldd + j] * t_dact(y[q * ldy + j], act);
}

// ---------------------------------------------------------------------------------------------------------
// host side: launch helpers
// ---------------------------------------------------------------------------------------------------------
static int pad16mod32(int v) { int p = (v + 15) / 16 * 16; if ((p & 31) != 16) p += 16; return p; }
static int mm(hipStream_t st, bool trans, const float* in, int ldi, const float* W, int ldw, const float* bias, int Q, int R,
              int J, int act, float* out, int ldo, const float* res = nullptr, int ldr = 0, const float* saved = nullptr,
              int lds = 0, int dact = 0, int accumulate = 0) {
  if (Q == 0) return GLORIE_OK;
  if (J > kMaxJT * 16) return GLORIE_EUNSUPPORTED;
  MmArgs a{in, ldi, W, ldw, bias, res, ldr, saved, lds, dact, out, ldo, Q, R, J, act, accumulate};
  // (staging the whole weight matrix once per persistent workgroup - no barrier in the reduction loop - was measured
  // slower: 5.39 vs 5.03 ms per mapping iteration; one workgroup per CU for the large layers)
  const dim3 grid((unsigned)((Q + 63) / 64));
  if (trans) hipLaunchKernelGGL(mm_rows_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(mm_rows_kernel<false>, grid, dim3(256), 0, st, a);
  return check_launch();
}
static int wgrad(hipStream_t st, const float* dY, int ldy, const float* X, int ldx, long Q, int N, int K, float* dW, int ldw,
                 float* db) {
  if (Q == 0 || !dW) return GLORIE_OK;
  const int KE = K + (db ? 1 : 0);
  if (N > 128 || KE > 256) return GLORIE_EUNSUPPORTED;
  // ~1024 workgroups at most, at least 256 rows each (keeps the atomics rare)
  long rows = (Q + 1023) / 1024;
  if (rows < 256) rows = 256;
  rows = (rows + 31) / 32 * 32;
  WgArgs a{dY, ldy, X, ldx, dW, ldw, db, Q, N, K, rows, 0, 0, 0, 0};
  const dim3 grid((unsigned)((Q + rows - 1) / rows));
  // columns of X per wave: KT tiles of 16, waves interleaved (kt = wave + 4 j)
  const int nnt = (N + 15) / 16, kt = ((KE + 15) / 16 + 3) / 4;
  if (nnt <= 2 && kt <= 1) hipLaunchKernelGGL((mm_wgrad_kernel<2, 1, 8>), grid, dim3(256), 0, st, a);
  else if (nnt <= 2 && kt <= 2) hipLaunchKernelGGL((mm_wgrad_kernel<2, 2, 8>), grid, dim3(256), 0, st, a);
  else if (nnt <= 2) hipLaunchKernelGGL((mm_wgrad_kernel<2, 4, 4>), grid, dim3(256), 0, st, a);
  else if (kt <= 1) hipLaunchKernelGGL((mm_wgrad_kernel<8, 1, 4>), grid, dim3(256), 0, st, a);
  else if (kt <= 2) hipLaunchKernelGGL((mm_wgrad_kernel<8, 2, 4>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((mm_wgrad_kernel<8, 4, 2>), grid, dim3(256), 0, st, a);
  return check_launch();
}

}  // namespace glorie

using namespace glorie;

// workspace layout (floats per sample); every buffer is [Q or 8Q][ld] row-major
namespace {
struct Ws {
  // geometry decoder
  float *g_c, *g_emb, *g_A[5], *g_H[5], *g_occ;     // H[2] is the tail of g_cat
  float* g_cat;                                     // [Q,125]: emb | H2
  // colour decoder
  float *n_X, *n_Z, *n_F, *c_c, *c_emb, *c_A[5], *c_H[5], *c_cat, *c_rgb;
  // backward temporaries
  float *t_a, *t_b, *t_c32, *t_n128, *t_n52, *t_n32, *t_q4, *t_cat, *t_emb;
  size_t total;
};
constexpr int G_HID = 32, G_EMB = 93, G_CAT = 125, C_HID = 128, C_EMB = 80, C_CAT = 208, NB_IN = 52;

Ws carve(float* base, long Q) {
  Ws w{};
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 3) / 4 * 4; return p; };
  const size_t q = (size_t)Q;
  w.g_c = take(q * 32); w.g_emb = take(q * G_EMB);
  for (int i = 0; i < 5; ++i) w.g_A[i] = take(q * G_HID);
  w.g_cat = take(q * G_CAT);
  for (int i = 0; i < 5; ++i) w.g_H[i] = (i == 2) ? nullptr : take(q * G_HID);
  w.g_occ = take(q);
  w.n_X = take(q * 8 * NB_IN); w.n_Z = take(q * 8 * C_HID); w.n_F = take(q * 8 * 32);
  w.c_c = take(q * 32); w.c_emb = take(q * C_EMB);
  for (int i = 0; i < 5; ++i) w.c_A[i] = take(q * C_HID);
  w.c_cat = take(q * C_CAT);
  for (int i = 0; i < 5; ++i) w.c_H[i] = (i == 2) ? nullptr : take(q * C_HID);
  w.c_rgb = take(q * 4);
  w.t_a = take(q * C_CAT); w.t_b = take(q * C_CAT); w.t_c32 = take(q * 32);
  w.t_n128 = take(q * 8 * C_HID); w.t_n52 = take(q * 8 * NB_IN); w.t_n32 = take(q * 8 * 32);
  w.t_q4 = take(q * 4); w.t_cat = take(q * C_CAT); w.t_emb = take(q * G_EMB);
  w.total = off;
  return w;
}
inline unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }

// The weight gradients of the backward pass run on a SECOND stream.  A layer's backward is  dW += dY^T X  (a reduction over all
// rows with fp32 atomics: 45-70 us at the sequence's 1000 rays) and  dX = dY W  (20 us) - both read dY, neither waits for the
// other, and at these sizes one of them does not fill the chip.  Every dY of a section gets its own buffer (no ping-pong), so
// the second stream only has to start behind the kernel that produced its dY (an event) and the first one only waits for it
// where a section's scratch is handed on (side_join).  The stream and the events are per device.
struct Side {
  hipStream_t s = nullptr;
  hipEvent_t fork[32];
  hipEvent_t join;
  std::atomic<unsigned> used{0};
};
// one Side per device (created on first use under a mutex): a process that drives a second GPU must not launch its
// weight gradients on a stream of the first one
int side_get(Side** out) {
  static Side table[16];
  static std::mutex mu;
  int dev = 0;
  GLORIE_TRY(check_hip(hipGetDevice(&dev)));
  if (dev < 0 || dev >= 16) return GLORIE_EUNSUPPORTED;
  Side& sd = table[dev];
  std::lock_guard<std::mutex> lock(mu);
  if (!sd.s) {
    hipStream_t s;
    GLORIE_TRY(check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)));
    for (auto& e : sd.fork) GLORIE_TRY(check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming)));
    GLORIE_TRY(check_hip(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming)));
    sd.s = s;
  }
  *out = &sd;
  return GLORIE_OK;
}
// the second stream, ordered behind everything `st` holds so far
int side_fork(hipStream_t st, hipStream_t* side) {
  Side* sd;
  GLORIE_TRY(side_get(&sd));
  hipEvent_t e = sd->fork[sd->used.fetch_add(1) % 32];
  GLORIE_TRY(check_hip(hipEventRecord(e, st)));
  GLORIE_TRY(check_hip(hipStreamWaitEvent(sd->s, e, 0)));
  *side = sd->s;
  return GLORIE_OK;
}
int side_join(hipStream_t st) {
  Side* sd;
  GLORIE_TRY(side_get(&sd));
  GLORIE_TRY(check_hip(hipEventRecord(sd->join, sd->s)));
  return check_hip(hipStreamWaitEvent(st, sd->join, 0));
}
}  // namespace

extern "C" size_t glorie_render_train_workspace(long Q) { return carve(nullptr, Q < 0 ? 0 : Q).total * sizeof(float); }

static int dact(hipStream_t st, const float* dy, int ldd, const float* y, int ldy, int act, long Q, int J, float* dz, int ldz) {
  if (Q == 0) return GLORIE_OK;
  hipLaunchKernelGGL(dact_kernel, dim3(blocks(Q * J)), dim3(256), 0, st, dy, ldd, y, ldy, act, Q, J, dz, ldz);
  return check_launch();
}

// the 5-layer trunk shared by both decoders (decoder.py:206-216, 414-426):
//   A_i = act(lin_i(h)); H_i = A_i + fc_c_i(c); after layer 2: h = cat(emb, H_2) (H_2 lives in the tail of `cat`)
static int trunk_fwd(hipStream_t st, long Q, int hid, int emb_w, int act, const float* emb, const float* c,
                     const float* const* Wl, const float* const* bl, const float* const* Ul, const float* const* ul,
                     float* const* A, float* const* H, float* cat) {
  const int catw = emb_w + hid;
  GLORIE_TRY(check_hip(hipMemcpy2DAsync(cat, sizeof(float) * catw, emb, sizeof(float) * emb_w, sizeof(float) * emb_w,
                                        (size_t)Q, hipMemcpyDeviceToDevice, st)));
  const float* h = emb;
  int hw = emb_w, ldh = emb_w;
  for (int i = 0; i < 5; ++i) {
    GLORIE_TRY(mm(st, true, h, ldh, Wl[i], hw, bl[i], (int)Q, hw, hid, act, A[i], hid));
    float* Hi = (i == 2) ? cat + emb_w : H[i];
    const int ldH = (i == 2) ? catw : hid;
    GLORIE_TRY(mm(st, true, c, 32, Ul[i], 32, ul[i], (int)Q, 32, hid, TACT_NONE, Hi, ldH, A[i], hid));
    if (i == 2) { h = cat; hw = catw; ldh = catw; }
    else { h = Hi; hw = hid; ldh = hid; }
  }
  return GLORIE_OK;
}

// backward of the trunk: dH_4 ([Q,hid] contiguous) -> parameter gradients (accumulated, on the second stream: the caller joins),
// dc [Q,32] (overwritten) and, if demb != NULL, the gradient of the embedding [Q,emb_w] (overwritten).
// scratch: 8 buffers [Q,hid] (every dH / dZ of the five layers keeps its own: the weight gradients read them later);
// dcat [Q,emb_w + hid]
static int trunk_bwd(hipStream_t st, long Q, int hid, int emb_w, int act, const float* emb, const float* c,
                     const float* const* Wl, const float* const* Ul, float* const* dWl, float* const* dbl,
                     float* const* dUl, float* const* dul, float* const* A, float* const* H, const float* cat,
                     float* dH4, float* scratch8, float* dcat, float* dc, float* demb) {
  const int catw = emb_w + hid;
  GLORIE_TRY(check_hip(hipMemsetAsync(dc, 0, sizeof(float) * (size_t)Q * 32, st)));
  const float* dh = dH4;
  int ldd = hid;
  hipStream_t sd;
  for (int i = 4; i >= 0; --i) {
    float* dz = scratch8 + (size_t)i * Q * hid;                    // dZ_i
    // H_i = A_i + c U_i^T + u_i
    GLORIE_TRY(side_fork(st, &sd));
    GLORIE_TRY(wgrad(sd, dh, ldd, c, 32, Q, hid, 32, dUl[i], 32, dul[i]));
    GLORIE_TRY(mm(st, false, dh, ldd, Ul[i], 32, nullptr, (int)Q, hid, 32, TACT_NONE, dc, 32, nullptr, 0, nullptr, 0, 0, 1));
    // A_i = act(Z_i):  dZ_i = dH_i * act'(A_i)
    GLORIE_TRY(dact(st, dh, ldd, A[i], hid, act, Q, hid, dz, hid));
    // Z_i = hin W_i^T + b_i
    const float* hin; int hw, ldh;
    if (i == 0) { hin = emb; hw = emb_w; ldh = emb_w; }
    else if (i == 3) { hin = cat; hw = catw; ldh = catw; }
    else if (i == 4) { hin = H[3]; hw = hid; ldh = hid; }
    else { hin = H[i - 1]; hw = hid; ldh = hid; }              // i = 1, 2: H_0, H_1
    GLORIE_TRY(side_fork(st, &sd));
    GLORIE_TRY(wgrad(sd, dz, hid, hin, ldh, Q, hid, hw, dWl[i], hw, dbl[i]));
    if (i == 0) {
      if (demb) GLORIE_TRY(mm(st, false, dz, hid, Wl[0], emb_w, nullptr, (int)Q, hid, emb_w, TACT_NONE, demb, emb_w,
                              nullptr, 0, nullptr, 0, 0, /*accumulate=*/1));
    } else if (i == 3) {
      GLORIE_TRY(mm(st, false, dz, hid, Wl[3], catw, nullptr, (int)Q, hid, catw, TACT_NONE, dcat, catw));
      if (demb)   // the embedding half of the skip connection; layer 0 adds its part on top
        GLORIE_TRY(check_hip(hipMemcpy2DAsync(demb, sizeof(float) * emb_w, dcat, sizeof(float) * catw,
                                              sizeof(float) * emb_w, (size_t)Q, hipMemcpyDeviceToDevice, st)));
      dh = dcat + emb_w; ldd = catw;
    } else {
      float* nxt = scratch8 + (size_t)(5 + (i == 4 ? 0 : (i == 2 ? 1 : 2))) * Q * hid;      // dH_3, dH_1, dH_0
      GLORIE_TRY(mm(st, false, dz, hid, Wl[i], hid, nullptr, (int)Q, hid, hid, TACT_NONE, nxt, hid));
      dh = nxt; ldd = hid;
    }
  }
  return GLORIE_OK;
}

static bool params_ok(const glorie_decoder_params* p, bool color) {
  if (!p || !p->g_B || !p->g_Wo || !p->g_bo) return false;
  for (int i = 0; i < 5; ++i) if (!p->g_W[i] || !p->g_b[i] || !p->g_U[i] || !p->g_u[i]) return false;
  if (!color) return true;
  if (!p->n_B || !p->n_W1 || !p->n_b1 || !p->n_W2 || !p->n_b2 || !p->c_Bp || !p->c_Bv || !p->c_Wo || !p->c_bo) return false;
  for (int i = 0; i < 5; ++i) if (!p->c_W[i] || !p->c_b[i] || !p->c_U[i] || !p->c_u[i]) return false;
  return true;
}

extern "C" int glorie_render_train_fwd(const glorie_decoder_params* P, const float* pts, const float* views,
                                       const float* cloud_pos, const float* geo_feats, const float* col_feats,
                                       const int64_t* I, const float* w, const uint8_t* has, long Q, int stage_color,
                                       float* workspace, float* raw, void* stream) {
  if (Q < 0) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!params_ok(P, stage_color != 0) || !pts || !geo_feats || !I || !w || !has || !workspace || !raw) return GLORIE_EINVAL;
  if (stage_color && (!views || !cloud_pos || !col_feats)) return GLORIE_EINVAL;
  if (Q > (1L << 27)) return GLORIE_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Ws W = carve(workspace, Q);
  // ---- geometry decoder (decoder.py:175-225) ----
  hipLaunchKernelGGL(idw_fwd_kernel, dim3(blocks(Q * 32)), dim3(256), 0, st, geo_feats, I, w, has, Q, W.g_c, 32);
  hipLaunchKernelGGL(fourier_fwd_kernel, dim3(blocks(Q * G_EMB)), dim3(256), 0, st, pts, 3, 0, P->g_B, G_EMB, 0, Q, W.g_emb, G_EMB);
  GLORIE_TRY(check_launch());
  float* gH[5] = {W.g_H[0], W.g_H[1], W.g_cat + G_EMB, W.g_H[3], W.g_H[4]};
  GLORIE_TRY(trunk_fwd(st, Q, G_HID, G_EMB, TACT_RELU, W.g_emb, W.g_c, P->g_W, P->g_b, P->g_U, P->g_u, W.g_A, gH, W.g_cat));
  GLORIE_TRY(mm(st, true, W.g_H[4], G_HID, P->g_Wo, G_HID, P->g_bo, (int)Q, G_HID, 1, TACT_NONE, W.g_occ, 1));
  if (stage_color) {
    // ---- per-neighbour F_theta + IDW sum (decoder.py:340-389, 228-243) ----
    hipLaunchKernelGGL(nb_rows_fwd_kernel, dim3(blocks(Q * 8 * 64)), dim3(256), 0, st, pts, cloud_pos, col_feats, I, P->n_B, Q, W.n_X, NB_IN);
    GLORIE_TRY(check_launch());
    GLORIE_TRY(mm(st, true, W.n_X, NB_IN, P->n_W1, NB_IN, P->n_b1, (int)(Q * 8), NB_IN, C_HID, TACT_SOFTPLUS, W.n_Z, C_HID));
    GLORIE_TRY(mm(st, true, W.n_Z, C_HID, P->n_W2, C_HID, P->n_b2, (int)(Q * 8), C_HID, 32, TACT_NONE, W.n_F, 32));
    hipLaunchKernelGGL(wsum_fwd_kernel, dim3(blocks(Q * 32)), dim3(256), 0, st, W.n_F, 32, w, has, Q, W.c_c, 32);
    // ---- colour decoder (decoder.py:391-433) ----
    hipLaunchKernelGGL(fourier_fwd_kernel, dim3(blocks(Q * 20)), dim3(256), 0, st, pts, 3, 0, P->c_Bp, 20, 1, Q, W.c_emb, C_EMB);
    hipLaunchKernelGGL(fourier_fwd_kernel, dim3(blocks(Q * 20)), dim3(256), 0, st, views, 3, 1, P->c_Bv, 20, 1, Q, W.c_emb + 40, C_EMB);
    GLORIE_TRY(check_launch());
    float* cH[5] = {W.c_H[0], W.c_H[1], W.c_cat + C_EMB, W.c_H[3], W.c_H[4]};
    GLORIE_TRY(trunk_fwd(st, Q, C_HID, C_EMB, TACT_SOFTPLUS, W.c_emb, W.c_c, P->c_W, P->c_b, P->c_U, P->c_u, W.c_A, cH, W.c_cat));
    GLORIE_TRY(mm(st, true, W.c_H[4], C_HID, P->c_Wo, C_HID, P->c_bo, (int)Q, C_HID, 3, TACT_SIGMOID, W.c_rgb, 4));
  }
  hipLaunchKernelGGL(raw_pack_kernel, dim3(blocks(Q)), dim3(256), 0, st, W.c_rgb, 4, W.g_occ, has, Q, stage_color ? 1 : 0, raw);
  return check_launch();
}

extern "C" int glorie_render_train_bwd(const glorie_decoder_params* P, const glorie_decoder_grads* G, const float* pts,
                                       const float* views, const float* cloud_pos, const float* geo_feats,
                                       const float* col_feats, const int64_t* I, const float* w, const uint8_t* has,
                                       long Q, int stage_color, float* workspace, const float* d_raw,
                                       float* d_geo_feats, float* d_col_feats, void* stream) {
  if (Q < 0) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!params_ok(P, stage_color != 0) || !G || !pts || !I || !w || !has || !workspace || !d_raw) return GLORIE_EINVAL;
  if (stage_color && (!views || !cloud_pos || !col_feats)) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Ws W = carve(workspace, Q);
  // d_raw -> d(occ), d(pre-sigmoid rgb)
  hipLaunchKernelGGL(raw_unpack_bwd_kernel, dim3(blocks(Q)), dim3(256), 0, st, d_raw, W.c_rgb, 4, has, Q, stage_color ? 1 : 0,
                     W.t_q4);
  GLORIE_TRY(check_launch());
  const float* docc = W.t_q4 + 3;         // column 3 of the [Q,4] scratch (row stride 4)
  // ---- geometry decoder ----
  {
    // occ = H_4 Wo^T + bo
    hipStream_t sd;
    GLORIE_TRY(side_fork(st, &sd));
    GLORIE_TRY(wgrad(sd, docc, 4, W.g_H[4], G_HID, Q, 1, G_HID, G->g_Wo, G_HID, G->g_bo));
    float* dH4 = W.t_a;
    GLORIE_TRY(mm(st, false, docc, 4, P->g_Wo, G_HID, nullptr, (int)Q, 1, G_HID, TACT_NONE, dH4, G_HID));
    float* gH[5] = {W.g_H[0], W.g_H[1], W.g_cat + G_EMB, W.g_H[3], W.g_H[4]};
    // scratch: the eight [Q,32] buffers of the trunk live in the (still unused) neighbour temporaries
    GLORIE_TRY(trunk_bwd(st, Q, G_HID, G_EMB, TACT_RELU, W.g_emb, W.g_c, P->g_W, P->g_U, G->g_W, G->g_b, G->g_U, G->g_u,
                         W.g_A, gH, W.g_cat, dH4, W.t_n128, W.t_cat, W.t_c32, G->g_B ? W.t_emb : nullptr));
    if (G->g_B) {
      const int rpb = 256;
      hipLaunchKernelGGL(fourier_bwd_kernel, dim3((unsigned)((Q + rpb - 1) / rpb)), dim3(256), 0, st, pts, 3, P->g_B, G_EMB, 0, Q,
                         W.t_emb, G_EMB, G->g_B, rpb);
    }
    if (d_geo_feats)
      hipLaunchKernelGGL(idw_bwd_kernel, dim3(blocks(Q * 32)), dim3(256), 0, st, W.t_c32, 32, I, w, has, Q, d_geo_feats);
    GLORIE_TRY(check_launch());
    GLORIE_TRY(side_join(st));             // the colour decoder reuses t_a, t_q4 stays, t_n128 is handed on
  }
  if (!stage_color) return GLORIE_OK;
  // ---- colour decoder ----
  {
    // rgb = sigmoid(H_4 Wo^T + bo): t_q4[:, 0:3] already holds dZ
    hipStream_t sd;
    GLORIE_TRY(side_fork(st, &sd));
    GLORIE_TRY(wgrad(sd, W.t_q4, 4, W.c_H[4], C_HID, Q, 3, C_HID, G->c_Wo, C_HID, G->c_bo));
    float* dH4 = W.t_a;
    GLORIE_TRY(mm(st, false, W.t_q4, 4, P->c_Wo, C_HID, nullptr, (int)Q, 3, C_HID, TACT_NONE, dH4, C_HID));
    float* cH[5] = {W.c_H[0], W.c_H[1], W.c_cat + C_EMB, W.c_H[3], W.c_H[4]};
    // scratch: the eight [Q,128] buffers of the trunk are exactly the [8Q,128] neighbour temporary
    GLORIE_TRY(trunk_bwd(st, Q, C_HID, C_EMB, TACT_SOFTPLUS, W.c_emb, W.c_c, P->c_W, P->c_U, G->c_W, G->c_b, G->c_U, G->c_u,
                         W.c_A, cH, W.c_cat, dH4, W.t_n128, W.t_cat, W.t_c32, nullptr));
    // c = has ? sum_k w F : 0;  F = Z L2^T + b2;  Z = softplus(X L1^T + b1)
    hipLaunchKernelGGL(wsum_bwd_kernel, dim3(blocks(Q * 32)), dim3(256), 0, st, W.t_c32, 32, w, has, Q, W.t_n32, 32);
    GLORIE_TRY(check_launch());
    GLORIE_TRY(side_join(st));             // the trunk's weight gradients read t_n128, which the next launch overwrites
    GLORIE_TRY(side_fork(st, &sd));
    GLORIE_TRY(wgrad(sd, W.t_n32, 32, W.n_Z, C_HID, Q * 8, 32, C_HID, G->n_W2, C_HID, G->n_b2));
    GLORIE_TRY(mm(st, false, W.t_n32, 32, P->n_W2, C_HID, nullptr, (int)(Q * 8), 32, C_HID, TACT_NONE, W.t_n128, C_HID, nullptr, 0,
                  W.n_Z, C_HID, TACT_SOFTPLUS));
    GLORIE_TRY(side_fork(st, &sd));
    GLORIE_TRY(wgrad(sd, W.t_n128, C_HID, W.n_X, NB_IN, Q * 8, C_HID, NB_IN, G->n_W1, NB_IN, G->n_b1));
    GLORIE_TRY(mm(st, false, W.t_n128, C_HID, P->n_W1, NB_IN, nullptr, (int)(Q * 8), C_HID, NB_IN, TACT_NONE, W.t_n52, NB_IN));
    if (d_col_feats || G->n_B) {
      // one launch serves both; a NULL target is replaced by a scratch sink
      float* sinkB = G->n_B ? G->n_B : W.t_q4;          // 30 floats
      if (!d_col_feats) return GLORIE_EINVAL;
      // rows per workgroup: ~2048 workgroups (fixed 1024 rows left the chip a quarter full at the mapper's 5000 rays and gave
      // 78 workgroups at 1000), at least 64 rows each so that the 30 atomics per workgroup on dB_rel stay rare
      long rpb = ((Q * 8 + 2047) / 2048 + 3) & ~3L;
      rpb = rpb < 64 ? 64 : (rpb > 1024 ? 1024 : rpb);
      hipLaunchKernelGGL(nb_rows_bwd_kernel, dim3((unsigned)((Q * 8 + rpb - 1) / rpb)), dim3(256), 0, st, W.t_n52, NB_IN, pts,
                         cloud_pos, I, P->n_B, Q, d_col_feats, sinkB, rpb);
      GLORIE_TRY(check_launch());
    }
  }
  return side_join(st);                    // the gradients (and the workspace) belong to the caller's stream again
}

extern "C" int glorie_composite_bwd(const float* raw, const float* z_vals, int R, int S, float coef, const float* g_depth,
                                    const float* g_rgb, float* d_raw, void* stream) {
  if (R < 0 || S < 1 || S > kMaxS) return GLORIE_EINVAL;
  if (R == 0) return GLORIE_OK;
  if (!raw || !z_vals || !d_raw) return GLORIE_EINVAL;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(blocks(R)), dim3(256), 0, (hipStream_t)stream, raw, z_vals, R, S, coef,
                     g_depth, g_rgb, d_raw);
  return check_launch();
}

extern "C" int glorie_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                float beta1, float beta2, float eps, int step, const uint8_t* row_mask, int row_len,
                                void* stream) {
  if (n < 0 || step < 1) return GLORIE_EINVAL;
  if (n == 0) return GLORIE_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || (row_mask && row_len < 1)) return GLORIE_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, bc1, bc2s, row_mask, row_len > 0 ? row_len : 1);
  return check_launch();
}

extern "C" int glorie_adam_multi(const void* table, int n_tensors, long max_numel, int step, const void* grad_base,
                                 void* stream) {
  if (n_tensors < 0 || max_numel < 0 || step < 1) return GLORIE_EINVAL;
  if (n_tensors == 0 || max_numel == 0) return GLORIE_OK;
  if (!table) return GLORIE_EINVAL;
  static_assert(sizeof(AdamEntry) == 80, "table layout");
  const dim3 grid((unsigned)((max_numel + 1023) / 1024), (unsigned)n_tensors);
  hipLaunchKernelGGL(adam_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const AdamEntry*>(table), step,
                     reinterpret_cast<const char*>(grad_base), (const int*)nullptr);
  return check_launch();
}

extern "C" int glorie_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                    float beta1, float beta2, float eps, const int* step_dev, const uint8_t* row_mask,
                                    int row_len, void* stream) {
  if (n < 0) return GLORIE_EINVAL;
  if (n == 0) return GLORIE_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || (row_mask && row_len < 1)) return GLORIE_EINVAL;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, step_dev, row_mask, row_len > 0 ? row_len : 1);
  return check_launch();
}

extern "C" int glorie_adam_multi_dev(const void* table, int n_tensors, long max_numel, const int* step_dev,
                                     const void* grad_base, void* stream) {
  if (n_tensors < 0 || max_numel < 0) return GLORIE_EINVAL;
  if (n_tensors == 0 || max_numel == 0) return GLORIE_OK;
  if (!table || !step_dev) return GLORIE_EINVAL;
  const dim3 grid((unsigned)((max_numel + 1023) / 1024), (unsigned)n_tensors);
  hipLaunchKernelGGL(adam_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const AdamEntry*>(table), 1,
                     reinterpret_cast<const char*>(grad_base), step_dev);
  return check_launch();
}

extern "C" int glorie_counter_add(int* counter, int delta, void* stream) {
  if (!counter) return GLORIE_EINVAL;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
  return check_launch();
}
