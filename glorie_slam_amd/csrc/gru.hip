// Fused element-wise stages of the ConvGRU update operator (scope row A4) for gfx950.
//
// The convolutions of UpdateModule (/root/reference/src/modules/droid_net/droid_net.py:69-139,
// gru.py:20-34) run through MIOpen; everything BETWEEN them -- bias add, ReLU / sigmoid / tanh,
// the global-context term, r*net, the [net|inp|corr|flow] concatenation and the final blend --
// is a chain of ~40 PyTorch element-wise launches per update (25 % of a BA-update).  These
// kernels collapse that chain: activations are channels-last fp16 ([pixels][C], what MIOpen's
// implicit-GEMM kernels consume natively), every thread moves 8 halfs (16 B), math is fp32 in
// registers with one rounding to fp16 on store, and outputs land directly in channel slices of
// the 448-channel GRU input buffer, so no torch.cat pass exists.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(8))) _Float16 h8;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) / (1.0f + e);
  return copysignf(t, x);
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_SOFTPLUS = 3 };

// y[p][0..C) = act(x[p][0..C) + bias) ; rows strided (slices of wider channels-last buffers).
// C8 = C/8 is a template parameter for the widths the update operator uses (a runtime 64-bit
// division per thread costs more than the 32 bytes the thread moves); C8T = 0 -> generic.
template <int C8T>
__global__ __launch_bounds__(256) void bias_act_kernel(const _Float16* __restrict__ x, int xs,
                                                       const float* __restrict__ bias,
                                                       _Float16* __restrict__ y, int ys, long P,
                                                       int C8R, int act) {
  const int C8 = C8T ? C8T : C8R;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * C8) return;
  long p;
  int c;
  if (C8T) {
    p = idx / C8T;
    c = (int)(idx - p * C8T) * 8;
  } else {
    p = (long)((unsigned long)idx / (unsigned)C8);
    c = (int)(idx - p * C8) * 8;
  }
  const h8 v = *reinterpret_cast<const h8*>(x + p * xs + c);
  float b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + c + 4);
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
    b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float f = (float)v[k] + b[k];
    if (act == ACT_RELU) f = fmaxf(f, 0.0f);
    else if (act == ACT_SIGMOID) f = sigmoidf_(f);
    o[k] = (_Float16)f;
  }
  *reinterpret_cast<h8*>(y + p * ys + c) = o;
}

// Global-context terms of the three gates (gru.py:25-31):
//   glo[n][c] = mean_p sigmoid(wn[p][c] + bw[c]) * net[p][c]
//   g[n][o]   = Gb[o] + sum_c glo[n][c] * G[c][o]          o < M (= 384: z | r | q terms)
// Pass 1, grid (N, parts): partial sums over a slice of the pixels of edge n -> partial[n][part][128].
// Pass 2, grid (N): fixed-order sum of the parts (deterministic, no atomics) + the [128 x M] product.
__global__ __launch_bounds__(256) void glo_partial_kernel(const _Float16* __restrict__ wn, int ws,
                                                          const float* __restrict__ bw,
                                                          const _Float16* __restrict__ net, int ns,
                                                          float* __restrict__ partial, int HW) {
  __shared__ float red[16][128];
  const int n = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
  const int c = (threadIdx.x & 15) * 8, prow = threadIdx.x >> 4;   // 16 pixel rows x 16 channel groups
  const int per = (HW + parts - 1) / parts;
  const int p0 = part * per, p1 = min(HW, p0 + per);
  float bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias[k] = bw[c + k];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = p0 + prow; p < p1; p += 16) {
    const long row = (long)n * HW + p;
    const h8 a = *reinterpret_cast<const h8*>(wn + row * ws + c);
    const h8 b = *reinterpret_cast<const h8*>(net + row * ns + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += sigmoidf_((float)a[k] + bias[k]) * (float)b[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[prow][c + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 128) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
    partial[((size_t)n * parts + part) * 128 + threadIdx.x] = s;
  }
}

// grid (N, M/128): every block re-sums the parts of its edge (2 KB each) and produces 128 outputs
__global__ __launch_bounds__(128) void glo_terms_kernel(const float* __restrict__ partial, int parts,
                                                        const float* __restrict__ G,
                                                        const float* __restrict__ Gb, int M,
                                                        float* __restrict__ g, int HW) {
  __shared__ float glo[128];
  const int n = blockIdx.x;
  float s = 0.0f;
  for (int q = 0; q < parts; ++q) s += partial[((size_t)n * parts + q) * 128 + threadIdx.x];
  glo[threadIdx.x] = s / (float)HW;
  __syncthreads();
  const int o = blockIdx.y * 128 + threadIdx.x;
  if (o >= M) return;
  float acc0 = Gb[o], acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
#pragma unroll 8
  for (int c = 0; c < 128; c += 4) {
    acc0 = fmaf(glo[c], G[(size_t)c * M + o], acc0);
    acc1 = fmaf(glo[c + 1], G[(size_t)(c + 1) * M + o], acc1);
    acc2 = fmaf(glo[c + 2], G[(size_t)(c + 2) * M + o], acc2);
    acc3 = fmaf(glo[c + 3], G[(size_t)(c + 3) * M + o], acc3);
  }
  g[(size_t)n * M + o] = (acc0 + acc1) + (acc2 + acc3);
}

// z = sigmoid(zc + gz[n]); rnet = sigmoid(rc + gr[n]) * net       (gru.py:28-30)
// zr: raw output of the merged (convz|convr) convolution, [P][256]; g: [N][256] = glo terms + biases
__global__ __launch_bounds__(256) void gru_gate_zr_kernel(const _Float16* __restrict__ zr, int zs,
                                                          const float* __restrict__ g, int gs,
                                                          const _Float16* __restrict__ net, int ns,
                                                          _Float16* __restrict__ z, int zos,
                                                          _Float16* __restrict__ rnet, int rs, long P,
                                                          int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * 16) return;
  const long p = idx >> 4;
  const int c = (int)(idx & 15) * 8;
  const int n = (int)(p / HW);
  const h8 zc = *reinterpret_cast<const h8*>(zr + p * zs + c);
  const h8 rc = *reinterpret_cast<const h8*>(zr + p * zs + 128 + c);
  const h8 nv = *reinterpret_cast<const h8*>(net + p * ns + c);
  h8 zo, ro;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    zo[k] = (_Float16)sigmoidf_((float)zc[k] + g[(size_t)n * gs + c + k]);
    ro[k] = (_Float16)(sigmoidf_((float)rc[k] + g[(size_t)n * gs + 128 + c + k]) * (float)nv[k]);
  }
  *reinterpret_cast<h8*>(z + p * zos + c) = zo;
  *reinterpret_cast<h8*>(rnet + p * rs + c) = ro;
}

// net' = (1 - z) * net + z * tanh(qc + gq[n])                      (gru.py:31-33)
// out2 (optional): second copy, e.g. the net slice of the GRU input buffer of the next update
__global__ __launch_bounds__(256) void gru_gate_q_kernel(const _Float16* __restrict__ qc, int qs,
                                                         const float* __restrict__ gq, int gs,
                                                         const _Float16* __restrict__ z, int zs,
                                                         const _Float16* __restrict__ net, int ns,
                                                         _Float16* __restrict__ out, int os,
                                                         _Float16* __restrict__ out2, int os2, long P,
                                                         int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * 16) return;
  const long p = idx >> 4;
  const int c = (int)(idx & 15) * 8;
  const int n = (int)(p / HW);
  const h8 qv = *reinterpret_cast<const h8*>(qc + p * qs + c);
  const h8 zv = *reinterpret_cast<const h8*>(z + p * zs + c);
  const h8 nv = *reinterpret_cast<const h8*>(net + p * ns + c);
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float q = tanhf_((float)qv[k] + gq[(size_t)n * gs + c + k]);
    const float zz = (float)zv[k];
    o[k] = (_Float16)((1.0f - zz) * (float)nv[k] + zz * q);
  }
  *reinterpret_cast<h8*>(out + p * os + c) = o;
  if (out2) *reinterpret_cast<h8*>(out2 + p * os2 + c) = o;
}

// out[g][p][c] = mean over the edges e with ix[e] == g of act(x[e][p][c] + bias[c])
// (scatter_mean of GraphAgg, droid_net.py:53-59).  The workgroup first compacts the ids of its group's edges,
// 256 candidates at a time and in ascending order (ballot + prefix count), then every thread walks that
// short list: fixed summation order, no atomics, no zero fill, and no O(N) scan per thread on graphs with
// thousands of edges.  grid (HW*16/256, G).
__global__ __launch_bounds__(256) void segment_mean_kernel(const _Float16* __restrict__ x, int xs,
                                                           const float* __restrict__ bias, int relu,
                                                           const int64_t* __restrict__ ix, int N,
                                                           _Float16* __restrict__ out, int os, int HW) {
  __shared__ int elist[256];
  __shared__ int wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int idx = blockIdx.x * 256 + tid;
  const bool live = idx < HW * 16;
  const int p = idx >> 4, c = (idx & 15) * 8, g = blockIdx.y;
  float b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (bias) {
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = bias[c + k];
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int cnt = 0;
  for (int base = 0; base < N; base += 256) {
    const int e = base + tid;
    const bool hit = e < N && ix[e] == g;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < wv) off += wcnt[q];
      total += wcnt[q];
    }
    if (hit) elist[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
    __syncthreads();
    if (live) {
      for (int q = 0; q < total; ++q) {
        const h8 v = *reinterpret_cast<const h8*>(x + ((long)elist[q] * HW + p) * xs + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f = (float)v[k] + b[k];
          if (relu) f = fmaxf(f, 0.0f);
          acc[k] += f;
        }
      }
    }
    cnt += total;
    __syncthreads();              // elist / wcnt are rewritten by the next batch
  }
  if (!live) return;
  const float inv = 1.0f / (float)max(cnt, 1);
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (_Float16)(acc[k] * inv);
  *reinterpret_cast<h8*>(out + ((long)g * HW + p) * os + c) = o;
}

// ---------------------------------------------------------------------------------------------
// 3x3 convolution 128 -> K (K <= 3) channels: the flow-revision / confidence heads and the damping
// head (droid_net.py:85-93,42-44).  A GEMM library pads K to a 32..128 wide tile (80 us per head on
// MIOpen for 44 MB of input); here the convolution is split into
//   taps[p][d*K+j] = < act(x[p] + in_bias), w[j][:, d] >      one MFMA pass over the input, and
//   out[p][j]      = act(bias[j] + sum_d taps[p + off(d)][d*K+j]) * scale   a 9-point stencil on
// the (tiny) tap planes.  `groups` independent heads read consecutive 128-channel slices of x.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NT>
__global__ __launch_bounds__(256) void conv_taps_kernel(const _Float16* __restrict__ x, int xs,
                                                        const float* __restrict__ in_bias, int in_relu,
                                                        const f16x8* __restrict__ wp,
                                                        float* __restrict__ taps, int ncols, int tstride,
                                                        long P) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int grp = blockIdx.y;
  f16x8 bfrag[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bfrag[t][kk] = wp[(((size_t)grp * NT + t) * 4 + kk) * 64 + lane];
  float ib[4][8];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int k = 0; k < 8; ++k) ib[kk][k] = in_bias ? in_bias[grp * 128 + kk * 32 + kg * 8 + k] : 0.0f;
  const long ntiles = (P + 15) / 16;
  for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long p0 = tile * 16;
    const long pa = min(p0 + col, P - 1);
    const f16x8* src = reinterpret_cast<const f16x8*>(x + pa * xs + grp * 128 + kg * 8);
    f16x8 afrag[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) afrag[kk] = src[kk * 4];
    if (in_bias || in_relu) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f = (float)afrag[kk][k] + ib[kk][k];
          if (in_relu) f = fmaxf(f, 0.0f);
          afrag[kk][k] = (_Float16)f;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[kk], bfrag[t][kk], acc, 0, 0, 0);
      const int n = t * 16 + col;
      if (n < ncols) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long pr = p0 + kg * 4 + r;
          if (pr < P) taps[pr * tstride + grp * ncols + n] = acc[r];
        }
      }
    }
  }
}

__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(__expf(x)); }

// PLANAR: taps = [groups*ncols][P] (written by glorie_conv_igemm_heads; a wave reads 64 consecutive pixels of one tap
// plane), else rows [P][groups*ncols] (conv_taps_kernel)
template <bool PLANAR>
__global__ __launch_bounds__(256) void conv_stencil_kernel(const float* __restrict__ taps, int ncols,
                                                           int tstride, const float* __restrict__ bias,
                                                           int K, int groups, int act_packed, float scale,
                                                           float* __restrict__ out, long P, int H, int W,
                                                           float* __restrict__ out_last = nullptr) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * K * groups) return;
  const int gk = PLANAR ? (int)(idx / P) : (int)(idx % (K * groups));
  const long p = PLANAR ? idx - (long)gk * P : idx / (K * groups);
  const int grp = gk / K, j = gk - grp * K;
  const int xw = (int)(p % W), yh = (int)((p / W) % H);
  float acc = bias ? bias[grp * K + j] : 0.0f;
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    const int dy = d / 3 - 1, dx = d % 3 - 1;
    if ((unsigned)(yh + dy) < (unsigned)H && (unsigned)(xw + dx) < (unsigned)W)
      acc += PLANAR ? taps[(size_t)(grp * ncols + d * K + j) * P + (p + dy * W + dx)]
                    : taps[(p + dy * W + dx) * tstride + grp * ncols + d * K + j];
  }
  const int act = (act_packed >> (4 * grp)) & 15;
  if (act == ACT_RELU) acc = fmaxf(acc, 0.0f);
  else if (act == ACT_SIGMOID) acc = sigmoidf_(acc);
  else if (act == ACT_SOFTPLUS) acc = softplusf_(acc);
  // out_last: the last group goes to a tensor of its own ([P][K]; e.g. the confidence weights straight into the caller's buffer)
  if (out_last && grp == groups - 1) out_last[(size_t)p * K + j] = acc * scale;
  else out[((size_t)grp * P + p) * K + j] = acc * scale;
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_bias_act(const void* x, int x_stride, const float* bias, void* y, int y_stride,
                               long P, int C, int act, void* stream) {
  if (P < 0 || C <= 0 || (C & 7) || (x_stride & 7) || (y_stride & 7)) return GLORIE_EINVAL;
  if (P == 0) return GLORIE_OK;
  if (!x || !y) return GLORIE_EINVAL;
  const long total = P * (C / 8);
  const dim3 grid((unsigned)((total + 255) / 256));
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(x), x_stride, bias,
                       reinterpret_cast<_Float16*>(y), y_stride, P, C / 8, act);
  };
  switch (C / 8) {
    case 8: launch(bias_act_kernel<8>); break;
    case 16: launch(bias_act_kernel<16>); break;
    case 48: launch(bias_act_kernel<48>); break;
    default: launch(bias_act_kernel<0>); break;
  }
  return check_launch();
}

// The same terms from the per-tile partial sums of glorie_conv_igemm's epilogue 3 (tiles [N][ceil(HW / 128)][128]: every
// map tiled on its own): map n sums its tiles in ascending order (deterministic), divides by HW and applies the [128 x M]
// product.
__global__ __launch_bounds__(128) void glo_from_tiles_kernel(const float* __restrict__ tiles, const float* __restrict__ G,
                                                             const float* __restrict__ Gb, int M,
                                                             float* __restrict__ g, int HW) {
  __shared__ float glo[128];
  const int n = blockIdx.x;
  const int tpm = (HW + 127) >> 7;
  // latency-bound (one small block per map and 128 outputs): loads are issued 8 / 16 at a time, the sums keep their order
  const float* tp = tiles + (size_t)n * tpm * 128 + threadIdx.x;
  float s = 0.0f;
  int t = 0;
  for (; t + 8 <= tpm; t += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tp[(size_t)(t + k) * 128];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; t < tpm; ++t) s += tp[(size_t)t * 128];
  const int o = blockIdx.y * 128 + threadIdx.x;
  const int oc = min(o, M - 1);
  float gv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) gv[k] = G[(size_t)k * M + oc];          // first batch of G in flight across the barrier
  glo[threadIdx.x] = s / (float)HW;
  __syncthreads();
  if (o >= M) return;
  float acc0 = Gb[o], acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
#pragma unroll
  for (int c0 = 0; c0 < 128; c0 += 16) {
    float nx[16];
    if (c0 + 16 < 128) {
#pragma unroll
      for (int k = 0; k < 16; ++k) nx[k] = G[(size_t)(c0 + 16 + k) * M + o];
    }
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      acc0 = fmaf(glo[c0 + k], gv[k], acc0);
      acc1 = fmaf(glo[c0 + k + 1], gv[k + 1], acc1);
      acc2 = fmaf(glo[c0 + k + 2], gv[k + 2], acc2);
      acc3 = fmaf(glo[c0 + k + 3], gv[k + 3], acc3);
    }
    if (c0 + 16 < 128) {
#pragma unroll
      for (int k = 0; k < 16; ++k) gv[k] = nx[k];
    }
  }
  g[(size_t)n * M + o] = (acc0 + acc1) + (acc2 + acc3);
}

extern "C" int glorie_gru_glo_from_tiles(const float* tiles, const float* G, const float* Gb, int M, float* g, int N,
                                         int HW, void* stream) {
  if (N < 0 || HW < 1 || M <= 0) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!tiles || !G || !Gb || !g) return GLORIE_EINVAL;
  hipLaunchKernelGGL(glo_from_tiles_kernel, dim3(N, (M + 127) / 128), dim3(128), 0, (hipStream_t)stream, tiles, G, Gb, M,
                     g, HW);
  return check_launch();
}

extern "C" int glorie_gru_glo_terms(const void* wn, int w_stride, const float* bw, const void* net,
                                    int n_stride, const float* G, const float* Gb, int M,
                                    float* partial, int parts, float* g, int N, int HW, void* stream) {
  if (N < 0 || HW <= 0 || M <= 0 || parts <= 0) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!wn || !bw || !net || !G || !Gb || !partial || !g) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(glo_partial_kernel, dim3(N, parts), dim3(256), 0, st,
                     reinterpret_cast<const _Float16*>(wn), w_stride, bw,
                     reinterpret_cast<const _Float16*>(net), n_stride, partial, HW);
  hipLaunchKernelGGL(glo_terms_kernel, dim3(N, (M + 127) / 128), dim3(128), 0, st, partial, parts, G, Gb, M, g, HW);
  return check_launch();
}

extern "C" int glorie_gru_gate_zr(const void* zr, int zr_stride, const float* g, int g_stride,
                                  const void* net, int n_stride, void* z, int z_stride, void* rnet, int r_stride,
                                  int N, int HW, void* stream) {
  if (N < 0 || HW <= 0) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!zr || !g || !net || !z || !rnet) return GLORIE_EINVAL;
  const long P = (long)N * HW;
  hipLaunchKernelGGL(gru_gate_zr_kernel, dim3((unsigned)((P * 16 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const _Float16*>(zr), zr_stride, g, g_stride,
                     reinterpret_cast<const _Float16*>(net), n_stride, reinterpret_cast<_Float16*>(z),
                     z_stride, reinterpret_cast<_Float16*>(rnet), r_stride, P, HW);
  return check_launch();
}

extern "C" int glorie_gru_gate_q(const void* qc, int q_stride, const float* gq, int gq_stride,
                                 const void* z, int z_stride, const void* net, int n_stride, void* out,
                                 int o_stride, void* out2, int o2_stride, int N, int HW, void* stream) {
  if (N < 0 || HW <= 0) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!qc || !gq || !z || !net || !out) return GLORIE_EINVAL;
  const long P = (long)N * HW;
  hipLaunchKernelGGL(gru_gate_q_kernel, dim3((unsigned)((P * 16 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const _Float16*>(qc), q_stride, gq, gq_stride,
                     reinterpret_cast<const _Float16*>(z), z_stride, reinterpret_cast<const _Float16*>(net),
                     n_stride, reinterpret_cast<_Float16*>(out), o_stride,
                     reinterpret_cast<_Float16*>(out2), o2_stride, P, HW);
  return check_launch();
}

// Bookkeeping between the update operator and the BA (factor_graph.py:219-223,248): target = coords1 + delta,
// damping[frames] = eta, the BA's damping input 0.2 * eta + EP, and the age of every edge - four element-wise torch
// launches and a gather in the reference's formulation, one launch here.
__global__ __launch_bounds__(256) void update_bookkeeping_kernel(
    const float* __restrict__ coords1, const float* __restrict__ delta, float* __restrict__ target, long n_target,
    const float* __restrict__ eta, const int64_t* __restrict__ frames, float* __restrict__ damping_table,
    float* __restrict__ damping_ba, long n_eta, int HW, float ep, int64_t* __restrict__ age, int n_edges) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n_target) target[i] = coords1[i] + delta[i];
  if (i < n_eta) {
    const float e = eta[i];
    const long g = i / HW;
    damping_table[(size_t)frames[g] * HW + (i - g * HW)] = e;
    const float m = e * 0.2f;
    damping_ba[i] = m + ep;
  }
  if (age && i < n_edges) age[i] += 1;
}

extern "C" int glorie_update_bookkeeping(const float* coords1, const float* delta, float* target, long n_target,
                                         const float* eta, const int64_t* frames, float* damping_table,
                                         float* damping_ba, int G, int HW, float ep, int64_t* age, int n_edges,
                                         void* stream) {
  if (n_target < 0 || G < 0 || HW <= 0 || n_edges < 0) return GLORIE_EINVAL;
  const long n_eta = (long)G * HW;
  const long n = n_target > n_eta ? n_target : n_eta;
  if (n == 0) return GLORIE_OK;
  if ((n_target && (!coords1 || !delta || !target)) || (n_eta && (!eta || !frames || !damping_table || !damping_ba)))
    return GLORIE_EINVAL;
  hipLaunchKernelGGL(update_bookkeeping_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     coords1, delta, target, n_target, eta, frames, damping_table, damping_ba, n_eta, HW, ep, age, n_edges);
  return check_launch();
}

extern "C" int glorie_segment_mean(const void* x, int x_stride, const float* bias, int relu,
                                   const int64_t* ix, int N, void* out, int o_stride, int G, int HW,
                                   void* stream) {
  if (N < 0 || G < 0 || HW <= 0 || (x_stride & 7) || (o_stride & 7)) return GLORIE_EINVAL;
  if (G == 0) return GLORIE_OK;
  if (!out || (N > 0 && (!x || !ix))) return GLORIE_EINVAL;
  hipLaunchKernelGGL(segment_mean_kernel, dim3((HW * 16 + 255) / 256, G), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const _Float16*>(x), x_stride, bias, relu,
                     ix, N, reinterpret_cast<_Float16*>(out), o_stride, HW);
  return check_launch();
}

extern "C" int glorie_conv3x3_small(const void* x, int x_stride, const float* in_bias, int in_relu,
                                    const void* w_packed, const float* out_bias, int groups, int K,
                                    int act_packed, float scale, float* taps, float* out, int N, int H,
                                    int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || groups < 1 || groups > 4 || K < 1 || K > 3 || (x_stride & 7))
    return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!x || !w_packed || !taps || !out) return GLORIE_EINVAL;
  const long P = (long)N * H * W;
  const int ncols = 9 * K, tstride = ncols * groups;
  const long ntiles = (P + 15) / 16;
  const unsigned gx = (unsigned)min((ntiles + 3) / 4, (long)2048);
  hipStream_t st = (hipStream_t)stream;
  if (ncols <= 16)
    hipLaunchKernelGGL(conv_taps_kernel<1>, dim3(gx, groups), dim3(256), 0, st,
                       reinterpret_cast<const _Float16*>(x), x_stride, in_bias, in_relu,
                       reinterpret_cast<const f16x8*>(w_packed), taps, ncols, tstride, P);
  else
    hipLaunchKernelGGL(conv_taps_kernel<2>, dim3(gx, groups), dim3(256), 0, st,
                       reinterpret_cast<const _Float16*>(x), x_stride, in_bias, in_relu,
                       reinterpret_cast<const f16x8*>(w_packed), taps, ncols, tstride, P);
  const long total = P * K * groups;
  hipLaunchKernelGGL(conv_stencil_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, taps,
                     ncols, tstride, out_bias, K, groups, act_packed, scale, out, P, H, W);
  return check_launch();
}

// second half of glorie_conv3x3_small alone, on the tap PLANES written by glorie_conv_igemm_heads
extern "C" int glorie_conv_stencil(const float* taps, const float* out_bias, int groups, int K, int act_packed,
                                   float scale, float* out, float* out_last, int N, int H, int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || groups < 1 || groups > 4 || K < 1 || K > 3) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  if (!taps || !out) return GLORIE_EINVAL;
  const long P = (long)N * H * W, total = P * K * groups;
  hipLaunchKernelGGL(conv_stencil_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     taps, 9 * K, 9 * K * groups, out_bias, K, groups, act_packed, scale, out, P, H, W, out_last);
  return check_launch();
}
