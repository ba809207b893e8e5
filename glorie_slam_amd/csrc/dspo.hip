// DSPO stage 2: disparity + per-frame scale/shift optimisation (scope rows B8, B9, B11).
//
// Replaces BA_with_scale_shift + schur_solve
//   (/root/reference/src/geom/ba.py:127-216, /root/reference/src/geom/chol.py:58-85) and the
//   jacobian=True branch of projective_transform (projective_ops.py:96-125, MIN_DEPTH 0.2).
//
// The reference materialises H_wq [M*M,2,2] and E_wq_d [M*M,2,HW] although only the M diagonal
// blocks are non-zero (ba.py:191-192 scatter with (ll,ll)) and then runs dense
// (2M x M*HW) matmuls -- 3.5 GB at M = 300.  Here the block-diagonal structure is used
// directly: ten pixel sums per depth frame give its 2x2 reduced system, solved in closed
// form in fp64, and dz follows per pixel.  Failure semantics of chol.py:10-17 are kept: if
// ANY frame's 2x2 block is not positive definite, the scale/shift update of ALL frames is
// zero (the reference factorises one dense matrix) while dz = Q w is still applied.
//
// Edge filtering by mono_thres (depth_video.py:228-261) is an `edge_on` byte mask: disabled
// edges are skipped and frames left without enabled edges keep their state, so the host never
// compacts tensors (no boolean-mask copies, no sync); eta keeps its unfiltered slot order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"
#include "se3.hiph"
#include "ba_common.hiph"

namespace glorie {

struct Dspo2Args {
  const float* poses;      // [B,7]
  float* disps;            // [B,HW] in/out
  const float* intr;       // [B,4]
  const float* mono;       // [B,HW]
  float* scales;           // [B] in/out
  float* shifts;           // [B] in/out
  const uint8_t* vmask;    // [B,HW] valid_depth_mask_small
  const float* target;     // [N,HW,2]
  const float* weight;     // [N,HW,2]
  const float* eta;        // [M,HW]
  const int64_t* ii;
  const int64_t* jj;
  const uint8_t* edge_on;  // [N] or null
  float alpha;
};

struct PriorJ { float Jd, Js, Jq; };

// Jacobians of the mono-prior residual sqrt(alpha') (d - (s m + q))   (ba.py:158-170)
__device__ __forceinline__ PriorJ prior_jacobians(float mono, bool vd, float alpha) {
  const bool invalid = mono < 1e-6f;
  const float sa = sqrtf(alpha) * (vd ? 10.0f : 1.0f);
  PriorJ j;
  j.Jd = (invalid && vd) ? 0.0f : sa;
  j.Js = invalid ? 0.0f : -mono * sa;
  j.Jq = invalid ? 0.0f : -sa;
  return j;
}

__device__ __forceinline__ bool frame_enabled(const BaWork& wk, const uint8_t* edge_on, int s) {
  if (!edge_on) return wk.csr_ptr[s + 1] > wk.csr_ptr[s];
  for (int ei = wk.csr_ptr[s]; ei < wk.csr_ptr[s + 1]; ++ei)
    if (edge_on[wk.csr_edge[ei]]) return true;
  return false;
}

// one workgroup: per-frame 2x2 solve -> wk.dx [M][2]; all-or-nothing failure.  (Folding this into the last workgroup
// of the accumulation launch was measured and is NOT used: the agent-scope release every workgroup then needs writes back
// its XCD's L2 - 9 + 7 us became 25.)
__global__ __launch_bounds__(256) void dspo2_solve_kernel(BaWork wk, Dspo2Args a, int M, int nchunks,
                                                          float lm, float ep) {
  __shared__ int fail;
  const int tid = threadIdx.x;
  if (tid == 0) fail = 0;
  __syncthreads();
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  for (int s = tid; s < M; s += 256) {
    const bool on = frame_enabled(wk, a.edge_on, s);
    double x1 = 0.0, x2 = 0.0;
    if (on) {
      double v[10];
      for (int q = 0; q < 10; ++q) v[q] = 0.0;
      for (int c = 0; c < nchunks; ++c)
        for (int q = 0; q < 10; ++q) v[q] += (double)wk.Hpart[((size_t)s * nchunks + c) * 10 + q];
      // damping BEFORE the Schur complement (chol.py:68-69)
      const double H11 = v[0] + ep + lm * v[0], H22 = v[2] + ep + lm * v[2];
      const double S11 = H11 - v[3], S12 = v[1] - v[4], S22 = H22 - v[5];
      const double b1 = v[6] - v[8], b2 = v[7] - v[9];
      const double l21 = (S11 > 0.0) ? S12 / sqrt(S11) : 0.0;
      const double l22sq = S22 - l21 * l21;
      if (!(S11 > 0.0) || !(l22sq > 0.0)) {
        atomicOr(&fail, 1);
      } else {
        const double det = S11 * S22 - S12 * S12;
        x1 = (S22 * b1 - S12 * b2) / det;
        x2 = (S11 * b2 - S12 * b1) / det;
      }
    }
    wk.dx[2 * s + 0] = (float)x1;
    wk.dx[2 * s + 1] = (float)x2;
  }
  __syncthreads();
  if (fail) {
    for (int s = tid; s < 2 * M; s += 256) wk.dx[s] = 0.0f;
    if (tid == 0) {
      atomicOr(&wk.status[0], BA_ST_CHOL_FAILED);
      atomicAdd(&wk.status[2], 1);
    }
  }
}

// per-frame pixel sums: H11 H12 H22 | G11 G12 G22 | u1 u2 | g1 g2   (grid: chunks x M)
__global__ __launch_bounds__(kBaThreads) void dspo2_accum_kernel(BaWork wk, Dspo2Args a, int HW,
                                                                 int w, int nchunks) {
  __shared__ float red[4][10];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int chunk = blockIdx.x, s = blockIdx.y;
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  const int k = wk.kx[s];
  const int e0 = wk.csr_ptr[s], e1 = wk.csr_ptr[s + 1];
  const int px = chunk * kBaThreads + tid;
  const bool live = px < HW;
  float Cp = 0.0f, Wp = 0.0f;
  int n_on = 0;
  const float fxi = a.intr[k * 4 + 0], fyi = a.intr[k * 4 + 1];
  const float cxi = a.intr[k * 4 + 2], cyi = a.intr[k * 4 + 3];
  const int yy = live ? px / w : 0, xx = live ? px - yy * w : 0;
  const float dsp = live ? a.disps[(size_t)k * HW + px] : 1.0f;
  for (int ei = e0; ei < e1; ++ei) {
    const int n = wk.csr_edge[ei];
    if (a.edge_on && !a.edge_on[n]) continue;
    ++n_on;
    const int jx = (int)a.jj[n];
    const Pose g = (jx == k) ? stereo_pose()
                             : relative_pose(load_pose(a.poses + k * 7), load_pose(a.poses + jx * 7));
    const float fxj = a.intr[jx * 4 + 0], fyj = a.intr[jx * 4 + 1];
    const float cxj = a.intr[jx * 4 + 2], cyj = a.intr[jx * 4 + 3];
    if (!live) continue;
    float X0[4], X1[4];
    X0[0] = ((float)xx - cxi) / fxi;
    X0[1] = ((float)yy - cyi) / fyi;
    X0[2] = 1.0f;
    X0[3] = dsp;
    se3_act(g, X0, X1);
    const float Z = (X1[2] < 0.1f) ? 1.0f : X1[2];
    const float d = 1.0f / Z;
    const float cu = fxj * (X1[0] * d) + cxj, cv = fyj * (X1[1] * d) + cyj;
    const float valid = (X1[2] > 0.2f) ? 1.0f : 0.0f;
    const float2 tg = reinterpret_cast<const float2*>(a.target)[(size_t)n * HW + px];
    const float2 wg = reinterpret_cast<const float2*>(a.weight)[(size_t)n * HW + px];
    const float wu = 0.001f * (valid * wg.x), wvv = 0.001f * (valid * wg.y);
    const float ru = tg.x - cu, rv = tg.y - cv;
    // Jz = Jp . (Gij * e4) = Jp . (t, 1)
    const float Jzu = fxj * d * g.t.x - fxj * X1[0] * d * d * g.t.z;
    const float Jzv = fyj * d * g.t.y - fyj * X1[1] * d * d * g.t.z;
    Cp += wu * Jzu * Jzu + wvv * Jzv * Jzv;
    Wp += wu * ru * Jzu + wvv * rv * Jzv;
  }
  if (n_on != 0) {          // a frame without enabled edges is not in kx of the reference: no sums
    float sums[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) sums[q] = 0.0f;
    if (live) {
      const float mono = a.mono[(size_t)k * HW + px];
      const PriorJ J = prior_jacobians(mono, a.vmask[(size_t)k * HW + px] != 0, a.alpha);
      const float rd = sqrtf(a.alpha) * (dsp - (a.scales[k] * mono + a.shifts[k]));
      const float C = Cp + J.Jd * J.Jd + a.eta[(size_t)s * HW + px];
      const float Wv = Wp - J.Jd * rd;
      const float Q = 1.0f / C;
      wk.Q[(size_t)s * HW + px] = Q;
      wk.W[(size_t)s * HW + px] = Wv;
      const float e1v = J.Js * J.Jd, e2v = J.Jq * J.Jd;
      sums[0] = J.Js * J.Js; sums[1] = J.Js * J.Jq; sums[2] = J.Jq * J.Jq;
      sums[3] = e1v * Q * e1v; sums[4] = e1v * Q * e2v; sums[5] = e2v * Q * e2v;
      sums[6] = -J.Js * rd; sums[7] = -J.Jq * rd;
      sums[8] = e1v * Q * Wv; sums[9] = e2v * Q * Wv;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) sums[q] = wave_sum(sums[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 10; ++q) red[wv][q] = sums[q];
    }
    __syncthreads();
    if (tid < 10)
      wk.Hpart[((size_t)s * nchunks + chunk) * 10 + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// INLINE (M <= 25 frames): every workgroup repeats the per-frame 2x2 solves of dspo2_solve_kernel itself (10 threads per
// frame sum the chunk partials in the same order, one thread per frame solves, all-or-nothing failure) instead of reading
// wk.dx from a launch of its own - these launches are a few microseconds each and launch-bound.
template <bool INLINE>
__global__ __launch_bounds__(kBaThreads) void dspo2_update_kernel(BaWork wk, Dspo2Args a, int HW,
                                                                  float* __restrict__ dz_out, int M, int nchunks,
                                                                  float lm, float ep) {
  __shared__ double vsum[INLINE ? 25 * 10 : 1];
  __shared__ float sdx[INLINE ? 2 * 25 : 1];
  __shared__ int fail;
  const int tid = threadIdx.x;
  const int s = blockIdx.y;
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  float ds, dq;
  if constexpr (INLINE) {
    if (tid == 0) fail = 0;
    if (tid < 10 * M) {
      const int f = tid / 10, q = tid - f * 10;
      double v = 0.0;
      for (int c = 0; c < nchunks; ++c) v += (double)wk.Hpart[((size_t)f * nchunks + c) * 10 + q];
      vsum[tid] = v;
    }
    __syncthreads();
    if (tid < M) {
      const bool on = frame_enabled(wk, a.edge_on, tid);
      double x1 = 0.0, x2 = 0.0;
      if (on) {
        const double* v = vsum + tid * 10;
        // damping BEFORE the Schur complement (chol.py:68-69)
        const double H11 = v[0] + ep + lm * v[0], H22 = v[2] + ep + lm * v[2];
        const double S11 = H11 - v[3], S12 = v[1] - v[4], S22 = H22 - v[5];
        const double b1 = v[6] - v[8], b2 = v[7] - v[9];
        const double l21 = (S11 > 0.0) ? S12 / sqrt(S11) : 0.0;
        const double l22sq = S22 - l21 * l21;
        if (!(S11 > 0.0) || !(l22sq > 0.0)) {
          atomicOr(&fail, 1);
        } else {
          const double det = S11 * S22 - S12 * S12;
          x1 = (S22 * b1 - S12 * b2) / det;
          x2 = (S11 * b2 - S12 * b1) / det;
        }
      }
      sdx[2 * tid + 0] = (float)x1;
      sdx[2 * tid + 1] = (float)x2;
    }
    __syncthreads();
    ds = fail ? 0.0f : sdx[2 * s + 0];
    dq = fail ? 0.0f : sdx[2 * s + 1];
    if (blockIdx.x == 0 && s == 0) {          // diagnostics and the record of the step, once
      if (tid < 2 * M) wk.dx[tid] = fail ? 0.0f : sdx[tid];
      if (tid == 0 && fail) {
        atomicOr(&wk.status[0], BA_ST_CHOL_FAILED);
        atomicAdd(&wk.status[2], 1);
      }
    }
  } else {
    ds = wk.dx[2 * s + 0];
    dq = wk.dx[2 * s + 1];
  }
  if (!frame_enabled(wk, a.edge_on, s)) return;
  const int k = wk.kx[s];
  if (blockIdx.x == 0 && tid == 0) {        // the frame's scale / shift step (nobody reads them in this launch)
    a.scales[k] += ds;
    a.shifts[k] += dq;
  }
  const int px = blockIdx.x * kBaThreads + tid;
  if (px >= HW) return;
  const PriorJ J = prior_jacobians(a.mono[(size_t)k * HW + px], a.vmask[(size_t)k * HW + px] != 0, a.alpha);
  const float dz = wk.Q[(size_t)s * HW + px] *
                   (wk.W[(size_t)s * HW + px] - (J.Js * J.Jd * ds + J.Jq * J.Jd * dq));
  // disp_retr then clamp(min=0) (ba.py:210-214)
  a.disps[(size_t)k * HW + px] = fmaxf(a.disps[(size_t)k * HW + px] + dz, 0.0f);
  if (dz_out) dz_out[(size_t)s * HW + px] = dz;
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_dspo_scale_shift(glorie_ctx* ctx, const float* poses, float* disps,
                                       const float* intrinsics, const float* mono_disps,
                                       float* scales, float* shifts, const uint8_t* valid_mask,
                                       const float* target, const float* weight, const float* eta,
                                       const int64_t* ii, const int64_t* jj, const uint8_t* edge_on,
                                       int B, int N, int M, int h, int w, int iterations, float lm,
                                       float ep, float alpha, float* dz_out, void* stream) {
  if (!ctx || B < 0 || N < 0 || M < 0 || h < 0 || w < 0 || iterations < 0) return GLORIE_EINVAL;
  const int HW = h * w;
  if (N == 0 || M == 0 || HW == 0 || iterations == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !mono_disps || !scales || !shifts || !valid_mask || !target ||
      !weight || !eta || !ii || !jj)
    return GLORIE_EINVAL;
  if (B > kMaxFramesLds || N > kMaxEdgesLds) return GLORIE_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nchunks = (HW + kBaThreads - 1) / kBaThreads;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_slot = carve(sizeof(int) * (size_t)B);
  const size_t o_kx = carve(sizeof(int) * (size_t)(M + 1));
  const size_t o_ptr = carve(sizeof(int) * (size_t)(M + 2));
  const size_t o_edge = carve(sizeof(int) * (size_t)N);
  const size_t o_Hp = carve(sizeof(float) * 10 * (size_t)M * nchunks);
  const size_t o_Q = carve(sizeof(float) * (size_t)M * HW);
  const size_t o_W = carve(sizeof(float) * (size_t)M * HW);
  const size_t o_dx = carve(sizeof(float) * 2 * (size_t)M);
  GLORIE_TRY(ctx_reserve(ctx, off));
  GLORIE_TRY(ctx_poison(ctx, off, st));
  char* base = reinterpret_cast<char*>(ctx->scratch);
  BaWork wk{};
  wk.slot_of_frame = reinterpret_cast<int*>(base + o_slot);
  wk.kx = reinterpret_cast<int*>(base + o_kx);
  wk.csr_ptr = reinterpret_cast<int*>(base + o_ptr);
  wk.csr_edge = reinterpret_cast<int*>(base + o_edge);
  wk.status = ctx->dstatus;
  wk.Hpart = reinterpret_cast<float*>(base + o_Hp);
  wk.Q = reinterpret_cast<float*>(base + o_Q);
  wk.W = reinterpret_cast<float*>(base + o_W);
  wk.dx = reinterpret_cast<float*>(base + o_dx);
  // kx = unique(ii): t0 = t1 = 0 adds no extra frames (ba.py:139)
  GLORIE_TRY(ba_prepare(wk, ii, B, N, M, 0, 0, st));
  Dspo2Args a{poses, disps, intrinsics, mono_disps, scales, shifts, valid_mask, target, weight, eta,
              ii, jj, edge_on, alpha};
  for (int it = 0; it < iterations; ++it) {
    // per iteration: pixel sums, per-frame solves (inside the update launch for small M), disparity + scale/shift steps
    hipLaunchKernelGGL(dspo2_accum_kernel, dim3(nchunks, M), dim3(kBaThreads), 0, st, wk, a, HW, w, nchunks);
    if (M <= 25) {
      hipLaunchKernelGGL(dspo2_update_kernel<true>, dim3(nchunks, M), dim3(kBaThreads), 0, st, wk, a, HW, dz_out, M,
                         nchunks, lm, ep);
    } else {
      hipLaunchKernelGGL(dspo2_solve_kernel, dim3(1), dim3(256), 0, st, wk, a, M, nchunks, lm, ep);
      hipLaunchKernelGGL(dspo2_update_kernel<false>, dim3(nchunks, M), dim3(kBaThreads), 0, st, wk, a, HW, dz_out, M,
                         nchunks, lm, ep);
    }
    GLORIE_TRY(check_launch());
  }
  return GLORIE_OK;
}
