// Dense bundle adjustment (DSPO stage 1 / DBA) for gfx950 -- scope rows B2..B7.
//
// Replaces droid_backends.ba -> ba_cuda (/root/reference/src/lib/droid_kernels.cu:1314-1437)
// including its helpers projective_transform_kernel (:176-424), accum_cuda (:948-998),
// EEt6x6 / Ev6x1 / EvT6x1 (:1001-1115), SparseBlock + Eigen LLT (:1117-1219), schur_block
// (:1222-1311) and the retraction kernels (:877-946).
//
// The reference round-trips to the host ~20 times per Gauss-Newton iteration (CPU CSR
// construction, CPU triplet enumeration, D2H of every block, Eigen solve, H2D).  This
// implementation keeps the whole iteration on the device, with no host synchronisation:
//
//   prepare   (1 WG)        frame slots (kx), CSR of edges by source frame -- once per call
//   jacobian  (chunks x N)  a workgroup owns (edge, pixel chunk): it stores E_ij (6 x HW), the edge's
//                           shares of C_k and w_k per pixel and the block-reduced H_jj (21) / v_j (6);
//                           also zeroes the reduced system.  (Until the end of round 2 a workgroup
//                           owned a depth frame and walked its edges serially.)
//   gram      (chunks x M)  prologue: C_k, w_k = sum of the frame's edge shares in CSR order (no atomics),
//                           Q_k = 1 / (C_k + damping); then G_k = F diag(Q) F^T on the matrix cores
//                           (v_mfma_f32_16x16x4_f32, exact fp32), F = the E_ij rows of frame
//                           k plus the w row; scattered into the dense reduced system through
//                           exact integer accumulators (acc_add: order-independent, bit-reproducible).  The pose-pose blocks from H_jj / v_j (edge i of a frame by the
//                           workgroup of chunk i mod nchunks; a launch of its own, `assemble`, in motion-only mode).
//   solve     (1 WG)        fp64 damping + Cholesky + triangular solves (zero update on
//                           failure, like Eigen's LLT info != Success branch).
//   update    (chunks x M)  dz = Q (w - sum_e E_e^T y_e), disparity and pose retraction.
//
// Algebra used to cut the per-pixel work: J_i = -Ad^T J_j is an edge-constant linear map
// L_e (6x6) of J_j, so H_ii = L H_jj L^T, H_ij = -L H_jj, v_i = -L v_j, E_ii = -L E_ij.
// Only H_jj, v_j and E_ij are accumulated per pixel (27 reductions per edge instead of the
// reference's 90) and the "self" row E_i of the Schur system never has to be formed:
//   S[k,k] = sum_ab L_a G_ab L_b^T,  S[k,j_b] = -sum_a L_a G_ab,  E_i^T dx_k = -sum_e E_e^T L_e^T dx_k.
//
// Reference quirks that are reproduced on purpose:
//   * MIN_DEPTH 0.25 (native) instead of 0.2 (python)           droid_kernels.cu:26,302-306
//   * damping  diag += ep + lm*diag  applied AFTER the Schur complement  :1196-1197
//   * back-substitution ignores rows whose pose index (p - t0) is <= 0   :1105
//   * stereo edges (ii == jj) use the fixed baseline and contribute only to C, w  :219-229,323
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "common.hiph"
#include "se3.hiph"
#include "ba_common.hiph"

namespace glorie {

// ------------------------------------------------------------------------------------
// prepare: one workgroup of 1024 threads; all tables are built in LDS and written once
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void block_inclusive_scan(int* a, int n) {
  // Hillis-Steele, n <= 4096 with 1024 threads (<= 4 entries per thread)
  const int tid = threadIdx.x;
  for (int off = 1; off < n; off <<= 1) {
    int tmp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + q * 1024;
      tmp[q] = (f < n) ? a[f] + (f >= off ? a[f - off] : 0) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + q * 1024;
      if (f < n) a[f] = tmp[q];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void ba_prepare_kernel(BaWork wk, const int64_t* __restrict__ ii,
                                                          int B, int N, int M, int t0, int t1) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  extern __shared__ int sm[];
  int* cnt = sm;             // [B] edges per frame
  int* scan = sm + B;        // [B] scan workspace
  int* slot = sm + 2 * B;    // [B] frame -> slot
  int* eii = sm + 3 * B;     // [N] source frame per edge
  const int tid = threadIdx.x;
  if (tid == 0 && wk.gate_hits) atomicAdd(wk.gate_hits, 1);
  for (int f = tid; f < B; f += 1024) cnt[f] = 0;
  if (tid < 4) wk.status[tid] = 0;
  __syncthreads();
  for (int n = tid; n < N; n += 1024) {
    const int f = (int)ii[n];
    eii[n] = f;
    if (f >= 0 && f < B) atomicAdd(&cnt[f], 1);
  }
  __syncthreads();
  for (int f = tid; f < B; f += 1024) scan[f] = (cnt[f] > 0 || (f >= t0 && f < t1)) ? 1 : 0;
  __syncthreads();
  block_inclusive_scan(scan, B);
  const int Mdev = B > 0 ? scan[B - 1] : 0;
  for (int f = tid; f < B; f += 1024) {
    const int present = (cnt[f] > 0 || (f >= t0 && f < t1)) ? 1 : 0;
    const int sl = present ? scan[f] - 1 : -1;
    slot[f] = sl;
    wk.slot_of_frame[f] = sl;
    if (sl >= 0 && sl < M) wk.kx[sl] = f;
  }
  if (tid == 0) {
    wk.status[1] = Mdev;
    if (Mdev != M) atomicOr(&wk.status[0], BA_ST_M_MISMATCH);
  }
  __syncthreads();
  if (Mdev != M) return;
  // per-slot edge counts -> inclusive scan (reuses scan[])
  for (int sl = tid; sl < M; sl += 1024) scan[sl] = 0;
  __syncthreads();
  for (int f = tid; f < B; f += 1024)
    if (slot[f] >= 0) scan[slot[f]] = cnt[f];
  __syncthreads();
  block_inclusive_scan(scan, M);
  for (int sl = tid; sl <= M; sl += 1024) wk.csr_ptr[sl] = (sl == 0) ? 0 : scan[sl - 1];
  // stable fill: each present frame walks the edge list in order (deterministic CSR)
  for (int f = tid; f < B; f += 1024) {
    const int sl = slot[f];
    if (sl < 0 || cnt[f] == 0) continue;
    int o = (sl == 0) ? 0 : scan[sl - 1];
    for (int n = 0; n < N; ++n)
      if (eii[n] == f) wk.csr_edge[o++] = n;
  }
}

// ------------------------------------------------------------------------------------
// jacobian pass
// ------------------------------------------------------------------------------------
struct PixJ {
  float Ju[6], Jv[6];
  float Jzu, Jzv;
  float ru, rv;
  float wu, wv;
};

__device__ __forceinline__ void pixel_terms(const Pose g, float fx, float fy, float cx, float cy,
                                            float u, float v, float disp, float tu, float tv,
                                            float wgt_u, float wgt_v, PixJ& o) {
  float Xi[4], Xj[4];
  Xi[0] = (u - cx) / fx;
  Xi[1] = (v - cy) / fy;
  Xi[2] = 1.0f;
  Xi[3] = disp;
  se3_act(g, Xi, Xj);
  const float x = Xj[0], y = Xj[1], hh = Xj[3];
  const bool near = Xj[2] < 0.25f;
  const float d = near ? 0.0f : 1.0f / Xj[2];
  const float d2 = d * d;
  o.wu = near ? 0.0f : 0.001f * wgt_u;
  o.wv = near ? 0.0f : 0.001f * wgt_v;
  o.ru = tu - (fx * d * x + cx);
  o.rv = tv - (fy * d * y + cy);
  o.Ju[0] = fx * (hh * d);
  o.Ju[1] = fx * 0.0f;
  o.Ju[2] = fx * (-x * hh * d2);
  o.Ju[3] = fx * (-x * y * d2);
  o.Ju[4] = fx * (1.0f + x * x * d2);
  o.Ju[5] = fx * (-y * d);
  o.Jzu = fx * (g.t.x * d - g.t.z * (x * d2));
  o.Jv[0] = fy * 0.0f;
  o.Jv[1] = fy * (hh * d);
  o.Jv[2] = fy * (-y * hh * d2);
  o.Jv[3] = fy * (-1.0f - y * y * d2);
  o.Jv[4] = fy * (x * y * d2);
  o.Jv[5] = fy * (x * d);
  o.Jzv = fy * (g.t.y * d - g.t.z * (y * d2));
}

// L (6x6, row-major): J_i = -L J_j.  Column m of L is adjT applied to unit vector e_m.
__device__ __forceinline__ void edge_adjoint(const Pose g, float L[36]) {
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    float X[6] = {0, 0, 0, 0, 0, 0}, Y[6];
    X[m] = 1.0f;
    se3_adjT(g, X, Y);
#pragma unroll
    for (int r = 0; r < 6; ++r) L[r * 6 + m] = Y[r];
  }
}

// grid (nchunks, M), 256 threads, `ppt` pixels per thread (chunk = 256*ppt pixels)
__global__ __launch_bounds__(kBaThreads) void ba_jacobian_kernel(
    BaWork wk, const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, const float* __restrict__ disps_sens,
    const float* __restrict__ targets, const float* __restrict__ weights,
    const float* __restrict__ eta, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, int HW, int w, int nchunks, int ppt, int motion_only, int hwc, long hd_doubles) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ float red[4][28];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int chunk = blockIdx.x;
  const int n = blockIdx.y;                       // ONE EDGE per workgroup row (a frame's edges used to be walked serially
                                                  // by one workgroup: 152 latency-bound workgroups at G8, now 36 x 19)
  // the reduced system [H | v] the gram launch accumulates into starts at zero (nobody reads it in this launch:
  // one memset node less per iteration)
  {
    const long nthreads = (long)gridDim.x * gridDim.y * kBaThreads;
    for (long i = ((long)n * gridDim.x + chunk) * kBaThreads + tid; i < 2 * hd_doubles; i += nthreads) wk.Hacc[i] = 0ull;
  }
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  const int k = (int)ii[n];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int pbase = chunk * kBaThreads * ppt;
  const Pose gk = load_pose(poses + k * 7);
  float dsp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dsp[t] = t < ppt ? disps[(size_t)k * HW + min(pbase + t * kBaThreads + tid, HW - 1)] : 1.0f;
  {
    const int jx = (int)jj[n];
    float2 tgv[4], wgv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ppt) {                 // uniform
        const int pxc = min(pbase + t * kBaThreads + tid, HW - 1);
        if (hwc) {                   // [N][h][w][2]: the layout FactorGraph keeps them in
          tgv[t] = reinterpret_cast<const float2*>(targets)[(size_t)n * HW + pxc];
          wgv[t] = reinterpret_cast<const float2*>(weights)[(size_t)n * HW + pxc];
        } else {                     // [N][2][h][w]: the layout of droid_backends.ba
          tgv[t] = make_float2(targets[((size_t)n * 2 + 0) * HW + pxc], targets[((size_t)n * 2 + 1) * HW + pxc]);
          wgv[t] = make_float2(weights[((size_t)n * 2 + 0) * HW + pxc], weights[((size_t)n * 2 + 1) * HW + pxc]);
        }
      }
    }
    const bool stereo = (jx == k);
    const Pose g = stereo ? stereo_pose() : relative_pose(gk, load_pose(poses + jx * 7));
    if (chunk == 0 && tid == 0) {
      float L[36];
      edge_adjoint(g, L);
#pragma unroll
      for (int q = 0; q < 36; ++q) wk.Ledge[(size_t)n * 36 + q] = L[q];
    }
    float Hjj[21], vj[6];
#pragma unroll
    for (int q = 0; q < 21; ++q) Hjj[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < 6; ++q) vj[q] = 0.0f;

#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int px = pbase + t * kBaThreads + tid;
      if (t < ppt && px < HW) {
        const int yy = px / w, xx = px - yy * w;
        PixJ J;
        const float2 tg = tgv[t], wg = wgv[t];
        pixel_terms(g, fx, fy, cx, cy, (float)xx, (float)yy, dsp[t], tg.x, tg.y, wg.x,
                    wg.y, J);
        // this edge's share of C_k and w_k: the frame's sum is formed, in CSR order, by the gram launch
        if (!motion_only)
          reinterpret_cast<float2*>(wk.CWpart)[(size_t)n * HW + px] =
              make_float2(J.wu * J.Jzu * J.Jzu + J.wv * J.Jzv * J.Jzv, J.wu * J.ru * J.Jzu + J.wv * J.rv * J.Jzv);
        const float wu = stereo ? 0.0f : J.wu;
        const float wvv = stereo ? 0.0f : J.wv;
        int l = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            Hjj[l] += wu * J.Ju[a] * J.Ju[b] + wvv * J.Jv[a] * J.Jv[b];
            ++l;
          }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          vj[a] += wu * J.ru * J.Ju[a] + wvv * J.rv * J.Jv[a];
          if (!motion_only)
            wk.Eij[((size_t)n * 6 + a) * HW + px] = wu * J.Jzu * J.Ju[a] + wvv * J.Jzv * J.Jv[a];
        }
      }
    }
    // block reduction of the 27 per-edge sums
#pragma unroll
    for (int q = 0; q < 21; ++q) Hjj[q] = wave_sum(Hjj[q]);
#pragma unroll
    for (int q = 0; q < 6; ++q) vj[q] = wave_sum(vj[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 21; ++q) red[wv][q] = Hjj[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) red[wv][21 + q] = vj[q];
    }
    __syncthreads();
    if (tid < 27)
      wk.Hpart[((size_t)n * nchunks + chunk) * 27 + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }

}

// ------------------------------------------------------------------------------------
// gram pass: G = F diag(Q) F^T with MFMA, then scatter into the dense system (exact accumulators)
// ------------------------------------------------------------------------------------
// Exact accumulation of the reduced system.  Many workgroups (pixel chunks x depth frames, and the edges' pose-pose
// blocks) add into the same entry of [H | v]; with fp64 atomics the ORDER of those additions, and with it the last
// bit of the sum, changed from run to run (measured in round 5: 999 of 1000 builds differed in some entry by one
// ulp).  The reference sums in a fixed order (sorted-edge accum, droid_kernels.cu:948-998; Eigen triplets,
// :1117-1219).  Here every term is rounded ONCE to a multiple of 2^-40 and added to a two-limb integer accumulator
// (value = (hi 2^32 + lo) 2^-40): integer addition is associative, so the sum does not depend on the order, and the
// limbs need no carry between them (lo takes < 2^32 per term: room for 2^31 terms; hi covers |sum| < 2^54).  The
// 9e-13 absolute rounding per term is far below the fp32 rounding of the terms themselves.
__device__ __forceinline__ void acc_add(const BaWork& wk, size_t idx, double v) {
  if (!(fabs(v) < 0x1p50)) {                     // also NaN
    atomicOr(&wk.status[0], BA_ST_NONFINITE);
    return;
  }
  const double t = v * 256.0;                    // in units of 2^-40 2^32
  const double fh = floor(t);
  const unsigned long long lo = (unsigned long long)rint((t - fh) * 4294967296.0);   // t - fh is exact, in [0, 1)
  atomicAdd(&wk.Hacc[2 * idx], (unsigned long long)(long long)fh);
  atomicAdd(&wk.Hacc[2 * idx + 1], lo);
}
__device__ __forceinline__ double acc_value(const unsigned long long* acc, size_t idx) {
  const long long hi = (long long)acc[2 * idx];
  return ((double)hi * 4294967296.0 + (double)acc[2 * idx + 1]) * 0x1p-40;
}

__device__ __forceinline__ void assemble_edge(const BaWork& wk, const int64_t* __restrict__ ii,
                                              const int64_t* __restrict__ jj, int n, int nchunks, int t0, int t1,
                                              double* sm);

constexpr int kGS = 8;                 // edges per row group (48 rows)
constexpr int kGramKC = 128;           // pixels staged per sub-chunk
constexpr int kGramLd = kGramKC + 2;   // +2 keeps the MFMA operand ds_read_b32 conflict free
constexpr int kRowsA = 48, kRowsB = 64;  // B side carries the extra w row (+pad to 16)

// One workgroup = (pixel chunk, depth frame k).  The edges leaving k are processed in groups
// of 8 (48 Jacobian rows); for every ordered pair of groups (A,B) the 48x64 product
// G_AB = F_A diag(Q) F_B'^T is accumulated over the chunk's pixels by 12 MFMA tiles
// (3 per wave), dropped into LDS and scattered.  Frames with <= 8 outgoing edges -- the
// normal case -- are a single pair.
__global__ __launch_bounds__(kBaThreads) void ba_gram_kernel(
    BaWork wk, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int HW, int chunk_px, int t0, int t1,
    int nchunks, const float* __restrict__ eta, const float* __restrict__ disps, const float* __restrict__ disps_sens) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ __attribute__((aligned(16))) float lds[(kRowsA + kRowsB) * kGramLd];
  __shared__ float Xs[kGS * 36];
  __shared__ int eA[kGS], eB[kGS];          // edge ids of the two row groups being multiplied
  __shared__ float qw[4 * kBaThreads][2];   // Q and w of this workgroup's pixels
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int chunk = blockIdx.x;
  const int s = blockIdx.y;
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  const int k = wk.kx[s];
  const int e0 = wk.csr_ptr[s];
  const int deg = wk.csr_ptr[s + 1] - e0;
  // C_k = sum of the per-edge shares (ba_jacobian_kernel, one workgroup row per edge) in CSR order + damping, Q = 1 / C, w:
  // for every depth frame, also those without outgoing edges (droid_kernels.cu:1385-1400)
  for (int l = tid; l < chunk_px; l += kBaThreads) {
    const int px = chunk * chunk_px + l;
    if (px >= HW) break;
    float C = 0.0f, Wv = 0.0f;
    for (int a0 = 0; a0 < deg; a0 += 4) {      // four edges' rows in flight
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = reinterpret_cast<const float2*>(wk.CWpart)[(size_t)wk.csr_edge[e0 + min(a0 + u, deg - 1)] * HW + px];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (a0 + u < deg) { C += v[u].x; Wv += v[u].y; }
    }
    const float et = eta[(size_t)s * HW + px];
    if (disps_sens) {  // depth-sensor prior, alpha = 0.05 (droid_kernels.cu:1397-1400)
      const float ds = disps_sens[(size_t)k * HW + px];
      const float m = ds > 0.0f ? 1.0f : 0.0f;
      C = C + m * 0.05f + (1.0f - m) * et;
      Wv = Wv - m * 0.05f * (disps[(size_t)k * HW + px] - ds);
    } else {
      C = C + et;
    }
    const float Q = 1.0f / C;
    wk.Q[(size_t)s * HW + px] = Q;
    wk.W[(size_t)s * HW + px] = Wv;
    qw[l][0] = Q;
    qw[l][1] = Wv;
  }
  if (deg == 0) return;
  const int P = t1 - t0;
  const int n6 = 6 * P;
  const int pk = k - t0;
  const bool self = (pk >= 0 && pk < P);
  const int pbeg = chunk * chunk_px;
  const int pend = min(pbeg + chunk_px, HW);
  if (pbeg >= HW) return;
  // the pose-pose blocks of this frame's edges (what ba_assemble_kernel did in a launch of its own): edge i of the frame
  // is taken by the workgroup of pixel chunk i mod nchunks, before its share of the gram products
  for (int i = chunk; i < deg; i += nchunks)
    assemble_edge(wk, ii, jj, wk.csr_edge[e0 + i], nchunks, t0, t1, reinterpret_cast<double*>(lds));
  const int NG = (deg + kGS - 1) / kGS;
  float* Fa = lds;
  float* Fb = lds + kRowsA * kGramLd;
  float* Gs = lds;  // [48][65] view used after the MFMA phase
  constexpr int GL = kRowsB + 1;

  for (int A = 0; A < NG; ++A) {
    const int degA = min(kGS, deg - A * kGS);
    const int rowsA = 6 * degA;
    for (int Bg = 0; Bg < NG; ++Bg) {
      const int degB = min(kGS, deg - Bg * kGS);
      const int wcol = 6 * degB;                 // column of the w row (only when Bg == 0)
      const int rowsB = wcol + (Bg == 0 ? 1 : 0);
      f32x4 acc[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      // slots past a group's degree repeat its last edge (their rows are never read); published by the barrier that
      // opens the pixel loop
      __syncthreads();
      if (tid < kGS) eA[tid] = wk.csr_edge[e0 + A * kGS + min(tid, degA - 1)];
      else if (tid < 2 * kGS) eB[tid - kGS] = wk.csr_edge[e0 + Bg * kGS + min(tid - kGS, degB - 1)];

      for (int p0 = pbeg; p0 < pend; p0 += kGramKC) {
        __syncthreads();
        {  // stage sqrt(Q)-scaled rows; thread = (pixel c, row parity): rows 6a + half + {0, 2, 4} of every edge a.
          // All loads of a side are issued before the first LDS store (edge slots past the group's degree re-read its
          // last edge and land in rows no tile reads): a loop "index load -> row load -> store" per row was a chain
          // of ~27 dependent memory round trips per sub-chunk and made this launch latency-bound (33 -> 22 us).
          const int c = tid & (kGramKC - 1);
          const int half = tid >> 7;
          const int px = p0 + c;
          const bool ok = px < pend;
          const int pxc = ok ? px : pbeg;
          const float sq = ok ? sqrtf(qw[pxc - pbeg][0]) : 0.0f;     // published by the barrier in front of the staging
          const float wv_ = qw[pxc - pbeg][1];
          float va[kGS][3];
#pragma unroll
          for (int a = 0; a < kGS; ++a) {
            const int n = eA[a];
#pragma unroll
            for (int q = 0; q < 3; ++q) va[a][q] = wk.Eij[((size_t)n * 6 + half + 2 * q) * HW + pxc];
          }
          if (A == Bg) {
#pragma unroll
            for (int a = 0; a < kGS; ++a)
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                const float v = va[a][q] * sq;
                Fa[(6 * a + half + 2 * q) * kGramLd + c] = v;
                Fb[(6 * a + half + 2 * q) * kGramLd + c] = v;
              }
          } else {
            float vb[kGS][3];
#pragma unroll
            for (int a = 0; a < kGS; ++a) {
              const int n = eB[a];
#pragma unroll
              for (int q = 0; q < 3; ++q) vb[a][q] = wk.Eij[((size_t)n * 6 + half + 2 * q) * HW + pxc];
            }
#pragma unroll
            for (int a = 0; a < kGS; ++a)
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                Fa[(6 * a + half + 2 * q) * kGramLd + c] = va[a][q] * sq;
                Fb[(6 * a + half + 2 * q) * kGramLd + c] = vb[a][q] * sq;
              }
          }
          // the w row of the B side (after the edge rows: slot degB of a full group would be row 48, also in range)
          if (Bg == 0 && half == (wcol & 1)) Fb[wcol * kGramLd + c] = wv_ * sq;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int t = q * 4 + wv;  // 12 tiles: I = t / 4 (A side), J = t % 4 (B side)
          const int I = t >> 2, J = t & 3;
          if (I * 16 < rowsA && J * 16 < rowsB) {
            const float* Ap = Fa + (I * 16 + (lane & 15)) * kGramLd + (lane >> 4);
            const float* Bp = Fb + (J * 16 + (lane & 15)) * kGramLd + (lane >> 4);
#pragma unroll 8
            for (int kk = 0; kk < kGramKC; kk += 4)
              acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[kk], Bp[kk], acc[q], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int t = q * 4 + wv;
        const int I = t >> 2, J = t & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Gs[(I * 16 + (lane >> 4) * 4 + r) * GL + J * 16 + (lane & 15)] = acc[q][r];
      }
      __syncthreads();

      // ---- scatter into the dense system: H -= S, v -= v_S (lower block triangle) ----
      const int ea = e0 + A * kGS, eb = e0 + Bg * kGS;
      for (int idx = tid; idx < degA * degB * 36; idx += kBaThreads) {
        const int c = idx % 6, r = (idx / 6) % 6, b = (idx / 36) % degB, a = idx / (36 * degB);
        const int pa = (int)jj[wk.csr_edge[ea + a]] - t0;
        const int pb = (int)jj[wk.csr_edge[eb + b]] - t0;
        if (pa >= 0 && pa < P && pb >= 0 && pb < P && pa >= pb)
          acc_add(wk, (size_t)(6 * pa + r) * n6 + 6 * pb + c, -(double)Gs[(6 * a + r) * GL + 6 * b + c]);
      }
      if (Bg == 0) {
        for (int idx = tid; idx < degA * 6; idx += kBaThreads) {
          const int r = idx % 6, a = idx / 6;
          const int pa = (int)jj[wk.csr_edge[ea + a]] - t0;
          if (pa >= 0 && pa < P)
            acc_add(wk, (size_t)n6 * n6 + 6 * pa + r, -(double)Gs[(6 * a + r) * GL + wcol]);
        }
      }
      if (self) {
        // X_b = -sum_{a in A} L_a G_ab   (= S[k, j_b] contribution)
        for (int idx = tid; idx < degB * 36; idx += kBaThreads) {
          const int c = idx % 6, r = (idx / 6) % 6, b = idx / 36;
          float sum = 0.0f;
          for (int a = 0; a < degA; ++a) {
            const float* L = wk.Ledge + (size_t)wk.csr_edge[ea + a] * 36;
#pragma unroll
            for (int m = 0; m < 6; ++m) sum += L[r * 6 + m] * Gs[(6 * a + m) * GL + 6 * b + c];
          }
          Xs[idx] = -sum;
        }
        __syncthreads();
        for (int idx = tid; idx < degB * 36; idx += kBaThreads) {
          const int c = idx % 6, r = (idx / 6) % 6, b = idx / 36;
          const int pb = (int)jj[wk.csr_edge[eb + b]] - t0;
          if (pb >= 0 && pb < P) {
            const double val = -(double)Xs[idx];
            if (pk >= pb) acc_add(wk, (size_t)(6 * pk + r) * n6 + 6 * pb + c, val);
            else          acc_add(wk, (size_t)(6 * pb + c) * n6 + 6 * pk + r, val);
          }
        }
        if (tid < 36) {  // S_kk = -sum_b X_b L_b^T ;  H -= S_kk
          const int c = tid % 6, r = tid / 6;
          float sum = 0.0f;
          for (int b = 0; b < degB; ++b) {
            const float* L = wk.Ledge + (size_t)wk.csr_edge[eb + b] * 36;
#pragma unroll
            for (int m = 0; m < 6; ++m) sum += Xs[b * 36 + r * 6 + m] * L[c * 6 + m];
          }
          acc_add(wk, (size_t)(6 * pk + r) * n6 + 6 * pk + c, (double)sum);
        } else if (Bg == 0 && tid >= 64 && tid < 70) {  // v_S[k] = -sum_a L_a g_a ; v -= v_S
          const int r = tid - 64;
          float sum = 0.0f;
          for (int a = 0; a < degA; ++a) {
            const float* L = wk.Ledge + (size_t)wk.csr_edge[ea + a] * 36;
#pragma unroll
            for (int m = 0; m < 6; ++m) sum += L[r * 6 + m] * Gs[(6 * a + m) * GL + wcol];
          }
          acc_add(wk, (size_t)n6 * n6 + 6 * pk + r, (double)sum);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// assemble: pose-pose blocks.  one 64-thread workgroup per edge
// ------------------------------------------------------------------------------------
// the pose-pose blocks of edge n: executed by threads 0..63 of the calling workgroup, every thread of the workgroup
// must call it (barriers); sm = 192 doubles of LDS
__device__ __forceinline__ void assemble_edge(const BaWork& wk, const int64_t* __restrict__ ii,
                                              const int64_t* __restrict__ jj, int n, int nchunks, int t0, int t1,
                                              double* sm) {
  double* Hjj = sm; double* Hij = sm + 36; double* Hii = sm + 72; double* L = sm + 108;
  double* vj = sm + 144; double* vi = sm + 150; double* red = sm + 156;      // 27
  const int tid = threadIdx.x;
  const int P = t1 - t0;
  const int pi = (int)ii[n] - t0, pj = (int)jj[n] - t0;
  const bool ai = pi >= 0 && pi < P, aj = pj >= 0 && pj < P;
  if (!ai && !aj) return;                          // uniform
  __syncthreads();                                 // sm may still be read from the previous edge
  if (tid < 27) {
    double acc = 0.0;
    for (int c = 0; c < nchunks; ++c) acc += (double)wk.Hpart[((size_t)n * nchunks + c) * 27 + tid];
    red[tid] = acc;
  }
  if (tid < 36) L[tid] = (double)wk.Ledge[(size_t)n * 36 + tid];
  __syncthreads();
  if (tid < 36) {
    const int r = tid / 6, c = tid % 6;
    const int a = r >= c ? r : c, b = r >= c ? c : r;
    Hjj[tid] = red[a * (a + 1) / 2 + b];
  }
  if (tid < 6) vj[tid] = red[21 + tid];
  __syncthreads();
  if (tid < 36) {  // Hij = -L Hjj
    const int r = tid / 6, c = tid % 6;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += L[r * 6 + m] * Hjj[m * 6 + c];
    Hij[tid] = -acc;
  }
  if (tid >= 36 && tid < 42) {  // vi = -L vj
    const int r = tid - 36;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += L[r * 6 + m] * vj[m];
    vi[r] = -acc;
  }
  __syncthreads();
  if (tid < 36) {  // Hii = L Hjj L^T = -Hij L^T
    const int r = tid / 6, c = tid % 6;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += Hij[r * 6 + m] * L[c * 6 + m];
    Hii[tid] = -acc;
  }
  __syncthreads();
  const int n6 = 6 * P;
  if (tid < 36) {
    const int r = tid / 6, c = tid % 6;
    if (ai) acc_add(wk, (size_t)(6 * pi + r) * n6 + 6 * pi + c, Hii[tid]);
    if (aj) acc_add(wk, (size_t)(6 * pj + r) * n6 + 6 * pj + c, Hjj[tid]);
    if (ai && aj) {
      if (pi >= pj) acc_add(wk, (size_t)(6 * pi + r) * n6 + 6 * pj + c, Hij[tid]);
      else          acc_add(wk, (size_t)(6 * pj + c) * n6 + 6 * pi + r, Hij[tid]);
    }
  }
  if (tid < 6) {
    if (ai) acc_add(wk, (size_t)n6 * n6 + 6 * pi + tid, vi[tid]);
    if (aj) acc_add(wk, (size_t)n6 * n6 + 6 * pj + tid, vj[tid]);
  }
}

// stand-alone form (motion-only BA, where the gram launch that otherwise does this work is skipped): one 64-thread
// workgroup per edge
__global__ __launch_bounds__(64) void ba_assemble_kernel(BaWork wk, const int64_t* __restrict__ ii,
                                                         const int64_t* __restrict__ jj,
                                                         int nchunks, int t0, int t1) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ double sm[192];
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  assemble_edge(wk, ii, jj, blockIdx.x, nchunks, t0, t1, sm);
}

// accumulators -> the fp64 system [H | v] the solvers (and a multi-GPU all-reduce) work on
__global__ __launch_bounds__(256) void ba_system_f64_kernel(BaWork wk, long total) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  // a dropped non-finite term poisons H[0][0]: after a multi-GPU all-reduce EVERY rank then sees the failure
  if (i < total) wk.Hd[i] = (i == 0 && (wk.status[0] & BA_ST_NONFINITE)) ? __builtin_nan("") : acc_value(wk.Hacc, (size_t)i);
}

// ------------------------------------------------------------------------------------
// solve: (H + damping) dx = v in fp64.  n <= kSolveMaxN and banded systems: ba_solve_band_kernel (LDS);
// dense systems up to kFusedMaxN: ba_solve_fused_kernel (one workgroup, matrix in L2); above: the
// multi-kernel blocked Cholesky below.
// ------------------------------------------------------------------------------------
// ---- large systems: blocked right-looking Cholesky on the dense fp64 matrix in HBM ----
constexpr int kNB = 32;

// trailing update A22 -= X X^T (lower part) + the right-hand side row n, 64 x 64 output tile per workgroup,
// 4 x 4 outputs per thread (rows ty + 16 i, columns tx + 16 j: broadcast / conflict-free LDS reads, coalesced
// updates); row index t of the (rem + 1)-row panel: t < rem -> matrix row j0 + nb + t, t == rem -> row n
constexpr int kTrailTile = 64;
__global__ __launch_bounds__(256) void chol_trail_kernel(BaWork wk, int n, int j0, int nb) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ double Ar[kTrailTile][kNB + 1], Ac[kTrailTile][kNB + 1];
  if (wk.status[0] & (BA_ST_SOLVE_ABORT | BA_ST_M_MISMATCH)) return;
  const int base = j0 + nb, rem = n - base;
  const int tr = blockIdx.y, tc = blockIdx.x;
  if (tc > tr) return;
  const int r0 = tr * kTrailTile, c0 = tc * kTrailTile;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < kTrailTile * kNB; idx += 256) {
    const int r = idx >> 5, m = idx & 31;
    const int t_r = r0 + r, t_c = c0 + r;
    Ar[r][m] = (t_r <= rem && m < nb) ? wk.Hd[(size_t)(t_r < rem ? base + t_r : n) * n + j0 + m] : 0.0;
    Ac[r][m] = (t_c < rem && m < nb) ? wk.Hd[(size_t)(base + t_c) * n + j0 + m] : 0.0;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
#pragma unroll 4
  for (int m = 0; m < kNB; ++m) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = Ar[ty + 16 * i][m];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = Ac[tx + 16 * j][m];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t_r = r0 + ty + 16 * i;
    if (t_r > rem) continue;
    double* row = wk.Hd + (size_t)(t_r < rem ? base + t_r : n) * n + base;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t_c = c0 + tx + 16 * j;
      if (t_c < rem && t_c <= t_r) row[t_c] -= acc[i][j];
    }
  }
}

__global__ __launch_bounds__(256) void chol_damp_kernel(BaWork wk, int n, float lm, float ep) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {                                                   // first launch of a large-system solve
    if (wk.status[0] & BA_ST_NONFINITE) {
      atomicOr(&wk.status[0], BA_ST_CHOL_FAILED | BA_ST_SOLVE_ABORT);
      atomicAdd(&wk.status[2], 1);
    } else {
      atomicAnd(&wk.status[0], ~BA_ST_SOLVE_ABORT);
    }
  }
  if (i < n) {
    double v = wk.Hd[(size_t)i * n + i];
    wk.Hd[(size_t)i * n + i] = v + (double)ep + (double)lm * v;
  }
}

// ---- banded systems: unblocked LDL^T-style elimination on band storage in LDS -------------------------
// A sliding-window graph (edges |i - j| <= r) couples poses at most 2r apart, so the reduced camera system
// is banded (half bandwidth bw = 41 for r = 3) and Cholesky fill stays inside the band: n bw^2 / 2 work and
// n (bw + 1) doubles of storage (99 KB at n = 294) instead of n^3 / 6 and n^2.  One workgroup of 256 threads:
//   * bw = largest r - c with H[r][c] != 0 (entries nobody accumulated into are exact zeros) comes from
//     ba_bandwidth_kernel through status[3]; ba_solve_fused_kernel does the work instead if the band does
//     not fit into LDS or bw >= 64.  H itself is never modified here.
//   * elimination: columns j, j + 1 update the window below / right of them and the right-hand side, ONE
//     barrier per TWO columns (band_eliminate).  Columns stay unscaled (U[r][j] = L[r][j] sqrt(d_j)); only 1 / d_j is needed, from
//     v_rcp_f64 + two Newton steps - no square root and no division on the critical path.  Every thread
//     owns the same <= 9 window positions (dr, dc) in every step, decoded once.
//   * back substitution x_j = (u_j - sum_k U[j + k][j] x_{j + k}) / d_j in ONE wave without barriers: lane
//     (r mod 64) keeps the running sum of row r in a register (bw < 64: rows sharing a lane are never
//     active together), x_j is broadcast with v_readlane.
// Same operation order on every rank of a sharded run -> bit-identical pose updates.
constexpr int kBandThreads = 256;

// half bandwidth of the system: one wave per row looks for its first non-zero; result in status[3] = bw << 1
// (status[3] is 0 between solves: ba_solve_fused_kernel resets it)
__global__ __launch_bounds__(64) void ba_bandwidth_kernel(BaWork wk, int n) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  const int r = blockIdx.x, lane = threadIdx.x;
  const double* row = wk.Hd + (size_t)r * n;
  int first = r;
  for (int c0 = 0; c0 < r; c0 += 256) {     // 4 independent loads per lane and round
    bool nz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * 64 + lane;
      nz[q] = c < r && row[c] != 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned long long m = __ballot(nz[q]);
      if (m && first == r) first = c0 + q * 64 + __ffsll((long long)m) - 1;
    }
    if (first != r) break;
  }
  if (lane == 0 && first < r) atomicMax(&wk.status[3], (r - first) << 1);
}

__device__ __forceinline__ double rcp_f64(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// Two columns per barrier.  With a = U[.][j] (column j, unscaled), a1 = a[j + 1], the next column after
// eliminating j is b[x] = H[x][j + 1] - a[x] a1 / d0 (d1 = b[j + 1]), and every entry right of it gets
//   H[r][c] -= a[r] a[c] / d0 + b[r] b[c] / d1 .
// Each thread forms the b's it needs itself (two more LDS reads per entry), so the dependent chain
// read -> 1/d0 -> d1 -> 1/d1 -> update -> write -> barrier is paid once per TWO columns; entries of column
// j + 1 itself (dc == 0) only take the first term, which is exactly b.
// Hazard (found in round 5, the cause of run-to-run differences of ~1e-6 .. 1e-3 in the poses whenever the waves
// of this workgroup did not run in lock step): column j + 1 is READ by every thread of the step (hr, hc, h11)
// and its new value b is WRITTEN by the owners of the dc == 0 entries in the same step - a wave that got through
// its two rcp chains before another wave had issued its loads made that wave read b instead of H[.][j + 1].  With
// one barrier per step the reads and writes of a step must not alias: the b column is therefore kept in registers
// and written one step LATE (step j + 2 reads columns j + 2, j + 3 and the window right of them, never column
// j + 1; the back substitution runs after the final barrier).  Everything else a step writes (dc > 0: columns
// >= j + 2) is read by nobody in that step.
template <int NQ>
__device__ __forceinline__ void band_eliminate(double* B, double* u, double* rinv, int* fail, int n, int bw) {
  const int tid = threadIdx.x;
  const int S = bw + 1;
  // window positions of this thread, the same in every step: entries (dr, dc), dc <= dr <= bw, of the triangle
  // below / right of (j + 1, j + 1), as LDS offsets relative to row j of the band (entry e = tid + 256 q of
  // the packed triangle); threads 0 .. bw also own one entry of the right-hand side
  const int T1 = (bw + 1) * (bw + 2) / 2;
  int o_ar[NQ], o_ac[NQ], o_w[NQ], e_dr[NQ], e_dc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = tid + q * kBandThreads;
    int dr = 1 << 20, dc = 0;        // inactive: never inside the matrix
    if (e < T1) {
      dr = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while ((dr + 1) * (dr + 2) / 2 <= e) ++dr;
      while (dr * (dr + 1) / 2 > e) --dr;
      dc = e - dr * (dr + 1) / 2;
    }
    e_dr[q] = dr; e_dc[q] = dc;
    o_ar[q] = (1 + dr) * S + dr + 1;     // U[r][j]     (one to the left: H[r][j + 1])
    o_ac[q] = (1 + dc) * S + dc + 1;     // U[c][j]     (one to the left: H[c][j + 1])
    o_w[q] = (1 + dr) * S + dr - dc;     // H[r][c]
  }
  const int rhs_dc = tid <= bw ? tid : (1 << 20);
  const int o_rc = (1 + rhs_dc) * S + rhs_dc + 1;
  double pend[NQ];
  bool pend_on[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { pend[q] = 0.0; pend_on[q] = false; }
  __syncthreads();
  int j = 0;
  for (; j + 1 < n; j += 2) {
    // every load of the step is issued before anything is used (branchy code would serialise an LDS round
    // trip per entry); entries in row j + 1 + bw lie outside the band of column j: a = 0 there
    const double* Bj = B + j * S;
    const double d0 = Bj[0], a1 = Bj[S + 1], h11 = Bj[S];
    double ar[NQ], ac[NQ], hr[NQ], hc[NQ], cur[NQ];
    bool ok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      ok[q] = j + 1 + e_dr[q] < n;
      const bool inr = ok[q] && e_dr[q] < bw, inc = ok[q] && e_dc[q] < bw;
      ar[q] = Bj[inr ? o_ar[q] : 0];
      ac[q] = Bj[inc ? o_ac[q] : 0];
      hr[q] = Bj[ok[q] ? o_ar[q] - 1 : 0];
      hc[q] = Bj[ok[q] ? o_ac[q] - 1 : 0];
      cur[q] = Bj[ok[q] ? o_w[q] : 0];
      if (!inr) ar[q] = 0.0;
      if (!inc) ac[q] = 0.0;
    }
    const bool rok = j + 1 + rhs_dc < n, rin = rok && rhs_dc < bw;
    double ra = Bj[rin ? o_rc : 0];
    const double rh = Bj[rok ? o_rc - 1 : 0];
    const double ru = u[rok ? j + 1 + rhs_dc : 0];
    const double u0 = u[j], u1h = u[j + 1];
    if (!rin) ra = 0.0;
    const double ri0 = rcp_f64(d0);
    const double d1 = h11 - a1 * ri0 * a1;
    if ((!(d0 > 0.0) || !(d1 > 0.0)) && tid == 0) *fail = 1;   // also NaN; the (garbage) result is discarded
    const double ri1 = rcp_f64(d1);
    if (tid == 0) { rinv[j] = ri0; rinv[j + 1] = ri1; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      // column j - 1 of the previous step (nobody reads it any more); the slot is rewritten below iff ok[q]
      if (pend_on[q]) B[(j - 2) * S + o_w[q]] = pend[q];
      pend_on[q] = false;
      if (!ok[q]) continue;
      double v = cur[q] - ar[q] * ri0 * ac[q];
      if (e_dc[q] > 0) {
        const double br = hr[q] - ar[q] * ri0 * a1, bc = hc[q] - ac[q] * ri0 * a1;
        v -= br * ri1 * bc;
        B[j * S + o_w[q]] = v;
      } else {
        pend[q] = v;              // column j + 1: read by other waves in THIS step, stored in the next one
        pend_on[q] = true;
      }
    }
    if (rok) {
      double v = ru - u0 * ri0 * ra;
      if (rhs_dc > 0) {
        const double u1 = u1h - u0 * ri0 * a1, bc = rh - ra * ri0 * a1;
        v -= u1 * ri1 * bc;
      }
      u[j + 1 + rhs_dc] = v;
    }
    __syncthreads();
  }
  // the b column of the last step (the barrier that closed it is behind us; the back substitution reads it)
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (pend_on[q]) B[(j - 2) * S + o_w[q]] = pend[q];
  if (j < n) {                       // odd n: the last column has nothing below it
    const double d = B[j * S];
    if (!(d > 0.0) && tid == 0) *fail = 1;
    if (tid == 0) rinv[j] = rcp_f64(d);
  }
  __syncthreads();
}

// from_acc: the system is read straight from the accumulators (single-GPU solve of a small system: no conversion launch)
__global__ __launch_bounds__(kBandThreads) void ba_solve_band_kernel(BaWork wk, int n, float lm, float ep,
                                                                    int lds_doubles, int force_bw, int from_acc) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  extern __shared__ double bsm[];
  __shared__ int fail, bw_s;
  const int tid = threadIdx.x;
  const double* A = wk.Hd;          // (n + 1) x n row-major, lower triangle; row n = right-hand side
  // force_bw >= 0: small dense systems (n <= 64) come here directly as a band of width n - 1, without the
  // bandwidth kernel and without the status[3] handshake
  if (tid == 0) {
    fail = (wk.status[0] & (BA_ST_M_MISMATCH | BA_ST_NONFINITE)) ? 1 : 0;
    bw_s = force_bw >= 0 ? force_bw : wk.status[3] >> 1;
  }
  __syncthreads();
  const int bw = bw_s;
  const int S = bw + 1;             // band row: B[r][k] = H[r][r - k], k = 0 .. bw
  if (bw >= 64 || (size_t)n * S + 2 * (size_t)n > (size_t)lds_doubles) {
    return;                                                   // not solved: the fused kernel takes over
  }
  double* B = bsm;
  double* u = B + (size_t)n * S;    // right-hand side
  double* rinv = u + n;             // 1 / d_j
  for (int base = tid; base < n * S; base += kBandThreads * 8) {
    double v[8];
    int rr[8], kk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = base + q * kBandThreads;
      const int r = idx / S, k = S - 1 - (idx - r * S);       // consecutive threads -> consecutive columns
      rr[q] = r; kk[q] = k;
      v[q] = (idx < n * S && k <= r) ? (from_acc ? acc_value(wk.Hacc, (size_t)r * n + r - k) : A[(size_t)r * n + r - k]) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (base + q * kBandThreads >= n * S) continue;
      double x = v[q];
      if (kk[q] == 0) x += (double)ep + (double)lm * x;
      B[rr[q] * S + kk[q]] = x;
    }
  }
  for (int i = tid; i < n; i += kBandThreads)
    u[i] = from_acc ? acc_value(wk.Hacc, (size_t)n * n + i) : A[(size_t)n * n + i];
  if ((bw + 1) * (bw + 2) / 2 <= 4 * kBandThreads) band_eliminate<4>(B, u, rinv, &fail, n, bw);
  else band_eliminate<9>(B, u, rinv, &fail, n, bw);      // bw <= 63: 2080 window entries at most
  if (fail) {
    for (int i = tid; i < n; i += kBandThreads) wk.dx[i] = 0.0f;
    if (tid == 0) {
      atomicOr(&wk.status[0], BA_ST_CHOL_FAILED);
      atomicAdd(&wk.status[2], 1);
      if (force_bw < 0) wk.status[3] = (bw << 1) | 1;
    }
    return;
  }
  if (tid < 64) {
    double srow = 0.0;                               // running sum of the row (mod 64) this lane owns
    // operands of column j are fetched one iteration ahead: the chain per column is sub, mul, readlane, fma
    double un = u[n - 1], rn = rinv[n - 1];
    int dn = (n - 1 - tid) & 63;                     // row (n - 1) - dn lives in this lane
    double bn = B[(n - 1) * S + min(dn, bw)];
    for (int j = n - 1; j >= 0; --j) {
      const double uc = un, rc = rn, bc = bn;
      const int dist = dn;
      if (j > 0) {
        un = u[j - 1]; rn = rinv[j - 1];
        dn = (j - 1 - tid) & 63;
        bn = B[(j - 1) * S + min(dn, bw)];
      }
      const int lo = j & 63;
      const double xj_own = (uc - srow) * rc;
      const unsigned lo32 = __builtin_amdgcn_readlane((int)__double2loint(xj_own), lo);
      const unsigned hi32 = __builtin_amdgcn_readlane((int)__double2hiint(xj_own), lo);
      const double xj = __hiloint2double((int)hi32, (int)lo32);
      if (tid == lo) { srow = 0.0; wk.dx[j] = (float)xj; }
      if (dist >= 1 && dist <= bw && j - dist >= 0) srow = fma(bc, xj, srow);
    }
  }
  if (tid == 0 && force_bw < 0) wk.status[3] = (bw << 1) | 1;   // solved: ba_solve_fused_kernel returns at once
}

// ---- medium systems (kSolveMaxN < n <= kFusedMaxN): the whole solve in ONE workgroup of 1024 threads -------------
// Blocked right-looking Cholesky (32-column blocks) on the fp64 system in HBM (it is L2 resident: 0.7 MB at
// n = 294), one launch instead of ~3 per block.  The right-hand side is row n of the same buffer (vd follows
// Hd), so it is carried through the panel / trailing update like any other row and ends up as y = L^-1 b;
// the back substitution is blocked the same way.  Fixed operation order: every rank of a sharded run
// factors the same all-reduced system to the same bits.
//   prologue:   damping; the half bandwidth bw comes from ba_solve_band_kernel (which solved the system itself
//               if the band fits into LDS - then this kernel returns at once).  Cholesky fill stays inside
//               the band, so every block only touches the bw rows below it (+ the rhs row); a graph with
//               loop closures degrades to the dense cost.
//   per block:  diagonal block -> LDS, 32 elimination steps with ONE barrier each on unscaled columns; the
//               same row operations applied to an identity give L11^-1 for free (threads (r, c <= j) are
//               idle otherwise), the square roots are taken once at the end.  L11^-1 is parked in the unused
//               strict upper triangle of the block for the back substitution.
//               panel X = A21 L11^-T as a small GEMM (row in registers, 8 independent dot products per
//               thread against L11^-1 read as LDS broadcasts) - a forward substitution per row is one long
//               dependent chain of LDS reads.
//               trailing update: 8 x 4 register tiles, the panel read from LDS (columns interleaved across
//               lanes: conflict-free reads, coalesced updates).
constexpr int kCB = 32;           // block size
constexpr int kCBP = kCB + 1;     // padded row of the LDS copies
constexpr int kFusedMaxN = 540;   // panel (n + 1 - 32 rows) x 33 doubles must fit into LDS next to the blocks

__global__ __launch_bounds__(1024) void ba_solve_fused_kernel(BaWork wk, int n, float lm, float ep) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  extern __shared__ double fsm[];
  double* Dl = fsm;                 // [32][33] diagonal block (unscaled columns), then L11
  double* Wl = Dl + kCB * kCBP;     // [32][33] row operations applied to I, then L11^-1
  double* P = Wl + kCB * kCBP;      // [rows below][33] panel of the current block; later x[n]
  __shared__ int fail, bw_s;
  const int tid = threadIdx.x;
  double* A = wk.Hd;                // (n + 1) x n row-major; row n = right-hand side (wk.vd)
  // status[3] = (half bandwidth << 1) | solved, left by ba_solve_band_kernel which always runs first
  if (tid == 0) { fail = (wk.status[0] & (BA_ST_M_MISMATCH | BA_ST_NONFINITE)) ? 1 : 0; bw_s = wk.status[3]; }
  __syncthreads();
  if (tid == 0) wk.status[3] = 0;      // consumed (everybody has read it above)
  if (bw_s & 1) return;
  for (int i = tid; i < n; i += 1024) {
    const double v = A[(size_t)i * n + i];
    A[(size_t)i * n + i] = v + (double)ep + (double)lm * v;
  }
  __syncthreads();
  const int bw = bw_s >> 1;
  const int br = tid >> 5, bc = tid & 31;   // element of the 32 x 32 block owned during the diagonal phase
  for (int j0 = 0; j0 < n; j0 += kCB) {
    const int nb = min(kCB, n - j0);
    const int rem = min(n - j0 - nb, bw);   // matrix rows below the block that can be non-zero; + the rhs row
    const int base = j0 + nb;
    // ---- diagonal block
    if (br < nb && bc <= br) {
      Dl[br * kCBP + bc] = A[(size_t)(j0 + br) * n + j0 + bc];
      Wl[br * kCBP + bc] = br == bc ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
      const double d = Dl[j * kCBP + j];
      if (!(d > 0.0) && tid == 0) fail = 1;   // also NaN; the (garbage) result is discarded below
      if (br < nb && br > j) {
        const double mlt = Dl[br * kCBP + j] / d;
        if (bc > j && bc <= br) Dl[br * kCBP + bc] -= mlt * Dl[bc * kCBP + j];
        else if (bc <= j) Wl[br * kCBP + bc] -= mlt * Wl[j * kCBP + bc];
      }
      __syncthreads();
    }
    double lval = 0.0, wval = 0.0;
    if (br < nb && bc <= br) {
      const double sc = sqrt(Dl[bc * kCBP + bc]), sr = sqrt(Dl[br * kCBP + br]);
      lval = (br == bc) ? sc : Dl[br * kCBP + bc] / sc;
      wval = Wl[br * kCBP + bc] / sr;
    }
    __syncthreads();
    if (br < nb && bc <= br) {
      Dl[br * kCBP + bc] = lval;
      Wl[br * kCBP + bc] = wval;
      A[(size_t)(j0 + br) * n + j0 + bc] = lval;
      if (bc < br) A[(size_t)(j0 + bc) * n + j0 + br] = wval;   // L11^-1 (strictly lower part), transposed
    }
    // ---- panel: X = A21 L11^-T.  Thread = (row t, parity g of its 16 columns); rows base .. base + rem - 1
    // and the rhs row n (t == rem)
    for (int idx = tid; idx < (rem + 1) * kCB; idx += 1024) {   // A21 (+ rhs row) -> LDS, coalesced
      const int t = idx >> 5, m = idx & 31;
      P[t * kCBP + m] = m < nb ? A[(size_t)(t < rem ? base + t : n) * n + j0 + m] : 0.0;
    }
    __syncthreads();                         // A21 staged, L11^-1 complete
    for (int t0 = 0; t0 <= rem; t0 += 256) {
      const int t = min(t0 + (tid & 255), rem), g = tid >> 8;   // columns 4 q + g: wave-uniform -> broadcast reads
      const bool on = t0 + (tid & 255) <= rem;
      double x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = 0.0;
      const double* wg = Wl + g * kCBP;
      const double* ap = P + t * kCBP;
#pragma unroll 2
      for (int m = 0; m < kCB; ++m) {          // (not fully unrolled: the 288 LDS reads get hoisted and spill)
        const double am = ap[m];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = fma(am, (m <= 4 * q + g) ? wg[4 * q * kCBP + m] : 0.0, x[q]);
      }
      __syncthreads();                       // every reader of these rows is done: overwrite them with X
      if (on) {
        double* arow = A + (size_t)(t < rem ? base + t : n) * n + j0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = 4 * q + g;
          if (c < nb) arow[c] = x[q];
          P[t * kCBP + c] = c < nb ? x[q] : 0.0;
        }
      }
    }
    __syncthreads();
    // ---- trailing update A22 -= X X^T on the lower triangle (+ the rhs row): tile = rows 8 tr .. 8 tr + 7,
    // columns tc + {0, 1, 2, 3} * TC
    if (rem > 0) {
      const int TR = (rem + 1 + 7) / 8, TC = (rem + 3) / 4;
      for (int t = tid; t < TR * TC; t += 1024) {
        const int tr = t / TC, tc = t - tr * TC;
        const int r0 = 8 * tr, rmax = min(r0 + 7, rem);
        if (tc > rmax) continue;              // entirely above the diagonal
        double acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[i][q] = 0.0;
        int roff[8], coff[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) roff[i] = min(r0 + i, rem) * kCBP;
#pragma unroll
        for (int q = 0; q < 4; ++q) coff[q] = min(tc + q * TC, rem - 1) * kCBP;
        for (int k = 0; k < kCB; ++k) {
          double av[8], bv[4];
#pragma unroll
          for (int i = 0; i < 8; ++i) av[i] = P[roff[i] + k];
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[q] = P[coff[q] + k];
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = fma(av[i], bv[q], acc[i][q]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = r0 + i;
          if (r > rem) continue;
          double* arow = A + (size_t)(r < rem ? base + r : n) * n + base;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = tc + q * TC;
            if (c < rem && c <= r) arow[c] -= acc[i][q];
          }
        }
      }
    }
    __syncthreads();
  }
  if (fail) {
    for (int i = tid; i < n; i += 1024) wk.dx[i] = 0.0f;
    if (tid == 0) {
      atomicOr(&wk.status[0], BA_ST_CHOL_FAILED);
      atomicAdd(&wk.status[2], 1);
    }
    return;
  }
  // ---- back substitution L^T x = y (y = row n), blocked from the last block up:
  // x_blk = L11^-T y_blk (L11^-1 from the upper triangle of the block), then y[r] -= sum_c L[j0 + c][r] x[j0 + c]
  // for the rows r < j0 inside the band
  double* xs = P;
  for (int i = tid; i < n; i += 1024) xs[i] = A[(size_t)n * n + i];
  __syncthreads();
  for (int j0 = ((n - 1) / kCB) * kCB; j0 >= 0; j0 -= kCB) {
    const int nb = min(kCB, n - j0);
    if (br < nb && bc < nb) {   // Wl[r][c] = L11^-1[r][c] for r >= c
      double v = 0.0;
      if (bc < br) v = A[(size_t)(j0 + bc) * n + j0 + br];
      else if (bc == br) v = 1.0 / A[(size_t)(j0 + br) * n + j0 + br];
      Wl[br * kCBP + bc] = v;
    }
    __syncthreads();
    if (tid < nb) {             // x[c] = sum_{r >= c} L11^-1[r][c] y[r]
      double v = 0.0;
      for (int r = tid; r < nb; ++r) v = fma(Wl[r * kCBP + tid], xs[j0 + r], v);
      Dl[tid] = v;
    }
    __syncthreads();
    if (tid < nb) xs[j0 + tid] = Dl[tid];
    __syncthreads();
    for (int r = max(0, j0 - bw) + tid; r < j0; r += 1024) {   // rows of L: coalesced in r
      double v = xs[r];
      for (int c = 0; c < nb; ++c) v -= A[(size_t)(j0 + c) * n + r] * xs[j0 + c];
      xs[r] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 1024) wk.dx[i] = (float)xs[i];
}

// ---- large systems (n > kFusedMaxN): the same blocked algorithm, one launch per phase and block so that the
// O(n^3) trailing update spreads over the chip.  The right-hand side rides along as row n here too.

// diagonal block: elimination on unscaled columns + the same row operations on an identity (L11^-1), like the
// diagonal phase of ba_solve_fused_kernel; L11 goes back into the lower triangle, L11^-1 (strictly lower
// part, transposed) into the unused upper triangle of the block
__global__ __launch_bounds__(1024) void chol_diag_inv_kernel(BaWork wk, int n, int j0, int nb) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ double Dl[kCB * kCBP], Wl[kCB * kCBP];
  __shared__ int fail;
  const int tid = threadIdx.x, br = tid >> 5, bc = tid & 31;
  double* A = wk.Hd;
  if (tid == 0) fail = 0;
  if (wk.status[0] & (BA_ST_SOLVE_ABORT | BA_ST_M_MISMATCH)) return;
  if (br < nb && bc <= br) {
    Dl[br * kCBP + bc] = A[(size_t)(j0 + br) * n + j0 + bc];
    Wl[br * kCBP + bc] = br == bc ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    const double d = Dl[j * kCBP + j];
    if (!(d > 0.0) && tid == 0) fail = 1;
    if (br < nb && br > j) {
      const double mlt = Dl[br * kCBP + j] / d;
      if (bc > j && bc <= br) Dl[br * kCBP + bc] -= mlt * Dl[bc * kCBP + j];
      else if (bc <= j) Wl[br * kCBP + bc] -= mlt * Wl[j * kCBP + bc];
    }
    __syncthreads();
  }
  if (fail) {
    if (tid == 0) {
      atomicOr(&wk.status[0], BA_ST_CHOL_FAILED | BA_ST_SOLVE_ABORT);
      atomicAdd(&wk.status[2], 1);
    }
    return;
  }
  if (br < nb && bc <= br) {
    const double sc = sqrt(Dl[bc * kCBP + bc]), sr = sqrt(Dl[br * kCBP + br]);
    A[(size_t)(j0 + br) * n + j0 + bc] = (br == bc) ? sc : Dl[br * kCBP + bc] / sc;
    if (bc < br) A[(size_t)(j0 + bc) * n + j0 + br] = Wl[br * kCBP + bc] / sr;
  }
}

// panel X = A21 L11^-T as a GEMM, 128 rows per workgroup (row t: matrix row j0 + nb + t, t == rem: rhs row n);
// thread = (row, residue g of its 4 columns 8 q + g): wave-uniform columns -> the L11^-1 reads are broadcasts
constexpr int kPanelRows = 128;
__global__ __launch_bounds__(1024) void chol_panel_gemm_kernel(BaWork wk, int n, int j0, int nb) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ double Wl[kCB * kCBP], P[kPanelRows * kCBP];
  if (wk.status[0] & (BA_ST_SOLVE_ABORT | BA_ST_M_MISMATCH)) return;
  const int tid = threadIdx.x, br = tid >> 5, bc = tid & 31;
  double* A = wk.Hd;
  const int base = j0 + nb, rem = n - base;
  if (br < nb && bc < nb) {              // Wl[r][c] = L11^-1[r][c], r >= c
    double v = 0.0;
    if (bc < br) v = A[(size_t)(j0 + bc) * n + j0 + br];
    else if (bc == br) v = 1.0 / A[(size_t)(j0 + br) * n + j0 + br];
    Wl[br * kCBP + bc] = v;
  }
  const int t0 = blockIdx.x * kPanelRows;
  for (int idx = tid; idx < kPanelRows * kCB; idx += 1024) {
    const int tl = idx >> 5, m = idx & 31, t = t0 + tl;
    P[tl * kCBP + m] = (t <= rem && m < nb) ? A[(size_t)(t < rem ? base + t : n) * n + j0 + m] : 0.0;
  }
  __syncthreads();
  const int tl = tid & (kPanelRows - 1), g = tid >> 7, t = t0 + tl;
  double x[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = 0.0;
  const double* wg = Wl + g * kCBP;
  const double* ap = P + tl * kCBP;
#pragma unroll 4
  for (int m = 0; m < kCB; ++m) {
    const double am = ap[m];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = fma(am, (m <= 8 * q + g) ? wg[8 * q * kCBP + m] : 0.0, x[q]);
  }
  if (t <= rem) {
    double* arow = A + (size_t)(t < rem ? base + t : n) * n + j0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 8 * q + g;
      if (c < nb) arow[c] = x[q];
    }
  }
}

// back substitution L^T x = y (y = row n of the buffer), one launch per 32-column block from the last block up:
// every workgroup forms x_blk = L11^-T y_blk itself (32 x 32, L11^-1 from the block's upper triangle) and
// subtracts L[blk rows][r] x_blk from its slice of y (rows of L: coalesced in r) - the O(n^2) part of the sweep
// spread over the chip instead of one workgroup; workgroup 0 also stores x_blk (and zeros after a failure).
__global__ __launch_bounds__(256) void chol_backsub_block_kernel(BaWork wk, int n, int j0, int nb) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  __shared__ double Wl[kCB * kCBP], yb[kCB], xb[kCB];
  const int tid = threadIdx.x;
  double* A = wk.Hd;
  double* y = A + (size_t)n * n;
  if (wk.status[0] & (BA_ST_SOLVE_ABORT | BA_ST_M_MISMATCH)) {
    if (blockIdx.x == 0 && tid < nb) wk.dx[j0 + tid] = 0.0f;
    return;
  }
  for (int idx = tid; idx < kCB * kCB; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    double v = 0.0;
    if (r < nb && c < nb) {
      if (c < r) v = A[(size_t)(j0 + c) * n + j0 + r];
      else if (c == r) v = 1.0 / A[(size_t)(j0 + r) * n + j0 + r];
    }
    Wl[r * kCBP + c] = v;
  }
  if (tid < kCB) yb[tid] = tid < nb ? y[j0 + tid] : 0.0;
  __syncthreads();
  if (tid < kCB) {                     // x[c] = sum_{r >= c} L11^-1[r][c] y[r]
    double v = 0.0;
    for (int r = tid; r < nb; ++r) v = fma(Wl[r * kCBP + tid], yb[r], v);
    xb[tid] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < nb) wk.dx[j0 + tid] = (float)xb[tid];
  const int r = blockIdx.x * 256 + tid;
  if (r < j0) {
    double v = y[r];
    for (int c = 0; c < nb; ++c) v -= A[(size_t)(j0 + c) * n + r] * xb[c];
    y[r] = v;
  }
}

// ------------------------------------------------------------------------------------
// update: back-substitution for dz, disparity add, pose retraction
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBaThreads) void ba_update_kernel(
    BaWork wk, float* __restrict__ poses, float* __restrict__ disps,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int HW, int t0, int t1,
    int motion_only, int depth_only, float* __restrict__ dx_out, float* __restrict__ dz_out) {
  if (ba_skipped(wk)) return;        // glorie_ba_set_gate
  constexpr int EB = 128;  // edges per batch
  __shared__ float ys[EB * 6];
  const int tid = threadIdx.x;
  const int s = blockIdx.y;
  if (wk.status[0] & BA_ST_M_MISMATCH) return;
  const int P = t1 - t0;
  // pose retraction by workgroup (0,0): nobody reads poses in this launch (L_e is cached)
  if (blockIdx.x == 0 && s == 0) {
    for (int p = tid; p < P; p += kBaThreads) {
      float xi[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) xi[q] = wk.dx[p * 6 + q];
      if (dx_out) {
#pragma unroll
        for (int q = 0; q < 6; ++q) dx_out[p * 6 + q] = xi[q];
      }
      if (!depth_only) {
        float* pp = poses + (size_t)(t0 + p) * 7;
        const Pose g = se3_retract(xi, load_pose(pp));
        pp[0] = g.t.x; pp[1] = g.t.y; pp[2] = g.t.z;
        pp[3] = g.q.x; pp[4] = g.q.y; pp[5] = g.q.z; pp[6] = g.q.w;
      }
    }
  }
  if (motion_only) return;
  const int k = wk.kx[s];
  const int e0 = wk.csr_ptr[s];
  const int deg = wk.csr_ptr[s + 1] - e0;
  const int pk = k - t0;
  const bool self_on = (pk > 0 && pk < P);  // "<= 0 skipped" quirk (droid_kernels.cu:1105)
  const int px = blockIdx.x * kBaThreads + tid;
  float dw = 0.0f;
  for (int a0 = 0; a0 < deg; a0 += EB) {
    const int nb = min(EB, deg - a0);
    __syncthreads();
    for (int idx = tid; idx < nb * 6; idx += kBaThreads) {
      const int a = idx / 6, r = idx % 6;
      const int n = wk.csr_edge[e0 + a0 + a];
      const int pj = (int)jj[n] - t0;
      float y = (pj > 0 && pj < P) ? wk.dx[pj * 6 + r] : 0.0f;
      if (self_on) {
        const float* L = wk.Ledge + (size_t)n * 36;
        float acc = 0.0f;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += L[m * 6 + r] * wk.dx[pk * 6 + m];  // (L^T dx_k)[r]
        y -= acc;
      }
      ys[idx] = y;
    }
    __syncthreads();
    if (px < HW) {
      for (int a = 0; a < nb; ++a) {
        const int n = wk.csr_edge[e0 + a0 + a];
#pragma unroll
        for (int r = 0; r < 6; ++r) dw += wk.Eij[((size_t)n * 6 + r) * HW + px] * ys[a * 6 + r];
      }
    }
  }
  if (px >= HW) return;
  const float dz = wk.Q[(size_t)s * HW + px] * (wk.W[(size_t)s * HW + px] - dw);
  disps[(size_t)k * HW + px] += dz;
  if (dz_out) dz_out[(size_t)s * HW + px] = dz;
}

int ba_prepare(const BaWork& wk, const int64_t* ii, int B, int N, int M, int t0, int t1,
               hipStream_t st) {
  const size_t prep_lds = sizeof(int) * ((size_t)3 * B + (size_t)N);
  if (prep_lds > 150 * 1024) return GLORIE_EUNSUPPORTED;
  static PerDeviceOnce attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_prepare_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  hipLaunchKernelGGL(ba_prepare_kernel, dim3(1), dim3(1024), prep_lds, st, wk, ii, B, N, M, t0, t1);
  return check_launch();
}

}  // namespace glorie

using namespace glorie;

namespace glorie {

struct BaPlan {
  BaWork wk;
  int B, N, M, h, w, HW, t0, t1, P, n6, ppt, chunk_px, nchunks;
  size_t scratch_used;
};

// validates sizes, carves the scratch arena (identical layout for identical sizes, so a
// build_system call and the following solve_update call see the same buffers)
static int ba_plan(glorie_ctx* ctx, int B, int N, int M, int h, int w, int t0, int t1,
                   double* hv_ext, BaPlan& pl) {
  if (!ctx || B < 0 || N < 0 || M < 0 || h < 0 || w < 0 || t1 < t0) return GLORIE_EINVAL;
  if (B > kMaxFramesLds || N > kMaxEdgesLds || t1 > B) return GLORIE_EUNSUPPORTED;
  pl.B = B; pl.N = N; pl.M = M; pl.h = h; pl.w = w; pl.HW = h * w; pl.t0 = t0; pl.t1 = t1;
  pl.P = t1 - t0; pl.n6 = 6 * pl.P;
#ifdef EXP_BA_CHUNK_WGS
  constexpr long kChunkWgs = EXP_BA_CHUNK_WGS;
#else
  constexpr long kChunkWgs = 256;
#endif
  // pixel chunking: enough workgroups to fill 256 CUs, at most 4 pixels per thread
  int ppt = 1;
  const int per1 = (pl.HW + kBaThreads - 1) / kBaThreads;
  while (ppt < 4 && (long)M * ((per1 + ppt - 1) / ppt) > kChunkWgs) ++ppt;       // (round 5: 1024 -> 256 - every (frame, chunk) workgroup scatters its block with two integer atomics per entry)
  pl.ppt = ppt;
  pl.chunk_px = kBaThreads * ppt;
  pl.nchunks = (pl.HW + pl.chunk_px - 1) / pl.chunk_px;
  const size_t HW = pl.HW, n6 = pl.n6;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_slot = carve(sizeof(int) * (size_t)B);
  const size_t o_kx = carve(sizeof(int) * (size_t)(M + 1));
  const size_t o_ptr = carve(sizeof(int) * (size_t)(M + 2));
  const size_t o_edge = carve(sizeof(int) * (size_t)N);
  const size_t o_L = carve(sizeof(float) * 36 * (size_t)N);
  const size_t o_E = carve(sizeof(float) * 6 * (size_t)N * HW);
  const size_t o_Hp = carve(sizeof(float) * 27 * (size_t)N * pl.nchunks);
  const size_t o_Q = carve(sizeof(float) * (size_t)M * HW);
  const size_t o_W = carve(sizeof(float) * (size_t)M * HW);
  const size_t o_CW = carve(sizeof(float) * 2 * (size_t)N * HW);
  const size_t o_Hd = carve(sizeof(double) * (n6 * n6 + n6));
  const size_t o_acc = carve(2 * sizeof(unsigned long long) * (n6 * n6 + n6));
  const size_t o_dx = carve(sizeof(float) * n6);
  GLORIE_TRY(ctx_reserve(ctx, off));
  pl.scratch_used = off;
  char* base = reinterpret_cast<char*>(ctx->scratch);
  BaWork& wk = pl.wk;
  wk.slot_of_frame = reinterpret_cast<int*>(base + o_slot);
  wk.kx = reinterpret_cast<int*>(base + o_kx);
  wk.csr_ptr = reinterpret_cast<int*>(base + o_ptr);
  wk.csr_edge = reinterpret_cast<int*>(base + o_edge);
  wk.status = ctx->dstatus;
  wk.Ledge = reinterpret_cast<float*>(base + o_L);
  wk.Eij = reinterpret_cast<float*>(base + o_E);
  wk.Hpart = reinterpret_cast<float*>(base + o_Hp);
  wk.Q = reinterpret_cast<float*>(base + o_Q);
  wk.W = reinterpret_cast<float*>(base + o_W);
  wk.CWpart = reinterpret_cast<float*>(base + o_CW);
  // the dense system [H (n6 x n6) | v (n6)] is one contiguous fp64 buffer so that a
  // multi-GPU caller can all-reduce it in a single collective
  wk.Hd = hv_ext ? hv_ext : reinterpret_cast<double*>(base + o_Hd);
  wk.vd = wk.Hd + n6 * n6;
  wk.Hacc = reinterpret_cast<unsigned long long*>(base + o_acc);
  wk.dx = reinterpret_cast<float*>(base + o_dx);
  wk.gate = ctx->ba_gate;
  wk.gate_hits = nullptr;
  static PerDeviceOnce attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_fused_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_band_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  }
  return GLORIE_OK;
}

// Jacobians + Schur gram + pose blocks -> dense reduced system in wk.Hd / wk.vd
static int ba_build_system(const BaPlan& pl, const float* poses, const float* disps,
                           const float* intrinsics, const float* disps_sens, const float* targets,
                           const float* weights, const float* eta, const int64_t* ii,
                           const int64_t* jj, int flags, bool to_f64, hipStream_t st) {
  const BaWork& wk = pl.wk;
  const int motion_only = flags & 1, hwc = (flags & GLORIE_BA_TARGETS_HWC) ? 1 : 0;
  hipLaunchKernelGGL(ba_jacobian_kernel, dim3(pl.nchunks, pl.N), dim3(kBaThreads), 0, st, wk, poses,
                     disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, pl.HW, pl.w,
                     pl.nchunks, pl.ppt, motion_only, hwc, (long)pl.n6 * pl.n6 + pl.n6);
  if (!motion_only)        // Schur products AND the pose-pose blocks (one launch less per iteration)
    hipLaunchKernelGGL(ba_gram_kernel, dim3(pl.nchunks, pl.M), dim3(kBaThreads), 0, st, wk, ii, jj, pl.HW,
                       pl.chunk_px, pl.t0, pl.t1, pl.nchunks, eta, disps, disps_sens);
  else
    hipLaunchKernelGGL(ba_assemble_kernel, dim3(pl.N), dim3(64), 0, st, wk, ii, jj, pl.nchunks, pl.t0, pl.t1);
  if (to_f64) {
    const long total = (long)pl.n6 * pl.n6 + pl.n6;
    hipLaunchKernelGGL(ba_system_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wk, total);
  }
  return check_launch();
}

// fp64 solve of the (all-reduced) system, then back-substitution + retraction
static int ba_solve_update(const BaPlan& pl, float* poses, float* disps, const int64_t* ii,
                           const int64_t* jj, float lm, float ep, int motion_only, int depth_only,
                           float* dx_out, float* dz_out, bool from_acc, hipStream_t st) {
  const BaWork& wk = pl.wk;
  const int n6 = pl.n6;
  if (n6 <= kSolveMaxN) {
    const int bwf = n6 > 1 ? n6 - 1 : 1;
    const int need = n6 * (bwf + 1) + 2 * n6;
    hipLaunchKernelGGL(ba_solve_band_kernel, dim3(1), dim3(kBandThreads), sizeof(double) * (size_t)need, st, wk, n6,
                       lm, ep, need, bwf, from_acc ? 1 : 0);
  } else if (n6 <= kFusedMaxN) {
    const size_t rows = (size_t)(n6 + 1 > kCB ? n6 + 1 - kCB : 1);
    const size_t lds = sizeof(double) * (2 * kCB * kCBP + (rows * kCBP > (size_t)n6 ? rows * kCBP : (size_t)n6));
    const int band_doubles = (160 * 1024 - 256) / (int)sizeof(double);
    hipLaunchKernelGGL(ba_bandwidth_kernel, dim3(n6), dim3(64), 0, st, wk, n6);
    hipLaunchKernelGGL(ba_solve_band_kernel, dim3(1), dim3(kBandThreads), sizeof(double) * (size_t)band_doubles, st,
                       wk, n6, lm, ep, band_doubles, -1, 0);
    hipLaunchKernelGGL(ba_solve_fused_kernel, dim3(1), dim3(1024), lds, st, wk, n6, lm, ep);
  } else {
    hipLaunchKernelGGL(chol_damp_kernel, dim3((n6 + 255) / 256), dim3(256), 0, st, wk, n6, lm, ep);
    for (int j0 = 0; j0 < n6; j0 += kCB) {
      const int nb = (n6 - j0 < kCB) ? (n6 - j0) : kCB;
      const int rem = n6 - j0 - nb;                     // + the rhs row
      hipLaunchKernelGGL(chol_diag_inv_kernel, dim3(1), dim3(1024), 0, st, wk, n6, j0, nb);
      hipLaunchKernelGGL(chol_panel_gemm_kernel, dim3((rem + kPanelRows) / kPanelRows), dim3(1024), 0, st, wk, n6, j0, nb);
      if (rem > 0) {
        const int tl = (rem + kTrailTile) / kTrailTile;
        hipLaunchKernelGGL(chol_trail_kernel, dim3(tl, tl), dim3(256), 0, st, wk, n6, j0, nb);
      }
    }
    for (int j0 = ((n6 - 1) / kCB) * kCB; j0 >= 0; j0 -= kCB) {
      const int nb = (n6 - j0 < kCB) ? (n6 - j0) : kCB;
      hipLaunchKernelGGL(chol_backsub_block_kernel, dim3(j0 > 0 ? (j0 + 255) / 256 : 1), dim3(256), 0, st, wk, n6, j0,
                         nb);
    }
  }
  const int upd_chunks = (pl.HW + kBaThreads - 1) / kBaThreads;
  hipLaunchKernelGGL(ba_update_kernel, dim3(upd_chunks, pl.M), dim3(kBaThreads), 0, st, wk, poses,
                     disps, ii, jj, pl.HW, pl.t0, pl.t1, motion_only, depth_only, dx_out, dz_out);
  return check_launch();
}

// the zero system of a rank without edges, behind the device gate of glorie_ba_set_gate
__global__ __launch_bounds__(256) void ba_zero_gated_kernel(const int* __restrict__ gate, int* __restrict__ dstatus,
                                                            double* __restrict__ hv, size_t n) {
  if (*gate != 0) return;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < 4) dstatus[i] = 0;
  if (i < n) hv[i] = 0.0;
}

}  // namespace glorie

extern "C" int glorie_ba(glorie_ctx* ctx, float* poses, float* disps, const float* intrinsics,
                         const float* disps_sens, const float* targets, const float* weights,
                         const float* eta, const int64_t* ii, const int64_t* jj, int B, int N,
                         int M, int h, int w, int t0, int t1, int iterations, float lm, float ep,
                         int motion_only, int depth_only, float* dx_out, float* dz_out,
                         void* stream) {
  if (iterations < 0) return GLORIE_EINVAL;
  const int flags = motion_only;          // bit 0: motion only, GLORIE_BA_TARGETS_HWC: target layout
  motion_only = flags & 1;
  BaPlan pl;
  GLORIE_TRY(ba_plan(ctx, B, N, M, h, w, t0, t1, nullptr, pl));
  if (N == 0 || pl.P == 0 || pl.HW == 0 || iterations == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !targets || !weights || !ii || !jj) return GLORIE_EINVAL;
  if (!motion_only && !eta) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  GLORIE_TRY(ctx_poison(ctx, pl.scratch_used, st));
  {
    BaWork w0 = pl.wk;                          // the first launch of a gated call counts itself (if it runs)
    if (ctx->ba_gate && ctx->ba_gate_count) { w0.gate_hits = ctx->ba_gate_hits; ctx->ba_gate_count = false; }
    GLORIE_TRY(ba_prepare(w0, ii, B, N, M, t0, t1, st));
  }
  // small systems are read by the band solver straight from the accumulators (no conversion launch)
  const bool direct = pl.n6 <= kSolveMaxN;
  for (int it = 0; it < iterations; ++it) {
    GLORIE_TRY(ba_build_system(pl, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
                               flags, !direct, st));
    GLORIE_TRY(ba_solve_update(pl, poses, disps, ii, jj, lm, ep, motion_only, depth_only, dx_out,
                               dz_out, direct, st));
  }
  return GLORIE_OK;
}

extern "C" int glorie_ba_build_system(glorie_ctx* ctx, const float* poses, const float* disps,
                                      const float* intrinsics, const float* disps_sens,
                                      const float* targets, const float* weights, const float* eta,
                                      const int64_t* ii, const int64_t* jj, int B, int N, int M,
                                      int h, int w, int t0, int t1, int motion_only, double* hv_out,
                                      void* stream) {
  BaPlan pl;
  GLORIE_TRY(ba_plan(ctx, B, N, M, h, w, t0, t1, hv_out, pl));
  if (!hv_out || pl.P == 0) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0 || pl.HW == 0) {  // a rank without edges contributes a zero system
    const size_t nhv = (size_t)pl.n6 * pl.n6 + pl.n6;
    if (ctx->ba_gate) {
      // gated call (glorie_ba_set_gate): the status word and hv_out stay untouched unless the gate is open
      hipLaunchKernelGGL(ba_zero_gated_kernel, dim3((unsigned)((nhv + 255) / 256)), dim3(256), 0, st, ctx->ba_gate,
                         ctx->dstatus, hv_out, nhv);
      return check_launch();
    }
    GLORIE_TRY(check_hip(hipMemsetAsync(ctx->dstatus, 0, 4 * sizeof(int), st)));
    return check_hip(hipMemsetAsync(hv_out, 0, sizeof(double) * nhv, st));
  }
  if (!poses || !disps || !intrinsics || !targets || !weights || !ii || !jj) return GLORIE_EINVAL;
  if (!(motion_only & 1) && !eta) return GLORIE_EINVAL;
  GLORIE_TRY(ctx_poison(ctx, pl.scratch_used, st));
  {
    BaWork w0 = pl.wk;                          // the first launch of a gated call counts itself (if it runs)
    if (ctx->ba_gate && ctx->ba_gate_count) { w0.gate_hits = ctx->ba_gate_hits; ctx->ba_gate_count = false; }
    GLORIE_TRY(ba_prepare(w0, ii, B, N, M, t0, t1, st));
  }
  return ba_build_system(pl, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
                         motion_only, /*to_f64=*/true, st);
}

extern "C" int glorie_ba_solve_update(glorie_ctx* ctx, float* poses, float* disps,
                                      const int64_t* ii, const int64_t* jj, int B, int N, int M,
                                      int h, int w, int t0, int t1, float lm, float ep,
                                      int motion_only, int depth_only, const double* hv,
                                      float* dx_out, float* dz_out, void* stream) {
  BaPlan pl;
  GLORIE_TRY(ba_plan(ctx, B, N, M, h, w, t0, t1, const_cast<double*>(hv), pl));
  if (!hv || !poses || !disps || pl.P == 0) return GLORIE_EINVAL;
  if (N > 0 && (!ii || !jj)) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0 || pl.HW == 0) {
    // no local edges: still solve and retract the (replicated) poses; no depth frames to update
    pl.M = 1;
    return ba_solve_update(pl, poses, disps, ii, jj, lm, ep, /*motion_only=*/1, depth_only, dx_out,
                           nullptr, /*from_acc=*/false, st);
  }
  return ba_solve_update(pl, poses, disps, ii, jj, lm, ep, motion_only, depth_only, dx_out, dz_out,
                         /*from_acc=*/false, st);
}

// Device-side gate for the BA entry points called next on this context (until cleared with run_if_zero = NULL): every kernel
// of glorie_ba / glorie_ba_build_system / glorie_ba_solve_update returns at once unless *run_if_zero == 0 - poses, disparities
// and the status word stay untouched.  The stage-1 fallback of a depth_scale stage (depth_video.py:290-294: "if not success:
// run pose_depth") is enqueued behind the stage unconditionally, gated on the stage's own `any edge left` word: no host
// decision, no stream drain per step, and the whole step stays one hipGraph.  hits (may be NULL): incremented once per gated
// call that did run.
extern "C" int glorie_ba_set_gate(glorie_ctx* ctx, const int* run_if_zero, int* hits) {
  if (!ctx) return GLORIE_EINVAL;
  ctx->ba_gate = run_if_zero;
  ctx->ba_gate_hits = run_if_zero ? hits : nullptr;
  ctx->ba_gate_count = run_if_zero && hits;
  return GLORIE_OK;
}

// ---- exchange format of the reduced system: lower triangle (row r: columns 0..r) followed by v ----
namespace glorie {
__global__ __launch_bounds__(256) void hv_pack_kernel(const double* __restrict__ hv, double* __restrict__ packed,
                                                      int n, int unpack) {
  const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  double* dense = const_cast<double*>(hv);
  if (r == n) {                                   // right-hand side
    if (c < n) {
      const size_t k = (size_t)n * (n + 1) / 2 + c;
      if (unpack) dense[(size_t)n * n + c] = packed[k];
      else packed[k] = hv[(size_t)n * n + c];
    }
    return;
  }
  if (c > r) return;
  const size_t k = (size_t)r * (r + 1) / 2 + c;
  if (unpack) dense[(size_t)r * n + c] = packed[k];
  else packed[k] = hv[(size_t)r * n + c];
}
}  // namespace glorie

extern "C" int glorie_ba_pack_system(const double* hv, double* packed, int n6, int unpack, void* stream) {
  if (!hv || !packed || n6 < 0) return GLORIE_EINVAL;
  if (n6 == 0) return GLORIE_OK;
  hipLaunchKernelGGL(glorie::hv_pack_kernel, dim3((n6 + 255) / 256, n6 + 1), dim3(256), 0, (hipStream_t)stream, hv,
                     packed, n6, unpack);
  return glorie::check_launch();
}

// diagnostic: blocks until `stream` drains, then returns the device status word
// (bit 0: M mismatch, bit 1: degree too large for the gram kernel, bit 2: Cholesky failed)
extern "C" int glorie_ba_status(glorie_ctx* ctx, int* status_out, void* stream) {
  if (!ctx || !status_out) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  GLORIE_TRY(check_hip(hipMemcpyAsync(status_out, ctx->dstatus, sizeof(int) * 4, hipMemcpyDeviceToHost, st)));
  return check_hip(hipStreamSynchronize(st));
}
