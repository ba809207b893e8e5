// Projective geometry kernels for gfx950 (scope rows A6, A7, B12 and the "next" rows
// iproj / depth_filter that share the same helpers).
//
// Reference behaviour restated (not translated) from
//   /root/reference/src/geom/projective_ops.py:18-125      (reproject, python path)
//   /root/reference/src/lib/droid_kernels.cu:518-657       (frame_distance)
//   /root/reference/src/lib/droid_kernels.cu:661-775       (depth_filter)
//   /root/reference/src/lib/droid_kernels.cu:779-850       (iproj)
//   /root/reference/src/modules/droid_net/droid_net.py:9-23 (cvx_upsample)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
// Every fp32 operation of this file (and of the SE3 helpers it inlines) is rounded on its own, in the
// source order of the reference kernels: hipcc's default -ffp-contract=fast would fuse mul+add pairs where
// IT likes (nvcc fuses where nvcc likes - neither is reproducible from the source).  Without contraction the
// integer-valued results (depth_filter counts, validity masks) and the thresholded ones (frame_distance,
// on which graph topology depends) are bit-identical to oracle/geom.py; the kernels are bandwidth-bound.
#pragma clang fp contract(off)
#include "common.hiph"
#include "se3.hiph"

namespace glorie {

// ------------------------------------------------------------------------------------
// reproject: coords1 = pi_j( Gj Gi^-1 * pi_i^-1(disp_i) ),  valid = Z1 > 0.2
// grid (ceil(HW/256), N); edge-uniform data (poses, intrinsics) are scalar loads.
// ------------------------------------------------------------------------------------
// target / motion (both or neither): the motion features of the update operator (factor_graph.py:219-221;
// motion_padded_kernel below) for the same pixel, written as the zero-padded fp16 map of glorie_flow_conv7_padded - the
// pixel grid coords0 is (x, y) itself, so they come for free with the reprojection
__global__ __launch_bounds__(256) void reproject_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, float* __restrict__ coords, float* __restrict__ valid,
    int h, int w, const float2* __restrict__ target = nullptr, _Float16* __restrict__ motion = nullptr,
    float lim = 64.0f) {
  const int n = blockIdx.y;
  const int HW = h * w;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int ix = static_cast<int>(ii[n]);
  const int jx = static_cast<int>(jj[n]);

  const float fxi = intr[ix * 4 + 0], fyi = intr[ix * 4 + 1];
  const float cxi = intr[ix * 4 + 2], cyi = intr[ix * 4 + 3];
  const float fxj = intr[jx * 4 + 0], fyj = intr[jx * 4 + 1];
  const float cxj = intr[jx * 4 + 2], cyj = intr[jx * 4 + 3];

  const Pose gij = (ix == jx) ? stereo_pose()
                              : relative_pose(load_pose(poses + ix * 7), load_pose(poses + jx * 7));
  if (k >= HW) return;
  const int y = k / w, x = k - y * w;

  float X0[4], X1[4];
  X0[0] = ((float)x - cxi) / fxi;
  X0[1] = ((float)y - cyi) / fyi;
  X0[2] = 1.0f;
  X0[3] = disps[(size_t)ix * HW + k];
  se3_act(gij, X0, X1);

  // proj(): Z < 0.5*MIN_DEPTH -> 1 (projective_ops.py:52), MIN_DEPTH = 0.2
  const float Z = (X1[2] < 0.1f) ? 1.0f : X1[2];
  const float d = 1.0f / Z;
  float2 c;
  c.x = fxj * (X1[0] * d) + cxj;
  c.y = fyj * (X1[1] * d) + cyj;
  reinterpret_cast<float2*>(coords)[(size_t)n * HW + k] = c;
  if (valid) valid[(size_t)n * HW + k] = (X1[2] > 0.2f) ? 1.0f : 0.0f;  // X0.z == 1 > 0.2
  if (motion) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const float2 t = target[(size_t)n * HW + k];
    h4 o;
    o[0] = (_Float16)fminf(fmaxf(c.x - (float)x, -lim), lim);
    o[1] = (_Float16)fminf(fmaxf(c.y - (float)y, -lim), lim);
    o[2] = (_Float16)fminf(fmaxf(t.x - c.x, -lim), lim);
    o[3] = (_Float16)fminf(fmaxf(t.y - c.y, -lim), lim);
    *reinterpret_cast<h4*>(motion + ((((size_t)n * (h + 6) + y + 3) * (w + 8)) + x + 3) * 4) = o;
  }
}

// ------------------------------------------------------------------------------------
// frame_distance: one 256-thread workgroup per pair; thread t owns pixels t, t+256, ...
// and the final tree follows the reference's order (128, 64, then 32..1) so the fp32 sum
// order -- which graph topology depends on -- is the same.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float block_tree_sum_256(float v, float* sdata) {
  const int tid = threadIdx.x;
  sdata[tid] = v;
  __syncthreads();
  if (tid < 128) sdata[tid] += sdata[tid + 128];
  __syncthreads();
  if (tid < 64) sdata[tid] += sdata[tid + 64];
  __syncthreads();
  // 64 -> 1 inside one wave; LDS keeps the reference's pairing (t += t+32, +16, ...)
  if (tid < 32) {
    volatile float* s = sdata;
    s[tid] += s[tid + 32];
    s[tid] += s[tid + 16];
    s[tid] += s[tid + 8];
    s[tid] += s[tid + 4];
    s[tid] += s[tid + 2];
    s[tid] += s[tid + 1];
  }
  __syncthreads();
  const float r = sdata[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void frame_distance_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, float* __restrict__ dist, int h, int w, float beta) {
  __shared__ float sdata[256];
  const int b = blockIdx.x;
  const int HW = h * w;
  const int ix = static_cast<int>(ii[b]);
  const int jx = static_cast<int>(jj[b]);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const Pose gij = relative_pose(load_pose(poses + ix * 7), load_pose(poses + jx * 7));

  float accum = 0.f, vld = 0.f, total = 0.f;
  const float* dp = disps + (size_t)ix * HW;
  for (int k = threadIdx.x; k < HW; k += 256) {
    const int y = k / w, x = k - y * w;
    const float u = (float)x, v = (float)y;
    float Xi[4], Xj[4];
    Xi[0] = (u - cx) / fx;
    Xi[1] = (v - cy) / fy;
    Xi[2] = 1.0f;
    Xi[3] = dp[k];
    se3_act(gij, Xi, Xj);
    float du = fx * (Xj[0] / Xj[2]) + cx - u;
    float dv = fy * (Xj[1] / Xj[2]) + cy - v;
    float d = sqrtf(du * du + dv * dv);
    total += beta;
    if (Xj[2] > 0.25f) {  // MIN_DEPTH of the native kernels
      accum += beta * d;
      vld += beta;
    }
    // translation-only term
    Xj[0] = Xi[0] + Xi[3] * gij.t.x;
    Xj[1] = Xi[1] + Xi[3] * gij.t.y;
    Xj[2] = Xi[2] + Xi[3] * gij.t.z;
    du = fx * (Xj[0] / Xj[2]) + cx - u;
    dv = fy * (Xj[1] / Xj[2]) + cy - v;
    d = sqrtf(du * du + dv * dv);
    total += (1.0f - beta);
    if (Xj[2] > 0.25f) {
      accum += (1.0f - beta) * d;
      vld += (1.0f - beta);
    }
  }
  const float A = block_tree_sum_256(accum, sdata);
  const float T = block_tree_sum_256(total, sdata);
  const float V = block_tree_sum_256(vld, sdata);
  if (threadIdx.x == 0) dist[b] = (V / (T + 1e-8f) < 0.75f) ? 1000.0f : A / V;
}

// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iproj_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, float* __restrict__ points, int h, int w) {
  const int b = blockIdx.y;
  const int HW = h * w;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const Pose g = load_pose(poses + b * 7);
  if (k >= HW) return;
  const int y = k / w, x = k - y * w;
  float Xi[4], Xj[4];
  Xi[0] = ((float)x - cx) / fx;
  Xi[1] = ((float)y - cy) / fy;
  Xi[2] = 1.0f;
  Xi[3] = disps[(size_t)b * HW + k];
  se3_act(g, Xi, Xj);
  float* o = points + ((size_t)b * HW + k) * 3;
  o[0] = Xj[0] / Xj[3];
  o[1] = Xj[1] / Xj[3];
  o[2] = Xj[2] / Xj[3];
}

// depth_filter: the reference launches (num, 6, chunks) blocks and atomically adds 1.0 per
// consistent neighbour; here one thread owns a pixel and walks its 6 neighbours, so the
// count is written once (integer-valued, hence identical to any atomic ordering).
__global__ __launch_bounds__(256) void depth_filter_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intr, const int64_t* __restrict__ inds,
    const float* __restrict__ thresh, float* __restrict__ counter, int B, int h, int w) {
  const int b = blockIdx.y;
  const int HW = h * w;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int ix = static_cast<int>(inds[b]);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float t = thresh[b];
  const Pose gi = load_pose(poses + ix * 7);
  if (k >= HW) return;
  const int y = k / w, x = k - y * w;
  float Xi[4], Xj[4];
  Xi[0] = ((float)x - cx) / fx;
  Xi[1] = ((float)y - cy) / fy;
  Xi[2] = 1.0f;
  Xi[3] = disps[(size_t)ix * HW + k];
  float cnt = 0.0f;
  for (int nb = 0; nb < 6; ++nb) {
    const int jx = (nb < 3) ? ix - nb - 1 : ix + nb;  // ix-1..-3, ix+3..+5
    if (jx < 0 || jx >= B) continue;
    const Pose gij = relative_pose(gi, load_pose(poses + jx * 7));
    se3_act(gij, Xi, Xj);
    const float uj = fx * (Xj[0] / Xj[2]) + cx;
    const float vj = fy * (Xj[1] / Xj[2]) + cy;
    const float dj = Xj[3] / Xj[2];
    const int u0 = static_cast<int>(floorf(uj));
    const int v0 = static_cast<int>(floorf(vj));
    if (u0 >= 0 && v0 >= 0 && u0 < w - 1 && v0 < h - 1) {
      const float* dj_map = disps + (size_t)jx * HW;
      const float d00 = dj_map[(v0 + 0) * w + u0 + 0];
      const float d01 = dj_map[(v0 + 0) * w + u0 + 1];
      const float d10 = dj_map[(v0 + 1) * w + u0 + 0];
      const float d11 = dj_map[(v0 + 1) * w + u0 + 1];
      // the reference evaluates abs(1.0/dj - 1.0/d) with double literals, i.e. in fp64
      const double zj = 1.0 / (double)dj;
      const double td = (double)t;
      if (fabs(zj - 1.0 / (double)d00) < td || fabs(zj - 1.0 / (double)d01) < td ||
          fabs(zj - 1.0 / (double)d10) < td || fabs(zj - 1.0 / (double)d11) < td)
        cnt += 1.0f;
    }
  }
  counter[(size_t)b * HW + k] = cnt;
}

// ------------------------------------------------------------------------------------
// cvx_upsample (8x convex upsampling of disparity).  Bandwidth bound on the mask:
// 576 values per low-res pixel.  A thread owns a PAIR of horizontally adjacent low-res
// pixels and one sub-row `a`: mask reads are 4-byte (half2) / 8-byte (float2) and
// coalesced along x, the 16 outputs per thread are 64 contiguous bytes.
// grid (ceil(HW/2/256), 8, M)
// ------------------------------------------------------------------------------------
template <typename MT> struct Pair;
template <> struct Pair<_Float16> { typedef __attribute__((ext_vector_type(2))) _Float16 type; };
template <> struct Pair<float> { typedef float2 type; };

template <typename MT>
__device__ __forceinline__ float round_like_mask(float v, bool softmax_f32) {
  if (softmax_f32) return v;
  return (float)(MT)v;
}

template <typename MT>
__global__ __launch_bounds__(256) void cvx_upsample_kernel(
    const float* __restrict__ disps, const int64_t* __restrict__ ix,
    const MT* __restrict__ mask, float* __restrict__ disps_up, int h, int w, int softmax_f32) {
  typedef typename Pair<MT>::type P2;
  const int HW = h * w;
  const int m = blockIdx.z;
  const int a = blockIdx.y;
  const int pp = blockIdx.x * blockDim.x + threadIdx.x;  // pixel-pair index
  const int p0 = pp * 2;
  if (p0 >= HW) return;
  const bool has2 = (p0 + 1) < HW;
  const int frame = static_cast<int>(ix[m]);
  const float* dmap = disps + (size_t)frame * HW;
  const MT* mk = mask + (size_t)m * 576 * HW;

  // 3x3 neighbourhoods (zero padded) of both pixels
  float nb[2][9];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = p0 + q;
    const int y = p / w, x = p - y * w;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      nb[q][k] = (p < HW && yy >= 0 && yy < h && xx >= 0 && xx < w) ? dmap[yy * w + xx] : 0.0f;
    }
  }

  float outv[2][8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    float lg[2][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const size_t off = (size_t)(k * 64 + a * 8 + b) * HW + p0;
      if (has2 && ((HW & 1) == 0)) {
        const P2 v = *reinterpret_cast<const P2*>(mk + off);
        lg[0][k] = (float)v.x;
        lg[1][k] = (float)v.y;
      } else {
        lg[0][k] = (float)mk[off];
        lg[1][k] = has2 ? (float)mk[off + 1] : 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float mx = lg[q][0];
#pragma unroll
      for (int k = 1; k < 9; ++k) mx = fmaxf(mx, lg[q][k]);
      float e[9], sum = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) { e[k] = expf(lg[q][k] - mx); sum += e[k]; }
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k)    // product and sum rounded separately (mask * up_data, then sum: droid_net.py:19-20)
        acc = __fadd_rn(acc, __fmul_rn(round_like_mask<MT>(e[k] / sum, softmax_f32 != 0), nb[q][k]));
      outv[q][b] = acc;
    }
  }
  const int W8 = 8 * w;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = p0 + q;
    if (p >= HW) break;
    const int y = p / w, x = p - y * w;
    float* o = disps_up + (size_t)frame * 64 * HW + (size_t)(8 * y + a) * W8 + 8 * x;
    reinterpret_cast<float4*>(o)[0] = make_float4(outv[q][0], outv[q][1], outv[q][2], outv[q][3]);
    reinterpret_cast<float4*>(o)[1] = make_float4(outv[q][4], outv[q][5], outv[q][6], outv[q][7]);
  }
}

// motion features of the update operator (factor_graph.py:219-221): per edge and pixel
//   [coords1 - coords0, target - coords1] clamped to +-64, as one channels-last float4
__global__ __launch_bounds__(256) void motion_kernel(const float2* __restrict__ coords1,
                                                     const float2* __restrict__ coords0,
                                                     const float2* __restrict__ target,
                                                     float4* __restrict__ out, long total, int HW, float lim) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float2 c1 = coords1[i], c0 = coords0[i % HW], t = target[i];
  float4 o;
  o.x = fminf(fmaxf(c1.x - c0.x, -lim), lim);
  o.y = fminf(fmaxf(c1.y - c0.y, -lim), lim);
  o.z = fminf(fmaxf(t.x - c1.x, -lim), lim);
  o.w = fminf(fmaxf(t.y - c1.y, -lim), lim);
  out[i] = o;
}

// the same features as the zero-padded fp16 map glorie_flow_conv7_padded reads ([N][h+6][w+8][4] halfs, interior at row 3,
// pixel 3; the borders are never written and must be zero): the flow encoder rounds its input to fp16 anyway
typedef __attribute__((ext_vector_type(4))) _Float16 mot4;
__global__ __launch_bounds__(256) void motion_padded_kernel(const float2* __restrict__ coords1,
                                                            const float2* __restrict__ coords0,
                                                            const float2* __restrict__ target,
                                                            _Float16* __restrict__ out, long total, int h, int w, float lim) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int HW = h * w;
  const long n = i / HW;
  const int q = (int)(i - n * HW);
  const int y = q / w, x = q - y * w;
  const float2 c1 = coords1[i], c0 = coords0[q], t = target[i];
  mot4 o;
  o[0] = (_Float16)fminf(fmaxf(c1.x - c0.x, -lim), lim);
  o[1] = (_Float16)fminf(fmaxf(c1.y - c0.y, -lim), lim);
  o[2] = (_Float16)fminf(fmaxf(t.x - c1.x, -lim), lim);
  o[3] = (_Float16)fminf(fmaxf(t.y - c1.y, -lim), lim);
  *reinterpret_cast<mot4*>(out + (((n * (h + 6) + y + 3) * (w + 8)) + x + 3) * 4) = o;
}

// Same operator on a CHANNELS-LAST fp16 mask ([pixel][576], what the 1x1 upmask convolution of
// GraphAgg writes): lane = (sub-row a, pixel q of a group of 8).  For tap k the 8 sub-column
// weights of (a) are 16 contiguous bytes, a wave reads 8 pixels x 128 contiguous bytes per tap and
// writes 8 sub-rows x 256 contiguous bytes.  grid (ceil(HW/8)*64/256, M)
typedef __attribute__((ext_vector_type(8))) _Float16 mask8;

__global__ __launch_bounds__(256) void cvx_upsample_nhwc_kernel(
    const float* __restrict__ disps, const int64_t* __restrict__ ix, const _Float16* __restrict__ mask,
    int ms, float* __restrict__ disps_up, int h, int w, int softmax_f32) {
  const int HW = h * w;
  const int m = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int a = (t >> 3) & 7;
  const int p = (t >> 6) * 8 + (t & 7);
  if (p >= HW) return;
  const int frame = static_cast<int>(ix[m]);
  const float* dmap = disps + (size_t)frame * HW;
  const int y = p / w, x = p - y * w;
  float nb[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    nb[k] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dmap[yy * w + xx] : 0.0f;
  }
  const _Float16* row = mask + ((size_t)m * HW + p) * ms + a * 8;
  mask8 v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = *reinterpret_cast<const mask8*>(row + k * 64);
  float outv[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    float mx = (float)v[0][b];
#pragma unroll
    for (int k = 1; k < 9; ++k) mx = fmaxf(mx, (float)v[k][b]);
    float e[9], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { e[k] = expf((float)v[k][b] - mx); sum += e[k]; }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k)      // product and sum rounded separately (glorie_conv_upsample does the same: same bits)
      acc = __fadd_rn(acc, __fmul_rn(round_like_mask<_Float16>(e[k] / sum, softmax_f32 != 0), nb[k]));
    outv[b] = acc;
  }
  float* o = disps_up + (size_t)frame * 64 * HW + (size_t)(8 * y + a) * (8 * w) + 8 * x;
  reinterpret_cast<float4*>(o)[0] = make_float4(outv[0], outv[1], outv[2], outv[3]);
  reinterpret_cast<float4*>(o)[1] = make_float4(outv[4], outv[5], outv[6], outv[7]);
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_reproject(const float* poses, const float* disps, const float* intrinsics,
                                const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                                int N, int h, int w, void* stream) {
  if (N < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  if (N == 0 || h * w == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !coords) return GLORIE_EINVAL;
  dim3 grid((h * w + 255) / 256, N);
  hipLaunchKernelGGL(reproject_kernel, grid, dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, ii, jj, coords, valid, h, w);
  return check_launch();
}

extern "C" int glorie_reproject_motion(const float* poses, const float* disps, const float* intrinsics,
                                       const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                                       const float* target, void* padded_motion, int N, int h, int w, float limit,
                                       void* stream) {
  if (N < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  if (N == 0 || h * w == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !coords || !target || !padded_motion) return GLORIE_EINVAL;
  dim3 grid((h * w + 255) / 256, N);
  hipLaunchKernelGGL(reproject_kernel, grid, dim3(256), 0, (hipStream_t)stream, poses, disps, intrinsics, ii, jj, coords,
                     valid, h, w, reinterpret_cast<const float2*>(target), reinterpret_cast<_Float16*>(padded_motion),
                     limit);
  return check_launch();
}

extern "C" int glorie_frame_distance(const float* poses, const float* disps,
                                     const float* intrinsics, const int64_t* ii,
                                     const int64_t* jj, float* dist, int K, int h, int w,
                                     float beta, void* stream) {
  if (K < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  if (K == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !dist) return GLORIE_EINVAL;
  hipLaunchKernelGGL(frame_distance_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, poses,
                     disps, intrinsics, ii, jj, dist, h, w, beta);
  return check_launch();
}

extern "C" int glorie_iproj(const float* poses, const float* disps, const float* intrinsics,
                            float* points, int num, int h, int w, void* stream) {
  if (num < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  if (num == 0 || h * w == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !points) return GLORIE_EINVAL;
  dim3 grid((h * w + 255) / 256, num);
  hipLaunchKernelGGL(iproj_kernel, grid, dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, points, h, w);
  return check_launch();
}

extern "C" int glorie_depth_filter(const float* poses, const float* disps,
                                   const float* intrinsics, const int64_t* ix,
                                   const float* thresh, float* count, int B, int num, int h,
                                   int w, void* stream) {
  if (B < 0 || num < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  if (num == 0 || h * w == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !ix || !thresh || !count) return GLORIE_EINVAL;
  dim3 grid((h * w + 255) / 256, num);
  hipLaunchKernelGGL(depth_filter_kernel, grid, dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, ix, thresh, count, B, h, w);
  return check_launch();
}

extern "C" int glorie_cvx_upsample(const float* disps, const int64_t* ix, const void* mask,
                                   float* disps_up, int M, int h, int w, int mask_dtype,
                                   int softmax_f32, void* stream) {
  if (M < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  if (M == 0 || h * w == 0) return GLORIE_OK;
  if (!disps || !ix || !mask || !disps_up) return GLORIE_EINVAL;
  const int pairs = (h * w + 1) / 2;
  dim3 grid((pairs + 255) / 256, 8, M);
  if (mask_dtype == GLORIE_F16)
    hipLaunchKernelGGL(cvx_upsample_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream,
                       disps, ix, reinterpret_cast<const _Float16*>(mask), disps_up, h, w,
                       softmax_f32);
  else if (mask_dtype == GLORIE_F32)
    hipLaunchKernelGGL(cvx_upsample_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       disps, ix, reinterpret_cast<const float*>(mask), disps_up, h, w, 1);
  else
    return GLORIE_EUNSUPPORTED;
  return check_launch();
}

extern "C" int glorie_cvx_upsample_nhwc(const float* disps, const int64_t* ix, const void* mask,
                                        int mask_stride, float* disps_up, int M, int h, int w,
                                        int softmax_f32, void* stream) {
  if (M < 0 || h < 0 || w < 0 || mask_stride < 576 || (mask_stride & 7)) return GLORIE_EINVAL;
  if (M == 0 || h * w == 0) return GLORIE_OK;
  if (!disps || !ix || !mask || !disps_up) return GLORIE_EINVAL;
  const int groups = (h * w + 7) / 8;
  hipLaunchKernelGGL(cvx_upsample_nhwc_kernel, dim3((groups * 64 + 255) / 256, M), dim3(256), 0,
                     (hipStream_t)stream, disps, ix, reinterpret_cast<const _Float16*>(mask), mask_stride,
                     disps_up, h, w, softmax_f32);
  return check_launch();
}

extern "C" int glorie_motion(const float* coords1, const float* coords0, const float* target, float* out,
                             int N, int h, int w, float limit, void* stream) {
  if (N < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  const long total = (long)N * h * w;
  if (total == 0) return GLORIE_OK;
  if (!coords1 || !coords0 || !target || !out) return GLORIE_EINVAL;
  hipLaunchKernelGGL(motion_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(coords1), reinterpret_cast<const float2*>(coords0),
                     reinterpret_cast<const float2*>(target), reinterpret_cast<float4*>(out), total, h * w,
                     limit);
  return check_launch();
}

extern "C" int glorie_motion_padded(const float* coords1, const float* coords0, const float* target, void* padded,
                                    int N, int h, int w, float limit, void* stream) {
  if (N < 0 || h < 0 || w < 0) return GLORIE_EINVAL;
  const long total = (long)N * h * w;
  if (total == 0) return GLORIE_OK;
  if (!coords1 || !coords0 || !target || !padded) return GLORIE_EINVAL;
  hipLaunchKernelGGL(motion_padded_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(coords1), reinterpret_cast<const float2*>(coords0),
                     reinterpret_cast<const float2*>(target), reinterpret_cast<_Float16*>(padded), total, h, w, limit);
  return check_launch();
}
