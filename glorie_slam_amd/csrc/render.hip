// Neural-point renderer kernels (scope rows R2, R5) for gfx950: inverse-distance feature
// interpolation over the k nearest neural points and per-ray alpha compositing.
//
// Reference behaviour restated from
//   /root/reference/src/modules/conv_onet/models/decoder.py:130-173, 340-389  (get_feature_at_pos)
//   /root/reference/src/utils/common.py:261-299                               (raw2outputs_nerf_color)
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

// ------------------------------------------------------------------------------------
// IDW gather:  c[q] = sum_k w_k feats[I_k],  w = normalize_L1( [D<=r^2] / (D + 1e-10) )
// One wave handles 2 samples: 32 lanes = the 32 feature channels of a 128-byte row, so
// every neighbour row is one fully coalesced 128 B read (the gather unit SURVEY.md 8(d)
// counts).  Lanes 0..7 of each half also carry the 8 (D, I) pairs.
// ------------------------------------------------------------------------------------
template <int K, int C>
__global__ __launch_bounds__(256) void idw_gather_kernel(
    const float* __restrict__ D, const int64_t* __restrict__ I, const int* __restrict__ nn,
    const float* __restrict__ feats, int Q, float radius, const float* __restrict__ radius_ptr,
    int min_nn, int expo_weighting, float* __restrict__ cout, float* __restrict__ wout,
    uint8_t* __restrict__ has_out) {
  static_assert(C == 32 && K <= 32, "layout assumes 32-channel features");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, ch = lane & 31;
  const int q = (blockIdx.x * 256 + threadIdx.x) / 32;
  const bool live = q < Q;
  const int qc = live ? q : Q - 1;
  // lanes ch < K load one neighbour each
  float d = FLT_MAX;
  int idx = -1;
  if (ch < K) {
    d = D[(size_t)qc * K + ch];
    idx = (int)I[(size_t)qc * K + ch];
  }
  const float r = radius_ptr ? radius_ptr[qc] : radius;
  const float r2 = r * r;
  float wgt = 0.0f;
  if (ch < K && idx >= 0 && !(d > r2))
    wgt = expo_weighting ? expf(-20.0f * sqrtf(d)) : 1.0f / (d + 1e-10f);
  // L1 normalisation over the K lanes of this half (F.normalize(p=1, eps=1e-12))
  float s = wgt;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);  // stays inside the half
  wgt = wgt / fmaxf(s, 1e-12f);
  float acc = 0.0f;
  if (cout) {                     // cout == NULL: weights / mask only (the geometry decoder gathers itself)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float wk = __shfl(wgt, half * 32 + k, 64);
      const int ik = __shfl(idx, half * 32 + k, 64);
      if (wk != 0.0f) acc += wk * feats[(size_t)ik * C + ch];
    }
  }
  if (!live) return;
  const bool has = nn[q] > min_nn - 1;
  // samples without enough neighbours get a N(0, 0.01) random feature in the reference
  // (decoder.py:170-171); here they get zeros (their occupancy is forced to -100 anyway)
  if (cout) cout[(size_t)q * C + ch] = has ? acc : 0.0f;
  if (wout && ch < K) wout[(size_t)q * K + ch] = wgt;
  if (has_out && ch == 0) has_out[q] = has ? 1 : 0;
}

// The same interpolation with 16-byte lanes and up to two feature tables per pass: 8 lanes per sample - lane j carries
// neighbour j's (D, I) and the j-th float4 of the 128-byte rows - so a wave gathers 8 rows (1 KB) per load instead of 2,
// and the geometry and the colour table share one read of (D, I, nn) and one set of weights.  Arithmetic and summation
// order are those of idw_gather_kernel (butterfly xor 4, 2, 1 for the norm; neighbours accumulated in order 0..7).
__global__ __launch_bounds__(256) void idw_gather2_kernel(
    const float* __restrict__ D, const int64_t* __restrict__ I, const int* __restrict__ nn,
    const float* __restrict__ feats_a, const float* __restrict__ feats_b, int Q, float radius,
    const float* __restrict__ radius_ptr, int min_nn, int expo_weighting, float* __restrict__ cout_a,
    float* __restrict__ cout_b, float* __restrict__ wout, uint8_t* __restrict__ has_out) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 7, grp = lane & ~7;
  const int q = (blockIdx.x * 256 + threadIdx.x) >> 3;
  const bool live = q < Q;
  const int qc = live ? q : Q - 1;
  const float d = D[(size_t)qc * 8 + j];
  const int idx = (int)I[(size_t)qc * 8 + j];
  const float r = radius_ptr ? radius_ptr[qc] : radius;
  const float r2 = r * r;
  float wgt = 0.0f;
  if (idx >= 0 && !(d > r2)) wgt = expo_weighting ? expf(-20.0f * sqrtf(d)) : 1.0f / (d + 1e-10f);
  float s = wgt;
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  wgt = wgt / fmaxf(s, 1e-12f);
  // all 8 (16) row loads are issued before the first use: a load under `if (w_k != 0)` would put a wait between
  // every pair of neighbours.  Rows of zero-weight neighbours are read (index clamped) and dropped by a select, so
  // the sum is the one of the conditional form bit for bit.
  float wk[8];
  float4 fa[8], fb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    wk[k] = __shfl(wgt, grp + k, 64);
    const int ik = max(__shfl(idx, grp + k, 64), 0);
    fa[k] = reinterpret_cast<const float4*>(feats_a)[(size_t)ik * 8 + j];
    if (feats_b) fb[k] = reinterpret_cast<const float4*>(feats_b)[(size_t)ik * 8 + j];
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool on = wk[k] != 0.0f;
    a.x = on ? a.x + wk[k] * fa[k].x : a.x; a.y = on ? a.y + wk[k] * fa[k].y : a.y;
    a.z = on ? a.z + wk[k] * fa[k].z : a.z; a.w = on ? a.w + wk[k] * fa[k].w : a.w;
    if (feats_b) {
      b.x = on ? b.x + wk[k] * fb[k].x : b.x; b.y = on ? b.y + wk[k] * fb[k].y : b.y;
      b.z = on ? b.z + wk[k] * fb[k].z : b.z; b.w = on ? b.w + wk[k] * fb[k].w : b.w;
    }
  }
  if (!live) return;
  const bool has = nn[q] > min_nn - 1;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  reinterpret_cast<float4*>(cout_a)[(size_t)q * 8 + j] = has ? a : zero;
  if (feats_b) reinterpret_cast<float4*>(cout_b)[(size_t)q * 8 + j] = has ? b : zero;
  if (wout) wout[(size_t)q * 8 + j] = wgt;
  if (has_out && j == 0) has_out[q] = has ? 1 : 0;
}

// the weights and the mask of idw_gather_kernel without the feature interpolation: one thread per sample
// (same arithmetic: w = [idx >= 0 and D <= r^2] / (D + 1e-10), L1-normalised with eps 1e-12)
__global__ __launch_bounds__(256) void idw_weights_kernel(
    const float* __restrict__ D, const int64_t* __restrict__ I, const int* __restrict__ nn, int Q, float radius,
    const float* __restrict__ radius_ptr, int min_nn, int expo_weighting, float* __restrict__ wout,
    uint8_t* __restrict__ has_out) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  const float r = radius_ptr ? radius_ptr[q] : radius;
  const float r2 = r * r;
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float d = D[(size_t)q * 8 + k];
    const long idx = I[(size_t)q * 8 + k];
    w[k] = (idx >= 0 && !(d > r2)) ? (expo_weighting ? expf(-20.0f * sqrtf(d)) : 1.0f / (d + 1e-10f)) : 0.0f;
  }
  // the L1 norm in the summation order of idw_gather_kernel's butterfly (xor 4, 2, 1 over the 8 weights), so
  // that both kernels produce the same bits
  const float s4[4] = {w[0] + w[4], w[1] + w[5], w[2] + w[6], w[3] + w[7]};
  const float s2[2] = {s4[0] + s4[2], s4[1] + s4[3]};
  const float s = s2[0] + s2[1];
  const float den = fmaxf(s, 1e-12f);
#pragma unroll
  for (int k = 0; k < 8; ++k) wout[(size_t)q * 8 + k] = w[k] / den;
  if (has_out) has_out[q] = nn[q] > min_nn - 1 ? 1 : 0;
}

// ------------------------------------------------------------------------------------
// compositing: one lane per ray, S samples scanned in registers
//   alpha = sigmoid(coef * occ), w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-10)
//   rgb = sum w rgb / (sum w + 1e-10), depth = sum w z / (sum w + 1e-10), var = sum w (z - depth)^2
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, int R, int S, float coef,
    float* __restrict__ depth, float* __restrict__ var, float* __restrict__ rgb,
    float* __restrict__ weights) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float T = 1.0f, wsum = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, dz = 0.0f;
  for (int s = 0; s < S; ++s) {
    const float4 v = reinterpret_cast<const float4*>(raw)[(size_t)r * S + s];
    const float alpha = 1.0f / (1.0f + expf(-coef * v.w));
    const float wgt = alpha * T;
    T *= (1.0f - alpha + 1e-10f);
    if (weights) weights[(size_t)r * S + s] = wgt;
    wsum += wgt;
    cr += wgt * v.x; cg += wgt * v.y; cb += wgt * v.z;
    dz += wgt * z_vals[(size_t)r * S + s];
  }
  const float den = wsum + 1e-10f;
  const float dm = dz / den;
  rgb[(size_t)r * 3 + 0] = cr / den;
  rgb[(size_t)r * 3 + 1] = cg / den;
  rgb[(size_t)r * 3 + 2] = cb / den;
  depth[r] = dm;
  float vv = 0.0f;
  T = 1.0f;
  for (int s = 0; s < S; ++s) {
    const float occ = raw[((size_t)r * S + s) * 4 + 3];
    const float alpha = 1.0f / (1.0f + expf(-coef * occ));
    const float wgt = alpha * T;
    T *= (1.0f - alpha + 1e-10f);
    const float t = z_vals[(size_t)r * S + s] - dm;
    vv += wgt * t * t;
  }
  var[r] = vv;
}

// sample placement of Renderer.render_batch_ray for rays with a depth prior (Renderer.py:106-125, 177-179):
//   z = near_s * d * (1 - t) + far_s * d * t  (t = linspace(0, 1, S), every product and the sum rounded
//   separately like the chain of torch ops), pts = o + dir * z, and the per-ray view direction / query
//   radius repeated per sample.  Rays with d <= 0 get z = 0 and are counted in *n_zero: the host sends such a
//   batch through the general path (they need the 25-probe search of sample_near_pcl).
// CAMERA: the rays are those of consecutive row-major pixels of a pinhole view (get_rays, common.py:302-322: OpenGL convention,
// dirs = ((i - cx) / fx, -(j - cy) / fy, -1), rays_d = sum(dirs * c2w[:3,:3], -1), rays_o = c2w[:3,3]) and are formed here
// instead of being read - scope row R7 fused into R4; `cam` = c2w rows 0..2 (12 floats), 1/fx, 1/fy, cx, cy.
template <bool CAMERA>
__global__ __launch_bounds__(256) void ray_samples_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ cam, int img_w,
    long first_pixel, const float* __restrict__ depth,
    const float* __restrict__ radius, const float* __restrict__ t_lin, int R, int S, float near_s, float far_s,
    float* __restrict__ z_vals, float* __restrict__ pts, float* __restrict__ views,
    float* __restrict__ radius_s, int* __restrict__ n_zero) {
#pragma clang fp contract(off)
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)R * S) return;
  const int r = (int)(idx / S), sidx = (int)(idx - (long)r * S);
  const float d = depth[r], t = t_lin[sidx];
  const float a = near_s * d, c = far_s * d;
  const float omt = 1.0f - t;
  const float p1 = a * omt, p2 = c * t;
  const float z = d > 0.0f ? p1 + p2 : 0.0f;
  if (sidx == 0 && !(d > 0.0f)) atomicAdd(n_zero, 1);
  z_vals[idx] = z;
  float o[3], dir[3];
  if (CAMERA) {
    const long px = first_pixel + r;
    const float fi = (float)(px % img_w), fj = (float)(px / img_w);
    // torch divides by a Python scalar as a multiplication with its fp32 reciprocal, and its 3-element row sum adds
    // element 2 before element 1 (vectorised partial accumulators): reproduced, so the rays are the bits of get_rays
    const float x = (fi - cam[14]) * cam[12];
    const float y = -((fj - cam[15]) * cam[13]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float q0 = x * cam[k * 4 + 0], q1 = y * cam[k * 4 + 1], q2 = -1.0f * cam[k * 4 + 2];
      float sum = q0 + q2;
      sum = sum + q1;
      dir[k] = sum;
      o[k] = cam[k * 4 + 3];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) { dir[k] = rays_d[(size_t)r * 3 + k]; o[k] = rays_o[(size_t)r * 3 + k]; }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float m = dir[k] * z;
    pts[idx * 3 + k] = o[k] + m;
    views[idx * 3 + k] = dir[k];
  }
  if (radius_s) radius_s[idx] = radius[r];
}

// z-buffer projection of a point set into a pinhole view (neural_point.py:446-506, proj_depth_map): camera
// coordinates X_c = w2c X (OpenGL convention: the camera looks along -z; the x axis is flipped before the
// projection), pixel (u, v) = trunc(K X_c / z), depth = -z; the closest point per pixel wins.  The reference
// sorts all points by depth and takes the first of every unique pixel; here every point does one atomicMin on
// the bit pattern of its (positive) depth.  depth_bits must be pre-filled with +inf (0x7f800000).
__global__ __launch_bounds__(256) void proj_depth_kernel(const float* __restrict__ pts,
                                                         const uint8_t* __restrict__ mask, long n,
                                                         const float* __restrict__ w2c, float fx, float fy,
                                                         float cx, float cy, int H, int W,
                                                         unsigned* __restrict__ depth_bits) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || (mask && !mask[i])) return;
  const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
  float xc = w2c[0] * x + w2c[1] * y + w2c[2] * z + w2c[3];
  const float yc = w2c[4] * x + w2c[5] * y + w2c[6] * z + w2c[7];
  const float zc = w2c[8] * x + w2c[9] * y + w2c[10] * z + w2c[11];
  xc = -xc;
  const float zz = zc + 1e-6f;
  const float u = (fx * xc + cx * zc) / zz, v = (fy * yc + cy * zc) / zz;
  if (!(u < (float)W && u >= 0.0f && v < (float)H && v >= 0.0f && -zz > 0.0f)) return;
  const int ui = (int)u, vi = (int)v;
  atomicMin(&depth_bits[(size_t)vi * W + ui], __float_as_uint(-zz));
}

// per-ray number of samples that have neighbours and the valid-ray flag (decoder.py:202-204)
__global__ __launch_bounds__(256) void ray_counts_kernel(const uint8_t* __restrict__ has, int R, int S,
                                                         int min_samples, int64_t* __restrict__ counts,
                                                         uint8_t* __restrict__ valid) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  int c = 0;
  for (int s = 0; s < S; ++s) c += has[(size_t)r * S + s] ? 1 : 0;
  counts[r] = c;
  valid[r] = c >= min_samples ? 1 : 0;
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_ray_samples(const float* rays_o, const float* rays_d, const float* depth,
                                  const float* radius, const float* t_lin, int R, int S, float near_s,
                                  float far_s, float* z_vals, float* pts, float* views, float* radius_s,
                                  int* n_zero, void* stream) {
  if (R < 0 || S < 1) return GLORIE_EINVAL;
  if (R == 0) return GLORIE_OK;
  if (!rays_o || !rays_d || !depth || !t_lin || !z_vals || !pts || !views || !n_zero) return GLORIE_EINVAL;
  if ((radius == nullptr) != (radius_s == nullptr)) return GLORIE_EINVAL;
  const long total = (long)R * S;
  hipLaunchKernelGGL(ray_samples_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     rays_o, rays_d, (const float*)nullptr, 1, 0L, depth, radius, t_lin, R, S, near_s, far_s, z_vals, pts,
                     views, radius_s, n_zero);
  return check_launch();
}

extern "C" int glorie_ray_samples_camera(const float* cam, int image_w, long first_pixel, const float* depth,
                                         const float* radius, const float* t_lin, int R, int S, float near_s,
                                         float far_s, float* z_vals, float* pts, float* views, float* radius_s,
                                         int* n_zero, void* stream) {
  if (R < 0 || S < 1 || image_w < 1 || first_pixel < 0) return GLORIE_EINVAL;
  if (R == 0) return GLORIE_OK;
  if (!cam || !depth || !t_lin || !z_vals || !pts || !views || !n_zero) return GLORIE_EINVAL;
  if ((radius == nullptr) != (radius_s == nullptr)) return GLORIE_EINVAL;
  const long total = (long)R * S;
  hipLaunchKernelGGL(ray_samples_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)nullptr, (const float*)nullptr, cam, image_w, first_pixel, depth, radius, t_lin, R, S,
                     near_s, far_s, z_vals, pts, views, radius_s, n_zero);
  return check_launch();
}

extern "C" int glorie_proj_depth(const float* points, const uint8_t* mask, long n, const float* w2c, float fx,
                                 float fy, float cx, float cy, int H, int W, float* depth_inf, void* stream) {
  if (n < 0 || H <= 0 || W <= 0) return GLORIE_EINVAL;
  if (n == 0) return GLORIE_OK;
  if (!points || !w2c || !depth_inf) return GLORIE_EINVAL;
  hipLaunchKernelGGL(proj_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points,
                     mask, n, w2c, fx, fy, cx, cy, H, W, reinterpret_cast<unsigned*>(depth_inf));
  return check_launch();
}

extern "C" int glorie_ray_counts(const uint8_t* has, int R, int S, int min_samples, int64_t* counts,
                                 uint8_t* valid, void* stream) {
  if (R < 0 || S < 1) return GLORIE_EINVAL;
  if (R == 0) return GLORIE_OK;
  if (!has || !counts || !valid) return GLORIE_EINVAL;
  hipLaunchKernelGGL(ray_counts_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, has, R, S,
                     min_samples, counts, valid);
  return check_launch();
}

extern "C" int glorie_idw_gather(const float* D, const int64_t* I, const int* nn,
                                 const float* feats, int Q, int k, int c_dim, float radius,
                                 const float* radius_ptr, int min_nn, int expo_weighting,
                                 float* c_out, float* w_out, uint8_t* has_out, void* stream) {
  if (Q < 0) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!D || !I || !nn || (c_out && !feats) || (!c_out && !w_out)) return GLORIE_EINVAL;
  if (k != 8 || c_dim != 32) return GLORIE_EUNSUPPORTED;
  if (!c_out) {                 // weights + mask only
    hipLaunchKernelGGL(idw_weights_kernel, dim3((Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, D, I, nn, Q,
                       radius, radius_ptr, min_nn, expo_weighting, w_out, has_out);
    return check_launch();
  }
  const long threads = (long)Q * 8;
  dim3 grid((unsigned)((threads + 255) / 256));
  hipLaunchKernelGGL(idw_gather2_kernel, grid, dim3(256), 0, (hipStream_t)stream, D, I, nn, feats, (const float*)nullptr,
                     Q, radius, radius_ptr, min_nn, expo_weighting, c_out, (float*)nullptr, w_out, has_out);
  return check_launch();
}

extern "C" int glorie_idw_gather2(const float* D, const int64_t* I, const int* nn, const float* feats_a,
                                  const float* feats_b, int Q, int k, int c_dim, float radius,
                                  const float* radius_ptr, int min_nn, int expo_weighting, float* c_out_a,
                                  float* c_out_b, float* w_out, uint8_t* has_out, void* stream) {
  if (Q < 0) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!D || !I || !nn || !feats_a || !feats_b || !c_out_a || !c_out_b) return GLORIE_EINVAL;
  if (k != 8 || c_dim != 32) return GLORIE_EUNSUPPORTED;
  const long threads = (long)Q * 8;
  dim3 grid((unsigned)((threads + 255) / 256));
  hipLaunchKernelGGL(idw_gather2_kernel, grid, dim3(256), 0, (hipStream_t)stream, D, I, nn, feats_a, feats_b, Q, radius,
                     radius_ptr, min_nn, expo_weighting, c_out_a, c_out_b, w_out, has_out);
  return check_launch();
}

extern "C" int glorie_composite(const float* raw, const float* z_vals, int R, int S, float coef,
                                float* depth, float* var, float* rgb, float* weights,
                                void* stream) {
  if (R < 0 || S < 0) return GLORIE_EINVAL;
  if (R == 0) return GLORIE_OK;
  if (!raw || !z_vals || !depth || !var || !rgb) return GLORIE_EINVAL;
  hipLaunchKernelGGL(composite_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw,
                     z_vals, R, S, coef, depth, var, rgb, weights);
  return check_launch();
}
