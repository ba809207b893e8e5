// Preparation of the DSPO depth_scale stage (scope row B) as 4 launches instead of ~70 PyTorch ones:
//   DepthVideo.update_valid_depth_mask(up=False)   /root/reference/src/depth_video.py:326-361
//   align_scale_and_shift(mono, est, valid)        /root/reference/src/utils/common.py:401-437
//   bad-frame / edge filtering by mono_thres       /root/reference/src/depth_video.py:228-247
//
//   1. prep_stats:  per frame mean depth (-> two-view threshold) and mean disparity
//   2. depth_filter (csrc/geom.hip): consistent-neighbour count per pixel
//   3. prep_align:  one workgroup per frame -- nanmedian of the depths that passed the filter by an
//      exact 4-pass radix select in LDS (torch.nanmedian = lower median), the validity mask
//      depth < 3*median, the 5 weighted sums of the least-squares fit, scale/shift, mean absolute
//      residual and the frame's `bad` flag
//   4. prep_edges:  edge_on[e] = !(bad[ii] | bad[jj]) and the "any edge on" flag
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "common.hiph"

extern "C" int glorie_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                                   const int64_t* ix, const float* thresh, float* count, int B, int num,
                                   int h, int w, void* stream);

namespace glorie {

__device__ __forceinline__ double block_sum(double v, double* red) {
  // 1024 threads = 16 waves; fixed order -> deterministic
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(1024) void prep_stats_kernel(const float* __restrict__ disps, int HW,
                                                          float mv_thresh, float* __restrict__ thresh,
                                                          float* __restrict__ avg_disp,
                                                          int64_t* __restrict__ ix, int* __restrict__ any_on) {
  __shared__ double red[16];
  const int f = blockIdx.x;
  if (f == 0 && threadIdx.x == 0) *any_on = 0;      // prep_edges_kernel ORs into it three launches later (no memset node)
  const float* d = disps + (size_t)f * HW;
  double sd = 0.0, sz = 0.0;
  for (int k = threadIdx.x; k < HW; k += blockDim.x) {
    const float v = d[k];
    sd += (double)v;
    sz += (double)(1.0f / v);
  }
  const double td = block_sum(sd, red);
  const double tz = block_sum(sz, red);
  if (threadIdx.x == 0) {
    thresh[f] = mv_thresh * (float)(tz / (double)HW);
    avg_disp[f] = (float)(td / (double)HW);
    ix[f] = f;
  }
}

// dynamic LDS: HW keys (uint32)
__global__ __launch_bounds__(1024) void prep_align_kernel(
    const float* __restrict__ disps, const float* __restrict__ mono, const float* __restrict__ count,
    const float* __restrict__ avg_disp, int HW, float visible, float mono_thres,
    uint8_t* __restrict__ valid_mask, float* __restrict__ scales, float* __restrict__ shifts,
    uint8_t* __restrict__ bad) {
  extern __shared__ uint32_t keys[];
  __shared__ double red[16];
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_k, wsum[4];
  const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* d = disps + (size_t)f * HW;
  const float* m = mono + (size_t)f * HW;
  const float* c = count + (size_t)f * HW;

  // ---- keys of the depths that passed the two-view filter (positive floats order like uints) ----
  double nv = 0.0;
  for (int k = tid; k < HW; k += nt) {
    const bool ok = c[k] >= visible;
    const float z = 1.0f / d[k];
    keys[k] = (ok && !isnan(z)) ? __float_as_uint(z) : 0xffffffffu;
    nv += (ok && !isnan(z)) ? 1.0 : 0.0;
  }
  const int nvalid = (int)block_sum(nv, red);
  float med = nanf("");
  if (nvalid > 0) {
    if (tid == 0) { sel_prefix = 0u; sel_k = (unsigned)((nvalid - 1) / 2); }   // lower median
    unsigned mask = 0u;
    for (int pass = 3; pass >= 0; --pass) {
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned prefix = sel_prefix;
      for (int k = tid; k < HW; k += nt) {
        const unsigned key = keys[k];
        if (key != 0xffffffffu && (key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
      }
      __syncthreads();
      {
        // which bin holds the k-th key: inclusive scan of the 256 bins over the first 4 waves
        const unsigned kk = sel_k;
        const unsigned hv = tid < 256 ? hist[tid] : 0u;
        unsigned incl = hv;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned v = __shfl_up(incl, off, 64);
          if ((tid & 63) >= off) incl += v;
        }
        if (tid < 256 && (tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        if (tid < 256) {
          for (int q = 0; q < (tid >> 6); ++q) incl += wsum[q];
          const unsigned excl = incl - hv;
          if (kk >= excl && kk < incl) {          // exactly one bin (the k-th key exists: nvalid > k)
            sel_k = kk - excl;
            sel_prefix = prefix | ((unsigned)tid << (8 * pass));
          }
        }
      }
      mask |= 0xffu << (8 * pass);
      __syncthreads();
    }
    med = __uint_as_float(sel_prefix);
  }
  // ---- mask + least squares  target(est disparity) ~ scale * prediction(mono) + shift ----
  const float lim = 3.0f * med;
  double a00 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0;
  for (int k = tid; k < HW; k += nt) {
    const unsigned key = keys[k];
    const bool on = key != 0xffffffffu && __uint_as_float(key) < lim;
    valid_mask[(size_t)f * HW + k] = on ? 1 : 0;
    if (on) {
      const float p = m[k], t = d[k];
      a00 += (double)(p * p);
      a01 += (double)p;
      a11 += 1.0;
      b0 += (double)(p * t);
      b1 += (double)t;
    }
  }
  const float A00 = (float)block_sum(a00, red), A01 = (float)block_sum(a01, red);
  const float A11 = (float)block_sum(a11, red), B0 = (float)block_sum(b0, red), B1 = (float)block_sum(b1, red);
  const float det = A00 * A11 - A01 * A01;
  const float scale = (A11 * B0 - A01 * B1) / det;
  const float shift = (-A01 * B0 + A00 * B1) / det;
  double es = 0.0;
  for (int k = tid; k < HW; k += nt) {
    const unsigned key = keys[k];
    if (key != 0xffffffffu && __uint_as_float(key) < lim) es += (double)fabsf(scale * m[k] + shift - d[k]);
  }
  const float err = (float)block_sum(es, red) / A11;
  if (tid == 0) {
    scales[f] = scale;
    shifts[f] = shift;
    bool b = false;
    if (mono_thres > 0.0f)
      b = (err / avg_disp[f] > mono_thres) || isnan(err) || (scale < 0.0f) || (A11 < 0.5f * (float)HW);
    bad[f] = b ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void prep_edges_kernel(const uint8_t* __restrict__ bad,
                                                         const int64_t* __restrict__ ii,
                                                         const int64_t* __restrict__ jj, int N, int n,
                                                         uint8_t* __restrict__ edge_on, int* __restrict__ any_on,
                                                         int* __restrict__ pub /* null | [launch count, arrivals] */,
                                                         int* host_word) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < N) {
    const int i = (int)ii[e], j = (int)jj[e];
    const bool bi = (i >= 0 && i < n) ? bad[i] != 0 : false;      // frames beyond the counter are never "bad"
    const bool bj = (j >= 0 && j < n) ? bad[j] != 0 : false;
    const bool on = !(bi || bj);
    edge_on[e] = on ? 1 : 0;
    if (on) atomicOr(any_on, 1);
  }
  if (!pub) return;
  // the last workgroup publishes (launch count << 1 | any_on) to pinned host memory (see publish_flag_kernel below)
  __shared__ int last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&pub[1], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    pub[1] = 0;
    const int c = pub[0] + 1;
    pub[0] = c;
    const int flag = __hip_atomic_load(any_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host_word, (c << 1) | (flag != 0 ? 1 : 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The one host decision of a depth_scale stage recorded into a hipGraph (`any edge enabled`, depth_video.py:290-294)
// without a stream synchronisation: the word (launch count << 1 | flag) is stored straight into pinned host memory; the
// host knows how many launches it has enqueued and polls the word until the count matches - 300 us before the replay
// ends, so the next replay is enqueued while this one still runs (a stream sync here left the GPU idle for 40 us per step).
__global__ void publish_flag_kernel(const int* __restrict__ flag, int* __restrict__ counter, int* host_word) {
  const int c = *counter + 1;
  *counter = c;
  __hip_atomic_store(host_word, (c << 1) | (*flag != 0 ? 1 : 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// Two-view validity mask for an arbitrary frame list and map size (scope row N4:
// DepthVideo.update_valid_depth_mask(up=True) at full resolution, depth_video.py:326-361): the
// nanmedian of 307,200 depths per frame does not fit LDS, so the radix select runs over global keys:
// per pass one histogram kernel (LDS bins -> global atomics) and one 1-thread-per-frame bin pick.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vmask_stats_kernel(const float* __restrict__ disps,
                                                           const int64_t* __restrict__ ix, int HW,
                                                           float mv_thresh, float* __restrict__ thresh) {
  __shared__ double red[16];
  const float* d = disps + (size_t)ix[blockIdx.x] * HW;
  double sz = 0.0;
  for (int k = threadIdx.x; k < HW; k += blockDim.x) sz += (double)(1.0f / d[k]);
  const double tz = block_sum(sz, red);
  if (threadIdx.x == 0) thresh[blockIdx.x] = mv_thresh * (float)(tz / (double)HW);
}

// count (float, in) -> key (uint32, in place): depth bits where the filter passed, else 0xffffffff
__global__ __launch_bounds__(256) void vmask_keys_kernel(const float* __restrict__ disps,
                                                         const int64_t* __restrict__ ix, int HW,
                                                         float visible, uint32_t* __restrict__ keys,
                                                         unsigned* __restrict__ sel /*[num][4]: nvalid,k,prefix,-*/) {
  const int f = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  unsigned ok = 0;
  if (p < HW) {
    const float cnt = __uint_as_float(keys[(size_t)f * HW + p]);
    const float z = 1.0f / disps[(size_t)ix[f] * HW + p];
    ok = (cnt >= visible && !isnan(z)) ? 1u : 0u;
    keys[(size_t)f * HW + p] = ok ? __float_as_uint(z) : 0xffffffffu;
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&sel[f * 4 + 0], (unsigned)__popcll(b));
}

__global__ __launch_bounds__(256) void vmask_hist_kernel(const uint32_t* __restrict__ keys, int HW, int pass,
                                                         const unsigned* __restrict__ sel,
                                                         unsigned* __restrict__ hist /*[num][256]*/) {
  __shared__ unsigned h[256];
  const int f = blockIdx.y;
  h[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned mask = pass == 3 ? 0u : (0xffffffffu << (8 * (pass + 1)));
  const unsigned prefix = sel[f * 4 + 2];
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const unsigned key = keys[(size_t)f * HW + p];
    if (key != 0xffffffffu && (key & mask) == prefix) atomicAdd(&h[(key >> (8 * pass)) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[f * 256 + threadIdx.x], h[threadIdx.x]);
}

// one thread per frame: pick the bin holding the k-th key, narrow the prefix, clear the histogram
__global__ void vmask_pick_kernel(unsigned* __restrict__ sel, unsigned* __restrict__ hist, int num, int pass) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= num) return;
  if (pass == 3) sel[f * 4 + 1] = sel[f * 4 + 0] ? (sel[f * 4 + 0] - 1) / 2 : 0;   // lower median
  unsigned kk = sel[f * 4 + 1], b = 0;
  for (; b < 256; ++b) {
    const unsigned c = hist[f * 256 + b];
    hist[f * 256 + b] = 0u;
    if (kk < c) break;
    kk -= c;
  }
  for (unsigned r = b + 1; r < 256; ++r) hist[f * 256 + r] = 0u;
  sel[f * 4 + 1] = kk;
  sel[f * 4 + 2] |= (b & 255u) << (8 * pass);
}

__global__ __launch_bounds__(256) void vmask_write_kernel(const uint32_t* __restrict__ keys, int HW,
                                                          const unsigned* __restrict__ sel,
                                                          uint8_t* __restrict__ mask) {
  const int f = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float med = sel[f * 4 + 0] ? __uint_as_float(sel[f * 4 + 2]) : nanf("");
  const unsigned key = keys[(size_t)f * HW + p];
  mask[(size_t)f * HW + p] = (key != 0xffffffffu && __uint_as_float(key) < 3.0f * med) ? 1 : 0;
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_dspo_prepare(const float* poses, const float* disps, const float* intrinsics,
                                   const float* mono_disps, int B, int n, int h, int w, float mv_thresh,
                                   int visible_num, float mono_thres, const int64_t* ii, const int64_t* jj,
                                   int N, uint8_t* valid_mask, float* scales, float* shifts,
                                   uint8_t* edge_on, int* any_on, void* scratch, int* publish_state,
                                   int* publish_host_word, void* stream) {
  if (B < 0 || n < 0 || n > B || h <= 0 || w <= 0 || N < 0) return GLORIE_EINVAL;
  const int HW = h * w;
  if ((size_t)HW * 4 > 150 * 1024) return GLORIE_EUNSUPPORTED;
  if (!any_on) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool publish = publish_state && publish_host_word;
  if (n == 0) {
    GLORIE_TRY(check_hip(hipMemsetAsync(any_on, 0, sizeof(int), st)));
    if (publish) hipLaunchKernelGGL(publish_flag_kernel, dim3(1), dim3(1), 0, st, any_on, publish_state, publish_host_word);
    return check_launch();
  }
  if (!poses || !disps || !intrinsics || !mono_disps || !valid_mask || !scales || !shifts || !scratch ||
      (N > 0 && (!ii || !jj || !edge_on)))
    return GLORIE_EINVAL;
  // scratch: count [n*HW] f32 | thresh [n] | avg [n] | ix [n] i64 | bad [n] u8
  char* sp = reinterpret_cast<char*>(scratch);
  float* count = reinterpret_cast<float*>(sp);
  float* thresh = count + (size_t)n * HW;
  float* avg = thresh + n;
  int64_t* ix = reinterpret_cast<int64_t*>(sp + ((((size_t)n * HW + 2 * n) * 4 + 7) / 8) * 8);
  uint8_t* bad = reinterpret_cast<uint8_t*>(ix + n);
  hipLaunchKernelGGL(prep_stats_kernel, dim3(n), dim3(1024), 0, st, disps, HW, mv_thresh, thresh, avg, ix, any_on);
  GLORIE_TRY(glorie_depth_filter(poses, disps, intrinsics, ix, thresh, count, B, n, h, w, stream));
  static PerDeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prep_align_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  hipLaunchKernelGGL(prep_align_kernel, dim3(n), dim3(1024), (size_t)HW * 4, st, disps, mono_disps, count, avg,
                     HW, (float)visible_num, mono_thres, valid_mask, scales, shifts, bad);
  if (N > 0)
    hipLaunchKernelGGL(prep_edges_kernel, dim3((N + 255) / 256), dim3(256), 0, st, bad, ii, jj, N, n, edge_on,
                       any_on, publish ? publish_state : nullptr, publish_host_word);
  else if (publish)
    hipLaunchKernelGGL(publish_flag_kernel, dim3(1), dim3(1), 0, st, any_on, publish_state, publish_host_word);
  return check_launch();
}

extern "C" int glorie_publish_flag(const int* flag, int* counter, int* host_word, void* stream) {
  if (!flag || !counter || !host_word) return GLORIE_EINVAL;
  hipLaunchKernelGGL(publish_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, counter, host_word);
  return check_launch();
}

extern "C" int glorie_valid_depth_mask(const float* poses, const float* disps, const float* intrinsics,
                                       const int64_t* ix, int B, int num, int h, int w, float mv_thresh,
                                       int visible_num, uint8_t* mask, void* scratch, void* stream) {
  if (B < 0 || num < 0 || h <= 0 || w <= 0) return GLORIE_EINVAL;
  if (num == 0) return GLORIE_OK;
  if (!poses || !disps || !intrinsics || !ix || !mask || !scratch) return GLORIE_EINVAL;
  const int HW = h * w;
  hipStream_t st = (hipStream_t)stream;
  // scratch: count/keys [num*HW] 4 B | thresh [num] | sel [num][4] | hist [num][256]
  float* count = reinterpret_cast<float*>(scratch);
  float* thresh = count + (size_t)num * HW;
  unsigned* sel = reinterpret_cast<unsigned*>(thresh + num);
  unsigned* hist = sel + (size_t)num * 4;
  GLORIE_TRY(check_hip(hipMemsetAsync(sel, 0, sizeof(unsigned) * ((size_t)num * 4 + (size_t)num * 256), st)));
  hipLaunchKernelGGL(vmask_stats_kernel, dim3(num), dim3(1024), 0, st, disps, ix, HW, mv_thresh, thresh);
  GLORIE_TRY(glorie_depth_filter(poses, disps, intrinsics, ix, thresh, count, B, num, h, w, stream));
  const dim3 pix((HW + 255) / 256, num);
  hipLaunchKernelGGL(vmask_keys_kernel, pix, dim3(256), 0, st, disps, ix, HW, (float)visible_num,
                     reinterpret_cast<uint32_t*>(count), sel);
  const int hb = (HW + 256 * 8 - 1) / (256 * 8);
  for (int pass = 3; pass >= 0; --pass) {
    hipLaunchKernelGGL(vmask_hist_kernel, dim3(hb, num), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(count),
                       HW, pass, sel, hist);
    hipLaunchKernelGGL(vmask_pick_kernel, dim3((num + 63) / 64), dim3(64), 0, st, sel, hist, num, pass);
  }
  hipLaunchKernelGGL(vmask_write_kernel, pix, dim3(256), 0, st, reinterpret_cast<const uint32_t*>(count), HW, sel,
                     mask);
  return check_launch();
}
