// Exact k-nearest-neighbour search over the neural point cloud (scope row R1) for gfx950.
//
// Replaces the faiss-gpu index used by NeuralPointCloud.find_neighbors_faiss
//   (/root/reference/src/neural_point.py:56-60,104-116,264-313):
//   IndexIVFFlat(IndexFlatL2(3), 3, nlist=400), nprobe=4 -- an APPROXIMATE search whose result
//   depends on an internal k-means.  This implementation returns the EXACT squared-L2 top-k
//   ordered by (distance, index), so results are reproducible and can be pinned bit-exactly
//   against a brute-force oracle.  Output conventions follow faiss: D = squared distances
//   ascending, I = int64 indices, missing results are I = -1 / D = FLT_MAX.
//
// Structure: uniform grid ("cell list").  Build = bounding box -> cell id per point ->
// counting sort (histogram, exclusive scan, scatter); everything on the device, no host
// synchronisation, rebuilt in well under a millisecond for 0.5 M points (the reference
// re-trains the IVF k-means on every insertion, neural_point.py:257,443).
// Query = one lane per query, expanding Chebyshev shells of cells around the query cell until
// the k-th best distance is provably inside the scanned cube.  The 256 queries of a workgroup
// (consecutive samples of ~25 neighbouring rays) are processed in the order of their cells, so
// the lanes of a wave share 2-3 cells and the sorted point array (16 B per point, x y z +
// original index) stays L1/L2 resident.
//
// Distances are evaluated as ((dx*dx + dy*dy) + dz*dz) with every operation rounded to fp32
// (no FMA contraction), which is what the numpy oracle computes.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

struct KnnGrid {       // 16 x 4 bytes, lives in device memory (written by the build)
  float ox, oy, oz;    // origin (min corner)
  float cs;            // cell size
  float inv_cs;
  int nx, ny, nz;
  int ncells;
  int npoints;
  int pad[6];
};

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void knn_bbox_init_kernel(unsigned* bb) {
  if (threadIdx.x < 3) bb[threadIdx.x] = 0xffffffffu;       // min
  else if (threadIdx.x < 6) bb[threadIdx.x] = 0u;            // max
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(const float* __restrict__ pts, int np,
                                                       unsigned* __restrict__ bb) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < np; i += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = pts[(size_t)i * 3 + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      atomicMin(&bb[a], f2ord(mn[a]));
      atomicMax(&bb[3 + a], f2ord(mx[a]));
    }
  }
}

__global__ void knn_grid_kernel(const unsigned* __restrict__ bb, KnnGrid* __restrict__ g, int np,
                                float cell_hint, int max_cells) {
  if (threadIdx.x != 0) return;
  float lo[3], ext[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = np > 0 ? ord2f(bb[a]) : 0.0f;
    const float hi = np > 0 ? ord2f(bb[3 + a]) : 0.0f;
    ext[a] = fmaxf(hi - lo[a], 0.0f);
  }
  float cs = fmaxf(cell_hint, 1e-6f);
  int nx, ny, nz;
  for (int it = 0; it < 64; ++it) {
    nx = (int)(ext[0] / cs) + 1;
    ny = (int)(ext[1] / cs) + 1;
    nz = (int)(ext[2] / cs) + 1;
    if ((double)nx * ny * nz <= (double)max_cells) break;
    cs *= 1.25f;
  }
  if ((double)nx * ny * nz > (double)max_cells) { nx = ny = nz = 1; cs = fmaxf(fmaxf(ext[0], ext[1]), ext[2]) + 1.0f; }
  g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2];
  g->cs = cs; g->inv_cs = 1.0f / cs;
  g->nx = nx; g->ny = ny; g->nz = nz;
  g->ncells = nx * ny * nz;
  g->npoints = np;
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_cs, int n) {
  int c = (int)floorf((v - o) * inv_cs);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ __launch_bounds__(256) void knn_count_kernel(const float* __restrict__ pts, int np,
                                                        const KnnGrid* __restrict__ g,
                                                        int* __restrict__ keys, int* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  const int cx = cell_coord(pts[(size_t)i * 3 + 0], g->ox, g->inv_cs, g->nx);
  const int cy = cell_coord(pts[(size_t)i * 3 + 1], g->oy, g->inv_cs, g->ny);
  const int cz = cell_coord(pts[(size_t)i * 3 + 2], g->oz, g->inv_cs, g->nz);
  const int key = (cz * g->ny + cy) * g->nx + cx;
  keys[i] = key;
  atomicAdd(&counts[key], 1);
}

// exclusive scan, 3 launches: per-block (4096 elements) scan + block sums + add-back
constexpr int kScanBlock = 4096;

__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* sh, int* total) {
  // v: this thread's value; returns exclusive prefix within the block (1024 threads)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    int s = lane < 16 ? sh[lane] : 0;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(s, off, 64);
      if (lane >= off) s += t;
    }
    if (lane < 16) sh[lane] = s;
  }
  __syncthreads();
  const int base = wv ? sh[wv - 1] : 0;
  if (total) *total = sh[15];
  return base + inc - v;
}

__global__ __launch_bounds__(1024) void knn_scan1_kernel(const int* __restrict__ counts,
                                                         const KnnGrid* __restrict__ g,
                                                         int* __restrict__ starts, int* __restrict__ bsum) {
  __shared__ int sh[16];
  const int n = g->ncells + 1;
  const int base = blockIdx.x * kScanBlock + threadIdx.x * 4;
  int v[4], s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = (base + q < n - 1) ? counts[base + q] : 0;
    s += v[q];
  }
  int total;
  int ex = block_exclusive_scan_1024(s, sh, &total);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (base + q < n) starts[base + q] = ex;
    ex += v[q];
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void knn_scan2_kernel(int* __restrict__ bsum, int nblocks) {
  __shared__ int sh[16];
  // nblocks <= 1024 (max_cells <= 4M)
  const int v = threadIdx.x < nblocks ? bsum[threadIdx.x] : 0;
  const int ex = block_exclusive_scan_1024(v, sh, nullptr);
  if (threadIdx.x < nblocks) bsum[threadIdx.x] = ex;
}

__global__ __launch_bounds__(1024) void knn_scan3_kernel(const KnnGrid* __restrict__ g,
                                                         int* __restrict__ starts,
                                                         const int* __restrict__ bsum,
                                                         int* __restrict__ fill) {
  const int n = g->ncells + 1;
  const int add = bsum[blockIdx.x];
  const int base = blockIdx.x * kScanBlock + threadIdx.x * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (base + q < n) {
      const int s = starts[base + q] + add;
      starts[base + q] = s;
      fill[base + q] = s;
    }
}

__global__ __launch_bounds__(256) void knn_scatter_kernel(const float* __restrict__ pts, int np,
                                                          const int* __restrict__ keys,
                                                          int* __restrict__ fill,
                                                          float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  const int pos = atomicAdd(&fill[keys[i]], 1);
  sorted[pos] = make_float4(pts[(size_t)i * 3 + 0], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2],
                            __int_as_float(i));
}

// ------------------------------------------------------------------------------------
// query
// ------------------------------------------------------------------------------------
template <int K>
struct TopK {
  float d[K];
  int i[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < K; ++s) { d[s] = FLT_MAX; i[s] = -1; }
  }
  __device__ __forceinline__ static bool less(float da, int ia, float db, int ib) {
    // (distance, index) order; empty slots (i = -1, d = FLT_MAX) sort last
    return da < db || (da == db && (unsigned)ia < (unsigned)ib);
  }
  __device__ __forceinline__ void push(float dd, int ii) {
    if (!less(dd, ii, d[K - 1], i[K - 1])) return;
    d[K - 1] = dd; i[K - 1] = ii;
#pragma unroll
    for (int s = K - 1; s > 0; --s) {
      const bool sw = less(d[s], i[s], d[s - 1], i[s - 1]);
      const float td = sw ? d[s - 1] : d[s];
      const int ti = sw ? i[s - 1] : i[s];
      d[s - 1] = sw ? d[s] : d[s - 1];
      i[s - 1] = sw ? i[s] : i[s - 1];
      d[s] = td; i[s] = ti;
    }
  }
};

__device__ __forceinline__ float dist2_exact(float qx, float qy, float qz, float4 p) {
#pragma clang fp contract(off)
  const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
  float s = dx * dx;
  const float t = dy * dy;
  s = s + t;
  const float u = dz * dz;
  s = s + u;
  return s;
}

template <int K>
__device__ __forceinline__ void knn_search(const float4* __restrict__ sorted,
                                           const int* __restrict__ starts, const KnnGrid& g,
                                           float qx, float qy, float qz, TopK<K>& top) {
  top.init();
  if (g.npoints <= 0) return;
  const int cx = cell_coord(qx, g.ox, g.inv_cs, g.nx);
  const int cy = cell_coord(qy, g.oy, g.inv_cs, g.ny);
  const int cz = cell_coord(qz, g.oz, g.inv_cs, g.nz);
  const int mmax = max(max(g.nx, g.ny), g.nz);
  for (int m = 0; m <= mmax; ++m) {
    const int x0 = max(cx - m, 0), x1 = min(cx + m, g.nx - 1);
    const int y0 = max(cy - m, 0), y1 = min(cy + m, g.ny - 1);
    const int z0 = max(cz - m, 0), z1 = min(cz + m, g.nz - 1);
    // Cells of one (z, y) row are consecutive in `starts` and their points consecutive in `sorted`: a
    // face row of the shell is ONE contiguous point range (2 dependent index loads per row instead
    // of 2 per cell), and the range is walked 4 points at a time with independent 16-byte loads
    // (a lane's search is a chain of dependent loads: this is what bounds the kernel).
    auto scan = [&](int b, int e) {
      int t = b;
      for (; t + 3 < e; t += 4) {
        const float4 p0 = sorted[t], p1 = sorted[t + 1], p2 = sorted[t + 2], p3 = sorted[t + 3];
        top.push(dist2_exact(qx, qy, qz, p0), __float_as_int(p0.w));
        top.push(dist2_exact(qx, qy, qz, p1), __float_as_int(p1.w));
        top.push(dist2_exact(qx, qy, qz, p2), __float_as_int(p2.w));
        top.push(dist2_exact(qx, qy, qz, p3), __float_as_int(p3.w));
      }
      for (; t < e; ++t) {
        const float4 p = sorted[t];
        top.push(dist2_exact(qx, qy, qz, p), __float_as_int(p.w));
      }
    };
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const bool face = (abs(z - cz) == m) || (abs(y - cy) == m);
        const int row = (z * g.ny + y) * g.nx;
        if (face) {                                   // every x of the row belongs to the shell
          scan(starts[row + x0], starts[row + x1 + 1]);
        } else {                                      // only the two end cells (if they are on the shell)
          if (abs(x0 - cx) == m) scan(starts[row + x0], starts[row + x0 + 1]);
          if (x1 != x0 && abs(x1 - cx) == m) scan(starts[row + x1], starts[row + x1 + 1]);
        }
      }
    // distance from q to the faces of the scanned cube that still have cells behind them
    float rho = FLT_MAX;
    if (cx - m > 0) rho = fminf(rho, qx - (g.ox + (cx - m) * g.cs));
    if (cx + m < g.nx - 1) rho = fminf(rho, (g.ox + (cx + m + 1) * g.cs) - qx);
    if (cy - m > 0) rho = fminf(rho, qy - (g.oy + (cy - m) * g.cs));
    if (cy + m < g.ny - 1) rho = fminf(rho, (g.oy + (cy + m + 1) * g.cs) - qy);
    if (cz - m > 0) rho = fminf(rho, qz - (g.oz + (cz - m) * g.cs));
    if (cz + m < g.nz - 1) rho = fminf(rho, (g.oz + (cz + m + 1) * g.cs) - qz);
    if (rho == FLT_MAX) break;                       // the cube covers the whole grid
    if (rho > 0.0f && top.d[K - 1] <= 0.998f * rho * rho) break;  // k-th best is inside
  }
}

template <int K>
__global__ __launch_bounds__(256) void knn_query_kernel(
    const float4* __restrict__ sorted, const int* __restrict__ starts,
    const KnnGrid* __restrict__ gp, const float* __restrict__ q, int Q, float radius,
    const float* __restrict__ radius_ptr, float* __restrict__ D, int64_t* __restrict__ I,
    int* __restrict__ nn) {
  // The 256 queries of a workgroup are processed in the order of their grid cells: lanes that sit in the same
  // cell walk the same shells over the same point ranges - identical trip counts and identical addresses
  // (one L1 transaction per wave instead of one per lane) - where in ray order a wave straddles ~5 cells
  // along its rays and every loop runs for the longest lane.  Each query is still searched by one lane and
  // written to its own row, so the result does not depend on the order.  Sort key = (cell id, local index)
  // in 30 bits (the build caps the grid at 2^22 cells), bitonic in LDS.
  __shared__ unsigned skey[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const KnnGrid g = *gp;
  {
    const int t = base + tid;
    unsigned key = 0xffffffffu;
    if (t < Q && g.npoints > 0) {
      const int cx = cell_coord(q[(size_t)t * 3 + 0], g.ox, g.inv_cs, g.nx);
      const int cy = cell_coord(q[(size_t)t * 3 + 1], g.oy, g.inv_cs, g.ny);
      const int cz = cell_coord(q[(size_t)t * 3 + 2], g.oz, g.inv_cs, g.nz);
      key = ((unsigned)((cz * g.ny + cy) * g.nx + cx) << 8) | (unsigned)tid;
    } else if (t < Q) {
      key = 0xffffff00u | (unsigned)tid;
    }
    skey[tid] = key;
  }
  __syncthreads();
  for (int k = 2; k <= 256; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = tid ^ j;
      if (p > tid) {
        const unsigned a = skey[tid], b = skey[p];
        const bool up = (tid & k) == 0;
        if ((a > b) == up) { skey[tid] = b; skey[p] = a; }
      }
      __syncthreads();
    }
  const unsigned mine = skey[tid];
  if (mine == 0xffffffffu) return;           // past the end of the query array
  const int t = base + (int)(mine & 255u);
  TopK<K> top;
  knn_search<K>(sorted, starts, g, q[(size_t)t * 3 + 0], q[(size_t)t * 3 + 1], q[(size_t)t * 3 + 2], top);
  const float r = radius_ptr ? radius_ptr[t] : radius;
  const float r2 = r * r;
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < K; ++s) {
    D[(size_t)t * K + s] = top.d[s];
    I[(size_t)t * K + s] = (int64_t)top.i[s];
    cnt += (top.d[s] < r2) ? 1 : 0;
  }
  if (nn) nn[t] = cnt;
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_knn_build(glorie_ctx* ctx, const float* points, int np, float cell_size,
                                int max_cells, float* sorted_pos, int* cell_start, void* grid,
                                void* stream) {
  if (!ctx || np < 0 || max_cells < 1 || max_cells >= (1 << 22)) return GLORIE_EINVAL;
  if (!sorted_pos || !cell_start || !grid || (np > 0 && !points)) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // scratch: bbox(8 u32) | keys[np] | counts[max_cells+1] | fill[max_cells+1] | bsum[1024]
  const size_t o_bb = 0, o_keys = 256;
  const size_t o_cnt = o_keys + ((sizeof(int) * (size_t)np + 255) & ~(size_t)255);
  const size_t o_fill = o_cnt + ((sizeof(int) * (size_t)(max_cells + 1) + 255) & ~(size_t)255);
  const size_t o_bsum = o_fill + ((sizeof(int) * (size_t)(max_cells + 1) + 255) & ~(size_t)255);
  const size_t total = o_bsum + sizeof(int) * 1024;
  GLORIE_TRY(ctx_reserve(ctx, total));
  char* base = reinterpret_cast<char*>(ctx->scratch);
  unsigned* bb = reinterpret_cast<unsigned*>(base + o_bb);
  int* keys = reinterpret_cast<int*>(base + o_keys);
  int* counts = reinterpret_cast<int*>(base + o_cnt);
  int* fill = reinterpret_cast<int*>(base + o_fill);
  int* bsum = reinterpret_cast<int*>(base + o_bsum);
  KnnGrid* g = reinterpret_cast<KnnGrid*>(grid);

  hipLaunchKernelGGL(knn_bbox_init_kernel, dim3(1), dim3(64), 0, st, bb);
  if (np > 0) {
    const int nb = min((np + 255) / 256, 1024);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb), dim3(256), 0, st, points, np, bb);
  }
  hipLaunchKernelGGL(knn_grid_kernel, dim3(1), dim3(64), 0, st, bb, g, np, cell_size, max_cells);
  GLORIE_TRY(check_hip(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)(max_cells + 1), st)));
  if (np > 0)
    hipLaunchKernelGGL(knn_count_kernel, dim3((np + 255) / 256), dim3(256), 0, st, points, np, g, keys, counts);
  // the number of cells is only known on the device: scan the full max_cells+1 range (zeros beyond)
  const int nblk = (max_cells + 1 + kScanBlock - 1) / kScanBlock;
  hipLaunchKernelGGL(knn_scan1_kernel, dim3(nblk), dim3(1024), 0, st, counts, g, cell_start, bsum);
  hipLaunchKernelGGL(knn_scan2_kernel, dim3(1), dim3(1024), 0, st, bsum, nblk);
  hipLaunchKernelGGL(knn_scan3_kernel, dim3(nblk), dim3(1024), 0, st, g, cell_start, bsum, fill);
  if (np > 0)
    hipLaunchKernelGGL(knn_scatter_kernel, dim3((np + 255) / 256), dim3(256), 0, st, points, np, keys,
                       fill, reinterpret_cast<float4*>(sorted_pos));
  return check_launch();
}

extern "C" int glorie_knn_query(const float* sorted_pos, const int* cell_start, const void* grid,
                                const float* queries, int Q, int k, float radius,
                                const float* radius_ptr, float* D, int64_t* I, int* nn,
                                void* stream) {
  if (Q < 0 || k < 1) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!sorted_pos || !cell_start || !grid || !queries || !D || !I) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 gridDim((Q + 255) / 256);
#define LAUNCH_K(KK)                                                                              \
  hipLaunchKernelGGL(knn_query_kernel<KK>, gridDim, dim3(256), 0, st,                             \
                     reinterpret_cast<const float4*>(sorted_pos), cell_start,                     \
                     reinterpret_cast<const KnnGrid*>(grid), queries, Q, radius, radius_ptr, D, I, nn)
  switch (k) {
    case 1: LAUNCH_K(1); break;
    case 4: LAUNCH_K(4); break;
    case 8: LAUNCH_K(8); break;
    case 16: LAUNCH_K(16); break;
    default: return GLORIE_EUNSUPPORTED;
  }
#undef LAUNCH_K
  return check_launch();
}
