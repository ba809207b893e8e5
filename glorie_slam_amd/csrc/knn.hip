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
// The k best (distance, index) pairs of a lane, ascending.  A pair is one 64-bit key (distance bits << 32 | index):
// squared distances are >= +0, so the fp32 bit pattern orders like the value and ONE unsigned 64-bit compare is the
// (distance, index) order - ties on distance fall to the smaller index, empty slots (FLT_MAX, -1) sort last.
// push() is what the search spends its VALU time in (it runs whenever ANY lane of the wave improves its list): with
// packed keys an insertion is k compares + 4k selects, no bubble pass and no separate tie logic.
template <int K>
struct TopK {
  unsigned long long key[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < K; ++s) key[s] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | 0xffffffffull;
  }
  __device__ __forceinline__ float dist(int s) const { return __uint_as_float((unsigned)(key[s] >> 32)); }
  __device__ __forceinline__ int index(int s) const { return (int)(unsigned)key[s]; }
  static __device__ __forceinline__ unsigned long long pack(float dd, int ii) {
    return ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)ii;
  }
  // the insertion network, for every lane (a key that does not beat the list - e.g. the all-ones filler - changes nothing)
  __device__ __forceinline__ void insert(unsigned long long x) {
    bool c = x < key[K - 1];
#pragma unroll
    for (int s = K - 1; s > 0; --s) {            // new[s] = x < old[s-1] ? old[s-1] : (x < old[s] ? x : old[s])
      const bool cm = x < key[s - 1];
      key[s] = cm ? key[s - 1] : (c ? x : key[s]);
      c = cm;
    }
    key[0] = c ? x : key[0];
  }
  __device__ __forceinline__ void push(float dd, int ii) {
    const unsigned long long x = pack(dd, ii);
    if (!(x < key[K - 1])) return;
    insert(x);
  }
};

// The list with a per-lane pending buffer in LDS.  push() runs its ~45-instruction network for the whole wave whenever ANY
// lane improves its list - with 64 lanes that is at practically every one of the ~120 candidates a lane looks at, although
// a single lane accepts only ~30-45 of them.  offer() instead appends an accepted key to the lane's own slots (one compare,
// one LDS store) and the network runs in flush(): once any lane holds more than kPendFlush keys, for as many rounds as the
// fullest lane needs - the wave executes max-over-lanes(accepted) networks instead of one per candidate.  Between flushes
// the acceptance bound is stale (too generous): keys that no longer beat the list are dropped by the network itself, so the
// final list is the same set in the same (distance, index) order.  Slot s of thread t is pend[s * 256 + t].
constexpr int kPendSlots = 8, kPendFlush = 4;     // a scan step offers up to 4 candidates: 4 + 4 <= 8
template <int K>
struct TopKBuf {
  TopK<K> top;
  unsigned long long* pend;
  int n;
#ifdef EXP_KNN_STATS
  int st_cand = 0, st_acc = 0, st_ins = 0, st_rows = 0, st_shells = 0;      // experiment: tools/knn_stats.py
#endif
  __device__ __forceinline__ void init(unsigned long long* lds_slot0) { top.init(); pend = lds_slot0; n = 0; }
  __device__ __forceinline__ float dist(int s) const { return top.dist(s); }
  __device__ __forceinline__ int index(int s) const { return top.index(s); }
  // branch-free: the key is stored at the lane's next slot whether it is taken or not, the count only moves if it is
  // (`valid` = false: a padding candidate of a scan step, never taken)
  __device__ __forceinline__ void offer(float dd, int ii, bool valid = true) {
    const unsigned long long x = TopK<K>::pack(dd, ii);
    const bool take = valid && x < top.key[K - 1];
#ifdef EXP_KNN_STATS
    st_cand += valid;
    st_acc += take;
#endif
    pend[n * 256] = x;
    n += take ? 1 : 0;
  }
  // callable under divergence: the ballots see the active lanes only, idle lanes keep their keys for a later flush
  __device__ __forceinline__ void flush() {
#pragma unroll 1
    for (int s = 0; s < kPendSlots; ++s) {
      if (__builtin_amdgcn_ballot_w64(s < n) == 0ull) break;
#ifdef EXP_KNN_STATS
      ++st_ins;
#endif
      top.insert(s < n ? pend[s * 256] : ~0ull);
    }
    n = 0;
  }
  __device__ __forceinline__ void relieve() {
    if (__builtin_amdgcn_ballot_w64(n > kPendFlush) != 0ull) flush();
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

// lim2: squared radius beyond which the caller has no use for a neighbour (FLT_MAX: plain k-NN).  With a limit the search
// treats min(k-th best, lim2) as its bound: cells outside the ball are skipped, and it stops as soon as the scanned cube
// covers the ball - a sample with fewer than k points inside its radius no longer walks shell after shell (with its whole
// wave waiting) to fill slots whose weight is zero anyway.  Every point with d <= lim2 is still found and ranked exactly.
template <int K>
__device__ __forceinline__ void knn_search(const float4* __restrict__ sorted,
                                           const int* __restrict__ starts, const KnnGrid& g,
                                           float qx, float qy, float qz, TopKBuf<K>& top, float lim2 = FLT_MAX) {
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
    // ONE scan site (the insertion is ~60 VALU instructions, inlined once per load slot): ranges are walked 4 points
    // at a time, the tail re-reads the last point with the insertion masked off.
    // (Requesting the next four points before looking at the current four - always legal, the indices are clamped into the
    // range - was measured: 111 instead of 96 VGPRs, 0.180 -> 0.197 ms; the ranges are too short for the extra loads to pay.)
    auto scan = [&](int b, int e) {
      for (int t = b; t < e; t += 4) {
        const int l = e - 1;
        const float4 p0 = sorted[t], p1 = sorted[min(t + 1, l)], p2 = sorted[min(t + 2, l)], p3 = sorted[min(t + 3, l)];
        top.offer(dist2_exact(qx, qy, qz, p0), __float_as_int(p0.w));
        top.offer(dist2_exact(qx, qy, qz, p1), __float_as_int(p1.w), t + 1 < e);
        top.offer(dist2_exact(qx, qy, qz, p2), __float_as_int(p2.w), t + 2 < e);
        top.offer(dist2_exact(qx, qy, qz, p3), __float_as_int(p3.w), t + 3 < e);
        top.relieve();
      }
    };
    // Once the list is full, cells whose box lies farther from q than the current k-th best cannot contribute
    // (push() would reject every point in them): rows are skipped on their (y, z) slab distance and face rows are
    // clipped in x to the ball.  A sample 10 cm off the surface otherwise reads ~400 points of a 5^3 shell to find
    // the handful inside its ball.  Bounds are shrunk by 1e-3 cell and 0.1 % so that the fp32 rounding of the cell
    // assignment can never exclude a point that would have been taken.
    const float slack = 1e-3f * g.cs;
#pragma unroll 1
    for (int z = z0; z <= z1; ++z) {
      const float zlo = g.oz + z * g.cs;
      const float bz = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.cs)) - slack, 0.0f);
#pragma unroll 1
      for (int y = y0; y <= y1; ++y) {
        const bool face = (abs(z - cz) == m) || (abs(y - cy) == m);
#ifdef EXP_KNN_STATS
        ++top.st_rows;
#endif
        const int row = (z * g.ny + y) * g.nx;
        const float ylo = g.oy + y * g.cs;
        const float by = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.cs)) - slack, 0.0f);
        const float byz2 = by * by + bz * bz;
        const float kth = fminf(top.dist(K - 1), lim2);
        const bool full = kth < FLT_MAX;
        if (full && byz2 > 1.001f * kth) continue;
        // up to two cell ranges of this row: the whole (clipped) row on a face, else the end cells that lie on the shell
        int sa = 0, sb = -1, ta = 0, tb = -1;
        if (face) {
          sa = x0; sb = x1;
          if (full) {
            const float rx = sqrtf(fmaxf(1.001f * kth - byz2, 0.0f)) + slack;
            sa = max(x0, cell_coord(qx - rx, g.ox, g.inv_cs, g.nx));
            sb = min(x1, cell_coord(qx + rx, g.ox, g.inv_cs, g.nx));
          }
        } else {
          auto far_cell = [&](int x) {
            const float xlo = g.ox + x * g.cs;
            const float bx = fmaxf(fmaxf(xlo - qx, qx - (xlo + g.cs)) - slack, 0.0f);
            return full && byz2 + bx * bx > 1.001f * kth;
          };
          if (abs(x0 - cx) == m && !far_cell(x0)) { sa = x0; sb = x0; }
          if (x1 != x0 && abs(x1 - cx) == m && !far_cell(x1)) { ta = x1; tb = x1; }
        }
#pragma unroll 1
        for (int seg = 0; seg < 2; ++seg) {
          const int ca = seg ? ta : sa, cb = seg ? tb : sb;
          if (ca <= cb) scan(starts[row + ca], starts[row + cb + 1]);
        }
      }
    }
    top.flush();                                     // the shell is complete: the stop test below needs the true k-th best
#ifdef EXP_KNN_STATS
    ++top.st_shells;
#endif
    // distance from q to the faces of the scanned cube that still have cells behind them
    float rho = FLT_MAX;
    if (cx - m > 0) rho = fminf(rho, qx - (g.ox + (cx - m) * g.cs));
    if (cx + m < g.nx - 1) rho = fminf(rho, (g.ox + (cx + m + 1) * g.cs) - qx);
    if (cy - m > 0) rho = fminf(rho, qy - (g.oy + (cy - m) * g.cs));
    if (cy + m < g.ny - 1) rho = fminf(rho, (g.oy + (cy + m + 1) * g.cs) - qy);
    if (cz - m > 0) rho = fminf(rho, qz - (g.oz + (cz - m) * g.cs));
    if (cz + m < g.nz - 1) rho = fminf(rho, (g.oz + (cz + m + 1) * g.cs) - qz);
    if (rho == FLT_MAX) break;                       // the cube covers the whole grid
    if (rho > 0.0f && fminf(top.dist(K - 1), lim2) <= 0.998f * rho * rho) break;  // k-th best (or the ball) is inside
  }
}

template <int K>
__global__ __launch_bounds__(256) void knn_query_kernel(
    const float4* __restrict__ sorted, const int* __restrict__ starts,
    const KnnGrid* __restrict__ gp, const float* __restrict__ q, int Q, float radius,
    const float* __restrict__ radius_ptr, float* __restrict__ D, int64_t* __restrict__ I,
    int* __restrict__ nn, int S, int image_w, float* __restrict__ wout, uint8_t* __restrict__ has_out, int min_nn,
    int expo_weighting, int ball_only) {
  // The 256 queries of a workgroup are processed in the order of their grid cells: lanes that sit in the same
  // cell walk the same shells over the same point ranges - identical trip counts and identical addresses
  // (one L1 transaction per wave instead of one per lane) - where in ray order a wave straddles ~5 cells
  // along its rays and every loop runs for the longest lane.  Each query is still searched by one lane and
  // written to its own row, so the result does not depend on the order.  Sort key = (cell id, local index)
  // in 30 bits (the build caps the grid at 2^22 cells), bitonic in LDS.
  //
  // image_w > 0: the queries are the S samples of the rays of an image strip (ray = y * image_w + x, query = ray * S + s,
  // what render_img hands over).  The workgroup then takes the s-th sample of a 16 x 16 pixel patch instead of 256
  // consecutive queries: ~25 rays x 10 depths straddle a 20 cm stretch of 6 cm cells, the patch at one depth is a
  // ~6 cm square - one or two cells, the coherence of a global sort by cell without the sort.
  __shared__ unsigned skey[256];
  __shared__ unsigned long long pend[kPendSlots * 256];
  const int tid = threadIdx.x;
  const KnnGrid g = *gp;
  auto query_of = [&](int l) -> int {
    if (image_w <= 0) {
      const int t = blockIdx.x * 256 + l;
      return t < Q ? t : -1;
    }
    const int R = Q / S;
    const int tiles_x = (image_w + 15) >> 4;
    const int s = blockIdx.x % S, tile = blockIdx.x / S;
    const int x = (tile % tiles_x) * 16 + (l & 15), y = (tile / tiles_x) * 16 + (l >> 4);
    const int r = y * image_w + x;
    return (x < image_w && r < R) ? r * S + s : -1;
  };
  {
    const int t = query_of(tid);
    unsigned key = 0xffffffffu;
    if (t >= 0 && g.npoints > 0) {
      const int cx = cell_coord(q[(size_t)t * 3 + 0], g.ox, g.inv_cs, g.nx);
      const int cy = cell_coord(q[(size_t)t * 3 + 1], g.oy, g.inv_cs, g.ny);
      const int cz = cell_coord(q[(size_t)t * 3 + 2], g.oz, g.inv_cs, g.nz);
      key = ((unsigned)((cz * g.ny + cy) * g.nx + cx) << 8) | (unsigned)tid;
    } else if (t >= 0) {
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
  const int t = query_of((int)(mine & 255u));
  TopKBuf<K> top;
  top.init(pend + tid);
  const float r = radius_ptr ? radius_ptr[t] : radius;
  const float r2 = r * r;
  // ball_only: the caller only uses neighbours with d <= r^2 (the IDW weight of the others is zero): bound the search by
  // the ball, a little generously so that the comparison d <= r2 below sees every candidate
  knn_search<K>(sorted, starts, g, q[(size_t)t * 3 + 0], q[(size_t)t * 3 + 1], q[(size_t)t * 3 + 2], top,
                ball_only ? r2 * 1.0001f + 1e-30f : FLT_MAX);
#ifdef EXP_KNN_STATS
  if (K == 8) {
    const float v[5] = {(float)top.st_cand, (float)top.st_acc, (float)top.st_ins, (float)top.st_rows, (float)top.st_shells};
    for (int s = 0; s < 5; ++s) D[(size_t)t * K + s] = v[s];
    return;
  }
#endif
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < K; ++s) {
    D[(size_t)t * K + s] = top.dist(s);
    I[(size_t)t * K + s] = (int64_t)top.index(s);
    cnt += (top.dist(s) < r2) ? 1 : 0;
  }
  if (nn) nn[t] = cnt;
  if (K == 8 && wout) {
    // the inverse-distance weights and the neighbour mask of get_feature_at_pos (decoder.py:130-173) while the list is
    // still in registers - the arithmetic and the summation order of idw_weights_kernel (csrc/render.hip): same bits,
    // one launch and one read of D / I / nn less per batch
    float w[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float d = top.dist(s & (K - 1));
      w[s] = (top.index(s & (K - 1)) >= 0 && !(d > r2)) ? (expo_weighting ? expf(-20.0f * sqrtf(d)) : 1.0f / (d + 1e-10f)) : 0.0f;
    }
    const float s4[4] = {w[0] + w[4], w[1] + w[5], w[2] + w[6], w[3] + w[7]};
    const float s2[2] = {s4[0] + s4[2], s4[1] + s4[3]};
    const float den = fmaxf(s2[0] + s2[1], 1e-12f);
#pragma unroll
    for (int s = 0; s < 8; ++s) wout[(size_t)t * 8 + s] = w[s] / den;
    if (has_out) has_out[t] = cnt > min_nn - 1 ? 1 : 0;
  }
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
  GLORIE_TRY(ctx_poison(ctx, total, st));
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

static int knn_query_launch(const float* sorted_pos, const int* cell_start, const void* grid,
                            const float* queries, int Q, int k, float radius,
                            const float* radius_ptr, float* D, int64_t* I, int* nn, int S, int image_w,
                            void* stream, float* wout = nullptr, uint8_t* has_out = nullptr, int min_nn = 0,
                            int expo = 0, int ball_only = 0) {
  if (Q < 0 || k < 1) return GLORIE_EINVAL;
  if (Q == 0) return GLORIE_OK;
  if (!sorted_pos || !cell_start || !grid || !queries || !D || !I) return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 gridDim((Q + 255) / 256);
  if (image_w > 0) {
    if (S < 1 || Q % S != 0) return GLORIE_EINVAL;
    const int R = Q / S, rows = (R + image_w - 1) / image_w;
    gridDim = dim3((unsigned)(((image_w + 15) / 16) * ((rows + 15) / 16) * S));
  }
#define LAUNCH_K(KK)                                                                              \
  hipLaunchKernelGGL(knn_query_kernel<KK>, gridDim, dim3(256), 0, st,                             \
                     reinterpret_cast<const float4*>(sorted_pos), cell_start,                     \
                     reinterpret_cast<const KnnGrid*>(grid), queries, Q, radius, radius_ptr, D, I, nn, S, image_w, \
                     wout, has_out, min_nn, expo, ball_only)
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

extern "C" int glorie_knn_query(const float* sorted_pos, const int* cell_start, const void* grid,
                                const float* queries, int Q, int k, float radius,
                                const float* radius_ptr, float* D, int64_t* I, int* nn,
                                void* stream) {
  return knn_query_launch(sorted_pos, cell_start, grid, queries, Q, k, radius, radius_ptr, D, I, nn, 1, 0, stream);
}

extern "C" int glorie_knn_query_image(const float* sorted_pos, const int* cell_start, const void* grid,
                                      const float* queries, int Q, int k, float radius,
                                      const float* radius_ptr, float* D, int64_t* I, int* nn,
                                      int samples_per_ray, int image_w, void* stream) {
  if (image_w < 1) return GLORIE_EINVAL;
  return knn_query_launch(sorted_pos, cell_start, grid, queries, Q, k, radius, radius_ptr, D, I, nn,
                          samples_per_ray, image_w, stream);
}

extern "C" int glorie_knn_query_weights(const float* sorted_pos, const int* cell_start, const void* grid,
                                        const float* queries, int Q, float radius, const float* radius_ptr, float* D,
                                        int64_t* I, int* nn, int samples_per_ray, int image_w, int min_nn,
                                        int expo_weighting, int ball_only, float* weights, uint8_t* has, void* stream) {
  if (!weights || !has || !nn || image_w < 0) return GLORIE_EINVAL;
  return knn_query_launch(sorted_pos, cell_start, grid, queries, Q, 8, radius, radius_ptr, D, I, nn,
                          image_w > 0 ? samples_per_ray : 1, image_w, stream, weights, has, min_nn, expo_weighting,
                          ball_only);
}
