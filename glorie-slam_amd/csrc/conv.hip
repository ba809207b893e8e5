// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores with fused epilogues: the wide
// convolutions of the update operator (scope row A4; /root/reference/src/modules/droid_net/
// droid_net.py:69-139, gru.py:5-34), 75 % of a BA-update iteration.
//
//   out[p][n] = epilogue( sum_{tap d} sum_{c} x[p + off(d)][c] * w[d][n][c] )        fp16 in, fp32 accumulate
//
// * Operands are channels-last fp16 rows ([pixel][C]).  The input may be two channel segments
//   with their own base pointer and row stride (segment A: ca channels, segment B: cb): the GRU
//   input [net | inp, corr, flow] is never concatenated, and r*net can live in its own buffer
//   while other workgroups still read net as halo rows (a fused r*net epilogue that overwrote a
//   net slice in place would race with them).
// * Workgroup tile = 128 output channels x 128 (or 256) pixels, 4 waves as 2 x 2, each 64 channels
//   x 64 (128) pixels = 4 x 4 (4 x 8) accumulator blocks of v_mfma_f32_16x16x32_f16.  The MFMA "A"
//   operand is the WEIGHT tile, "B" the pixel tile, so a lane ends up with 4 consecutive output
//   channels of one pixel (8-byte epilogue loads/stores) instead of 4 pixels of one channel.
// * K loop = 64-channel chunks x taps (taps innermost: the 9 shifted reads of a chunk hit L2).
//   Both tiles are staged with buffer_load_dwordx4 ... lds (HBM/L2 -> LDS without touching VGPRs)
//   into ONE LDS stage (32 KB), 3 workgroups per CU.  Per step: barrier (tile landed) -> all 16
//   fragment reads -> barrier (every wave holds its fragments: the stage is free) -> DMA of the
//   next tile into the same stage -> 32 MFMAs from registers while it streams in.  Occupancy plus
//   this register-level double buffering measured faster than two or three LDS stages with fewer
//   resident workgroups (ST = 2 is kept as a template variant; the other measurements are listed at
//   the dispatch below).  The LDS image is
//   lane-linear as the DMA requires; the 16-byte slot of a row is XOR-swizzled on the SOURCE
//   address and on the fragment read (conflict-free b128 reads).
// * Zero padding: a lane whose row is outside the map for the current tap sets bit 31 of its
//   buffer offset; the hardware range check then writes zeros into LDS (tools/probes/
//   buffer_lds_probe.hip pins that behaviour) -- no branch and no zero page in the K loop.
// * Epilogues: bias + activation; the GRU z/r gates (sigmoid, r * net); the GRU blend
//   (1 - z) * net + z * tanh(.).  The per-edge global-context terms come from glorie_gru_glo_terms.
// * Consecutive workgroup ids are remapped so that one XCD (private L2) owns a contiguous range of
//   pixel tiles: the 9 taps, the halo rows and the output-channel tiles of a pixel range hit L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hiph"

namespace glorie {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { EPI_BIAS_ACT = 0, EPI_GRU_ZR = 1, EPI_GRU_Q = 2 };
enum { CACT_NONE = 0, CACT_RELU = 1, CACT_SIGMOID = 2 };

struct ConvArgs {
  const _Float16* xa; int xa_stride; int cha;   // segment A: cha 64-channel chunks
  const _Float16* xb; int xb_stride; int chb;   // segment B
  const _Float16* w;                            // [taps][npad][C] halfs (+ 64 halfs of padding)
  int taps, npad, nout;
  long P; int H, W, HW;
  _Float16* out; int out_stride;
  _Float16* out2; int out2_stride;
  const float* terms; int terms_stride; int act;
  const _Float16* net; int net_stride;
  const _Float16* z; int z_stride;
  const _Float16* pre; int pre_stride;          // per-pixel term added before the gate non-linearity (or null)
};

constexpr int kTileN = 128;       // output channels per workgroup

__device__ __forceinline__ float csigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float ctanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - e) / (1.0f + e), x);
}

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// NB = 16-pixel blocks per wave, BK = channels per K step (32 | 64), NW = waves (2 channel halves x
// NW/2 pixel groups: pixel tile = NW/2 * 16*NB), ST = LDS stages (1: single stage + register-resident
// fragments, the shipped form; 2: classic double buffering with one __syncthreads per step)
template <int EPI, int NB, int BK, int NW, int ST, int MB = 4>
__global__ __launch_bounds__(64 * NW, ST == 1 ? 3 : 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int TN = 32 * MB;               // output channels per workgroup (MB 16-channel blocks per wave)
  static_assert(ST == 1 || ST == 2, "LDS stages");
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource types are device-only)
  constexpr int PT = (NW / 2) * 16 * NB;      // pixels per workgroup
  constexpr int RB = BK * 2;                  // bytes per staged row
  constexpr int SL = RB / 16;                 // 16-byte slots per row
  constexpr int RPI = 64 / SL;                // rows per wave-wide DMA instruction
  constexpr int XI = PT / RPI / NW;           // DMA instructions per wave per step: pixel tile
  constexpr int WI = TN / RPI / NW;           //                                      weight tile
  constexpr int XBYTES = PT * RB, WBYTES = TN * RB;
  constexpr int KK = BK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ST x (pixel tile, weight tile), one array
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int wm = wv & 1, wn = wv >> 1;

  // 16-byte slot swizzle key of an LDS row: ds_read_b128 of 16 consecutive rows is conflict-free for
  // the hardware's lane groups ({0-3,12-15,20-27}, ...) with these keys (brute-force checked)
  auto key = [](int row) { return RB == 128 ? (row & 7) : ((row >> 1) & 3); };

  // XCD-aware (bijective) remap of the workgroup id, then (pixel tile, output-channel tile)
  const int nwg = gridDim.x, ntn = (a.nout + TN - 1) / TN;
  const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  const int pt = lid / ntn, nt = lid - pt * ntn;
  const long p0 = (long)pt * PT;
  const int n0 = nt * TN;

  const int cpc = 64 / BK;                    // K steps per 64-channel chunk
  const int nsteps_tap = (a.cha + a.chb) * cpc;
  const int C = (a.cha + a.chb) * 64;
  const int T = a.taps * nsteps_tap;

  // Staging through buffer descriptors: address = base + SGPR offset (tap shift, channel chunk:
  // wave-uniform, changes per step) + VGPR offset (row, swizzled slot: per lane, loop-invariant).
  // A lane whose row falls outside the map for this tap gets bit 31 set in its VGPR offset: the
  // hardware range check fails and the DMA writes ZEROS into LDS -- zero padding costs three VALU
  // instructions per load and no branch.  The descriptor base sits `back` rows before the tensor
  // so that the SGPR offset of the (-1,-1) tap is not negative.
  const int back = a.taps == 9 ? a.W + 1 : 0;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xa - (long)back * a.xa_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.xb - (long)back * a.xb_stride), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);

  // staging roles: DMA instruction i of wave wv fills rows (i*4 + wv)*RPI .. +RPI-1; lane -> (row, slot)
  const int srow = lane / SL, slot = lane % SL;
  unsigned voffA[XI], voffB[XI], woff[WI];
  int vmask[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (i * NW + wv) * RPI + srow;
    const int sw = (slot ^ key(row)) << 3;          // swizzled 16-byte slot, in halfs
    const long p = p0 + row;
    int m = 0;
    if (p < a.P) {
      const int xw = (int)(p % a.W), yh = (int)((p / a.W) % a.H);
      if (a.taps == 9) {
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          const int dy = d / 3 - 1, dx = d % 3 - 1;
          if ((unsigned)(yh + dy) < (unsigned)a.H && (unsigned)(xw + dx) < (unsigned)a.W) m |= 1 << d;
        }
      } else {
        m = 1;
      }
    }
    vmask[i] = m;
    const long pc = p < a.P ? p : 0;
    voffA[i] = (unsigned)((pc * a.xa_stride + sw) * 2);
    voffB[i] = (unsigned)((pc * a.xb_stride + sw) * 2);
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (i * NW + wv) * RPI + srow;
    woff[i] = (unsigned)(((size_t)row * C + ((slot ^ key(row)) << 3)) * 2);
  }

  auto stage = [&](int t, int buf) {
    // K order: 64-channel chunk outermost, the taps inside it -- the 9 shifted reads of a chunk of
    // the workgroup's pixel rows follow each other, so all but the first hit L2 (tap-major order
    // has a reuse distance of the whole [pixels x C] panel of every resident workgroup: > L2)
    const int per_chunk = a.taps * cpc;
    const int ch = t / per_chunk, rem = t - ch * per_chunk;
    const int d = rem / cpc, sub = rem - d * cpc;        // tap, BK-wide part of the 64-channel chunk
    const int shift = (a.taps == 9 ? (d / 3 - 1) * a.W + (d % 3 - 1) : 0) + back;
    const bool segA = ch < a.cha;
    const int xs = segA ? a.xa_stride : a.xb_stride;
    const unsigned xsoff = (unsigned)((shift * xs + (segA ? ch : ch - a.cha) * 64 + sub * BK) * 2);
    const unsigned wsoff = (unsigned)((((size_t)d * a.npad + n0) * C + ch * 64 + sub * BK) * 2);
    char* lx = smem + buf * (XBYTES + WBYTES);
    char* lw = lx + XBYTES;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const unsigned inv = ~((unsigned)vmask[i] >> d);
      const unsigned vo = (inv << 31) | (segA ? voffA[i] : voffB[i]);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(segA ? rA : rB,
          (__attribute__((address_space(3))) void*)(lx + (i * NW + wv) * RPI * RB), 16, vo, xsoff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW,
          (__attribute__((address_space(3))) void*)(lw + (i * NW + wv) * RPI * RB), 16, woff[i], wsoff, 0, 0);
  };

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = 16*blk + col, logical slot kk*4 + kg, swizzle key(row) = key(col)
  int foff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) foff[kk] = col * RB + (((kk * 4 + kg) ^ key(col)) << 4);
  const int wbase = XBYTES + wm * (16 * MB) * RB, xbase_l = wn * (16 * NB) * RB;

  stage(0, 0);
  if (EPI != EPI_BIAS_ACT && a.pre) {
    // the hoisted per-pixel term seeds the accumulators: its loads travel together with the first tile
    // (an add in the epilogue would be an exposed round trip at the tail of every workgroup)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
      const long p = p0 + wn * (16 * NB) + ni * 16 + col;
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) {
        const int n = n0 + wm * (16 * MB) + mi * 16 + kg * 4;
        if (p < a.P && n < a.nout) {
          const f16x4 pv = *reinterpret_cast<const f16x4*>(a.pre + p * a.pre_stride + n);
          acc[mi][ni] = f32x4{(float)pv[0], (float)pv[1], (float)pv[2], (float)pv[3]};
        }
      }
    }
  }
  int cur = 0;                                 // LDS stage holding tile t
  for (int t = 0; t < T; ++t) {
    __syncthreads();                           // tile t landed (vmcnt(0) + barrier); ST = 2: the other stage is free
    const char* base = smem + cur * (XBYTES + WBYTES);
    // all fragment reads of the step go out first (one exposed LDS latency per step, not per kk), the
    // DMA of the next tile is issued in their shadow, then the MFMAs run back to back
    f16x8 wf[KK][MB], xf[KK][NB];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
        wf[kk][mi] = *reinterpret_cast<const f16x8*>(base + wbase + mi * 16 * RB + foff[kk]);
#pragma unroll
      for (int ni = 0; ni < NB; ++ni)
        xf[kk][ni] = *reinterpret_cast<const f16x8*>(base + xbase_l + ni * 16 * RB + foff[kk]);
    }
    if (ST == 1) {
      // single LDS stage, 3 workgroups per CU.  Every fragment of the step is in registers now, so
      // once all waves got theirs the buffer is free: tile t+1 streams into it under the MFMAs.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 1 < T) stage(t + 1, 0);
    } else {
      if (t + 1 < T) stage(t + 1, cur ^ 1);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][mi], xf[kk][ni], acc[mi][ni], 0, 0, 0);
    if (ST == 2) cur ^= 1;
  }

  // ---- epilogue: lane owns channels n0 + wm*64 + mi*16 + kg*4 .. +3 of pixel p0 + wn*16*NB + ni*16 + col ----
#pragma unroll
  for (int ni = 0; ni < NB; ++ni) {
    const long p = p0 + wn * (16 * NB) + ni * 16 + col;
    if (p >= a.P) continue;
    const int e = (int)(p / a.HW);
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const int n = n0 + wm * (16 * MB) + mi * 16 + kg * 4;
      if (n >= a.nout) continue;
      const f32x4 v = acc[mi][ni];
      f16x4 o;
      if (EPI == EPI_BIAS_ACT) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.terms) b = *reinterpret_cast<const float4*>(a.terms + n);
        float f[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (a.act == CACT_RELU) f[k] = fmaxf(f[k], 0.0f);
          else if (a.act == CACT_SIGMOID) f[k] = csigmoid(f[k]);
          o[k] = (_Float16)f[k];
        }
        *reinterpret_cast<f16x4*>(a.out + p * a.out_stride + n) = o;
      } else if (EPI == EPI_GRU_ZR) {
        // channels 0..127: z = sigmoid(.) ; 128..255: r -> r * net          (gru.py:28-30)
        const float4 g = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + n);
        const float s[4] = {csigmoid(v[0] + g.x), csigmoid(v[1] + g.y), csigmoid(v[2] + g.z),
                            csigmoid(v[3] + g.w)};
        if (n < 128) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (_Float16)s[k];
          *reinterpret_cast<f16x4*>(a.out + p * a.out_stride + n) = o;
        } else {
          const f16x4 nv = *reinterpret_cast<const f16x4*>(a.net + p * a.net_stride + (n - 128));
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (_Float16)(s[k] * (float)nv[k]);
          *reinterpret_cast<f16x4*>(a.out2 + p * a.out2_stride + (n - 128)) = o;
        }
      } else {
        // net' = (1 - z) * net + z * tanh(.)                                (gru.py:31-33)
        const float4 g = *reinterpret_cast<const float4*>(a.terms + (size_t)e * a.terms_stride + n);
        const f16x4 nv = *reinterpret_cast<const f16x4*>(a.net + p * a.net_stride + n);
        const f16x4 zv = *reinterpret_cast<const f16x4*>(a.z + p * a.z_stride + n);
        const float qv[4] = {ctanh(v[0] + g.x), ctanh(v[1] + g.y), ctanh(v[2] + g.z), ctanh(v[3] + g.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zz = (float)zv[k];
          o[k] = (_Float16)((1.0f - zz) * (float)nv[k] + zz * qv[k]);
        }
        *reinterpret_cast<f16x4*>(a.out + p * a.out_stride + n) = o;
      }
    }
  }
#endif
}

template <int EPI, int NB, int BK, int NW, int ST, int MB>
static void launch_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr int RB = BK * 2, PT = (NW / 2) * 16 * NB;
  constexpr size_t lds = (size_t)ST * (PT * RB + 32 * MB * RB);
  static bool attr = false;            // > 64 KB of dynamic LDS needs the opt-in once per kernel
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<EPI, NB, BK, NW, ST, MB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<EPI, NB, BK, NW, ST, MB>), grid, dim3(64 * NW), lds, st, a);
}

template <int NB, int BK, int NW, int ST, int MB = 4>
static int launch_conv(const ConvArgs& a, int epilogue, hipStream_t st) {
  constexpr int PT = (NW / 2) * 16 * NB;
  const long ptiles = (a.P + PT - 1) / PT;
  const long nwg = ptiles * ((a.nout + 32 * MB - 1) / (32 * MB));
  if (nwg > 0x7fffffffL) return GLORIE_EINVAL;
  const dim3 grid((unsigned)nwg);
  switch (epilogue) {
    case EPI_BIAS_ACT: launch_one<EPI_BIAS_ACT, NB, BK, NW, ST, MB>(a, grid, st); break;
    case EPI_GRU_ZR: launch_one<EPI_GRU_ZR, NB, BK, NW, ST, MB>(a, grid, st); break;
    default: launch_one<EPI_GRU_Q, NB, BK, NW, ST, MB>(a, grid, st); break;
  }
  return check_launch();
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_conv_igemm(const void* xa, int xa_stride, int ca, const void* xb, int xb_stride,
                                 int cb, const void* w_packed, int taps, int nout, int epilogue,
                                 const float* terms, int terms_stride, int act, const void* net,
                                 int net_stride, const void* z, int z_stride, void* out, int out_stride,
                                 void* out2, int out2_stride, const void* pre, int pre_stride, int N, int H,
                                 int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || (taps != 1 && taps != 9) || nout <= 0 || (nout & 3)) return GLORIE_EINVAL;
  if (ca < 0 || cb < 0 || (ca % 64) || (cb % 64) || ca + cb == 0) return GLORIE_EINVAL;
  if (epilogue < 0 || epilogue > 2) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;               // nothing to do: pointers of empty maps may be null
  if ((ca && (!xa || (xa_stride & 7))) || (cb && (!xb || (xb_stride & 7)))) return GLORIE_EINVAL;
  if (!w_packed || !out || (out_stride & 3)) return GLORIE_EINVAL;
  if (epilogue == EPI_GRU_ZR && (nout != 256 || !terms || !net || !out2 || (terms_stride & 3))) return GLORIE_EINVAL;
  if (epilogue == EPI_GRU_Q && (nout != 128 || !terms || !net || !z || (terms_stride & 3))) return GLORIE_EINVAL;
  if (pre && (epilogue == EPI_BIAS_ACT || (pre_stride & 3))) return GLORIE_EINVAL;
  ConvArgs a;
  a.xa = reinterpret_cast<const _Float16*>(xa); a.xa_stride = xa_stride; a.cha = ca / 64;
  a.xb = reinterpret_cast<const _Float16*>(xb); a.xb_stride = xb_stride; a.chb = cb / 64;
  a.w = reinterpret_cast<const _Float16*>(w_packed);
  a.taps = taps; a.nout = nout; a.npad = (nout + kTileN - 1) / kTileN * kTileN;
  a.P = (long)N * H * W; a.H = H; a.W = W; a.HW = H * W;
  a.out = reinterpret_cast<_Float16*>(out); a.out_stride = out_stride;
  a.out2 = reinterpret_cast<_Float16*>(out2); a.out2_stride = out2_stride;
  a.terms = terms; a.terms_stride = terms_stride; a.act = act;
  a.net = reinterpret_cast<const _Float16*>(net); a.net_stride = net_stride;
  a.z = reinterpret_cast<const _Float16*>(z); a.z_stride = z_stride;
  a.pre = reinterpret_cast<const _Float16*>(pre); a.pre_stride = pre_stride;
  // buffer-descriptor addressing: 31-bit byte offsets per input segment and for the weights
  const long lim = 0x7fffffffL;
  if (((a.P + W + 2) * (long)xa_stride + 64) * 2 > lim || ((a.P + W + 2) * (long)xb_stride + 64) * 2 > lim ||
      ((long)taps * a.npad * (ca + cb) + 64) * 2 > lim)
    return GLORIE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // Measured on the update operator's layers at 36x60x80 (tools/bench_conv.py): the single-stage
  // 128x128 tile at 3 workgroups per CU (1050 TFLOP/s on the 448->256 layer) beats every variant with
  // more LDS stages and fewer resident workgroups: 2 stages 930, 2 stages + register-resident fragments
  // (loads two steps ahead, ST = 4) 915, 8 waves with a 3-stage ring and counted vmcnt 830, 256-pixel
  // tiles 850, 32-channel steps at 4 workgroups per CU 824; the same wave tile on v_mfma_f32_32x32x16_f16
  // (16 instead of 32 MFMAs per step) 945.
  // layers with <= 64 output channels (flow_encoder[2]) use a 64-channel tile instead of padding to 128
  if (nout <= 64 && epilogue == EPI_BIAS_ACT) return launch_conv<4, 64, 4, 1, 2>(a, epilogue, st);
  return launch_conv<4, 64, 4, 1>(a, epilogue, st);
}
