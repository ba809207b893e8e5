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
// * Workgroup tile = 128 output channels x 128 pixels, 4 waves as 2 x 2, each 64 x 64 =
//   4 x 4 accumulator blocks of v_mfma_f32_16x16x32_f16.  The MFMA "A" operand is the WEIGHT
//   tile, "B" the pixel tile, so a lane ends up with 4 consecutive output channels of one pixel
//   (8-byte epilogue loads/stores) instead of 4 pixels of one channel.
// * K loop = taps x 64-channel chunks.  Both tiles ([128 rows][64 halfs] = 16 KB each) are staged
//   with global_load_lds_dwordx4 (HBM/L2 -> LDS without touching VGPRs), double buffered, one
//   barrier per step.  The LDS image is lane-linear as the DMA requires; the 16-byte slot of a row
//   is XOR-swizzled with (row & 7) on the SOURCE address and on the fragment read, which turns the
//   8-way bank conflict of 128-byte rows into 2-way.
// * Zero padding: a tap that falls outside the map (or a pixel row beyond P) sources its 128
//   bytes from a zero block appended to the packed weights -- no branches in the K loop.
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
  const _Float16* w;                            // [taps][npad][C] halfs, then 64 zero halfs
  int taps, npad, nout;
  long P; int H, W, HW;
  _Float16* out; int out_stride;
  _Float16* out2; int out2_stride;
  const float* terms; int terms_stride; int act;
  const _Float16* net; int net_stride;
  const _Float16* z; int z_stride;
};

constexpr int kTile = 128;        // pixels and output channels per workgroup
constexpr int kBK = 64;           // channels per K step
constexpr int kTileBytes = kTile * kBK * 2;

__device__ __forceinline__ float csigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float ctanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - e) / (1.0f + e), x);
}

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];   // 2 buffers x (pixel tile, weight tile)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;

  // XCD-aware (bijective) remap of the workgroup id, then (pixel tile, output-channel tile)
  const int nwg = gridDim.x, ntn = a.npad / kTile;
  const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  const int pt = lid / ntn, nt = lid - pt * ntn;
  const long p0 = (long)pt * kTile;
  const int n0 = nt * kTile;

  const int nchunks = a.cha + a.chb;
  const int C = nchunks * kBK;
  const int T = a.taps * nchunks;
  const _Float16* zeros = a.w + (size_t)a.taps * a.npad * C;

  // staging roles: instruction i of wave wv fills rows (i*4 + wv)*8 .. +7, lane -> (row, 16-byte slot)
  const int srow = lane >> 3, slot = lane & 7;
  long prow[4];
  int vmask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wv) * 8 + srow;
    const long p = p0 + row;
    prow[i] = p;
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
  }
  const int sw_src = (slot ^ srow) << 3;     // (row & 7) == srow: swizzled 16-byte slot, in halfs

  auto stage = [&](int t, int buf) {
    const int d = t / nchunks, ch = t - d * nchunks;
    const int shift = a.taps == 9 ? (d / 3 - 1) * a.W + (d % 3 - 1) : 0;
    const bool segA = ch < a.cha;
    const _Float16* xbase = segA ? a.xa : a.xb;
    const int xs = segA ? a.xa_stride : a.xb_stride;
    const int coff = (segA ? ch : ch - a.cha) * kBK + sw_src;
    char* lx = smem + buf * 2 * kTileBytes;
    char* lw = lx + kTileBytes;
    const _Float16* wsrc = a.w + ((size_t)d * a.npad + n0) * C + ch * kBK + sw_src;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = (i * 4 + wv) * 8;
      const _Float16* src = ((vmask[i] >> d) & 1) ? xbase + (prow[i] + shift) * xs + coff : zeros + slot * 8;
      glds16(src, lx + rb * 128);
      glds16(wsrc + (size_t)(rb + srow) * C, lw + rb * 128);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = 64*w? + 16*blk + col, logical slot kk*4 + kg, swizzled with (row & 7) = col & 7
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = col * 128 + (((kk * 4 + kg) ^ (col & 7)) << 4);
  const int wbase = kTileBytes + wm * 64 * 128, xbase_l = wn * 64 * 128;

  stage(0, 0);
  for (int t = 0; t < T; ++t) {
    __syncthreads();                           // tile t landed (vmcnt(0) + barrier); buffer (t+1)&1 is free
    if (t + 1 < T) stage(t + 1, (t + 1) & 1);
    const char* base = smem + (t & 1) * 2 * kTileBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 wf[4], xf[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        wf[mi] = *reinterpret_cast<const f16x8*>(base + wbase + mi * 16 * 128 + foff[kk]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        xf[ni] = *reinterpret_cast<const f16x8*>(base + xbase_l + ni * 16 * 128 + foff[kk]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi], xf[ni], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns channels n0 + wm*64 + mi*16 + kg*4 .. +3 of pixel p0 + wn*64 + ni*16 + col ----
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const long p = p0 + wn * 64 + ni * 16 + col;
    if (p >= a.P) continue;
    const int e = (int)(p / a.HW);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int n = n0 + wm * 64 + mi * 16 + kg * 4;
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
}

}  // namespace glorie

using namespace glorie;

extern "C" int glorie_conv_igemm(const void* xa, int xa_stride, int ca, const void* xb, int xb_stride,
                                 int cb, const void* w_packed, int taps, int nout, int epilogue,
                                 const float* terms, int terms_stride, int act, const void* net,
                                 int net_stride, const void* z, int z_stride, void* out, int out_stride,
                                 void* out2, int out2_stride, int N, int H, int W, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || (taps != 1 && taps != 9) || nout <= 0 || (nout & 3)) return GLORIE_EINVAL;
  if (ca < 0 || cb < 0 || (ca % kBK) || (cb % kBK) || ca + cb == 0) return GLORIE_EINVAL;
  if ((ca && (!xa || (xa_stride & 7))) || (cb && (!xb || (xb_stride & 7)))) return GLORIE_EINVAL;
  if (!w_packed || !out || (out_stride & 3)) return GLORIE_EINVAL;
  if (epilogue == EPI_GRU_ZR && (nout != 256 || !terms || !net || !out2 || (terms_stride & 3))) return GLORIE_EINVAL;
  if (epilogue == EPI_GRU_Q && (nout != 128 || !terms || !net || !z || (terms_stride & 3))) return GLORIE_EINVAL;
  if (epilogue < 0 || epilogue > 2) return GLORIE_EINVAL;
  if (N == 0) return GLORIE_OK;
  ConvArgs a;
  a.xa = reinterpret_cast<const _Float16*>(xa); a.xa_stride = xa_stride; a.cha = ca / kBK;
  a.xb = reinterpret_cast<const _Float16*>(xb); a.xb_stride = xb_stride; a.chb = cb / kBK;
  a.w = reinterpret_cast<const _Float16*>(w_packed);
  a.taps = taps; a.nout = nout; a.npad = (nout + kTile - 1) / kTile * kTile;
  a.P = (long)N * H * W; a.H = H; a.W = W; a.HW = H * W;
  a.out = reinterpret_cast<_Float16*>(out); a.out_stride = out_stride;
  a.out2 = reinterpret_cast<_Float16*>(out2); a.out2_stride = out2_stride;
  a.terms = terms; a.terms_stride = terms_stride; a.act = act;
  a.net = reinterpret_cast<const _Float16*>(net); a.net_stride = net_stride;
  a.z = reinterpret_cast<const _Float16*>(z); a.z_stride = z_stride;
  const long ptiles = (a.P + kTile - 1) / kTile;
  const long nwg = ptiles * (a.npad / kTile);
  if (nwg > 0x7fffffffL) return GLORIE_EINVAL;
  const dim3 grid((unsigned)nwg), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case EPI_BIAS_ACT: hipLaunchKernelGGL(conv_igemm_kernel<EPI_BIAS_ACT>, grid, block, 0, st, a); break;
    case EPI_GRU_ZR: hipLaunchKernelGGL(conv_igemm_kernel<EPI_GRU_ZR>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(conv_igemm_kernel<EPI_GRU_Q>, grid, block, 0, st, a); break;
  }
  return check_launch();
}
