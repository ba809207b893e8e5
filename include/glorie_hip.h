/*
 * glorie_hip.h -- C ABI of libglorie_hip.so, the MI355X (gfx950) implementation of
 * GlORIE-SLAM's dense hot path.
 *
 * This is the drop-in boundary.  It replaces the reference's pybind11 module
 * `droid_backends` (/root/reference/src/lib/droid.cpp:239-252) and the three third-party
 * boundaries the hot path crosses (lietorch SE3 retraction inside `ba`, faiss-gpu
 * `IndexIVFFlat.search`, torch_scatter segment sums).  Every entry point
 *
 *   - is `extern "C"`, takes raw DEVICE pointers + explicit sizes, never a torch type;
 *   - writes into CALLER-OWNED output buffers (allocation stays with the host framework);
 *   - enqueues on the `hipStream_t` passed as the trailing `void* stream` (NULL = default
 *     stream) and does not synchronise with the host;
 *   - returns GLORIE_OK (0) or a negative glorie_status; no exception crosses the ABI.
 *
 * Index tensors are int64 (`long` in the reference kernels), images are row-major,
 * poses are [tx ty tz qx qy qz qw], intrinsics are [fx fy cx cy].
 *
 * Each declaration cites the reference interface it replaces.
 */
#ifndef GLORIE_HIP_H
#define GLORIE_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum glorie_status {
  GLORIE_OK = 0,
  GLORIE_EINVAL = -1,    /* bad argument (null pointer, negative size, unsupported dtype) */
  GLORIE_EHIP = -2,      /* a HIP runtime call failed; see glorie_last_hip_error()         */
  GLORIE_ENOMEM = -3,    /* scratch arena too small / allocation failed                    */
  GLORIE_EUNSUPPORTED = -4
} glorie_status;

typedef enum glorie_dtype { GLORIE_F16 = 0, GLORIE_F32 = 1 } glorie_dtype;

/* library identification: "glorie_hip <version> gfx950" */
const char* glorie_version(void);
/* hipError_t of the last failing HIP call on this thread (0 if none) */
int glorie_last_hip_error(void);

/* ------------------------------------------------------------------------------------ */
/* Context: scratch arena + (optional) multi-GPU state.  One per process / GPU.          */
/* ------------------------------------------------------------------------------------ */
typedef struct glorie_ctx glorie_ctx;

/* scratch_bytes: initial size of the device scratch arena (grown on demand outside of
 * stream capture). */
int glorie_ctx_create(glorie_ctx** out, size_t scratch_bytes);
int glorie_ctx_destroy(glorie_ctx* ctx);
/* Grow the arena to at least scratch_bytes now (never inside a stream capture).  Growing re-allocates:
 * kernels recorded into a hipGraph before the move still point into the old block. */
int glorie_ctx_reserve(glorie_ctx* ctx, size_t scratch_bytes);
/* Number of times the arena of `ctx` has moved.  An owner of recorded launches (FactorGraph's hipGraph replay)
 * compares it with the value at capture time and re-captures when it differs. */
unsigned long long glorie_ctx_generation(const glorie_ctx* ctx);

/* ------------------------------------------------------------------------------------ */
/* A. correlation lookup                                                                 */
/* ------------------------------------------------------------------------------------ */

/* droid_backends.corr_index_forward(volume, coords, radius)
 *   reference: src/lib/droid.cpp:172-180, src/lib/correlation_kernels.cu:19-70,126-155
 * volume [N,h1,w1,h2,w2] (dtype f16|f32), coords [N,2,h1,w1] f32 (already scaled to this
 * level), out [N,2r+1,2r+1,h1,w1] same dtype as volume (fully overwritten; the reference
 * zero-fills and accumulates).  Channel order: x-offset major (out[n][i][j], i<->x).
 * f16 results are bit-identical to the reference's rounding sequence. */
int glorie_corr_index_fwd(const void* volume, const float* coords, void* out,
                          int N, int h1, int w1, int h2, int w2, int radius,
                          int dtype, void* stream);

/* Fused 4-level form of CorrBlock.__call__ (src/modules/droid_net/corr.py:43-53):
 * volumes[l] is level l ([N,h1,w1,h2>>l,w2>>l]); coords [N,2,h1,w1] are UNscaled
 * (the kernel applies /2^l, exact in fp32); out is the concatenated
 * [N, L*(2r+1)^2, h1, w1] tensor the reference builds with torch.cat. */
int glorie_corr_lookup_pyramid(const void* const* volumes, int num_levels,
                               const float* coords, void* out,
                               int N, int h1, int w1, int h2, int w2, int radius,
                               int dtype, void* stream);

/* Same lookup (radius 3, fp16) on a TILED pyramid: level l of every source pixel is stored as
 * [ceil(h2l/4)][ceil(w2l/8)][4][8] halfs (64-byte blocks of 4 rows x 8 columns, zero padded), which
 * cuts the HBM sectors touched per window from ~9.7 to ~5.2.  Values are bit-identical to
 * glorie_corr_lookup_pyramid on the same data.  Each volume needs >= 16 bytes of readable slack
 * before and after it.  h1*w1 % 8 == 0, out and coords 16-byte aligned. */
int glorie_corr_lookup_pyramid_tiled(const void* const* volumes, int num_levels, const float* coords,
                                     void* out, int N, int h1, int w1, int h2, int w2, void* stream);

/* CorrBlock.__init__ for a batch of new edges (reference: src/modules/droid_net/corr.py:26-41,67-76 - all-pairs
 * <f1/4, f2/4> as an fp16 GEMM with fp32 accumulation, then avg_pool2d level by level on the fp16 values), written
 * directly in the tiled layout above into SLOTS of an arena: levels[l] is [capacity*h*w][plane_l] fp16 (plane_l =
 * ceil((h>>l)/4)*ceil((w>>l)/8)*32 halfs, zero-initialised once: padding is never written), slots[e] the arena slot of
 * new edge e.  fmaps_cl [F][h*w][128] fp16 channels-last, already scaled by 1/4 (the layout of glorie_corr_otf's
 * level 0); ii / jj [n_new] select the source / target map.  C must be 128, w % 8 == 0.
 * Edges keep their slot for life: add_factors / rm_factors / cat / __getitem__ of the reference (factor_graph.py:126,
 * 161 re-copy every volume through a mask) become updates of the slot list. */
int glorie_corr_build(const void* fmaps_cl, const int64_t* ii, const int64_t* jj, const int* slots,
                      void* const* levels, int num_levels, int n_new, int h, int w, int C, void* stream);

/* glorie_corr_lookup_pyramid_tiled on an arena: edge n reads the volumes of slot slots[n] (int32 [N]) */
int glorie_corr_lookup_arena(const void* const* levels, int num_levels, const int* slots, const float* coords,
                             void* out, int N, int h1, int w1, int h2, int w2, void* stream);

/* The tiled lookup with a channels-last result for the fused update operator: out = fp16 [N*h1*w1][256], channel
 * l * 64 + dy * 8 + dx holds the reference's channel l * 49 + dx * 7 + dy (corr.py:43-53: CorrSampler output
 * [.., 2r+1 (x), 2r+1 (y), h, w], levels concatenated), the 8th row / tap of every level is zero.  corr_encoder[0]
 * (droid_net.py:73-75, 1x1, 196 -> 128) then runs as glorie_conv_igemm with taps = 1 on weight columns permuted the same way
 * (update_ops.pack_corr_encoder).  slots may be NULL (volumes stacked in edge order); 4 levels, radius 3 only.
 * coords_xy != 0: coords are [N][h1*w1][2] (x, y interleaved, the layout the reprojection produces, factor_graph.py:205-215)
 * instead of the planar [N][2][h1*w1]. */
int glorie_corr_lookup_tiled_cl(const void* const* levels, int num_levels, const int* slots, const float* coords,
                                int coords_xy, void* out, int N, int h1, int w1, int h2, int w2, void* stream);

/* The pyramid in the DISPLACEMENT-MAJOR, source-tiled layout (round 3; csrc/corr_dm.hip) - the same fp16 values as the
 * reference's per-pixel planes [N,h,w,h>>l,w>>l] (corr.py:26-41), stored so that neighbouring source pixels that look at the
 * same displacement share a 128-byte line:
 *   levels[l] = [capacity][ntiles][(h>>l) * Wp][64] fp16,  ntiles = ceil(h/8)*ceil(w/8)  (8 x 8 source tiles, row-major),
 *   Wp = (w>>l) rounded up to even;  element [slot][tile][dy][dx >> 1][(sy&7)*8 + (sx&7)][dx & 1] =
 *   volume_l[slot][sy][sx][ty][tx] with  dy = (ty - (sy>>l) + ((h>>l)>>1)) mod (h>>l),
 *   dx = (tx - (sx>>l) + ((w>>l)>>1)) mod Wp  (round 4: a lane's two neighbouring displacement columns are one dword, so a
 *   window row is five dword gathers instead of eight 2-byte ones; for odd w>>l the displacement that would name target
 *   column w>>l holds a zero).
 * glorie_corr_dm_level_halfs: halfs per slot of level l (-1 for bad arguments).
 * glorie_corr_dm_build: CorrBlock.__init__ for a batch of new edges into arena slots; arguments as glorie_corr_build (any
 *   width whose staging fits LDS, w <= ~112; C must be 128); lanes of padding source pixels (sy >= h or sx >= w) are
 *   never read by the lookup.
 * glorie_corr_dm_lookup: CorrBlock.__call__ (corr.py:43-53 -> correlation_kernels.cu:19-70; radius 3, 4 levels, fp16,
 *   bit-identical to glorie_corr_lookup_pyramid on the same volumes) with, optionally, corr_encoder[0] of the update
 *   operator (droid_net.py:73-77: 1x1 convolution 196 -> 128, bias, ReLU) as an MFMA epilogue of the same launch.
 *   coords: UNscaled, [N][h*w][2] (coords_xy != 0, the reprojection's layout) or planar [N][2][h*w]; slots int32 [N] or NULL.
 *   corr_cl: NULL or fp16 [N*h*w][256], channel l*64 + j*8 + i = the reference's channel l*49 + i*7 + j (i <-> x), 8th row /
 *   tap zero (the layout of glorie_corr_lookup_tiled_cl).
 *   enc_out: NULL or fp16 rows of enc_stride halfs per edge-pixel (a channels-last [N,128,h,w] map or a 128-channel slice of
 *   a wider one; enc_stride % 8 == 0, 16-byte aligned): enc_out[(n*h*w + p)*enc_stride + o] = relu(sum_k enc_w[o][k] corr[k] +
 *   enc_b[o]), fp32 accumulation over the fp16 lookup values; enc_w fp16 [128][224] with column l*56 + j*8 + i =
 *   W[o][l*49 + i*7 + j], zero for i == 7 (the packing of glorie_corr_otf_encode: update_ops.pack_corr_encoder_dm);
 *   enc_b f32 [128].
 *   At least one of corr_cl / enc_out must be given. */
long glorie_corr_dm_level_halfs(int h, int w, int level);
int glorie_corr_dm_build(const void* fmaps_cl, const int64_t* ii, const int64_t* jj, const int* slots,
                         void* const* levels, int num_levels, int n_new, int h, int w, int C, void* stream);
int glorie_corr_dm_lookup(const void* const* levels, const int* slots, const float* coords, int coords_xy, int N,
                          int h, int w, void* corr_cl, const void* enc_w, const float* enc_b, void* enc_out,
                          int enc_stride, void* stream);

/* Volume-free form of CorrBlock.__call__ / AltCorrBlock.__call__
 *   reference: src/modules/droid_net/corr.py:43-53 (volume lookup), :79-145 (alt-corr),
 *   src/lib/altcorr_kernel.cu:27-149
 * fmap1 [F,h*w,C] fp16 channel-last, pre-scaled by 1/4 (corr.py:70-71); fmap2_levels[l]
 * [F,(h>>l)*(w>>l),C] fp16 = avg-pooled pyramid of the same maps; coords [N,2,h,w] f32
 * (unscaled); ii/jj [N] int64 select the source / target frame of every edge;
 * out [N, L*49, h, w] fp16 (radius 3).  C must be 128.  Dot products run on
 * v_mfma_f32_16x16x32_f16 (fp32 accumulate) and are rounded to fp16 before the fp16 bilinear
 * blend, i.e. the values a materialised fp16 volume would hold. */
int glorie_corr_otf(const void* fmap1, const void* const* fmap2_levels, int num_levels,
                    const float* coords, const int64_t* ii, const int64_t* jj, void* out,
                    int N, int h, int w, int C, void* stream);

/* The same lookup with corr_encoder[0] of the update operator fused behind it
 *   reference: src/modules/droid_net/droid_net.py:73-77 (1x1 convolution 196 -> 128, bias, ReLU on the looked-up
 *   features) - the first consumer of CorrBlock.__call__ in UpdateModule.forward (droid_net.py:121)
 * enc_w: fp16 [128][224], enc_w[o][l*56 + j*8 + i] = W[o][l*49 + i*7 + j] for i < 7, zero for i == 7 (the order the
 * kernel stages a pixel's looked-up features in LDS); enc_b: f32 [128];
 * enc_out: fp16 rows of `enc_stride` halfs per edge-pixel (channels-last map or a channel slice of one,
 * enc_stride % 4 == 0, 8-byte aligned): enc_out[(n*h*w + p) * enc_stride + o] = relu(sum_k W[o][k] corr[n][k][p] + b[o]),
 * corr rounded to fp16 like the lookup's own output, fp32 accumulation.
 * corr_out: NULL (the 196-channel map never goes to HBM) or [N,196,h,w] fp16 as for glorie_corr_otf.
 * num_levels must be 4, C 128. */
int glorie_corr_otf_encode(const void* fmap1, const void* const* fmap2_levels, int num_levels,
                           const float* coords, const int64_t* ii, const int64_t* jj, void* corr_out,
                           int N, int h, int w, int C, const void* enc_w, const float* enc_b,
                           void* enc_out, int enc_stride, void* stream);

/* droid_backends.altcorr_forward(fmap1, fmap2, coords, radius)
 *   reference: src/lib/droid.cpp:195-205, src/lib/altcorr_kernel.cu:27-149,290-319
 * fmap1 [B,H,W,C], fmap2 [B,H2,W2,C], coords [B,S,H,W,2] f32, out [B,S,(2r+1)^2,H,W].
 * dtype f32 (the reference casts to float at the call site, corr.py:125). */
int glorie_altcorr_fwd(const float* fmap1, const float* fmap2, const float* coords,
                       float* out, int B, int S, int H, int W, int H2, int W2, int C,
                       int radius, void* stream);

/* Bookkeeping of FactorGraph.update between the update operator and the BA, in one launch
 *   reference: src/factor_graph.py:219-223 (target = coords1 + delta, damping[unique(ii)] = damping), :248
 *   (BA damping 0.2 * damping + EP), :256 (age += 1)
 * coords1 / delta / target: n_target floats ([N,h,w,2]); eta [G,HW] = the operator's damping output for the G frames
 * `frames` (int64 [G]); damping_table [B,HW] receives the rows, damping_ba [G,HW] = 0.2 * eta + ep (each op rounded
 * separately, as the torch chain); age int64 [n_edges] is incremented (may be NULL). */
int glorie_update_bookkeeping(const float* coords1, const float* delta, float* target, long n_target,
                              const float* eta, const int64_t* frames, float* damping_table, float* damping_ba,
                              int G, int HW, float ep, int64_t* age, int n_edges, void* stream);

/* Fused stages of UpdateModule / ConvGRU / GraphAgg between the (MIOpen) convolutions
 *   reference: src/modules/droid_net/gru.py:20-34, src/modules/droid_net/droid_net.py:34-66,106-139
 * All activations are channels-last fp16: row p (= edge*HW + pixel) holds C contiguous halfs;
 * `*_stride` is the distance between rows in halfs, so operands may be channel slices of wider
 * buffers (e.g. the 448-channel GRU input), which is how torch.cat disappears.
 * act codes: 0 none, 1 ReLU, 2 sigmoid, 3 softplus.
 *   glorie_bias_act      y = act(x + bias); C % 8 == 0
 *   glorie_gru_glo_terms glo[n][c] = mean_p sigmoid(wn + bw) * net (gru.py:25-26), then
 *                        g[n][o] = Gb[o] + sum_c glo[n][c] G[c][o], o < M: the convz_glo | convr_glo |
 *                        convq_glo terms with the conv biases folded into Gb.  `partial` is a
 *                        workspace of N*parts*128 floats (parts = pixel slices reduced in parallel;
 *                        the final sum runs in a fixed order: deterministic).
 *   glorie_gru_gate_zr   z = sigmoid(zc + g[n][0:128]); rnet = sigmoid(rc + g[n][128:256]) * net;
 *                        zr = raw output [P,256] of the merged convz|convr     (gru.py:28-30)
 *   glorie_gru_gate_q    out = (1 - z) * net + z * tanh(qc + gq[n])            (gru.py:31-33);
 *                        out2 (nullable) receives a second copy (net slice of the next GRU input)
 *   glorie_segment_mean  out[g] = mean_{e: ix[e]==g} act(x[e] + bias), C = 128: scatter_mean of
 *                        GraphAgg (droid_net.py:53-59) with the bias+ReLU of conv1 folded in
 *   glorie_conv3x3_small 3x3 convolution, zero padding, 128 -> K<=3 channels for `groups`<=4 heads
 *                        reading consecutive 128-channel slices of x (delta / weight / eta heads,
 *                        droid_net.py:85-93,42-44): out[grp][p][j] = scale * act_grp(out_bias +
 *                        conv(act_in(x + in_bias))), fp32.  act_packed = act codes, 4 bits per
 *                        group.  `taps` = workspace of P*groups*9K floats.  w_packed = fp16 MFMA B
 *                        fragments [groups][NT][4][64][8], NT = 1 if 9K <= 16 else 2: element
 *                        [grp][t][kk][lane][i] = w[grp][j][32kk + 8(lane>>4) + i][d] for column
 *                        16t + (lane&15) = 3*3 tap d (row-major ky,kx) * K + j, zero beyond 9K. */
int glorie_bias_act(const void* x, int x_stride, const float* bias, void* y, int y_stride,
                    long P, int C, int act, void* stream);
int glorie_gru_glo_terms(const void* wn, int w_stride, const float* bw, const void* net, int n_stride,
                         const float* G, const float* Gb, int M, float* partial, int parts, float* g,
                         int N, int HW, void* stream);
/* glorie_gru_glo_terms from the per-tile partial sums that glorie_conv_igemm's epilogue 3 leaves (the 1x1 convolution w of
 * gru.py:25 with the sigmoid, the product with net and the pixel reduction in its epilogue): tiles = float
 * [ceil(N*HW/128)][2][128]. */
int glorie_gru_glo_from_tiles(const float* tiles, const float* G, const float* Gb, int M, float* g, int N, int HW,
                              void* stream);
int glorie_gru_gate_zr(const void* zr, int zr_stride, const float* g, int g_stride, const void* net,
                       int n_stride, void* z, int z_stride, void* rnet, int r_stride, int N, int HW,
                       void* stream);
int glorie_gru_gate_q(const void* qc, int q_stride, const float* gq, int gq_stride, const void* z,
                      int z_stride, const void* net, int n_stride, void* out, int o_stride, void* out2,
                      int o2_stride, int N, int HW, void* stream);
int glorie_segment_mean(const void* x, int x_stride, const float* bias, int relu, const int64_t* ix,
                        int N, void* out, int o_stride, int G, int HW, void* stream);
int glorie_conv3x3_small(const void* x, int x_stride, const float* in_bias, int in_relu,
                         const void* w_packed, const float* out_bias, int groups, int K,
                         int act_packed, float scale, float* taps, float* out, int N, int H, int W,
                         void* stream);

/* Implicit-GEMM convolution (3x3 zero-padded, or 1x1) on the matrix cores with fused epilogues:
 * the wide convolutions of UpdateModule / ConvGRU (droid_net.py:73-104, gru.py:10-19).
 *   out[p][n] = epilogue( sum_taps sum_c x[p + off(tap)][c] * w[tap][n][c] ),  fp16 in, fp32 accumulate
 * Input = channels-last fp16 rows in up to two channel segments (xa: ca channels, xb: cb channels,
 * each with its own row stride; ca, cb multiples of 64; either may be 0) -- the GRU input
 * [net | inp, corr, flow] without a concatenation.  N maps of H x W pixels.
 * w_packed: fp16 [taps][npad][ca+cb] (npad = nout rounded up to 128, padding rows zero, tap order
 * row-major ky,kx, channel order segment A then B) followed by 64 zero halfs.  nout % 4 == 0.
 * epilogue 0: out[p][n] = act(acc + terms[n])                 terms = bias [nout] or NULL
 * epilogue 1: GRU gates, nout = 256 (convz | convr merged, gru.py:28-30):
 *             out[p][c]  = z = sigmoid(acc[c] + terms[e][c]),             c < 128
 *             out2[p][c] = sigmoid(acc[128+c] + terms[e][128+c]) * net[p][c]
 * epilogue 2: GRU blend, nout = 128 (gru.py:31-33):
 *             out[p][c] = (1 - z[p][c]) * net[p][c] + z[p][c] * tanh(acc[c] + terms[e][c])
 * epilogue 3: global-context reduction (gru.py:25-26), nout = 128, taps = 1: nothing is stored per pixel; every map is
 *             tiled on its own (ceil(H*W / 128) tiles of 128 pixels) and out = float [N][tiles per map][128] receives, per
 *             tile and channel, the sum of sigmoid(acc[c] + terms[c]) * net[p][c] over the tile's pixels;
 *             glorie_gru_glo_from_tiles turns them into the gate terms.
 * e = p / (H*W) is the map (edge) index; terms rows are terms_stride floats apart (gates) and come
 * from glorie_gru_glo_terms.  out / out2 / net / z are fp16 rows with their own strides (halfs).
 * pre (may be NULL; gate epilogues only): fp16 rows [p][nout] added to the accumulator before the
 * non-linearity, like terms but per pixel -- the convolution over the context features `inp` of an edge,
 * which never change while the edge lives (factor_graph.py:125-130: inp = video.inps[ii] at add_factors) and
 * is therefore evaluated once per edge instead of once per iteration (a convolution is linear in its input
 * channels: conv([net|inp|corr|flow]) = conv_dyn([net|corr|flow]) + conv_inp(inp)).
 * pre_map (may be NULL): int32 [N], map e reads the rows of map pre_map[e] of `pre` - edges with the same source
 * keyframe have the same context features (inp = video.inps[ii]), so the term is stored once per keyframe.
 * `epilogue` carries two more fields above its low byte (round 4):
 *   bit 8  (GLORIE_CONV_PAIR16): the rows of w_packed are PAIRED - within every group of 32 output channels, row 16 blk + r
 *          holds channel 8 (r / 4) + 4 blk + r % 4 - so that a lane of the kernel owns 8 consecutive channels of its pixel and
 *          the epilogue moves 16-byte pieces (epilogues 0-2, nout % 32 == 0; update_ops.pack_conv_igemm(pair=True)).  Same
 *          products and sums: the output is bit-identical to the unpaired packing's.
 *   bits 12-15: tile policy forced by the caller, 0 = automatic (what every product call passes); 1 = 128 x 128 tiles,
 *          2 = 64-pixel tiles, 3 = whole rounds of 128-pixel tiles + the remainder as 64-pixel tiles, 4 = 128 x 256,
 *          5 = conv_igemm_kernel's own choice (no haloed pixel tile, no ping-pong tile), 6 = the 256-channel x 256-pixel
 *          ping-pong tile of round 5 (conv_pp_kernel: 3x3 / 1x1 layers with nout % 256 == 0, epilogues 0 and 1; the automatic
 *          policy takes it for such 3x3 layers on maps of >= 65,536 pixels), 7 = the 128-channel x 512-pixel ping-pong tile of
 *          round 6 (conv_ppw_kernel: nout % 128 == 0, epilogues 0 and 2, and the heads entry point; the automatic policy takes
 *          it for the q gate and the heads on maps of >= 262,144 pixels - a 55-edge graph at 60 x 80 - where it is 4-5 %
 *          faster; on G8 it equals the shipped tiles; GLORIE_CONV_PPW = 0 / 1 overrides).  Tests (bit-identity of the
 *          variants) and tools/bench_conv.py. */
#define GLORIE_CONV_PAIR16 0x100
int glorie_conv_igemm(const void* xa, int xa_stride, int ca, const void* xb, int xb_stride, int cb,
                      const void* w_packed, int taps, int nout, int epilogue, const float* terms,
                      int terms_stride, int act, const void* net, int net_stride, const void* z,
                      int z_stride, void* out, int out_stride, void* out2, int out2_stride,
                      const void* pre, int pre_stride, const int* pre_map, int N, int H, int W, void* stream);

/* The hidden layers of the update operator's heads and their tap GEMMs in one launch (droid_net.py:85-93 delta / weight
 * heads: conv3x3 -> ReLU -> conv3x3 128 -> K; droid_net.py:38 the first convolution of GraphAgg shares the input).
 * A 3x3 / 1x1 convolution x [N*H*W][c] -> nout channels (nout % 128 == 0) with bias + ReLU as glorie_conv_igemm's
 * epilogue 0, except that the first `groups` 128-channel slices of its output are NOT stored: each is the input of a 3x3
 * head with K (<= 3) output channels, whose tap planes  tap_out[grp*9K + d*K + j][p] = < w2[grp][j][:, d], hidden[p][grp] >
 * (the `taps` workspace of glorie_conv3x3_small transposed: float [groups*9K][N*H*W]) are formed from the accumulators;
 * glorie_conv_stencil finishes the heads.  Channels 128*groups .. nout-1 are stored to out (fp16 rows of nout - 128*groups
 * channels, out_stride halfs apart; may be NULL when there are none).
 * tap_w: fp16 MFMA A fragments [groups][2][2][2][64][8]: element [grp][half][chunk][rb][lane][s] = w2[grp][j][ch][d] for
 * tap row 16rb + (lane&15) = d*K + j (zero beyond 9K) and hidden channel ch = 64half + 16(2chunk + s/4) + 4(lane>>4) + s%4
 * - the order in which the convolution's accumulator registers hold the hidden channels. */
int glorie_conv_igemm_heads(const void* x, int x_stride, int c, const void* w_packed, int taps, int nout,
                            const float* bias, const void* tap_w, int groups, int K, float* tap_out, void* out,
                            int out_stride, int N, int H, int W, void* stream);
/* GraphAgg's upmask convolution (droid_net.py:46-48: 1x1, c -> 576 = 9 taps x 8 x 8 sub-pixels, bias) and the convex
 * upsampling of depth_video.py:140-144 / droid_net.py:9-23 in one launch:
 *   disps_up[ix[m]] = cvx_upsample(disps[ix[m]], conv1x1(x[m]) + bias),   m < M,
 * the logits rounded to fp16 as the stored map of glorie_conv_igemm + glorie_cvx_upsample_nhwc would be, and the same
 * arithmetic in the same order: same bits, without writing and re-reading M*H*W*576 halfs.
 * x: fp16 rows [M*H*W][c] (x_stride halfs apart).  w_packed: glorie_conv_igemm's 1x1 operand for 1024 output rows,
 * row a*128 + b*16 + t = the convolution's output channel t*64 + a*8 + b for tap t < 9 (sub-row a, sub-column b), zero
 * rows for t >= 9; bias [1024] in the same order.  disps [B][H][W], disps_up [B][8H][8W] float, ix int64 [M]. */
int glorie_conv_upsample(const void* x, int x_stride, int c, const void* w_packed, const float* bias,
                         const float* disps, const int64_t* ix, float* disps_up, int softmax_f32, int M, int H, int W,
                         void* stream);
/* second half of glorie_conv3x3_small on the tap planes of glorie_conv_igemm_heads: out float [groups][N*H*W][K];
 * out_last (may be NULL): the LAST group is written there ([N*H*W][K]) instead of into its slice of out. */
int glorie_conv_stencil(const float* taps, const float* out_bias, int groups, int K, int act_packed, float scale,
                        float* out, float* out_last, int N, int H, int W, void* stream);

/* flow_encoder[0] (droid_net.py:79-81): 7x7 convolution, zero padding 3, 4 -> 128 channels, + bias
 * + ReLU.  flow: float32 channels-last motion map [N*H*W][4]; out: fp16 rows of 128 channels,
 * out_stride halfs apart.  w_packed: fp16 [128][224], column ky*32 + kx*4 + c = weight[n][c][ky][kx],
 * the 8th tap of every stencil row (kx = 7) zero. */
int glorie_flow_conv7(const float* flow, const void* w_packed, const float* bias, void* out,
                      int out_stride, int N, int H, int W, void* stream);

/* The same layer on the motion map as ZERO-PADDED fp16: padded = halfs [N][H+6][W+8][4], map pixel (y, x) at row y + 3,
 * pixel x + 3, every other element zero (3 rows above / below each map, 3 pixels left, 5 right).  A stencil row's taps are
 * then the MFMA fragment as loaded (no conversion, no boundary selects): 43 -> ~20 us at 36x60x80.  The flow encoder rounds its
 * input to fp16 in both forms: same results.  glorie_flow_pad fills the interior from the fp32 map [N][H][W][4] (the
 * borders are never written: zero them once), glorie_motion_padded (below) writes the motion features there directly.
 * out_stride % 8 == 0. */
int glorie_flow_pad(const float* flow, void* padded, int N, int H, int W, void* stream);
int glorie_flow_conv7_padded(const void* padded, const void* w_packed, const float* bias, void* out, int out_stride,
                             int N, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------ */
/* A/B. projective geometry                                                              */
/* ------------------------------------------------------------------------------------ */

/* pops.projective_transform(jacobian=False) via DepthVideo.reproject
 *   reference: src/geom/projective_ops.py:96-125, src/depth_video.py:156-164
 * poses [B,7], disps [B,h,w], intrinsics [B,4], ii/jj [N] int64.
 * coords [N,h,w,2], valid [N,h,w] (1.0/0.0).  MIN_DEPTH 0.2, Z<0.1 -> 1 (python path). */
int glorie_reproject(const float* poses, const float* disps, const float* intrinsics,
                     const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                     int N, int h, int w, void* stream);

/* droid_backends.frame_distance(poses, disps, intrinsics, ii, jj, beta)
 *   reference: src/lib/droid.cpp:122-138, src/lib/droid_kernels.cu:518-657,1441-1463 */
int glorie_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                          const int64_t* ii, const int64_t* jj, float* dist,
                          int K, int h, int w, float beta, void* stream);

/* droid_backends.iproj(poses, disps, intrinsics)
 *   reference: src/lib/droid.cpp:161-169, src/lib/droid_kernels.cu:779-850,1521-1544
 * points [num,h,w,3] */
int glorie_iproj(const float* poses, const float* disps, const float* intrinsics,
                 float* points, int num, int h, int w, void* stream);

/* droid_backends.depth_filter(poses, disps, intrinsics, ix, thresh)
 *   reference: src/lib/droid.cpp:208-224, src/lib/droid_kernels.cu:661-775,1494-1518
 * disps [B,h,w]; ix [num] int64; thresh [num]; count [num,h,w] (overwritten). */
int glorie_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                        const int64_t* ix, const float* thresh, float* count,
                        int B, int num, int h, int w, void* stream);

/* cvx_upsample(disps[ix,...,None], mask) via DepthVideo.upsample
 *   reference: src/modules/droid_net/droid_net.py:9-23, src/depth_video.py:140-144
 * disps [B,h,w] f32, ix [M] int64 (rows to read and rows of disps_up to write),
 * mask [M,576,h,w] (dtype f16|f32), disps_up [B,8h,8w] f32.
 * softmax_f32: 0 = round the softmax weights to the mask dtype (what torch.softmax does on
 * an fp16 mask outside autocast, factor_graph.py:231-254); 1 = keep them in fp32 (softmax
 * under autocast, factor_graph.py:286-291). */
int glorie_cvx_upsample(const float* disps, const int64_t* ix, const void* mask,
                        float* disps_up, int M, int h, int w, int mask_dtype,
                        int softmax_f32, void* stream);
/* motion features of the update operator (factor_graph.py:219-221):
 * out[n][p] = clamp([coords1 - coords0, target - coords1], -limit, limit) as float4 per pixel
 * (coords1, target [N,h,w,2]; coords0 [h,w,2]; out [N,h,w,4] = a channels-last 4-channel map) */
int glorie_motion(const float* coords1, const float* coords0, const float* target, float* out, int N,
                  int h, int w, float limit, void* stream);
/* glorie_motion with the zero-padded fp16 map of glorie_flow_conv7_padded as its output (interior only) */
int glorie_motion_padded(const float* coords1, const float* coords0, const float* target, void* padded, int N, int h,
                         int w, float limit, void* stream);
/* glorie_reproject and glorie_motion_padded in one launch (coords0 of factor_graph.py:219 is the pixel grid itself):
 * coords / valid as glorie_reproject, padded_motion as glorie_motion_padded for target [N][h][w][2]. */
int glorie_reproject_motion(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                            const int64_t* jj, float* coords, float* valid, const float* target, void* padded_motion,
                            int N, int h, int w, float limit, void* stream);

/* same operator, mask given channels-last in fp16: row (m*h*w + pixel) holds the 576 logits,
 * rows `mask_stride` halfs apart -- the layout the 1x1 upmask convolution produces */
int glorie_cvx_upsample_nhwc(const float* disps, const int64_t* ix, const void* mask, int mask_stride,
                             float* disps_up, int M, int h, int w, int softmax_f32, void* stream);

/* ------------------------------------------------------------------------------------ */
/* B. dense bundle adjustment                                                            */
/* ------------------------------------------------------------------------------------ */

/* droid_backends.ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
 *                   t0, t1, iterations, lm, ep, motion_only, depth_only)
 *   reference: src/lib/droid.cpp:89-119,241; src/lib/droid_kernels.cu:1314-1437
 * poses [B,7] and disps [B,h,w] are updated IN PLACE.  targets/weights [N,2,h,w],
 * eta [M,h,w] with M = #unique(cat(arange(t0,t1), ii)) (sorted order), disps_sens may be
 * NULL (= all zeros, which is what the reference passes, depth_video.py:217).
 * dx_out [t1-t0,6] / dz_out [M,h*w] receive the last iteration's updates (may be NULL).
 * Whole Gauss-Newton loop runs on the device with no host synchronisation.
 * `motion_only` is a flag word: bit 0 = the reference's motion_only; GLORIE_BA_TARGETS_HWC set =
 * targets/weights are given as [N,h,w,2] - the layout FactorGraph.target / .weight have before the
 * permute(0,3,1,2).contiguous() of depth_video.py:215-216 - which saves those two copies per call.
 * A caller that passes the reference's bool gets the reference's layout. */
#define GLORIE_BA_TARGETS_HWC 2
int glorie_ba(glorie_ctx* ctx, float* poses, float* disps, const float* intrinsics,
              const float* disps_sens, const float* targets, const float* weights,
              const float* eta, const int64_t* ii, const int64_t* jj,
              int B, int N, int M, int h, int w, int t0, int t1, int iterations,
              float lm, float ep, int motion_only, int depth_only,
              float* dx_out, float* dz_out, void* stream);

/* Multi-GPU form of glorie_ba: one Gauss-Newton iteration split at its
 * single exchange step.  Edges are partitioned by SOURCE keyframe, so every per-edge term, the
 * depth blocks C_k / w_k and every Schur product of frame k are local to the rank owning k;
 * only the dense reduced system [H (6P x 6P, lower block triangle) | v (6P)] must be summed.
 *   glorie_ba_build_system : local edges -> hv_out, (6P*6P + 6P) doubles, caller-owned
 *   <caller: all-reduce(sum) of hv over RCCL>
 *   glorie_ba_solve_update : damping + fp64 Cholesky of hv, pose retraction (replicated,
 *                            identical on all ranks) and dz for the local depth frames.
 * Both calls take the SAME local edge set / sizes on the same ctx, back to back; M counts
 * unique(cat(arange(t0,t1), ii_local)) and eta holds those frames' rows.  A rank without
 * edges passes N = 0.  With one rank the pair is equivalent to one iteration of glorie_ba. */
int glorie_ba_build_system(glorie_ctx* ctx, const float* poses, const float* disps,
                           const float* intrinsics, const float* disps_sens,
                           const float* targets, const float* weights, const float* eta,
                           const int64_t* ii, const int64_t* jj, int B, int N, int M, int h, int w,
                           int t0, int t1, int motion_only, double* hv_out, void* stream);
int glorie_ba_solve_update(glorie_ctx* ctx, float* poses, float* disps, const int64_t* ii,
                           const int64_t* jj, int B, int N, int M, int h, int w, int t0, int t1,
                           float lm, float ep, int motion_only, int depth_only, const double* hv,
                           float* dx_out, float* dz_out, void* stream);

/* Device-side gate for the bundle-adjustment entry points called next on `ctx` (sticky until cleared with run_if_zero = NULL):
 * every kernel of glorie_ba / glorie_ba_build_system / glorie_ba_solve_update returns at once unless *run_if_zero == 0 - poses,
 * disparities and the status word stay untouched.  It carries the reference's host decision `if not success: run pose_depth`
 * (src/depth_video.py:290-294) onto the device: the stage-1 fallback of a depth_scale stage is enqueued behind the stage,
 * gated on the stage's own "any edge left" word (glorie_dspo_prepare's any_on), so a BA-update step needs no host round trip
 * and replays as one hipGraph.  hits (may be NULL) is incremented once per gated call that did run.
 * The gate is state of the CONTEXT, not of a call: a context with a gate set must not be shared with another thread's BA calls
 * between the set and the clear (use one glorie_ctx per thread of BA callers). */
int glorie_ba_set_gate(glorie_ctx* ctx, const int* run_if_zero, int* hits);

/* The exchange step itself (glorie_allreduce_normal_eq(ctx, S_v_buf, n)).  The context owns an RCCL
 * communicator: rank 0 draws an id (glorie_comm_unique_id, GLORIE_COMM_ID_BYTES bytes), hands it to every rank by whatever
 * channel the host has (the Python side broadcasts it through torch.distributed's store), and every rank calls
 * glorie_comm_init(ctx, id, rank, world) once - collective, like ncclCommInitRank.  glorie_allreduce_normal_eq sums `n`
 * doubles in place over the ranks on `stream` (the packed or the full [H | v]); glorie_allgather_rows gathers
 * bytes_per_rank bytes from every rank into recv (rank-major) - the owned disparity rows after an iteration.  Both are plain
 * stream work: they may be recorded into a hipGraph together with the launches around them.  RCCL (librccl.so.1) is bound
 * at first use; GLORIE_EUNSUPPORTED if it cannot be loaded, GLORIE_EINVAL if the context has no communicator.
 * glorie_comm_world: ranks of the context's communicator, 0 without one.  The reference is single-GPU: no counterpart. */
#define GLORIE_COMM_ID_BYTES 128
int glorie_comm_unique_id(void* id_out);
int glorie_comm_init(glorie_ctx* ctx, const void* id, int rank, int world);
int glorie_comm_destroy(glorie_ctx* ctx);
int glorie_comm_world(const glorie_ctx* ctx);
int glorie_allreduce_normal_eq(glorie_ctx* ctx, double* hv, size_t n, void* stream);
int glorie_allgather_rows(glorie_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank, void* stream);

/* Exchange format of the reduced system: the solve reads the lower triangle of H only, so ranks all-reduce
 * n6*(n6+1)/2 + n6 doubles (row r: columns 0..r, then v) instead of n6*n6 + n6 - 13 MB instead of 26 MB at
 * P = 300.  unpack = 0: hv -> packed; unpack = 1: packed -> the lower triangle and v of hv (the rest of hv is
 * left as it is). */
int glorie_ba_pack_system(const double* hv, double* packed, int n6, int unpack, void* stream);

/* DSPO stage 2, `BA_with_scale_shift(target, weight, eta, poses, disps, intrinsics, ii, jj,
 *                mono_disps, scales, shifts, valid_depth_mask, ignore_frames=0, lm, ep, alpha)`
 *   reference: src/geom/ba.py:127-216, src/geom/chol.py:58-85, call site
 *   src/depth_video.py:262-276 (alpha = 0.01, `itrs` repetitions).
 * Optimises disparities + per-frame scale/shift of the mono prior with poses fixed.
 * disps [B,h,w], scales/shifts [B] are updated IN PLACE (the reference rebinds
 * them).  target/weight are in the python layout [N,h,w,2]; intrinsics [B,4];
 * valid_mask [B,h,w] uint8 (valid_depth_mask_small); eta [M,h,w], M = #unique(ii) in sorted
 * order.  edge_on [N] uint8 (may be NULL = all on) replaces the reference's edge filtering by
 * boolean-mask copies (depth_video.py:228-261); frames whose edges are all off are untouched
 * and their eta rows ignored.  dz_out [M,h*w] may be NULL. */
int glorie_dspo_scale_shift(glorie_ctx* ctx, const float* poses, float* disps,
                            const float* intrinsics, const float* mono_disps, float* scales,
                            float* shifts, const uint8_t* valid_mask, const float* target,
                            const float* weight, const float* eta, const int64_t* ii,
                            const int64_t* jj, const uint8_t* edge_on, int B, int N, int M, int h,
                            int w, int iterations, float lm, float ep, float alpha, float* dz_out,
                            void* stream);

/* Preparation of the depth_scale stage in 4 launches (depth_video.py:228-247,326-361,
 * common.py:401-437): two-view validity mask of frames [0,n) at BA resolution (threshold
 * mv_thresh * mean depth, >= visible_num consistent neighbours, depth < 3 * nanmedian), per-frame
 * least-squares scale/shift of mono_disps onto disps under that mask, and the mono_thres filter:
 * bad(f) = err/mean(disp) > mono_thres | isnan(err) | scale < 0 | valid pixels < HW/2;
 * edge_on[e] = !(bad[ii[e]] | bad[jj[e]]); *any_on = any(edge_on).  mono_thres <= 0 disables the
 * filter (all edges on).  Outputs: valid_mask [n,h,w] bytes, scales/shifts [n], edge_on [N] bytes,
 * any_on device int.  scratch: >= n*h*w*4 + n*32 + 64 bytes.  No host synchronisation.
 * publish_state / publish_host_word (both or neither; may be NULL): the last launch also does what glorie_publish_flag
 * does with counter = &publish_state[0] (publish_state = device int[2], zero-initialised: launch count, arrivals). */
int glorie_dspo_prepare(const float* poses, const float* disps, const float* intrinsics,
                        const float* mono_disps, int B, int n, int h, int w, float mv_thresh,
                        int visible_num, float mono_thres, const int64_t* ii, const int64_t* jj, int N,
                        uint8_t* valid_mask, float* scales, float* shifts, uint8_t* edge_on,
                        int* any_on, void* scratch, int* publish_state, int* publish_host_word, void* stream);

/* *host_word = (++*counter << 1) | (*flag != 0), stored by the device into pinned (device-mapped) host memory: a host
 * that counts its launches can poll the word for the flag of a given launch instead of synchronising the stream
 * (the stage-1 fallback decision of a depth_scale stage replayed from a hipGraph, depth_video.py:290-294). */
int glorie_publish_flag(const int* flag, int* counter, int* host_word, void* stream);

/* Two-view validity mask of the frames ix[0..num) of a depth video at any resolution
 * (DepthVideo.update_valid_depth_mask, depth_video.py:326-361):
 * thresh = mv_thresh * mean(1/disp); a pixel survives if >= visible_num of its 6 neighbour frames
 * agree within thresh and its depth is below 3 x the (lower) median of the surviving depths.
 * disps [B,h,w] is the map the mask is computed on (disps_up with intrinsics scaled by 8, or disps),
 * mask [num,h,w] bytes.  scratch: >= num*h*w*4 + num*(4 + 16 + 1024) bytes.  No host sync. */
int glorie_valid_depth_mask(const float* poses, const float* disps, const float* intrinsics,
                            const int64_t* ix, int B, int num, int h, int w, float mv_thresh,
                            int visible_num, uint8_t* mask, void* scratch, void* stream);

/* ------------------------------------------------------------------------------------ */
/* C. neural point cloud renderer                                                        */
/* ------------------------------------------------------------------------------------ */

/* Search structure replacing the faiss index of NeuralPointCloud
 *   reference: src/neural_point.py:56-60 (IndexIVFFlat construction), :104-116
 *   (index_train / index_reset / index_add), :254-259, :441-444 (re-train + add).
 * Builds a uniform cell list over `points` [np,3] entirely on the device.
 * Caller-owned outputs: sorted_pos [np,4] f32 (x,y,z, original index bit-cast to f32),
 * cell_start [max_cells+1] int32, grid: 64 bytes (origin, cell size, dims).
 * cell_size is a hint; it is enlarged on the device until the grid fits max_cells (< 2^22). */
int glorie_knn_build(glorie_ctx* ctx, const float* points, int np, float cell_size,
                     int max_cells, float* sorted_pos, int* cell_start, void* grid,
                     void* stream);

/* NeuralPointCloud.find_neighbors_faiss(pos, step, ..., dynamic_radius) -> (D, I, neighbor_num)
 *   reference: src/neural_point.py:264-313 (faiss IndexIVFFlat.search, k = nn_num = 8)
 * EXACT squared-L2 top-k ordered by (distance, index) -- the faiss index is approximate and
 * not reproducible, see DESIGN.md.  queries [Q,3]; radius scalar, or radius_ptr [Q] (per-query
 * `dynamic_radius`) when non-NULL.  D [Q,k] f32 ascending, I [Q,k] int64 (-1 / FLT_MAX when
 * fewer than k points exist, like faiss), nn [Q] int32 = #(D < r^2) (may be NULL). k in {1,4,8,16} */
int glorie_knn_query(const float* sorted_pos, const int* cell_start, const void* grid,
                     const float* queries, int Q, int k, float radius, const float* radius_ptr,
                     float* D, int64_t* I, int* nn, void* stream);

/* glorie_knn_query for the samples of an image strip: queries = the samples_per_ray samples of rays laid out row-major
 * with image_w rays per row (query = (y * image_w + x) * samples_per_ray + s; what Renderer.render_img,
 * src/utils/Renderer.py:221-306, evaluates).  Same outputs in the same rows, bit for bit; a workgroup searches one
 * depth of a 16 x 16 pixel patch instead of 256 consecutive queries, so its lanes share cells and point ranges.
 * The last row may be partial (Q / samples_per_ray need not be a multiple of image_w). */
int glorie_knn_query_image(const float* sorted_pos, const int* cell_start, const void* grid,
                           const float* queries, int Q, int k, float radius, const float* radius_ptr,
                           float* D, int64_t* I, int* nn, int samples_per_ray, int image_w, void* stream);

/* glorie_knn_query (image_w == 0) / glorie_knn_query_image (image_w > 0) for k = 8 with the inverse-distance weights and the
 * neighbour mask of get_feature_at_pos (reference: src/modules/conv_onet/models/decoder.py:130-173) produced by the same
 * launch: weights [Q,8] = [I >= 0 and D <= r^2] / (D + 1e-10) (or exp(-20 sqrt(D)) with expo_weighting), L1-normalised
 * (eps 1e-12); has [Q] = neighbour count (D < r^2) >= min_nn.  Bit-identical to glorie_idw_gather's weights / mask.
 * ball_only != 0: the search is bounded by the query's radius - everything the decoders use (weights, has, nn, and every
 * (D, I) slot with D <= r^2) is unchanged, slots BEYOND the radius (weight 0) hold whatever the bounded search had seen:
 * farther points in (distance, index) order among those seen, or (FLT_MAX, -1).  A sample with fewer than 8 points in
 * its ball then stops at the ball instead of walking outward until it has found 8. */
int glorie_knn_query_weights(const float* sorted_pos, const int* cell_start, const void* grid,
                             const float* queries, int Q, float radius, const float* radius_ptr, float* D,
                             int64_t* I, int* nn, int samples_per_ray, int image_w, int min_nn,
                             int expo_weighting, int ball_only, float* weights, uint8_t* has, void* stream);

/* Feature interpolation of MLP_geometry/MLP_color.get_feature_at_pos
 *   reference: src/modules/conv_onet/models/decoder.py:130-173 (geometry), :340-389 (colour)
 * w = L1-normalise( [D <= r^2] / (D + 1e-10) )  (or exp(-20 sqrt(D)) when expo_weighting),
 * c[q] = sum_k w_k feats[I_k]; has[q] = nn[q] >= min_nn.  Samples without enough neighbours
 * get c = 0 (the reference draws N(0,0.01) noise there, decoder.py:170-171).
 * c_out [Q,c_dim], w_out [Q,k] (may be NULL), has_out [Q] uint8 (may be NULL). k=8, c_dim=32. */
int glorie_idw_gather(const float* D, const int64_t* I, const int* nn, const float* feats,
                      int Q, int k, int c_dim, float radius, const float* radius_ptr,
                      int min_nn, int expo_weighting, float* c_out, float* w_out,
                      uint8_t* has_out, void* stream);

/* glorie_idw_gather on the geometry and the colour table in one pass (decoder.py:130-173 and :340-389 share the
 * neighbours and the weights): c_out_a = sum_k w_k feats_a[I_k], c_out_b likewise.  Same bits as two calls. */
int glorie_idw_gather2(const float* D, const int64_t* I, const int* nn, const float* feats_a,
                       const float* feats_b, int Q, int k, int c_dim, float radius,
                       const float* radius_ptr, int min_nn, int expo_weighting, float* c_out_a,
                       float* c_out_b, float* w_out, uint8_t* has_out, void* stream);

/* Fused decoders: POINT.forward(p, npc, stage, ...) minus the neighbour search
 *   reference: src/modules/conv_onet/models/decoder.py:175-225 (MLP_geometry.forward),
 *   :340-389 + :228-243 (per-neighbour F_theta + IDW sum), :391-433 (MLP_color.forward),
 *   :460-501 (POINT.forward), src/utils/Renderer.py:206-207 (occupancy -100 without neighbours)
 * packed: glorie_decoder_pack_floats() floats, layout in csrc/mlp.hip (built by
 * glorie_slam_amd.point_ops.pack_decoders from the state dict).  pts/views [Q,3]; cloud_pos
 * [Np,3]; col_feats [Np,32]; c_geo [Q,32] + weights [Q,8] + has [Q] from glorie_idw_gather;
 * I [Q,8] from glorie_knn_query; c_col_scratch [Q,32]; raw [Q,4] = (r,g,b,occ), rgb written
 * only when (stage_flags & 1) (caller zero-fills otherwise).
 * Arithmetic: fp32-accurate.  By default the matmuls run on the fp16 matrix cores as 3-term hi/lo splits with fp32
 * accumulation (22 mantissa bits; csrc/mlp.hip); stage_flags & 2 (or GLORIE_MLP_F32=1) selects the exact-fp32 MFMA 16x16x4
 * kernels.  The split needs every operand within the fp16 range: range_flag (NULL or a device int, zero-initialised by the
 * caller) gets bit 0 set when a kernel met |x| > 65504 or a NaN - the results of that call are then invalid and the caller
 * repeats it with stage_flags | 2 (glorie_slam_amd.renderer does).
 * stage_flags & 4 (measurement, bench.py's roofline_knn): only the feature-pull phases of the geometry and per-neighbour
 * kernels run - ids, weights, the 8 neighbour rows, positions and the interpolation, the networks removed - so the R2 traffic
 * can be timed the way the product performs it (inside these kernels); needs geo_feats (c_geo = NULL), not combinable with
 * bit 1; `raw` / c_col_scratch then hold sums of the gathered values, not decoder outputs.
 * c_geo [Q,32] is the IDW-interpolated geometry feature (glorie_idw_gather); pass NULL and the feature table
 * geo_feats [Np,32] instead to have the geometry kernel interpolate it itself from (I, weights) - then
 * glorie_idw_gather is only needed for the weights and the mask (c_out = NULL). */
size_t glorie_decoder_pack_floats(void);
int glorie_render_mlp(const float* packed, const float* pts, const float* views,
                      const float* cloud_pos, const float* col_feats, const float* c_geo,
                      const float* geo_feats, const int64_t* I, const float* weights, const uint8_t* has,
                      int Q, float* c_col_scratch, float* raw, int stage_flags, int* range_flag, void* stream);

/* Sample placement of Renderer.render_batch_ray for rays with a depth prior
 *   reference: src/utils/Renderer.py:106-125 (z_vals), 177-179 (pts, per-sample view directions),
 *   183-184 (per-sample query radius)
 * z[r][s] = near_s*d*(1 - t[s]) + far_s*d*t[s] (t_lin = linspace(0,1,S), [S] device floats; each op rounded
 * separately like the torch chain), pts = o + dir*z [R*S,3], views [R*S,3] = dir, radius_s [R*S] = radius[r]
 * (both NULL or both given).  *n_zero (device int, accumulated) counts rays with d <= 0: those get z = 0 and
 * need the reference's sample_near_pcl path (Renderer.py:127-174), which the host handles. */
int glorie_ray_samples(const float* rays_o, const float* rays_d, const float* depth,
                       const float* radius, const float* t_lin, int R, int S, float near_s, float far_s,
                       float* z_vals, float* pts, float* views, float* radius_s, int* n_zero,
                       void* stream);

/* glorie_ray_samples for the rays of R consecutive row-major pixels of a pinhole view, starting at pixel first_pixel, formed
 * in the kernel instead of read: get_rays (reference: src/utils/common.py:302-322, OpenGL convention) fused into the sample
 * placement (scope rows R7 + R4; Renderer.render_img, src/utils/Renderer.py:221-306).
 * cam: 16 device floats = c2w rows 0..2 (3 x 4, row-major), 1/fx, 1/fy (the double reciprocal rounded to fp32: how torch divides a
 * tensor by a Python scalar), cx, cy.  Same bits as get_rays + glorie_ray_samples. */
int glorie_ray_samples_camera(const float* cam, int image_w, long first_pixel, const float* depth, const float* radius,
                              const float* t_lin, int R, int S, float near_s, float far_s, float* z_vals, float* pts,
                              float* views, float* radius_s, int* n_zero, void* stream);

/* proj_depth_map(c2w, npc, ...): z-buffer projection of a point set into a view
 *   reference: src/neural_point.py:446-506
 * points [n,3] world coordinates, mask [n] uint8 (NULL = all points), w2c = inverse of the camera-to-world
 * matrix, row-major [>= 12 floats] on the device; camera looks along -z, x flipped before the projection,
 * pixel = trunc(u, v), depth = -z.  depth_inf [H,W] must hold +inf on entry; on return every pixel hit by a
 * point holds the smallest depth, the rest still +inf (the caller writes 0 there). */
int glorie_proj_depth(const float* points, const uint8_t* mask, long n, const float* w2c, float fx, float fy,
                      float cx, float cy, int H, int W, float* depth_inf, void* stream);

/* Per-ray count of samples with neighbours and the valid-ray flag (count >= min_samples)
 *   reference: src/modules/conv_onet/models/decoder.py:202-204 (ray_counter, ~(counter < 3))
 * has [R*S] uint8 -> counts [R] int64, valid [R] uint8. */
int glorie_ray_counts(const uint8_t* has, int R, int S, int min_samples, int64_t* counts,
                      uint8_t* valid, void* stream);

/* raw2outputs_nerf_color(raw, z_vals, rays_d, coef)
 *   reference: src/utils/common.py:261-299
 * raw [R,S,4] (rgb, occupancy), z_vals [R,S] -> depth [R], var [R], rgb [R,3],
 * weights [R,S] (may be NULL). */
int glorie_composite(const float* raw, const float* z_vals, int R, int S, float coef,
                     float* depth, float* var, float* rgb, float* weights, void* stream);

/* Diagnostic: waits for `stream`, then copies the 4-int device status of the last glorie_ba
 * call on this context to host memory: [0] bit0 = M disagrees with the device-side count,
 * bit2 = a Cholesky factorisation failed (update zeroed, as the reference's
 * `solver.info() != Success` branch, droid_kernels.cu:1202-1210); [1] = M seen on the
 * device; [2] = number of failed factorisations. */
int glorie_ba_status(glorie_ctx* ctx, int* status_out, void* stream);

/* ------------------------------------------------------------------------------------ */
/* N1. training path of the renderer (forward with saved activations, backward, Adam)   */
/* ------------------------------------------------------------------------------------ */

/* Parameters of the decoders `POINT` (src/modules/conv_onet/models/decoder.py:436-501) as the tensors of its
 * state dict, fp32, in torch's own layouts (nn.Linear weight = [out][in]; no packing, no copies):
 *   geometry decoder (hidden 32): g_B = embedder._B [3][93]; g_W/g_b = pts_linears[i] ([32][93], [32][32],
 *     [32][32], [32][125], [32][32]); g_U/g_u = fc_c[i] ([32][32]); g_Wo/g_bo = output_linear ([1][32])
 *   per-neighbour F_theta of the colour decoder: n_B = embedder_rel_pos._B [3][10]; n_W1 [128][52], n_W2 [32][128]
 *   colour decoder (hidden 128): c_Bp / c_Bv = embedder._B / embedder_view_direction._B [3][20] (not learnable);
 *     c_W/c_b = pts_linears[i] ([128][80], [128][128], [128][128], [128][208], [128][128]); c_U/c_u = fc_c[i]
 *     ([128][32]); c_Wo/c_bo = output_linear ([3][128])
 * glorie_decoder_grads has the same fields; a NULL gradient pointer skips that gradient (c_Bp / c_Bv are ignored).
 * Gradients are ACCUMULATED (+=) like autograd's .grad. */
typedef struct glorie_decoder_params {
  const float* g_B; const float* g_W[5]; const float* g_b[5]; const float* g_U[5]; const float* g_u[5];
  const float* g_Wo; const float* g_bo;
  const float* n_B; const float* n_W1; const float* n_b1; const float* n_W2; const float* n_b2;
  const float* c_Bp; const float* c_Bv; const float* c_W[5]; const float* c_b[5]; const float* c_U[5];
  const float* c_u[5]; const float* c_Wo; const float* c_bo;
} glorie_decoder_params;
typedef struct glorie_decoder_grads {
  float* g_B; float* g_W[5]; float* g_b[5]; float* g_U[5]; float* g_u[5];
  float* g_Wo; float* g_bo;
  float* n_B; float* n_W1; float* n_b1; float* n_W2; float* n_b2;
  float* c_Bp; float* c_Bv; float* c_W[5]; float* c_b[5]; float* c_U[5];
  float* c_u[5]; float* c_Wo; float* c_bo;
} glorie_decoder_grads;

/* bytes of the activation workspace for Q samples (saved forward activations + backward temporaries) */
size_t glorie_render_train_workspace(long Q);

/* POINT.forward for a training batch: raw [Q,4] = (rgb, occ; occ = -100 where has == 0), every layer's output kept
 * in `workspace` for the backward pass
 *   reference: decoder.py:175-225 (MLP_geometry.forward), :340-389 + :228-243 (get_feature_at_pos + F_theta),
 *   :391-433 (MLP_color.forward), src/utils/Renderer.py:206-207
 * pts / views [Q,3]; cloud_pos [Np,3]; geo_feats / col_feats [Np,32]; I [Q,8] int64, w [Q,8], has [Q] uint8 from
 * glorie_knn_query + glorie_idw_gather.  stage_color = 0: geometry stage (rgb = 0; views, cloud_pos, col_feats unused). */
int glorie_render_train_fwd(const glorie_decoder_params* params, const float* pts, const float* views,
                            const float* cloud_pos, const float* geo_feats, const float* col_feats,
                            const int64_t* I, const float* w, const uint8_t* has, long Q, int stage_color,
                            float* workspace, float* raw, void* stream);

/* what loss.backward() computes below `raw` (src/mapper.py:511): d_raw [Q,4] -> parameter gradients (accumulated
 * into `grads`), d_geo_feats / d_col_feats [Np,32] (accumulated with fp32 atomics: rows are shared by samples).
 * Same arguments and the workspace of the matching glorie_render_train_fwd call.  Samples without neighbours pass
 * no gradient to the feature tables (their feature is a constant) but, like in the reference (the -100 is assigned
 * under no_grad, Renderer.py:206-207), the gradient of their occupancy reaches the geometry decoder. */
int glorie_render_train_bwd(const glorie_decoder_params* params, const glorie_decoder_grads* grads, const float* pts,
                            const float* views, const float* cloud_pos, const float* geo_feats,
                            const float* col_feats, const int64_t* I, const float* w, const uint8_t* has, long Q,
                            int stage_color, float* workspace, const float* d_raw, float* d_geo_feats,
                            float* d_col_feats, void* stream);

/* backward of raw2outputs_nerf_color (src/utils/common.py:261-299) for depth and colour:
 * g_depth [R], g_rgb [R,3] (either may be NULL = zero) -> d_raw [R,S,4].  S <= 32. */
int glorie_composite_bwd(const float* raw, const float* z_vals, int R, int S, float coef, const float* g_depth,
                         const float* g_rgb, float* d_raw, void* stream);

/* torch.optim.Adam.step() for one tensor (amsgrad off, no weight decay; src/mapper.py:612-624, :512):
 * step = 1-based step count of this tensor; row_mask [n / row_len] uint8 (may be NULL) restricts the update to the
 * marked rows of a [rows, row_len] table (the frustum-selected feature rows, mapper.py:586-611). */
int glorie_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                     float beta1, float beta2, float eps, int step, const uint8_t* row_mask, int row_len, void* stream);

/* glorie_adam_step for many small tensors that are at the same step, in one launch.  table: n_tensors entries of 80 bytes on
 * the device - { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64 numel; float lr, beta1, beta2, eps;
 * 24 bytes padding } (nothing step-dependent: it is re-sent only when a pointer moved); max_numel = the largest numel.
 * grad_base != NULL: the `grad` field of every entry is a BYTE OFFSET from grad_base instead of a pointer - the gradients of a
 * backward pass are views of one buffer that the allocator places anew every iteration; with offsets the table stays valid. */
int glorie_adam_multi(const void* table, int n_tensors, long max_numel, int step, const void* grad_base, void* stream);

/* The two Adam entry points with the step count read from DEVICE memory (*step_dev, 1-based, one word shared by the tensors of
 * an optimizer) and glorie_counter_add(counter, delta) to advance it on the stream: a mapping iteration - forward, loss,
 * backward, Adam - recorded into a hipGraph per keyframe (the loop of src/mapper.py:586-624 runs 150-1500 iterations on the
 * same tensors) must not bake a by-value step into the recording.  The bias corrections 1 - beta^step are formed on the device
 * (powf) as glorie_adam_multi does. */
int glorie_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                         float beta1, float beta2, float eps, const int* step_dev, const uint8_t* row_mask, int row_len,
                         void* stream);
int glorie_adam_multi_dev(const void* table, int n_tensors, long max_numel, const int* step_dev, const void* grad_base,
                          void* stream);
int glorie_counter_add(int* counter, int delta, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLORIE_HIP_H */
