"""Fused element-wise stages of the update operator (csrc/gru.hip) on channels-last fp16 maps.

Every tensor argument is a 4-D fp16 map [N, C, h, w] whose memory is channels-last
([pixel][channel]) -- either a whole `torch.channels_last` tensor or a CHANNEL SLICE of a wider
one (`buf[:, 128:256]`), which is how the `torch.cat([net, inp, corr, flow])` of the reference
(gru.py:22-23, droid_net.py:121-122) is replaced by writing producers into slices.
reference: src/modules/droid_net/gru.py:20-34, droid_net.py:106-139.
"""
import torch

from . import _lib as L

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SOFTPLUS = 0, 1, 2, 3


def _rows(t, name):
    """(row stride in halfs) of a channels-last map or channel slice thereof"""
    if t.dtype != torch.float16 or t.dim() != 4:
        raise RuntimeError(f"{name}: expected a 4-D float16 map, got {t.dtype} {tuple(t.shape)}")
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if c > 1 and sc != 1:
        raise RuntimeError(f"{name}: not channels-last (channel stride {sc})")
    if (w > 1 and sh != w * sw) or (h * w > 1 and n > 1 and sn != h * w * sw) or sw < c:
        raise RuntimeError(f"{name}: rows are not uniformly strided {tuple(t.stride())}")
    if sw % 8 or t.data_ptr() % 16:
        raise RuntimeError(f"{name}: rows must be 16-byte aligned")
    return sw


def bias_act(x, bias, act, out=None):
    """out = act(x + bias[c]); `out` may alias x (in place) or be a channel slice"""
    L.need_cuda(x)
    out = x if out is None else out
    xs, ys = _rows(x, "x"), _rows(out, "out")
    n, c, h, w = x.shape
    if tuple(out.shape) != tuple(x.shape):
        raise RuntimeError("bias_act: shape mismatch")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != c or not bias.is_contiguous()):
        raise RuntimeError("bias_act: bias must be contiguous float32 [C]")
    L.check(L.load().glorie_bias_act(L.ptr(x), xs, L.ptr(bias) if bias is not None else None, L.ptr(out),
                                     ys, n * h * w, c, act, L.stream_ptr()), "glorie_bias_act")
    return out


def _terms(g, width, name):
    if g.dtype != torch.float32 or g.dim() != 2 or g.shape[1] != width or g.stride(1) != 1:
        raise RuntimeError(f"{name}: expected float32 [N,{width}] rows (column slices allowed)")
    return g.stride(0)


def gru_glo_terms(wn, bw, net, G, Gb, parts=16):
    """glo[n] = mean_pixels sigmoid(wn + bw) * net (gru.py:25-26); returns g = glo @ G + Gb,
    float32 [N, M]: the convz_glo | convr_glo | convq_glo terms (G [128, M], biases in Gb)"""
    L.need_cuda(wn, net, G, Gb)
    n, c, h, w = net.shape
    if c != 128 or tuple(wn.shape) != tuple(net.shape) or G.shape[0] != 128 or not G.is_contiguous() \
            or G.dtype != torch.float32 or Gb.dtype != torch.float32 or Gb.numel() != G.shape[1]:
        raise RuntimeError("gru_glo_terms: bad shapes")
    M = G.shape[1]
    partial = torch.empty((n, parts, 128), dtype=torch.float32, device=net.device)
    g = torch.empty((n, M), dtype=torch.float32, device=net.device)
    L.check(L.load().glorie_gru_glo_terms(L.ptr(wn), _rows(wn, "wn"), L.ptr(bw), L.ptr(net), _rows(net, "net"),
                                          L.ptr(G), L.ptr(Gb), M, L.ptr(partial), parts, L.ptr(g), n, h * w,
                                          L.stream_ptr()), "glorie_gru_glo_terms")
    return g


def gru_glo_terms_fused(net, w_packed, bw, G, Gb):
    """gru_glo_terms with the 1x1 convolution w (gru.py:25) inside: conv_igemm's epilogue 3 reduces
    sigmoid(w(net) + bw) * net over 128-pixel tiles, glorie_gru_glo_from_tiles finishes per map - the intermediate
    [N,128,h,w] map never exists.  w_packed = pack_conv_igemm(gru.w.weight)."""
    L.need_cuda(net, w_packed, G, Gb)
    n, c, h, w = net.shape
    if c != 128 or G.shape[0] != 128 or not G.is_contiguous() or G.dtype != torch.float32 \
            or Gb.dtype != torch.float32 or Gb.numel() != G.shape[1]:
        raise RuntimeError("gru_glo_terms_fused: bad shapes")
    M = G.shape[1]
    tiles = torch.empty((n, (h * w + 127) // 128, 128), dtype=torch.float32, device=net.device)
    g = torch.empty((n, M), dtype=torch.float32, device=net.device)
    if n == 0:
        return g
    lib = L.load()
    L.check(lib.glorie_conv_igemm(L.ptr(net), _rows(net, "net"), 128, None, 0, 0, L.ptr(w_packed), 1, 128, EPI_GLO,
                                  L.ptr(bw), 0, ACT_NONE, L.ptr(net), _rows(net, "net"), None, 0, L.ptr(tiles), 128,
                                  None, 0, None, 0, None, n, h, w, L.stream_ptr()), "glorie_conv_igemm (epilogue 3)")
    L.check(lib.glorie_gru_glo_from_tiles(L.ptr(tiles), L.ptr(G), L.ptr(Gb), M, L.ptr(g), n, h * w, L.stream_ptr()),
            "glorie_gru_glo_from_tiles")
    return g


def gru_gate_zr(zr, g, net, z, rnet):
    """z = sigmoid(zr[:, :128] + g[n, :128]); rnet = sigmoid(zr[:, 128:] + g[n, 128:]) * net
    (gru.py:28-30).  zr: raw [N,256,h,w] output of the merged convz|convr; g float32 [N,256]"""
    L.need_cuda(zr, net, z, rnet, g)
    n, c, h, w = net.shape
    if c != 128 or zr.shape[1] != 256 or g.shape[0] != n:
        raise RuntimeError("gru_gate_zr: bad shapes")
    L.check(L.load().glorie_gru_gate_zr(L.ptr(zr), _rows(zr, "zr"), L.ptr(g), _terms(g, 256, "g"), L.ptr(net),
                                        _rows(net, "net"), L.ptr(z), _rows(z, "z"), L.ptr(rnet),
                                        _rows(rnet, "rnet"), n, h * w, L.stream_ptr()), "glorie_gru_gate_zr")


def gru_gate_q(qc, gq, z, net, out, out2=None):
    """out = (1 - z) * net + z * tanh(qc + gq[n])   (gru.py:31-33); gq float32 [N,128];
    out2: optional second destination (e.g. the net slice of the GRU input buffer)"""
    L.need_cuda(qc, z, net, out, gq)
    n, c, h, w = net.shape
    if c != 128 or gq.shape[0] != n:
        raise RuntimeError("gru_gate_q: bad shapes")
    L.check(L.load().glorie_gru_gate_q(L.ptr(qc), _rows(qc, "qc"), L.ptr(gq), _terms(gq, 128, "gq"), L.ptr(z),
                                       _rows(z, "z"), L.ptr(net), _rows(net, "net"), L.ptr(out), _rows(out, "out"),
                                       L.ptr(out2), _rows(out2, "out2") if out2 is not None else 0, n, h * w,
                                       L.stream_ptr()), "glorie_gru_gate_q")
    return out


def segment_mean(x, ix, groups, bias=None, relu=False):
    """out[g] = mean over edges e with ix[e] == g of act(x[e] + bias): scatter_mean of GraphAgg
    (droid_net.py:53-59).  x [N,128,h,w] channels-last fp16 (slice allowed), ix int64 [N]
    -> [groups,128,h,w] channels-last fp16"""
    L.need_cuda(x, ix)
    n, c, h, w = x.shape
    if c != 128 or ix.dtype != torch.int64 or ix.numel() != n or not ix.is_contiguous():
        raise RuntimeError("segment_mean: x must be [N,128,h,w], ix int64 [N]")
    out = torch.empty((groups, 128, h, w), dtype=torch.float16, device=x.device,
                      memory_format=torch.channels_last)
    L.check(L.load().glorie_segment_mean(L.ptr(x), _rows(x, "x"), L.ptr(bias), int(relu), L.ptr(ix), n,
                                         L.ptr(out), 128, groups, h * w, L.stream_ptr()), "glorie_segment_mean")
    return out


def pack_conv3x3_small(weights):
    """weights: list (one per group) of conv weights [K,128,3,3] -> fp16 MFMA B fragments
    [groups][NT][4][64][8] (layout documented in include/glorie_hip.h)"""
    K = weights[0].shape[0]
    ncols = 9 * K
    nt = 1 if ncols <= 16 else 2
    packs = []
    for wgt in weights:
        if tuple(wgt.shape) != (K, 128, 3, 3):
            raise RuntimeError("pack_conv3x3_small: expected [K,128,3,3] weights")
        cols = torch.zeros(nt * 16, 128, dtype=torch.float32, device=wgt.device)
        # column d*K + j  <-  w[j, :, d]
        cols[:ncols] = wgt.detach().float().reshape(K, 128, 9).permute(2, 0, 1).reshape(ncols, 128)
        # [t][col][kk][kg][i] -> [t][kk][kg][col][i]   (lane = kg*16 + col)
        packs.append(cols.view(nt, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).reshape(nt, 4, 64, 8))
    return torch.stack(packs).half().contiguous()


def conv3x3_small(x, w_packed, out_bias, K, acts, scale=1.0, in_bias=None, in_relu=False):
    """3x3 convolution 128 -> K (<= 3) channels per group on consecutive 128-channel slices of the
    channels-last fp16 map x [N, 128*groups(+), h, w] -> float32 [groups, N, h, w, K]
    (heads of the update operator, droid_net.py:85-93,42-44)"""
    L.need_cuda(x, w_packed)
    groups = w_packed.shape[0]
    n, c, h, w = x.shape
    if c < 128 * groups or len(acts) != groups:
        raise RuntimeError("conv3x3_small: bad shapes")
    P = n * h * w
    taps = torch.empty((P, groups * 9 * K), dtype=torch.float32, device=x.device)
    out = torch.empty((groups, n, h, w, K), dtype=torch.float32, device=x.device)
    packed = 0
    for gidx, a in enumerate(acts):
        packed |= int(a) << (4 * gidx)
    L.check(L.load().glorie_conv3x3_small(L.ptr(x), _rows(x, "x"), L.ptr(in_bias), int(in_relu), L.ptr(w_packed),
                                          L.ptr(out_bias), groups, K, packed, float(scale), L.ptr(taps),
                                          L.ptr(out), n, h, w, L.stream_ptr()), "glorie_conv3x3_small")
    return out


EPI_BIAS_ACT, EPI_GRU_ZR, EPI_GRU_Q, EPI_GLO = 0, 1, 2, 3


EPI_PAIR16 = 0x100          # flag on glorie_conv_igemm's `epilogue`: the weights come from pack_conv_igemm(pair=True)


def pack_conv_igemm(weight, pair=False):
    """conv weight [Nout, C, k, k] (k = 1 or 3) -> the fp16 operand of glorie_conv_igemm:
    [taps][npad][C] (npad = Nout rounded up to 128, zero rows) followed by 64 zero halfs.
    pair: within every group of 32 output channels, row 16 blk + r holds channel 8 (r // 4) + 4 blk + r % 4, so that a lane of
    the kernel owns 8 consecutive channels of its pixel (16-byte epilogue loads / stores, csrc/conv.hip: conv_epilogue_pair);
    the tensor remembers it (conv_igemm passes EPI_PAIR16).  Bias / gate epilogues only, Nout % 32 == 0."""
    nout, C, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3) or C % 64:
        raise RuntimeError("pack_conv_igemm: need 1x1 or 3x3 weights with C % 64 == 0")
    taps, npad = kh * kw, (nout + 127) // 128 * 128
    w = torch.zeros(taps, npad, C, dtype=torch.float16, device=weight.device)
    w[:, :nout] = weight.detach().permute(2, 3, 0, 1).reshape(taps, nout, C).half()
    if pair:
        if nout % 32:
            raise RuntimeError("pack_conv_igemm: pair needs Nout % 32 == 0")
        R = torch.arange(npad, device=weight.device)
        r = R % 16
        chan = 32 * (R // 32) + 8 * (r // 4) + 4 * ((R % 32) // 16) + r % 4
        w = w[:, chan]
    out = torch.cat([w.reshape(-1), torch.zeros(64, dtype=torch.float16, device=weight.device)])
    out._glorie_pair = bool(pair)
    return out


def pack_corr_encoder(weight):
    """corr_encoder[0] weight [128, 196, 1, 1] (droid_net.py:73-75) -> the 1x1 operand of glorie_conv_igemm over the
    channels-last lookup of glorie_corr_lookup_tiled_cl: column l*64 + dy*8 + dx takes the reference's column
    l*49 + dx*7 + dy, the padding columns are zero"""
    nout, C = weight.shape[0], weight.shape[1]
    if C != 196 or tuple(weight.shape[2:]) != (1, 1):
        raise RuntimeError("pack_corr_encoder: expected a [n, 196, 1, 1] weight")
    w = weight.detach().reshape(nout, 4, 7, 7)                 # [n][level][dx][dy]
    wp = torch.zeros(nout, 4, 8, 8, dtype=w.dtype, device=w.device)
    wp[:, :, :7, :7] = w.permute(0, 1, 3, 2)                  # [n][level][dy][dx]
    return pack_conv_igemm(wp.reshape(nout, 256, 1, 1))


def pack_corr_encoder_dm(weight):
    """corr_encoder[0] weight [128, 196, 1, 1] -> fp16 [128, 224] for the fused encoder of glorie_corr_dm_lookup (and of
    glorie_corr_otf_encode): column l*56 + j*8 + i takes the reference's column l*49 + i*7 + j (i <-> x offset, j <-> y
    offset), column i == 7 of every row is zero"""
    if tuple(weight.shape) != (128, 196, 1, 1):
        raise RuntimeError("pack_corr_encoder_dm: expected a [128, 196, 1, 1] weight")
    w = weight.detach().reshape(128, 4, 7, 7)                  # [n][level][i][j]
    wp = torch.zeros(128, 4, 7, 8, dtype=torch.float16, device=w.device)
    wp[..., :7] = w.permute(0, 1, 3, 2).half()                # [n][level][j][i]
    return wp.reshape(128, 224).contiguous()


CONV_POLICY = {None: 0, "auto": 0, "128": 1, "64": 2, "split": 3, "wide": 4, "nohalo": 5, "pp": 6, "ppw": 7}   # bits 12-15 of `epilogue`


def conv_igemm(xa, xb, w_packed, taps, nout, out, epilogue=EPI_BIAS_ACT, terms=None, act=ACT_NONE,
               net=None, z=None, out2=None, pre=None, pre_map=None, policy=None, pair=None):
    """Implicit-GEMM convolution with fused epilogue (csrc/conv.hip, include/glorie_hip.h).
    xa / xb: channels-last fp16 maps [N,Ca,h,w] / [N,Cb,h,w] (either may be None); writes `out`
    (and `out2` for the GRU gates) and returns `out`.  pre: fp16 channels-last [N,nout,h,w] added before the
    gate non-linearity (the hoisted convolution over the context features); with pre_map (int32 [N]) map e reads map
    pre_map[e] of `pre`, which then holds one map per distinct context (source keyframe) instead of one per edge.
    policy: tile choice forced by the caller (CONV_POLICY; tests and tools/bench_conv.py - the product passes None = auto).
    pair: the row layout of `w_packed` (pack_conv_igemm(pair=...)).  None reads the flag pack_conv_igemm left on the tensor
    object and RAISES when it is gone: a copy (.to / .clone / state_dict round trip) drops Python attributes, and the
    unpaired epilogue on paired rows would permute the output channels without any error."""
    ref = xa if xa is not None else xb
    L.need_cuda(ref, w_packed, out)
    n, _, h, w = ref.shape
    ca = xa.shape[1] if xa is not None else 0
    cb = xb.shape[1] if xb is not None else 0
    npad = (nout + 127) // 128 * 128
    if w_packed.dtype != torch.float16 or w_packed.numel() != taps * npad * (ca + cb) + 64:
        raise RuntimeError("conv_igemm: packed weights do not match (taps, nout, channels)")
    if epilogue == EPI_BIAS_ACT:
        if terms is not None and (terms.dtype != torch.float32 or terms.numel() != nout or not terms.is_contiguous()):
            raise RuntimeError("conv_igemm: bias must be contiguous float32 [nout]")
        ts = 0
    else:
        ts = _terms(terms, nout, "terms")
        if terms.shape[0] != n:
            raise RuntimeError("conv_igemm: one row of gate terms per map")
    opt = lambda t, name: (L.ptr(t), _rows(t, name) if t is not None else 0)
    (pa, sa), (pb, sb) = opt(xa, "xa"), opt(xb, "xb")
    (pn, sn), (pz, sz), (po2, so2) = opt(net, "net"), opt(z, "z"), opt(out2, "out2")
    pp, sp = opt(pre, "pre")
    if pre is not None and (epilogue == EPI_BIAS_ACT or (pre_map is None and pre.shape[0] != n) or pre.shape[1] != nout
                            or tuple(pre.shape[2:]) != (h, w)):
        raise RuntimeError("conv_igemm: pre needs a gate epilogue and the shape [N,nout,h,w]")
    if pre_map is not None and (pre is None or pre_map.dtype != torch.int32 or pre_map.numel() != n or
                                not pre_map.is_contiguous() or pre_map.device != ref.device):
        raise RuntimeError("conv_igemm: pre_map must be a contiguous int32 [N] on the device, next to pre")
    if out.shape[1] != (128 if (epilogue & 0xff) != EPI_BIAS_ACT else nout) or out.shape[0] != n:
        raise RuntimeError("conv_igemm: bad output shape")
    if pair is None:
        pair = getattr(w_packed, "_glorie_pair", None)
        if pair is None:
            raise RuntimeError("conv_igemm: w_packed carries no row-layout flag (a copy of a pack_conv_igemm result loses "
                               "it): pack again or pass pair=True/False")
    if pair:
        epilogue |= EPI_PAIR16
    epilogue |= CONV_POLICY[policy] << 12
    L.check(L.load().glorie_conv_igemm(pa, sa, ca, pb, sb, cb, L.ptr(w_packed), taps, nout, epilogue,
                                       L.ptr(terms), ts, act, pn, sn, pz, sz, L.ptr(out), _rows(out, "out"),
                                       po2, so2, pp, sp, L.ptr(pre_map), n, h, w, L.stream_ptr()), "glorie_conv_igemm")
    return out


def pack_head_taps(weights):
    """weights: list (one per head) of the heads' second conv weights [K,128,3,3] -> the fp16 MFMA A fragments of
    glorie_conv_igemm_heads [groups][2][2][2][64][8] (layout in include/glorie_hip.h): tap row d*K + j of hidden channel
    64*half + 16*(2*chunk + s//4) + 4*(lane>>4) + s%4"""
    K = weights[0].shape[0]
    ncols = 9 * K
    packs = []
    for wgt in weights:
        if tuple(wgt.shape) != (K, 128, 3, 3):
            raise RuntimeError("pack_head_taps: expected [K,128,3,3] weights")
        rows = torch.zeros(32, 128, dtype=torch.float32, device=wgt.device)
        rows[:ncols] = wgt.detach().float().reshape(K, 128, 9).permute(2, 0, 1).reshape(ncols, 128)
        # channel = 64*half + 32*chunk + 16*blk + 4*kg + q, fragment slot s = 4*blk + q, lane = 16*kg + i, row = 16*rb + i
        r = rows.view(2, 16, 2, 2, 2, 4, 4)                  # [rb][i][half][chunk][blk][kg][q]
        packs.append(r.permute(2, 3, 0, 5, 1, 4, 6).reshape(2, 2, 2, 64, 8))   # [half][chunk][rb][kg*16+i][blk*4+q]
    return torch.stack(packs).half().contiguous()


def conv_igemm_heads(x, w_packed, taps, nout, bias, tap_w, K, out=None):
    """glorie_conv_igemm_heads: the hidden layers of `groups` 3x3 heads (+ trailing channels stored to `out`) and the
    heads' tap planes in one launch; returns the float32 planes [groups*9K, N*h*w] for conv_stencil"""
    L.need_cuda(x, w_packed, tap_w)
    n, c, h, w = x.shape
    groups = tap_w.shape[0]
    if w_packed.dtype != torch.float16 or w_packed.numel() != taps * nout * c + 64 or nout % 128:
        raise RuntimeError("conv_igemm_heads: packed weights do not match (taps, nout, channels)")
    if tap_w.dtype != torch.float16 or tuple(tap_w.shape[1:]) != (2, 2, 2, 64, 8) or not tap_w.is_contiguous():
        raise RuntimeError("conv_igemm_heads: tap_w must come from pack_head_taps")
    if bias.dtype != torch.float32 or bias.numel() != nout or not bias.is_contiguous():
        raise RuntimeError("conv_igemm_heads: bias must be contiguous float32 [nout]")
    rest = nout - 128 * groups
    if rest < 0 or (rest > 0 and (out is None or out.shape[0] != n or out.shape[1] != rest)):
        raise RuntimeError("conv_igemm_heads: `out` must hold the nout - 128*groups trailing channels")
    rows = torch.empty((groups * 9 * K, n * h * w), dtype=torch.float32, device=x.device)
    L.check(L.load().glorie_conv_igemm_heads(L.ptr(x), _rows(x, "x"), c, L.ptr(w_packed), taps, nout, L.ptr(bias),
                                             L.ptr(tap_w), groups, K, L.ptr(rows), L.ptr(out) if rest else None,
                                             _rows(out, "out") if rest else 0, n, h, w, L.stream_ptr()),
            "glorie_conv_igemm_heads")
    return rows


def conv_stencil(rows, out_bias, n, h, w, groups, K, acts, scale=1.0, out_last=None):
    """the 9-point stencil half of conv3x3_small on tap planes [groups*9K, n*h*w] -> float32 [groups, n, h, w, K];
    out_last: contiguous float32 tensor of n*h*w*K elements that receives the LAST group instead of its slice of the result"""
    L.need_cuda(rows)
    if rows.dtype != torch.float32 or tuple(rows.shape) != (groups * 9 * K, n * h * w) or not rows.is_contiguous() \
            or len(acts) != groups:
        raise RuntimeError("conv_stencil: bad tap planes")
    out = torch.empty((groups, n, h, w, K), dtype=torch.float32, device=rows.device)
    packed = 0
    for gidx, a in enumerate(acts):
        packed |= int(a) << (4 * gidx)
    if out_last is not None and (out_last.dtype != torch.float32 or not out_last.is_contiguous() or not out_last.is_cuda
                                 or out_last.numel() != n * h * w * K):
        raise RuntimeError("conv_stencil: out_last must be a contiguous float32 CUDA tensor of n*h*w*K elements")
    L.check(L.load().glorie_conv_stencil(L.ptr(rows), L.ptr(out_bias), groups, K, packed, float(scale), L.ptr(out),
                                         L.ptr(out_last), n, h, w, L.stream_ptr()), "glorie_conv_stencil")
    return out


def pack_flow_conv7(weight):
    """flow_encoder[0] weight [128,4,7,7] -> fp16 [128][224], column ky*32 + kx*4 + c (kx = 7 zero)"""
    if tuple(weight.shape) != (128, 4, 7, 7):
        raise RuntimeError("pack_flow_conv7: expected [128,4,7,7]")
    w = torch.zeros(128, 7, 8, 4, dtype=torch.float16, device=weight.device)
    w[:, :, :7] = weight.detach().permute(0, 2, 3, 1).half()
    return w.reshape(128, 224).contiguous()


def pack_upmask_conv(weight, bias):
    """GraphAgg.upmask[0] (droid_net.py:46-48) weight [576, C, 1, 1], bias [576] -> (packed 1x1 operand of 1024 rows,
    bias [1024]) for glorie_conv_upsample: row a*128 + b*16 + t = channel t*64 + a*8 + b (t < 9), zero rows for t >= 9"""
    if weight.shape[0] != 576 or tuple(weight.shape[2:]) != (1, 1):
        raise RuntimeError("pack_upmask_conv: expected a [576, C, 1, 1] weight")
    C = weight.shape[1]
    w = weight.detach().reshape(9, 8, 8, C)                    # [t][a][b][C]
    wp = torch.zeros(8, 8, 16, C, dtype=w.dtype, device=w.device)
    wp[:, :, :9] = w.permute(1, 2, 0, 3)                        # [a][b][t][C]
    bp = torch.zeros(8, 8, 16, dtype=torch.float32, device=w.device)
    bp[:, :, :9] = bias.detach().float().reshape(9, 8, 8).permute(1, 2, 0)
    return pack_conv_igemm(wp.reshape(1024, C, 1, 1)), bp.reshape(1024).contiguous()


class LazyUpmask:
    """the upmask logits of FusedUpdate NOT evaluated: the input of the 1x1 convolution and its packed weights.
    DepthVideo.upsample runs convolution + convex upsampling as one launch (conv_upsample)"""

    def __init__(self, x, w_packed, bias):
        self.x, self.w_packed, self.bias = x, w_packed, bias

    def evaluate(self):
        """the logits themselves, fp16 [M, 576, h, w] in the reference's channel order (tests, callers that need the map)"""
        m, c, h, w = self.x.shape
        wide = torch.empty((m, 1024, h, w), dtype=torch.float16, device=self.x.device, memory_format=torch.channels_last)
        conv_igemm(self.x, None, self.w_packed, 1, 1024, wide, terms=self.bias)
        return wide.view(m, 8, 8, 16, h, w)[:, :, :, :9].permute(0, 3, 1, 2, 4, 5).reshape(m, 576, h, w)


def conv_upsample(up, disps, ix, disps_up, softmax_f32=False):
    """disps_up[ix] = cvx_upsample(disps[ix], conv1x1(up.x) + bias) (glorie_conv_upsample); up: LazyUpmask"""
    L.need_cuda(up.x, up.w_packed, up.bias, disps, ix, disps_up)
    m, c, h, w = up.x.shape
    if ix.dtype != torch.int64 or ix.numel() != m or not ix.is_contiguous():
        raise RuntimeError("conv_upsample: ix must be a contiguous int64 [M]")
    if tuple(disps.shape[1:]) != (h, w) or tuple(disps_up.shape[1:]) != (8 * h, 8 * w) or disps.dtype != torch.float32 \
            or disps_up.dtype != torch.float32 or not disps.is_contiguous() or not disps_up.is_contiguous():
        raise RuntimeError("conv_upsample: disps [B,h,w] / disps_up [B,8h,8w] float32 contiguous expected")
    if up.w_packed.numel() != 1024 * c + 64 or up.bias.numel() != 1024:
        raise RuntimeError("conv_upsample: weights must come from pack_upmask_conv")
    L.check(L.load().glorie_conv_upsample(L.ptr(up.x), _rows(up.x, "x"), c, L.ptr(up.w_packed), L.ptr(up.bias),
                                          L.ptr(disps), L.ptr(ix), L.ptr(disps_up), int(bool(softmax_f32)), m, h, w,
                                          L.stream_ptr()), "glorie_conv_upsample")
    return disps_up


class PaddedFlow:
    """the motion map of N edges as zero-padded fp16 [N, h+6, w+8, 4] (include/glorie_hip.h: glorie_flow_conv7_padded);
    the borders are zeroed once here and never written again"""

    def __init__(self, n, h, w, device):
        self.n, self.h, self.w = n, h, w
        self.buf = torch.zeros((n, h + 6, w + 8, 4), dtype=torch.float16, device=device)

    def fits(self, n, h, w, device):
        return (self.n, self.h, self.w) == (n, h, w) and self.buf.device == torch.device(device)

    def interior(self):
        """[N, h, w, 4] view of the map itself (tests)"""
        return self.buf[:, 3:3 + self.h, 3:3 + self.w]


def flow_pad(flow, padded):
    """fp32 channels-last motion map [N,h,w,4] -> the interior of `padded` (PaddedFlow)"""
    L.need_cuda(flow, padded.buf)
    n, h, w, c = flow.shape
    if c != 4 or flow.dtype != torch.float32 or not flow.is_contiguous() or not padded.fits(n, h, w, flow.device):
        raise RuntimeError("flow_pad: flow must be contiguous float32 [N,h,w,4] matching the padded map")
    L.check(L.load().glorie_flow_pad(L.ptr(flow), L.ptr(padded.buf), n, h, w, L.stream_ptr()), "glorie_flow_pad")
    return padded


def flow_conv7_padded(padded, w_packed, bias, out):
    """flow_conv7 on a PaddedFlow (droid_net.py:79-81): out channels-last fp16 [N,128,h,w] (slice allowed)"""
    L.need_cuda(padded.buf, w_packed, bias, out)
    if out.shape[1] != 128 or out.shape[0] != padded.n or tuple(out.shape[2:]) != (padded.h, padded.w):
        raise RuntimeError("flow_conv7_padded: bad output shape")
    L.check(L.load().glorie_flow_conv7_padded(L.ptr(padded.buf), L.ptr(w_packed), L.ptr(bias), L.ptr(out),
                                              _rows(out, "out"), padded.n, padded.h, padded.w, L.stream_ptr()),
            "glorie_flow_conv7_padded")
    return out


def flow_conv7(flow, w_packed, bias, out):
    """out = relu(conv7x7(flow) + bias); flow float32 [N,h,w,4] contiguous (channels-last motion
    map), out channels-last fp16 [N,128,h,w] (slice allowed)   (droid_net.py:79-81)"""
    L.need_cuda(flow, w_packed, bias, out)
    n, h, w, c = flow.shape
    if c != 4 or flow.dtype != torch.float32 or not flow.is_contiguous() or out.shape[1] != 128:
        raise RuntimeError("flow_conv7: flow must be contiguous float32 [N,h,w,4]")
    L.check(L.load().glorie_flow_conv7(L.ptr(flow), L.ptr(w_packed), L.ptr(bias), L.ptr(out), _rows(out, "out"),
                                       n, h, w, L.stream_ptr()), "glorie_flow_conv7")
    return out
