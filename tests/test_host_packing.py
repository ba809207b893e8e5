"""Host-side operand packing of the update-operator kernels and the tiled correlation pyramid (CPU):
the layouts documented in include/glorie_hip.h, checked element by element."""
import numpy as np
import torch

from glorie_slam_amd import droid_backends as db
from glorie_slam_amd import update_ops as U


def test_tile_corr_level_layout_and_slack():
    g = torch.Generator().manual_seed(0)
    P, h2, w2 = 5, 7, 10                                     # not multiples of the 4 x 8 block
    vol = torch.randn(P, h2, w2, generator=g).half()
    t = db.tile_corr_level(vol)
    nby, nbx = 2, 2
    assert tuple(t.shape) == (P, nby * nbx * 32) and t.is_contiguous()
    blocks = t.view(P, nby, nbx, 4, 8)
    for (p, y, x) in [(0, 0, 0), (4, 6, 9), (2, 3, 8), (1, 4, 7)]:
        assert blocks[p, y // 4, x // 8, y % 4, x % 8] == vol[p, y, x]
    pad = blocks.clone()
    for y in range(h2):
        for x in range(w2):
            pad[:, y // 4, x // 8, y % 4, x % 8] = 0
    assert float(pad.abs().sum()) == 0.0                      # padding rows / columns are zero
    # one spare plane on either side of the view (read slack of the unaligned block-row loads)
    assert t.storage_offset() == t.shape[1] and t.untyped_storage().nbytes() >= (P + 2) * t.shape[1] * 2


def test_pack_conv_igemm_layout():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(70, 128, 3, 3, generator=g)
    pk = U.pack_conv_igemm(w)
    assert pk.dtype == torch.float16 and pk.numel() == 9 * 128 * 128 + 64
    body = pk[:-64].view(9, 128, 128)
    assert torch.equal(body[4, 33, :], w[33, :, 1, 1].half())           # tap (ky,kx)=(1,1) -> index 4
    assert torch.equal(body[2, 69, :], w[69, :, 0, 2].half())
    assert float(body[:, 70:].abs().sum()) == 0.0 and float(pk[-64:].abs().sum()) == 0.0
    w1 = torch.randn(576, 128, 1, 1, generator=g)
    assert U.pack_conv_igemm(w1).numel() == 640 * 128 + 64               # 576 -> 5 tiles of 128


def test_pack_conv_igemm_paired_rows():
    """pair=True (include/glorie_hip.h: GLORIE_CONV_PAIR16): within every group of 32 output channels, packed row
    16 blk + r holds channel 8 (r // 4) + 4 blk + r % 4 - the MFMA hands lane group kg rows 4 kg .. 4 kg + 3 of a 16-row
    block, so the lane's rows of blocks 2b and 2b + 1 are the 8 consecutive channels 32 b + 8 kg .. + 7"""
    import pytest
    g = torch.Generator().manual_seed(2)
    w = torch.randn(96, 64, 3, 3, generator=g)
    pk = U.pack_conv_igemm(w, pair=True)
    assert pk._glorie_pair and not U.pack_conv_igemm(w)._glorie_pair
    body = pk[:-64].view(9, 128, 64)
    plain = U.pack_conv_igemm(w)[:-64].view(9, 128, 64)
    for R in range(128):
        r = R % 16
        chan = 32 * (R // 32) + 8 * (r // 4) + 4 * ((R % 32) // 16) + r % 4
        assert torch.equal(body[:, R], plain[:, chan])
    for b in range(4):                                           # what a lane (kg) owns across a pair of blocks
        for kg in range(4):
            rows = [32 * b + 16 * blk + 4 * kg + k for blk in (0, 1) for k in range(4)]
            chans = [32 * b + 8 * (r % 16 // 4) + 4 * ((r % 32) // 16) + r % 4 for r in rows]
            assert chans == list(range(32 * b + 8 * kg, 32 * b + 8 * kg + 8))
    with pytest.raises(RuntimeError):
        U.pack_conv_igemm(torch.randn(70, 64, 3, 3), pair=True)          # Nout % 32 != 0
    assert U.CONV_POLICY[None] == 0 and sorted(U.CONV_POLICY.values())[-1] == 7 and U.EPI_PAIR16 == 0x100


def test_pack_conv3x3_small_and_flow_layouts():
    g = torch.Generator().manual_seed(2)
    ws = [torch.randn(2, 128, 3, 3, generator=g), torch.randn(2, 128, 3, 3, generator=g)]
    pk = U.pack_conv3x3_small(ws)
    assert tuple(pk.shape) == (2, 2, 4, 64, 8)
    for (grp, t, kk, lane, i) in [(0, 0, 0, 0, 0), (1, 1, 2, 37, 5), (0, 0, 3, 63, 7), (1, 1, 1, 1, 2)]:
        col, kg = lane & 15, lane >> 4
        n = 16 * t + col
        ref = 0.0
        if n < 18:
            d, j = divmod(n, 2)
            ref = float(ws[grp][j, 32 * kk + 8 * kg + i, d // 3, d % 3].half())
        assert float(pk[grp, t, kk, lane, i]) == ref
    w1 = torch.randn(1, 128, 3, 3, generator=g)
    assert tuple(U.pack_conv3x3_small([w1]).shape) == (1, 1, 4, 64, 8)   # 9 columns: one 16-column tile
    wf = torch.randn(128, 4, 7, 7, generator=g)
    pf = U.pack_flow_conv7(wf)
    assert tuple(pf.shape) == (128, 224)
    assert float(pf[17, 3 * 32 + 5 * 4 + 2]) == float(wf[17, 2, 3, 5].half())
    assert float(pf.view(128, 7, 8, 4)[:, :, 7].abs().sum()) == 0.0        # the 8th tap of every row


def test_se3_inverse_helper():
    from glorie_slam_amd.neural_point import se3_inv
    from oracle import se3 as ose3
    rng = np.random.default_rng(0)
    q = rng.standard_normal((6, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([rng.standard_normal((6, 3)), q], 1).astype(np.float32)
    inv = se3_inv(torch.from_numpy(poses)).numpy()
    for a, b in zip(poses, inv):
        np.testing.assert_allclose(ose3.matrix(a) @ ose3.matrix(b), np.eye(4), atol=2e-6)


def test_dm_layout_definition_and_roundtrip():
    """droid_backends.dm_corr_level / dm_to_rowmajor against the layout's definition (include/glorie_hip.h:
    glorie_corr_dm_build), element by element, on CPU - incl. odd level widths (a zero padding column)"""
    import numpy as np
    import torch
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(0)
    N, h, w = 2, 11, 18
    for l in range(4):
        hl, wl = h >> l, w >> l
        wp = (wl + 1) & ~1
        vol = torch.from_numpy(rng.standard_normal((N, h, w, hl, wl)).astype(np.float16))
        dm = db.dm_corr_level(vol, l)
        nt, hh, ww = db.dm_shape(h, w, l)
        assert (hh, ww) == (hl, wp) and dm.shape == (N, nt * hl * wp * 64)
        assert torch.equal(db.dm_to_rowmajor(dm, h, w, l), vol)
        d = dm.view(N, (h + 7) // 8, (w + 7) // 8, hl, wp // 2, 64, 2)
        for _ in range(200):
            n, sy, sx, ty, tx = (int(rng.integers(0, k)) for k in (N, h, w, hl, wl))
            dy = (ty - (sy >> l) + (hl >> 1)) % hl
            dx = (tx - (sx >> l) + (wl >> 1)) % wp
            assert d[n, sy // 8, sx // 8, dy, dx >> 1, (sy & 7) * 8 + (sx & 7), dx & 1] == vol[n, sy, sx, ty, tx]
        if wl & 1:      # the displacement that would name target column wl holds a zero for every live source pixel
            for _ in range(50):
                n, sy, sx, dy = (int(rng.integers(0, k)) for k in (N, h, w, hl))
                dx = (wl - (sx >> l) + (wl >> 1)) % wp
                assert d[n, sy // 8, sx // 8, dy, dx >> 1, (sy & 7) * 8 + (sx & 7), dx & 1] == 0
