"""Pins the corr-lookup oracle against an independent dense formulation (no GPU)."""
import numpy as np

from oracle import corr as ocorr


def test_index_forward_matches_bilinear_window_fp32():
    rng = np.random.default_rng(0)
    N, h1, w1, h2, w2 = 2, 5, 6, 9, 11
    vol = rng.standard_normal((N, h1, w1, h2, w2)).astype(np.float32)
    coords = np.stack([rng.uniform(-3, w2 + 3, (N, h1, w1)), rng.uniform(-3, h2 + 3, (N, h1, w1))], 1).astype(np.float32)
    got = ocorr.corr_index_forward(vol, coords, 3)
    ref = ocorr.window_bilinear_reference(vol, coords, 3)
    np.testing.assert_allclose(got, ref, atol=1e-5)


def test_index_forward_fp16_close_to_fp32_and_channel_order():
    rng = np.random.default_rng(1)
    N, h1, w1, h2, w2 = 1, 4, 4, 8, 8
    vol = rng.standard_normal((N, h1, w1, h2, w2)).astype(np.float16)
    coords = np.stack([rng.uniform(1, 6, (N, h1, w1)), rng.uniform(1, 6, (N, h1, w1))], 1).astype(np.float32)
    a = ocorr.corr_index_forward(vol, coords, 3).astype(np.float32)
    b = ocorr.window_bilinear_reference(vol.astype(np.float32), coords, 3)
    np.testing.assert_allclose(a, b, atol=2e-2)
    # integer coords: out[n,i,j,y,x] == vol[n,y,x, y0+j-3, x0+i-3]  (i <-> x offset)
    c = np.zeros((1, 2, 4, 4), np.float32)
    c[:, 0] = 4.0
    c[:, 1] = 3.0
    o = ocorr.corr_index_forward(vol, c, 3)
    assert o[0, 5, 2, 1, 1] == vol[0, 1, 1, 3 + 2 - 3, 4 + 5 - 3]


def test_altcorr_equals_volume_lookup():
    rng = np.random.default_rng(2)
    B, H, W, C = 1, 6, 7, 64
    f1 = rng.standard_normal((B, H, W, C)).astype(np.float32)
    f2 = rng.standard_normal((B, H, W, C)).astype(np.float32)
    coords = np.stack([rng.uniform(-2, W + 2, (B, 1, H, W)), rng.uniform(-2, H + 2, (B, 1, H, W))], -1).astype(np.float32)
    alt = ocorr.altcorr_forward(f1, f2, coords, 3)[:, 0].reshape(B, 7, 7, H, W)
    vol = np.einsum("bhwc,bijc->bhwij", f1, f2).astype(np.float32)
    ref = ocorr.window_bilinear_reference(vol, coords[:, 0].transpose(0, 3, 1, 2), 3)
    np.testing.assert_allclose(alt, ref, rtol=1e-4, atol=1e-4)
