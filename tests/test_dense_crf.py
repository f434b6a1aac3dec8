"""Dense CRF post-processing (test.py:309-322).  pydensecrf is a third-party dependency outside the reference tree and is not installed:
the CPU tests pin the oracle (oracle/dense_crf_numpy.py) by closed-form cases of the published recursion, the GPU tests compare the
HIP implementation with it on images small enough for the brute-force N^2 evaluation, and check the full-size run by properties.
Parity against pydensecrf itself (permutohedral-lattice filtering) is UNPINNED."""
import importlib

import numpy as np
import pytest

from oracle import dense_crf_numpy as OC


def test_oracle_closed_form_cases():
    rng = np.random.default_rng(0)
    p = rng.uniform(0.05, 0.95, (6, 7))
    rgb = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8)
    # no iterations, or zero pairwise weights: the marginals are the sigmoid map itself
    for q in (OC.dense_crf(p, rgb, iters=0), OC.dense_crf(p, rgb, compat_g=0.0, compat_b=0.0, iters=3)):
        assert np.allclose(q[1], p, atol=1e-12) and np.allclose(q[0], 1 - p, atol=1e-12)
    # a uniform map on a uniform image stays uniform in space (symmetric normalisation), and the pairwise term pulls towards the majority
    q = OC.dense_crf(np.full((5, 5), 0.7), np.full((5, 5, 3), 90, np.uint8))
    assert np.allclose(q.sum(0), 1) and q[1].min() > 0.7 and np.allclose(q[1], q[1][::-1, ::-1])
    # one iteration by hand on two pixels: K = [[1, k], [k, 1]], D = 1 + k, Ktilde Q = (Q_i + k Q_j) / (1 + k)
    p2 = np.array([[0.8, 0.3]])
    im2 = np.array([[[10, 10, 10], [12, 11, 10]]], np.uint8)
    kg, kb = np.exp(-0.5 / 9.0), np.exp(-0.5 * (1 / 400.0 + (4 + 1) / 9.0))
    U = OC.unary_from_sigmoid(p2)
    Q0 = np.stack([1 - p2.ravel(), p2.ravel()])
    e = -U + 3.0 * (Q0 + kg * Q0[:, ::-1]) / (1 + kg) + 10.0 * (Q0 + kb * Q0[:, ::-1]) / (1 + kb)
    want = np.exp(e - e.max(0)); want /= want.sum(0)
    assert np.allclose(OC.dense_crf(p2, im2, iters=1).reshape(2, 2), want, atol=1e-12)
    # a strong edge in the image stops the bilateral term: the two halves keep opposite labels
    p3 = np.concatenate([np.full((8, 4), 0.9), np.full((8, 4), 0.1)], 1)
    im3 = np.concatenate([np.full((8, 4, 3), 200, np.uint8), np.full((8, 4, 3), 20, np.uint8)], 1)
    m = OC.dense_crf(p3, im3).argmax(0)
    assert m[:, :4].all() and not m[:, 4:].any()


@pytest.mark.gpu
def test_hip_dense_crf_matches_oracle_and_full_size_properties():
    import torch
    CRF = importlib.import_module("cmpc-refseg_amd.crf")
    rng = np.random.default_rng(1)
    for (H, W, kw) in ((20, 24, {}), (13, 37, dict(sxy_g=2.0, compat_g=1.5, sxy_b=9.0, srgb=6.0, compat_b=4.0, iters=3)), (9, 9, dict(iters=0)), (16, 16, dict(iters=1))):
        yy, xx = np.mgrid[0:H, 0:W]
        p = 1 / (1 + np.exp(-(((xx - W / 2) ** 2 + (yy - H / 2) ** 2 < (min(H, W) / 3) ** 2) * 3.0 - 1.5 + rng.normal(0, 0.8, (H, W)))))
        rgb = np.clip(rng.normal(0, 6, (H, W, 3)) + np.where(p[..., None] > 0.5, 150, 60), 0, 255).astype(np.uint8)
        ref = OC.dense_crf(p, rgb, **kw)
        q = CRF.dense_crf(p.astype(np.float32), rgb, **kw).cpu().numpy()
        m = CRF.dense_crf_mask(p.astype(np.float32), rgb, **kw).cpu().numpy()
        assert np.abs(q - ref).max() < 2e-4, (H, W, np.abs(q - ref).max())
        sure = np.abs(ref[1] - ref[0]) > 1e-3
        assert np.array_equal(m[sure], ref.argmax(0)[sure].astype(np.uint8))
    # full size (320 x 320, the reference's parameters): a noisy disc on a two-tone image is cleaned up, Q is a distribution
    H = W = 320
    yy, xx = np.mgrid[0:H, 0:W]
    disc = (xx - 170) ** 2 + (yy - 150) ** 2 < 70 ** 2
    p = np.clip(np.where(disc, 0.8, 0.2) + rng.normal(0, 0.25, (H, W)), 0.01, 0.99).astype(np.float32)
    rgb = np.clip(rng.normal(0, 4, (H, W, 3)) + np.where(disc[..., None], 180, 40), 0, 255).astype(np.uint8)
    q = CRF.dense_crf(p, rgb)
    torch.cuda.synchronize()
    q = q.cpu().numpy()
    assert np.isfinite(q).all() and np.allclose(q.sum(0), 1, atol=1e-5)
    raw_err = ((p > 0.5) != disc).mean(); crf_err = ((q[1] > q[0]) != disc).mean()
    assert raw_err > 0.05 and crf_err < 0.002, (raw_err, crf_err)
    with pytest.raises(ValueError):
        CRF.dense_crf(p, rgb[:10])
