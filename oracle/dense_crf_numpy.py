"""TEST INFRASTRUCTURE (oracle): fully connected CRF post-processing as the reference's evaluation script applies it
(/root/reference/test.py:309-322: `densecrf.DenseCRF2D(W, H, 2)`, `setUnaryEnergy`, `addPairwiseGaussian(sxy=3, compat=3)`,
`addPairwiseBilateral(sxy=20, srgb=3, rgbim, compat=10)`, `inference(5)`, argmax).

The algorithm lives in a third-party dependency that is NOT part of the reference tree (`pydensecrf`, a wrapper of P. Kraehenbuehl's
densecrf library; the reference pins no version) and is not installed here, so this file restates the published algorithm
(Kraehenbuehl & Koltun, "Efficient Inference in Fully Connected CRFs with Gaussian Edge Potentials", NIPS 2011; densecrf
src/densecrf.cpp `DenseCRF::inference`, src/pairwise.cpp `PairwisePotential::apply`, `PottsCompatibility::apply`,
`NORMALIZE_SYMMETRIC`):

    Q   <- softmax_l(-U)
    repeat n times:   Q <- softmax_l( -U + sum_m w_m * Ktilde_m Q )           (Potts compatibility: -w on equal labels)
    Ktilde = D^-1/2 K D^-1/2,  D = diag(K 1 + 1e-20),   K_ij = exp(-|f_i - f_j|^2 / 2)   (the term j = i included, as in the library)
    f = (x, y) / sxy   for the Gaussian,   f = (x / sxy, y / sxy, r / srgb, g / srgb, b / srgb)   for the bilateral kernel

with K evaluated EXACTLY (all N^2 pairs) where the library filters through a permutohedral lattice, an approximation of the same
Gaussian.  PARITY UNPINNED: no pydensecrf output exists to compare with; masks can differ from the library's at object borders.
Brute force, O(N^2) memory: small images only."""
import numpy as np


def unary_from_sigmoid(sigm):
    """test.py:312-315: label 0 = -log(1 - p), label 1 = -log(p)   (probabilities clamped away from 0 so that the energies stay finite)."""
    p = np.asarray(sigm, dtype=np.float64).reshape(-1)
    return np.stack([-np.log(np.maximum(1.0 - p, 1e-30)), -np.log(np.maximum(p, 1e-30))])


def _softmax0(e):
    e = e - e.max(axis=0, keepdims=True)
    q = np.exp(e)
    return q / q.sum(axis=0, keepdims=True)


def _kernel(feat):
    d2 = ((feat[:, None, :] - feat[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d2)


def dense_crf(sigm, rgb, sxy_g=3.0, compat_g=3.0, sxy_b=20.0, srgb=3.0, compat_b=10.0, iters=5):
    """sigm [H, W] probabilities of label 1, rgb [H, W, 3] uint8 -> Q [2, H, W] float64 after `iters` mean-field iterations."""
    sigm = np.asarray(sigm, dtype=np.float64)
    H, W = sigm.shape
    ys, xs = np.mgrid[0:H, 0:W]
    pos = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.float64)
    col = np.asarray(rgb, dtype=np.float64).reshape(-1, 3)
    kernels = []
    for w, feat in ((compat_g, pos / sxy_g), (compat_b, np.concatenate([pos / sxy_b, col / srgb], 1))):
        K = _kernel(feat)
        nrm = 1.0 / np.sqrt(K.sum(1) + 1e-20)
        kernels.append((w, K, nrm))
    U = unary_from_sigmoid(sigm)
    Q = _softmax0(-U)
    for _ in range(iters):
        e = -U
        for w, K, nrm in kernels:
            e = e + w * (nrm[None, :] * ((Q * nrm[None, :]) @ K.T))
        Q = _softmax0(e)
    return Q.reshape(2, H, W)
