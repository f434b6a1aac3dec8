"""Dense CRF post-processing of the reference's evaluation script (test.py:309-322, the `-c` / dcrf option): pydensecrf's
`DenseCRF2D(W, H, 2)` with the unary energies of the sigmoid map, a Gaussian and a bilateral Potts term and 5 mean-field iterations.

    Q = dense_crf(sigm, proc_im)                     # [2, H, W]; np.argmax(Q, 0) is the refined mask
    mask = dense_crf_mask(sigm, proc_im)             # [H, W] uint8 in {0, 1}

Runs on the GPU through `cmpc_dense_crf` (csrc/ops_crf.hip): the Gaussian kernels are evaluated exactly in a 4-sigma window where the
library filters through a permutohedral lattice, so masks can differ from pydensecrf's at object borders (parity-unpinned: pydensecrf
is a third-party dependency outside the reference tree and is not installed here).  `oracle/dense_crf_numpy.py` is the brute-force
CPU restatement the tests compare with.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

DEFAULTS = dict(sxy_g=3.0, compat_g=3.0, sxy_b=20.0, srgb=3.0, compat_b=10.0, iters=5)        # test.py:317-319


def _run(sigm, rgb, want_q, want_mask, device, **kw):
    p = dict(DEFAULTS); p.update(kw)
    dev = torch.device(device)
    s = torch.as_tensor(np.asarray(sigm) if not torch.is_tensor(sigm) else sigm).to(dev, torch.float32).contiguous()
    if s.dim() != 2:
        raise ValueError("sigm must be [H, W], got %s" % (tuple(s.shape),))
    H, W = s.shape
    im = torch.as_tensor(np.asarray(rgb) if not torch.is_tensor(rgb) else rgb)
    if tuple(im.shape) != (H, W, 3):
        raise ValueError("rgb must be [H, W, 3] matching sigm, got %s" % (tuple(im.shape),))
    if im.dtype != torch.uint8:
        raise ValueError("rgb must be uint8 (the resized / padded image the reference passes as rgbim)")
    im = im.to(dev).contiguous()
    q = torch.empty(2, H, W, device=dev, dtype=torch.float32) if want_q else None
    m = torch.empty(H, W, device=dev, dtype=torch.uint8) if want_mask else None
    with torch.cuda.device(dev):
        _lib.call("cmpc_dense_crf", s.data_ptr(), im.data_ptr(), H, W, float(p["sxy_g"]), float(p["compat_g"]), float(p["sxy_b"]), float(p["srgb"]),
                  float(p["compat_b"]), int(p["iters"]), q.data_ptr() if q is not None else None, m.data_ptr() if m is not None else None,
                  torch.cuda.current_stream(dev).cuda_stream)
    return q, m


def dense_crf(sigm, rgb, device="cuda:0", **params) -> torch.Tensor:
    """d.inference(iters) of test.py:309-320: the label marginals Q [2, H, W] (fp32, on `device`)."""
    return _run(sigm, rgb, True, False, device, **params)[0]


def dense_crf_mask(sigm, rgb, device="cuda:0", **params) -> torch.Tensor:
    """np.argmax(Q, axis=0) of test.py:321: [H, W] uint8 in {0, 1}."""
    return _run(sigm, rgb, False, True, device, **params)[1]
