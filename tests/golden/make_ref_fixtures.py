"""Golden vectors from the REFERENCE's own code (run in the build container, where /root/reference exists).

The only parts of the hot path's reference that are executable here are its pure-NumPy helpers (TensorFlow,
nltk and skimage are not installed, so CMPC_model.py, util/text_processing.py and util/im_processing.py cannot be
imported):
    util/processing_tools.py:5-17    generate_spatial_batch   
    util/processing_tools.py:24-42   compute_accuracy, compute_meanIoU   (metric half of row S)
    util/eval_tools.py:31-35         compute_mask_IU          (the I/U accumulators of trainval_model.py:267-303)
This script imports those two files from /root/reference, calls them on seeded inputs and writes inputs + outputs
to tests/golden/ref_processing_tools.npz.  Only that data file travels; no reference source does.

    python tests/golden/make_ref_fixtures.py
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    pt = _load("util/processing_tools.py", "ref_processing_tools")
    et = _load("util/eval_tools.py", "ref_eval_tools")
    out = {}
    # spatial grids: config 2 (40x40), the tiny test graph (8x8, B=2), config 4 (64x64), a non-square one
    for (n, h, w) in ((1, 40, 40), (2, 8, 8), (1, 64, 64), (3, 5, 7)):
        out[f"grid/{n}x{h}x{w}"] = pt.generate_spatial_batch(n, h, w)
    rng = np.random.default_rng(20261004)
    cases = []
    for i, (h, w) in enumerate(((320, 320), (64, 64), (17, 23), (320, 320), (8, 8), (40, 40))):
        scores = rng.normal(size=(h, w)).astype(np.float32)
        if i == 3:
            scores[:] = -1.0                     # empty prediction: intersection 0
        labels = np.zeros((h, w), dtype=np.float32)
        y0, x0 = rng.integers(0, h // 2), rng.integers(0, w // 2)
        labels[y0:y0 + h // 3 + 1, x0:x0 + w // 3 + 1] = 1.0
        if i == 4:
            scores = np.where(labels != 0, 2.0, -2.0).astype(np.float32)    # perfect prediction: IoU 1
        out[f"iou/{i}/scores"], out[f"iou/{i}/labels"] = scores, labels
        out[f"iou/{i}/meanIoU"] = np.float64(pt.compute_meanIoU(scores, labels))
        out[f"iou/{i}/accuracy"] = np.asarray(pt.compute_accuracy(scores, labels), dtype=np.float64)
        I, U = et.compute_mask_IU(scores > 0, labels != 0)
        out[f"iou/{i}/IU"] = np.asarray([I, U], dtype=np.int64)
        cases.append(i)
    out["iou/n"] = np.int64(len(cases))
    # a batch the in-graph mIoU (CMPC_model.py:486-490: mean over the batch of per-image I/U) is checked against
    B, H, W = 4, 64, 64
    up = rng.normal(size=(B, H, W, 1)).astype(np.float32)
    tg = np.zeros((B, H, W, 1), dtype=np.float32)
    for b in range(B):
        y0, x0 = rng.integers(0, H // 2, size=2)
        tg[b, y0:y0 + 20 + 3 * b, x0:x0 + 15 + 2 * b] = 1.0
    out["batch/up"], out["batch/target"] = up, tg
    out["batch/meanIoU_per_image"] = np.asarray([pt.compute_meanIoU(up[b], tg[b]) for b in range(B)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_processing_tools.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_processing_tools.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
