"""Host-side helpers around the hot path: what the reference's drivers do to a sample before
`sess.run` and to a mask after it.  Pure Python / NumPy, no GPU.

  tokeniser + padding   util/text_processing.py:9-67   (vocabulary file: one word per line, '<pad>' = line 0 and
                                                         '<unk>' present, e.g. data/vocabulary_Gref.txt)
  image feed            trainval_model.py:90-91,229-230,371   (RGB -> BGR, minus the channel means)
  mask IoU bookkeeping  util/eval_tools.py:31-35, trainval_model.py:267-296   (I/U, cumulative IoU, precision@X)

Resizing (util/im_processing.py:7-41) goes through scikit-image in the reference (third-party, not installed here, and its
`resize` defaults changed between releases).  `resize_and_pad_geometry` / `resize_and_crop_geometry` return the geometry those
functions compute (pinned by hand-worked cases); `resize_and_pad` / `resize_and_crop` fill it with a restatement of
`skimage.transform.resize` as scikit-image >= 0.19 documents it (order 1, mode "reflect", anti-aliasing Gaussian when shrinking,
uint8 scaled to [0, 1]) on scipy.ndimage -- the resampled pixel values are PARITY-UNPINNED (no scikit-image to compare with).
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

MU = np.array((104.00698793, 116.66876762, 122.67891434))          # trainval_model.py:371 (B, G, R)
UNK, PAD = "<unk>", "<pad>"
_SPLIT = re.compile(r"(\W+)")


def load_vocab_dict_from_file(path: str) -> Dict[str, int]:
    """word -> line number (util/text_processing.py:9-13)."""
    with open(path) as f:
        return {w.strip(): i for i, w in enumerate(f.readlines())}


def sentence2vocab_indices(sentence: str, vocab: Dict[str, int]) -> List[int]:
    """Split on runs of non-word characters (the separators are kept as tokens unless blank), lower-case, drop one
    trailing '.', map unknown tokens to '<unk>' (util/text_processing.py:17-25)."""
    toks = [t.lower() for t in _SPLIT.split(sentence.strip()) if t.strip()]
    if toks and toks[-1] == ".":
        toks = toks[:-1]
    unk = vocab[UNK]
    return [vocab.get(t, unk) for t in toks]


def preprocess_sentence_lstm(sentence: str, vocab: Dict[str, int], T: int) -> Tuple[List[int], int]:
    """The form CMPC_model is fed with (test.py:267): truncate to T, pad AT THE END with '<pad>', return the ids and the
    unpadded length = `seq_len` (util/text_processing.py:55-67)."""
    ids = sentence2vocab_indices(sentence, vocab)[:T]
    n = len(ids)
    return ids + [vocab[PAD]] * (T - n), n


def preprocess_sentence(sentence: str, vocab: Dict[str, int], T: int) -> List[int]:
    """Front-padded variant used by the non-LSTM readers (util/text_processing.py:42-53)."""
    ids = sentence2vocab_indices(sentence, vocab)[:T]
    return [vocab[PAD]] * (T - len(ids)) + ids


def image_feed(rgb: np.ndarray, mu: Sequence[float] = MU) -> np.ndarray:
    """uint8/float RGB [H,W,3] (or [B,H,W,3]) -> float32 BGR minus the per-channel means (trainval_model.py:90-91)."""
    return (np.asarray(rgb, dtype=np.float32)[..., ::-1] - np.asarray(mu, dtype=np.float32)).astype(np.float32)


def resize_and_pad_geometry(im_h: int, im_w: int, out_h: int, out_w: int) -> Tuple[int, int, int, int]:
    """(resized_h, resized_w, pad_top, pad_left) of im_processing.resize_and_pad (util/im_processing.py:7-23)."""
    scale = min(out_h / im_h, out_w / im_w)
    rh, rw = int(np.round(im_h * scale)), int(np.round(im_w * scale))
    return rh, rw, int(np.floor(out_h - rh) / 2), int(np.floor(out_w - rw) / 2)


def resize_and_crop_geometry(im_h: int, im_w: int, out_h: int, out_w: int) -> Tuple[int, int, int, int]:
    """(resized_h, resized_w, crop_top, crop_left) of im_processing.resize_and_crop (util/im_processing.py:25-41)."""
    scale = max(out_h / im_h, out_w / im_w)
    rh, rw = int(np.round(im_h * scale)), int(np.round(im_w * scale))
    return rh, rw, int(np.floor(rh - out_h) / 2), int(np.floor(rw - out_w) / 2)


def skimage_like_resize(im: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """skimage.transform.resize(im, [out_h, out_w]) as scikit-image >= 0.19 documents its defaults: integer images become floats in
    [0, 1] (img_as_float), bilinear (order=1), mode="reflect" (numpy.pad's: ndimage "mirror"), anti_aliasing on when an axis shrinks
    (Gaussian, sigma = (scale - 1) / 2 per shrinking axis), pixel-area grid (ndimage.zoom grid_mode=True); channels are not resized.
    Parity-unpinned (module header)."""
    from scipy import ndimage as ndi
    a = np.asarray(im)
    if a.dtype == np.bool_:
        a = a.astype(np.float64)
    elif np.issubdtype(a.dtype, np.integer):
        a = a.astype(np.float64) / float(np.iinfo(a.dtype).max)
    else:
        a = a.astype(np.float64)
    h, w = a.shape[:2]
    if (h, w) == (out_h, out_w):
        return a
    factors = (h / out_h, w / out_w)
    extra = (1,) * (a.ndim - 2)
    if max(factors) > 1:
        sigma = tuple(max(0.0, (f - 1) / 2) for f in factors) + (0,) * (a.ndim - 2)
        a = ndi.gaussian_filter(a, sigma, mode="mirror")
    return ndi.zoom(a, (out_h / h, out_w / w) + extra, order=1, mode="mirror", grid_mode=True)


def resize_and_pad(im: np.ndarray, input_h: int, input_w: int) -> np.ndarray:
    """im_processing.resize_and_pad (util/im_processing.py:7-23): aspect-preserving resize to fit, centred on a zero canvas."""
    rh, rw, top, left = resize_and_pad_geometry(im.shape[0], im.shape[1], input_h, input_w)
    r = skimage_like_resize(im, rh, rw)
    out = np.zeros((input_h, input_w) + r.shape[2:], dtype=r.dtype)
    out[top: top + rh, left: left + rw, ...] = r
    return out


def resize_and_crop(im: np.ndarray, input_h: int, input_w: int) -> np.ndarray:
    """im_processing.resize_and_crop (util/im_processing.py:25-41): aspect-preserving resize to cover, centre crop."""
    rh, rw, top, left = resize_and_crop_geometry(im.shape[0], im.shape[1], input_h, input_w)
    r = skimage_like_resize(im, rh, rw)
    return np.ascontiguousarray(r[top: top + input_h, left: left + input_w, ...])


def compute_mask_IU(masks: np.ndarray, target: np.ndarray) -> Tuple[int, int]:
    """|masks AND target|, |masks OR target| (util/eval_tools.py:31-35)."""
    if masks.shape[-2:] != target.shape[-2:]:
        raise ValueError("mask and target sizes differ: %s vs %s" % (masks.shape, target.shape))
    return int(np.logical_and(masks, target).sum()), int(np.logical_or(masks, target).sum())


class SegEval:
    """The evaluation loop's running numbers (trainval_model.py:198-203,267-296): cumulative I and U, per-sample IoU
    mean, and precision@{.5,.6,.7,.8,.9}."""

    def __init__(self, thresholds: Sequence[float] = (.5, .6, .7, .8, .9)):
        self.thresholds = tuple(thresholds)
        self.cum_I = self.cum_U = 0
        self.sum_iou, self.total = 0.0, 0
        self.correct = np.zeros(len(self.thresholds), dtype=np.int64)

    def add(self, pred_mask: np.ndarray, gt_mask: np.ndarray) -> Tuple[int, int]:
        I, U = compute_mask_IU(pred_mask, gt_mask)
        iou = float(I) / U
        self.cum_I += I
        self.cum_U += U
        self.sum_iou += iou
        self.correct += np.array([iou >= t for t in self.thresholds])
        self.total += 1
        return I, U

    def result(self) -> Dict[str, float]:
        out = {"precision@%s" % t: float(c) / self.total for t, c in zip(self.thresholds, self.correct)}
        out["overall_IoU"] = self.cum_I / self.cum_U
        out["mean_IoU"] = self.sum_iou / self.total
        return out
