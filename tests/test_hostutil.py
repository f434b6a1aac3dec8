"""Host helpers (tokeniser / feeds / IoU bookkeeping): known answers worked out by hand from the reference's rules
(util/text_processing.py:17-67, trainval_model.py:90-91,267-296, util/im_processing.py:7-41, util/eval_tools.py:31-35)."""
import importlib

import numpy as np
import pytest

H = importlib.import_module("cmpc-refseg_amd.hostutil")
VOCAB = {w: i for i, w in enumerate(["<pad>", "<go>", "<eos>", "<unk>", "the", "man", "in", "red", ",", "left", "-", "most", "'", "s"])}


def test_tokeniser_rules():
    # separators survive as tokens unless blank -- WITH their surrounding blanks, so ", " is not "," and falls to <unk>
    # (util/text_processing.py:18-19 tests w.strip() but keeps w); lower-cased; one trailing "." dropped
    assert H.sentence2vocab_indices("The man, in red.", VOCAB) == [4, 5, 3, 6, 7]
    assert H.sentence2vocab_indices("man,red", VOCAB) == [5, 8, 7]
    assert H.sentence2vocab_indices("  LEFT-most   zebra ", VOCAB) == [9, 10, 11, 3]
    assert H.sentence2vocab_indices("man's", VOCAB) == [5, 12, 13]
    assert H.sentence2vocab_indices("red .", VOCAB) == [7, 3]             # " ." is not "."
    assert H.sentence2vocab_indices("red ...", VOCAB) == [7, 3]           # '...' is one separator token, not '.'


def test_padding_forms():
    ids, n = H.preprocess_sentence_lstm("the man in red", VOCAB, 6)
    assert ids == [4, 5, 6, 7, 0, 0] and n == 4                               # pad at the END, seq_len = 4
    ids, n = H.preprocess_sentence_lstm("the man in red the man in red", VOCAB, 6)
    assert ids == [4, 5, 6, 7, 4, 5] and n == 6                               # truncated to T
    assert H.preprocess_sentence("the man", VOCAB, 5) == [0, 0, 0, 4, 5]      # front-padded variant


def test_vocab_file(tmp_path):
    p = tmp_path / "vocab.txt"
    p.write_text("<pad>\n<go>\n<eos>\n<unk>\nzebra \n")
    v = H.load_vocab_dict_from_file(str(p))
    assert v["<pad>"] == 0 and v["<unk>"] == 3 and v["zebra"] == 4


def test_image_feed():
    rgb = np.zeros((2, 3, 3), dtype=np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 10, 20, 30
    x = H.image_feed(rgb)
    assert x.dtype == np.float32 and x.shape == (2, 3, 3)
    np.testing.assert_allclose(x[0, 0], np.array([30, 20, 10]) - H.MU, rtol=0, atol=1e-5)     # B, G, R order


def test_resize_geometry():
    assert H.resize_and_pad_geometry(480, 640, 320, 320) == (240, 320, 40, 0)
    assert H.resize_and_pad_geometry(333, 500, 320, 320) == (213, 320, 53, 0)
    assert H.resize_and_crop_geometry(320, 320, 480, 640) == (640, 640, 80, 0)
    assert H.resize_and_crop_geometry(320, 320, 333, 500) == (500, 500, 83, 0)


def test_seg_eval_accumulators():
    gt = np.zeros((4, 4), bool); gt[:2] = True                      # 8 pixels
    p1 = np.zeros((4, 4), bool); p1[:2, :2] = True                  # I=4, U=8 -> 0.5
    p2 = gt.copy()                                                  # I=8, U=8 -> 1.0
    p3 = np.zeros((4, 4), bool); p3[1:3] = True                     # I=4, U=12 -> 1/3
    ev = H.SegEval()
    assert ev.add(p1, gt) == (4, 8) and ev.add(p2, gt) == (8, 8) and ev.add(p3, gt) == (4, 12)
    r = ev.result()
    assert r["overall_IoU"] == pytest.approx(16 / 28) and r["mean_IoU"] == pytest.approx((0.5 + 1.0 + 1 / 3) / 3)
    assert r["precision@0.5"] == pytest.approx(2 / 3) and r["precision@0.6"] == pytest.approx(1 / 3) and r["precision@0.9"] == pytest.approx(1 / 3)
    with pytest.raises(ValueError):
        H.compute_mask_IU(np.zeros((3, 4), bool), gt)
