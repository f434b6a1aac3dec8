"""Generates tests/golden/tiny_case.npz: seeded inputs, parameters and the oracle's outputs /
gradients for the shrunken configuration (tests/util.tiny_cfg).  PARITY UNPINNED: the vectors come
from our own restatement (oracle/cmpc_torch.py), because the reference (TF1) cannot run here and
holds no fixtures of its own; they pin the oracle against regressions and travel to the GPU box.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import util as U          # noqa: E402
from tests.util import O             # noqa: E402

KEEP_TAPS = ("words_parse", "gw_w_c5", "gw_v_c5", "vis_la_sp_c4", "spa_graph_c3", "fusion_c5", "exg_c4_2", "fused",
             "pred", "up", "up_c3", "up_c4", "up_c5")
KEEP_GRADS = ("text_objseg/fusion_c5/DW", "text_objseg/rnn/lstm_cell/kernel", "text_objseg/words_parse_2/DW",
              "text_objseg/rnn/conv_lstm_cell/W_co", "text_objseg/gconv_feat_ln_spa_graph_c4/gamma",
              "text_objseg/vis_trans_c3_head2/DW", "text_objseg/spa_graph_key_c4_2gv_f1/DW", "text_objseg/score/DW")


def main():
    torch.set_num_threads(1)
    cfg = U.tiny_cfg()
    hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
    words, im, sl, tgt = O.synth_batch(cfg)
    feats = O.backbone_forward(bp, im, cfg)
    scal, grads, taps = O.grads_of(hp, feats, words, sl, tgt, cfg)
    out = {"words": words.numpy(), "im": im.numpy(), "seq_len": sl.numpy(), "target": tgt.numpy()}
    for i, n in enumerate(("c3", "c4", "c5")):
        out["feat_" + n] = feats[i].numpy()
    for k in KEEP_TAPS:
        out["tap/" + k] = taps[k].numpy()
    for k in KEEP_GRADS:
        out["grad/" + k] = grads[k].numpy()
    for k, v in scal.items():
        out["scal/" + k] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, "tiny_case.npz"), **out)
    print("wrote tiny_case.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
