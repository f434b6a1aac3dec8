"""CPU tests of the oracle itself: the two independent restatements agree, the known-answer
properties derivable from the reference source hold, and the committed
golden vectors are reproduced."""
import math
import os

import numpy as np
import pytest
import torch

from tests import util as U
from tests.util import O
from oracle import cmpc_numpy as NP

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tiny64():
    torch.set_num_threads(2)
    cfg = U.tiny_cfg()
    hp = O.init_head_params(cfg, dtype=torch.float64)
    bp = O.init_backbone_params(cfg, dtype=torch.float64)
    words, im, sl, tgt = O.synth_batch(cfg)
    feats = O.backbone_forward(bp, im.double(), cfg)
    taps = O.head_forward(hp, feats, words, sl, cfg)
    return cfg, hp, feats, words, sl, tgt, taps


def test_numpy_and_torch_restatements_agree(tiny64):
    cfg, hp, feats, words, sl, tgt, taps = tiny64
    dims = dict(B=cfg.batch_size, T=cfg.num_steps, h=cfg.vf_h, w=cfg.vf_w, H=cfg.H, W=cfg.W, C=cfg.v_emb_dim,
                M=cfg.mlp_dim, R=cfg.rnn_size)
    tn = NP.head_forward({k: v.numpy() for k, v in hp.items()}, [f.numpy() for f in feats], words.numpy(), sl.numpy(), dims)
    for k, v in tn.items():
        assert np.abs(v - taps[k].numpy()).max() < 1e-11, k


def test_backbone_same_padding_matches_numpy():
    rng = np.random.default_rng(0)
    for (hw, k, s, d) in ((9, 7, 2, 1), (10, 3, 1, 2), (8, 3, 1, 4), (7, 1, 2, 1)):
        x = rng.normal(size=(1, hw, hw, 3)); w = rng.normal(size=(k, k, 3, 4))
        a = NP.conv_same(x, w, s, d)
        b = O.tf_conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), s, d).permute(0, 2, 3, 1).numpy()
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-12


def test_spatial_grid_kat():
    # util/processing_tools.py:5-17
    g = O.generate_spatial_batch(1, 40, 40)
    assert np.allclose(g[0, 0, 0].numpy(), [-1, -1, -0.95, -0.95, -0.975, -0.975, 0.025, 0.025])
    assert np.allclose(g[0, 39, 39].numpy(), [0.95, 0.95, 1.0, 1.0, 0.975, 0.975, 0.025, 0.025])


def test_adjacency_rows_sum_to_one_and_masks(tiny64):
    cfg, hp, feats, words, sl, tgt, taps = tiny64
    B, T = cfg.batch_size, cfg.num_steps
    for lv in ("c5", "c4", "c3"):
        adj = taps[f"gw_w_{lv}"] @ taps[f"gw_v_{lv}"].transpose(1, 2)      # CMPC_model.py:400-401
        assert torch.allclose(adj.sum(2), torch.ones_like(adj.sum(2)), atol=1e-10)
    for b in range(B):
        n = int(sl[b])
        assert torch.all(taps["words_feat"][b, 0, n:] == 0)                 # dynamic_rnn zero outputs past length
        assert torch.all(taps["seq_mask"][b, 0, :n] == 1) and torch.all(taps["seq_mask"][b, 0, n:] == 0)
        assert torch.all(taps["gw_w_c5"][b, :, n:] == 0) and torch.all(taps["gw_v_c5"][b, :, n:] == 0)
    assert torch.allclose(taps["words_parse"].sum(3, keepdim=True), taps["seq_mask"], atol=1e-12)


def test_unit_norms(tiny64):
    cfg, hp, feats, words, sl, tgt, taps = tiny64
    for k in ("lat_c5", "vis_la_sp_c4", "spa_graph_c3", "exg_c3", "exg_c5_2"):
        n = taps[k].pow(2).sum(-1)
        assert torch.allclose(n, torch.ones_like(n), atol=1e-9), k


def test_poly_lr_endpoints():
    cfg = O.Cfg()
    assert abs(O.poly_lr(0, cfg) - 2.5e-4) < 1e-15                        # CMPC_model.py:451-452
    assert abs(O.poly_lr(800000, cfg) - 1e-5) < 1e-15
    assert abs(O.poly_lr(10 ** 7, cfg) - 1e-5) < 1e-15
    assert abs(O.poly_lr(400000, cfg) - (2.4e-4 * 0.5 ** 0.9 + 1e-5)) < 1e-15


def test_resize_bilinear_legacy_kat():
    x = torch.arange(4, dtype=torch.float32).view(1, 2, 2, 1)            # [[0,1],[2,3]]
    y = O.resize_bilinear(x, 4, 4)[0, :, :, 0]
    # src = dst * 0.5: rows/cols 0, .5, 1, 1(clamped: hi=min(lo+1,1))
    ref = torch.tensor([[0, .5, 1, 1], [1, 1.5, 2, 2], [2, 2.5, 3, 3], [2, 2.5, 3, 3]])
    assert torch.allclose(y, ref)
    c = torch.full((1, 5, 5, 2), 3.25)
    assert torch.all(O.resize_bilinear(c, 40, 40) == 3.25)


def test_sigmoid_xent_and_adam_kat():
    x = torch.tensor([-3.0, 0.0, 2.5]); z = torch.tensor([0.0, 1.0, 1.0])
    ref = -(z * torch.log(torch.sigmoid(x)) + (1 - z) * torch.log(1 - torch.sigmoid(x)))
    assert torch.allclose(O.sigmoid_xent(x, z), ref, atol=1e-6)
    p = {"a": torch.tensor([1.0, -2.0])}
    opt = O.TFAdam(p)
    opt.step(p, {"a": torch.tensor([0.5, -4.0])}, 0.1)
    # first TF-Adam step moves every coordinate by lr * g / (|g| + eps*sqrt(1-b2))  ~ lr * sign(g)
    assert torch.allclose(p["a"], torch.tensor([0.9, -1.9]), atol=1e-6)


def test_param_manifest_total():
    assert sum(int(np.prod(s)) for _, s, _, _ in O.head_param_specs(O.Cfg())) == 76055608


def test_oracle_reproduces_golden():
    g = np.load(os.path.join(HERE, "golden", "tiny_case.npz"))
    torch.set_num_threads(1)
    cfg = U.tiny_cfg()
    hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
    words, im, sl, tgt = (torch.from_numpy(g[k]) for k in ("words", "im", "seq_len", "target"))
    feats = O.backbone_forward(bp, im, cfg)
    for i, n in enumerate(("c3", "c4", "c5")):
        # the fixture was written by this oracle in round 1; torch may pick another fp32 conv algorithm, so the bound is relative to the map's scale
        assert np.abs(feats[i].numpy() - g["feat_" + n]).max() <= 1e-5 * np.abs(g["feat_" + n]).max()
    scal, grads, taps = O.grads_of(hp, feats, words, sl, tgt, cfg)
    for k in g.files:
        if k.startswith("tap/"):
            assert np.allclose(taps[k[4:]].numpy(), g[k], rtol=1e-3, atol=1e-5), k
        elif k.startswith("grad/"):
            ref = g[k]
            assert np.abs(grads[k[5:]].numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, k
        elif k.startswith("scal/"):
            assert abs(scal[k[5:]] - float(g[k])) <= 1e-4 * abs(float(g[k])) + 1e-9, k
