"""CPU tests of the CMPCv5_BiLSTM oracle (BASELINE config 4): the two independent restatements (torch, NumPy float64) agree in both
batch-norm modes and with the HSV branch; known-answer properties derivable from the reference source hold
(CMPCv5_BiLSTM_model.py; citations in oracle/cmpc_v5_torch.py).  PARITY UNPINNED against TensorFlow (absent)."""
import numpy as np
import pytest
import torch

from oracle import cmpc_torch as O
from oracle import cmpc_v5_numpy as NP5
from oracle import cmpc_v5_torch as V


def _case(hsv, train, dtype=torch.float64, seed=0):
    torch.set_num_threads(2)
    cfg = V.tiny_cfg(hsv=hsv, train_mode=train)
    hp = V.init_head_params(cfg, dtype=dtype)
    bp = V.init_backbone_params(cfg, dtype=dtype)
    bn = V.init_bn_state(cfg, dtype=dtype)
    g = torch.Generator().manual_seed(5)
    for k in bn:                                   # non-trivial moving statistics so that the inference mode is exercised
        bn[k] = (torch.rand(bn[k].shape, generator=g, dtype=torch.float64) * 0.5 + (0.75 if k.endswith("variance") else -0.25)).to(dtype)
    for k in hp:                                   # non-zero biases / beta: padded words then do NOT vanish after conv + tanh
        if k.endswith(("biases", "beta")):
            hp[k] = (torch.randn(hp[k].shape, generator=g, dtype=torch.float64) * 0.05).to(dtype)
    words, im, sl, tgt = O.synth_batch(cfg, seed=seed)
    feats = V.backbone_taps(bp, im.to(dtype), cfg)
    return cfg, hp, bn, feats, words, im, sl, tgt


@pytest.mark.parametrize("hsv,train", [(False, True), (True, True), (False, False), (True, False)])
def test_numpy_and_torch_restatements_agree(hsv, train):
    cfg, hp, bn, feats, words, im, sl, tgt = _case(hsv, train)
    taps = V.head_forward(hp, bn, feats, words, sl, cfg, im=im.double())
    dims = dict(B=cfg.batch_size, T=cfg.num_steps, h=cfg.vf_h, w=cfg.vf_w, H=cfg.H, W=cfg.W, C=cfg.v_emb_dim, M=cfg.mlp_dim, R=cfg.rnn_size,
                hsv=hsv, train=train, rates=cfg.aspp_rates)
    tn = NP5.head_forward({k: v.numpy() for k, v in hp.items()}, {k: v.numpy() for k, v in bn.items()}, [f.numpy() for f in feats],
                          words.numpy(), sl.numpy(), dims, im=im.numpy())
    for k, v in tn.items():
        assert np.abs(v - taps[k].numpy()).max() < 1e-9 * max(1.0, np.abs(v).max()), k


def test_bilstm_reverse_sequence_semantics():
    """bidirectional_dynamic_rnn: the backward output at position t < len is the LSTM state after reading words len-1 .. t; positions
    past the length are zero in both directions; a sample of full length equals a plain time reversal."""
    cfg, hp, bn, feats, words, im, sl, tgt = _case(False, True)
    wf, mask, cat = V.bilstm(hp, words, sl, cfg)
    R, T = cfg.rnn_size, cfg.num_steps
    for b in range(cfg.batch_size):
        n = int(sl[b])
        assert torch.all(cat[b, 0, n:] == 0) and torch.all(mask[b, 0, :n] == 1) and torch.all(mask[b, 0, n:] == 0)
    emb = hp["text_objseg/Variable"][words.long()]
    pre = "text_objseg/bidirectional_rnn/bw/lstm_cell/"
    b = 0
    assert int(sl[b]) == T
    plain = V._lstm_dir(torch.flip(emb[b:b + 1], dims=[1]), hp[pre + "kernel"], hp[pre + "bias"], sl[b:b + 1], R)
    assert torch.allclose(torch.flip(plain, dims=[1])[0], cat[b, 0, :, R:], atol=1e-12)
    x = torch.arange(2 * 5 * 3, dtype=torch.float64).view(2, 5, 3)
    r = V.reverse_sequence(x, torch.tensor([3, 5]))
    assert torch.equal(r[0, :3], x[0, :3].flip(0)) and torch.equal(r[0, 3:], x[0, 3:]) and torch.equal(r[1], x[1].flip(0))
    assert torch.equal(V.reverse_sequence(r, torch.tensor([3, 5])), x)            # an involution


def test_padded_words_after_words_feat_conv():
    """v5:181-185: seq_mask comes from the raw BiLSTM outputs, but words_feat of a padded word is l2norm(tanh(bias)) -- NOT zero once
    the bias is trained; it still carries no weight anywhere, because words_parse and both graph softmaxes are masked."""
    cfg, hp, bn, feats, words, im, sl, tgt = _case(False, True)
    taps = V.head_forward(hp, bn, feats, words, sl, cfg)
    b = 1
    n = int(sl[b])
    assert n < cfg.num_steps and float(taps["words_feat"][b, 0, n:].abs().max()) > 0
    assert torch.all(taps["words_parse"][b, 0, n:] == 0)
    for lv in V.LEVELS:
        assert torch.all(taps[f"gw_w_{lv}"][b, :, n:] == 0) and torch.all(taps[f"gw_v_{lv}"][b, :, n:] == 0)
        # mask AFTER softmax_T (v5:486-487): rows of gw_w sum to (valid words' share) <= 1, < 1 for the padded sample
        s = taps[f"gw_w_{lv}"].sum(2)
        assert torch.all(s <= 1 + 1e-12) and float(s[b].max()) < 1 and torch.allclose(s[0], torch.ones_like(s[0]), atol=1e-12)


def test_rgb_to_hsv_known_answers():
    x = torch.tensor([[255., 0, 0], [0, 255., 0], [0, 0, 255.], [10., 10, 10], [0, 0, 0], [255., 0, 255.], [200., 100, 50]], dtype=torch.float64)
    h = V.rgb_to_hsv(x)
    exp = torch.tensor([[0, 1, 255.], [1 / 3, 1, 255.], [2 / 3, 1, 255.], [0, 0, 10.], [0, 0, 0], [5 / 6, 1, 255.], [(100 - 50) / (6 * 150), 0.75, 200.]],
                       dtype=torch.float64)
    assert torch.allclose(h, exp, atol=1e-12)
    assert np.allclose(NP5._hsv(x.numpy()), exp.numpy(), atol=1e-12)


def test_batch_norm_modes_and_moving_statistics():
    """Training mode normalises with batch statistics (every BN+relu input has zero mean / unit biased variance per channel before the
    affine), updates moving = decay * moving + (1 - decay) * batch with the UNBIASED variance; inference mode leaves the state alone."""
    cfg, hp, bn, feats, words, im, sl, tgt = _case(False, True)
    new = {}
    V.head_forward(hp, bn, feats, words, sl, cfg, new_state=new)
    assert set(new) == set(bn)
    # recompute one layer by hand: decoder/low_level_features on c2
    sc = "decoder/low_level_features/conv_1x1"
    y = feats[0] @ hp[f"text_objseg/{sc}/weights"][0, 0]
    n = y.shape[0] * y.shape[1] * y.shape[2]
    mean, var = y.mean(dim=(0, 1, 2)), y.var(dim=(0, 1, 2), unbiased=False)
    d = cfg.batch_norm_decay
    assert torch.allclose(new[f"text_objseg/{sc}/BatchNorm/moving_mean"], bn[f"text_objseg/{sc}/BatchNorm/moving_mean"] * d + mean * (1 - d), atol=1e-12)
    assert torch.allclose(new[f"text_objseg/{sc}/BatchNorm/moving_variance"],
                          bn[f"text_objseg/{sc}/BatchNorm/moving_variance"] * d + var * n / (n - 1) * (1 - d), atol=1e-12)
    cfg_e, hp_e, bn_e, feats_e, *_ = _case(False, False)
    new_e = {}
    V.head_forward(hp_e, bn_e, feats_e, words, sl, cfg_e, new_state=new_e)
    assert new_e == {}


def test_manifest_and_train_step():
    """Variable names / order follow the graph's creation order; 70.26 M trainable scalars at config 4's sizes (HSV variant); one
    TF-Adam step moves every trainable variable and the moving statistics."""
    cfg = V.Cfg(batch_size=1, num_steps=25, vf_h=64, vf_w=64, H=512, W=512, hsv=True)
    specs = V.head_param_specs(cfg)
    names = [s[0] for s in specs]
    assert names[0] == "text_objseg/Variable" and names[1] == "text_objseg/bidirectional_rnn/fw/lstm_cell/kernel"
    assert names.index("text_objseg/words_feat/DW") < names.index("text_objseg/c5_lateral/DW") < names.index("text_objseg/score_c5/DW")
    assert names[-1] == "text_objseg/decoder/upsampling_logits/conv_1x1/biases" and "text_objseg/c3_lateral/DW" not in names
    assert dict((s[0], s[1]) for s in specs)["text_objseg/c5_lateral/DW"] == (1, 1, 2051, 1000)             # hsv:129
    assert sum(int(np.prod(s[1])) for s in specs) == 70262763
    reg = [s[0] for s in specs if "reg" in s[3]]
    assert "text_objseg/aspp/conv_3x3_2/weights" in reg and "text_objseg/aspp/conv_1x1/BatchNorm/gamma" not in reg       # v5:530
    assert [s[0] for s in specs if "x2" in s[3]][-1] == "text_objseg/decoder/upsampling_logits/conv_1x1/biases"           # v5:561
    cfg, hp, bn, feats, words, im, sl, tgt = _case(True, True, dtype=torch.float32)
    before = {k: v.clone() for k, v in hp.items()}
    bn0 = {k: v.clone() for k, v in bn.items()}
    opt = O.TFAdam(hp)
    scal = V.train_step(hp, bn, opt, 0, feats, words, sl, tgt, cfg, im=im)
    assert abs(scal["loss_all"] - (0.8 * scal["loss_last"] + 0.1 * scal["loss_c5"] + 0.1 * scal["loss_c4"])) < 1e-3 * scal["loss_all"]   # v5:541-542
    assert all(not torch.equal(hp[k], before[k]) for k in hp if k != "text_objseg/Variable")
    assert all(not torch.equal(bn[k], bn0[k]) for k in bn)
