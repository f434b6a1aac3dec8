"""conv5=True (reference CMPC_model.py:427-430: the res3 / res4 / res5 convolution weights train with the head): the product's backbone
backward (cmpc-refseg_amd/backbone_train.py, the library's kernels through the op-level C ABI) and the taps' gradients the handle returns
(cmpc_cfg.conv5, taps "dc5" / "dc4" / "dc3") against autograd through the oracle's backbone + head.  Parity unpinned against TensorFlow."""
import numpy as np
import pytest
import torch

from tests import util as U
from tests.util import O

pytestmark = pytest.mark.gpu


def cfg32():
    # backbone width 32: every res3-res5 convolution has cin % 64 == 0 (the library's convolution path); taps 256 / 512 / 1024 channels
    return O.Cfg(batch_size=2, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=1024, c4_dim=512, c3_dim=256, vocab_size=50, v_emb_dim=40, mlp_dim=24,
                 rnn_size=40, glove_dim=12, parse_dim=20, backbone_width=32, backbone_blocks=(1, 2, 2, 1))


def make():
    torch.set_num_threads(8)
    cfg = cfg32()
    hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
    words, im, sl, tgt = O.synth_batch(cfg)
    return cfg, hp, bp, words, im, sl, tgt


def test_conv5_gradients_fp32_match_autograd():
    cfg, hp, bp, words, im, sl, tgt = make()
    scal, grads, gb = O.grads_of_conv5(hp, bp, im, words, sl, tgt, cfg)
    P = U.pkg()
    m = P.LSTM_model(head_params=hp, backbone_params=bp, conv5=True, **U.model_kwargs(cfg, "f32"))
    tr = m.bb_trainer
    assert sorted(L["name"] for L in tr.layers) == sorted(gb)
    imd = torch.as_tensor(im).to(m.device)
    feats = tr.forward(imd)
    ref_feats = O.backbone_forward(bp, torch.as_tensor(im), cfg)
    for f, r in zip(feats, ref_feats):
        assert U.rel_err(f.float().cpu(), r) < 2e-5
    o = m.loss_and_grads(feats, words, tgt, sl)
    torch.cuda.synchronize()
    assert abs(float(o["loss_all"]) - scal["loss_all"]) <= 1e-4 * abs(scal["loss_all"])
    B, h, w = cfg.batch_size, cfg.vf_h, cfg.vf_w
    tr.backward({5: m.eng.tap("dc5").view(B, h, w, -1), 4: m.eng.tap("dc4").view(B, h, w, -1), 3: m.eng.tap("dc3").view(B, h, w, -1)})
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for L in tr.layers:
        g = tr.view(tr.grads, L).cpu() / m.eng.loss_scale
        ref = gb[L["name"]] - cfg.weight_decay * bp[L["name"]]           # the product adds the L2 term inside the Adam kernel
        err = U.rel_err(g, ref)
        worst = max(worst, (L["name"], err), key=lambda kv: kv[1])
        assert err < 2e-4, (L["name"], err)
    print("conv5 fp32 worst backbone gradient error:", worst)
    # the head's own gradients are unchanged by the extra outputs
    gh = m.store.grad_dict()
    flags = {k: f for k, _, _, f in O.head_param_specs(cfg)}
    for n in ("text_objseg/c5_lateral/DW", "text_objseg/c3_lateral/DW", "text_objseg/fusion_c4/DW"):
        ref = grads[n] / (2.0 if "x2" in flags[n] else 1.0) - cfg.weight_decay * hp[n]
        assert U.rel_err(gh[n], ref) < 2e-4, n


def test_conv5_train_steps_and_checkpoint(tmp_path):
    cfg, hp, bp, words, im, sl, tgt = make()
    hp_o = {k: v.clone() for k, v in hp.items()}
    bp_o = {k: v.clone() for k, v in bp.items()}
    opt, opt_b = O.TFAdam(hp_o), O.TFAdam({n: bp_o[n] for n in O.conv5_trainable(bp_o)})
    P = U.pkg()
    m = P.LSTM_model(head_params=hp, backbone_params=bp, conv5=True, **U.model_kwargs(cfg, "f32"))
    for step in range(2):
        ref = O.train_step_conv5(hp_o, bp_o, opt, opt_b, step, torch.as_tensor(im), words, sl, tgt, cfg)
        s, sc = m.train_step(words, im, tgt, sl)
        torch.cuda.synchronize()
        assert s == step + 1 and abs(float(sc["loss_all"]) - ref["loss_all"]) <= 2e-4 * abs(ref["loss_all"]), (step, float(sc["loss_all"]), ref["loss_all"])
    got = m.bb_trainer.named_weights()
    for n in O.conv5_trainable(bp):
        d = (got[n] - bp_o[n]).abs().flatten()
        moved = (bp[n] - bp_o[n]).abs().max()
        assert float(moved) > 0.5 * cfg.start_lr                                     # the weights did train
        # Adam's first steps move every element by ~lr: elements whose tiny gradient changes sign under rounding differ by up to 2 lr
        assert float(torch.quantile(d[:200000], 0.99)) <= 0.5 * cfg.start_lr and float(d.max()) <= 4.5 * cfg.start_lr, (n, float(d.max()))
    # frozen parts stayed frozen
    for n in bp:
        if n.startswith(("conv1", "res2", "bn")):
            assert torch.equal(torch.as_tensor(m.backbone_vars[n]), bp[n]), n
    # checkpoint round trip: trained backbone weights + their Adam slots
    CK = __import__("importlib").import_module("cmpc-refseg_amd.checkpoint")
    saver = CK.Saver(fmt="npz")
    path = saver.save(m, str(tmp_path / "c5"), global_step=2)
    m2 = P.LSTM_model(head_params=hp, backbone_params=bp, conv5=True, **U.model_kwargs(cfg, "f32"))
    CK.Saver().restore(m2, path)
    for n, v in m.bb_trainer.named_weights().items():
        assert torch.equal(m2.bb_trainer.named_weights()[n], v), n
    for n, v in m.bb_trainer.named_slots().items():
        assert torch.equal(m2.bb_trainer.named_slots()[n], v), n
    a = m.train_step(words, im, tgt, sl)[1]
    b = m2.train_step(words, im, tgt, sl)[1]
    torch.cuda.synchronize()
    assert float(a["loss_all"]) == float(b["loss_all"])


def test_conv5_full_size_f16_steps():
    """BASELINE's sizes (320x320, L = 20, ResNet-101, B = 2) with conv5=True in the default f16 storage: two train steps run, the loss is
    finite, no gradient overflowed, the res3-res5 weights moved and the frozen variables did not."""
    from bench import synth_batch
    P = U.pkg()
    m = P.LSTM_model(batch_size=2, mode="train", conv5=True)
    w, im, sl, tg = (torch.from_numpy(x) for x in synth_batch(2, 20, 320, 320, m.cfg.vocab_size, 5))
    before = {k: torch.as_tensor(v).clone() for k, v in m.backbone_vars.items()}
    for _ in range(2):
        s, sc = m.train_step(w, im, tg, sl)
    torch.cuda.synchronize()
    assert s == 2 and np.isfinite(float(sc["loss_all"])) and m.grad_nonfinite() == 0 and int(m.bb_trainer.nonfinite.item()) == 0
    after = m.backbone_vars
    moved = [n for n in before if not torch.equal(torch.as_tensor(after[n]), before[n])]
    assert sorted(moved) == sorted(n for n in before if n.startswith(("res3", "res4", "res5")))
    assert len(moved) == 3 * 30 + 3            # 30 bottlenecks x (2a, 2b, 2c) + one branch1 per stage


def _backbone_grad_check(m, gb, bp, wd, tol=3e-4):
    tr = m.bb_trainer
    assert sorted(L["name"] for L in tr.layers) == sorted(gb)
    worst = ("", 0.0)
    for L in tr.layers:
        g = tr.view(tr.grads, L).cpu() / m.eng.loss_scale
        ref = gb[L["name"]] - wd * bp[L["name"]]
        err = U.rel_err(g, ref)
        worst = max(worst, (L["name"], err), key=lambda kv: kv[1])
        if err >= tol:
            print("  backbone gradient", L["name"], "error %.2e" % err)
    assert worst[1] < tol, worst
    return worst


def test_conv5_v5_model_gradients_fp32():
    """conv5=True of the CMPCv5_BiLSTM graph (v5:521-525): taps res2b (frozen side), res4b22, res5c; tanh laterals with the HSV K-segment."""
    import tests.test_gpu_v5 as T5
    from oracle import cmpc_v5_torch as V
    torch.set_num_threads(8)
    cfg = V.Cfg(batch_size=4, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=1024, c4_dim=512, c3_dim=256, vocab_size=50, v_emb_dim=40, mlp_dim=24,
                rnn_size=40, glove_dim=12, parse_dim=20, backbone_width=32, backbone_blocks=(2, 1, 2, 1), hsv=True, aspp_depth=16, low_dim=8,
                aspp_rates=(1, 3, 6), train_mode=True)
    hp, bp, bn = V.init_head_params(cfg), V.init_backbone_params(cfg), V.init_bn_state(cfg)
    words, im, sl, tgt = O.synth_batch(cfg, seed=3)
    im_t = torch.as_tensor(im)
    names_b = O.conv5_trainable(bp)
    # the oracle in float64: the image-level batch-norm of the ASPP (moments over the B samples of a 1x1 map) is ill-conditioned at small B and
    # every backbone gradient passes through it -- fp32 autograd differs from itself by percents there (tests/test_gpu_v5.py has the same note)
    D = lambda d: {k: v.double() for k, v in d.items()}
    bl = D(bp)
    for n in names_b:
        bl[n] = bl[n].detach().clone().requires_grad_(True)
    hp64, bn64, im64 = D(hp), D(bn), im_t.double()
    taps = V.head_forward(hp64, bn64, V.backbone_taps(bl, im64, cfg), words, sl, cfg, im=im64, new_state={})
    cost = V.losses(hp64, taps, torch.as_tensor(tgt).double(), cfg)["cost"] + cfg.weight_decay * sum(0.5 * (bl[n] ** 2).sum() for n in names_b)
    gb = {n: g.float() for n, g in zip(names_b, torch.autograd.grad(cost, [bl[n] for n in names_b]))}
    P = U.pkg()
    m = P.get_segmentation_model("CMPCv5_BiLSTM_HSV_model", head_params=hp, backbone_params=bp, conv5=True, **T5.model_kwargs(cfg, "f32", "train"))
    m.load_extra_vars({k: v.numpy() for k, v in bn.items()})
    imd = im_t.to(m.device)
    feats = m.bb_trainer.forward(imd)
    m.loss_and_grads(feats, words, tgt, sl, im=imd)
    h, w = cfg.vf_h, cfg.vf_w
    m.bb_trainer.backward({5: m.eng.tap("dc5").view(4, h, w, -1), 4: m.eng.tap("dc4").view(4, h, w, -1)})
    torch.cuda.synchronize()
    print("conv5 (CMPCv5_BiLSTM_HSV) worst backbone gradient error:", _backbone_grad_check(m, gb, bp, cfg.weight_decay, tol=2e-3))
    s, sc = m.train_step(words, im, tgt, sl)
    assert s == 1 and np.isfinite(float(sc["loss_all"]))


def test_finetune_video_model_gradients_fp32():
    """finetune=True of CMPC_video_mm_tgraph_allvec (vid:554-557): the 5 sampled frames' taps get gradients from the per-frame laterals."""
    from oracle import cmpc_video_torch as VD
    torch.set_num_threads(8)
    cfg = VD.Cfg(batch_size=1, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=1024, c4_dim=512, c3_dim=256, vocab_size=50, v_emb_dim=40, mlp_dim=24,
                 rnn_size=40, glove_dim=12, parse_dim=20, backbone_width=32, backbone_blocks=(1, 2, 2, 1))
    hp, bp = VD.init_head_params(cfg), O.init_backbone_params(cfg)
    words, clip, tgt = VD.synth_clip(cfg, seed=1)
    vi = torch.tensor([[int((words[0] == 0).sum())]], dtype=torch.int32)
    names_b = O.conv5_trainable(bp)
    bl = dict(bp)
    for n in names_b:
        bl[n] = bp[n].detach().clone().requires_grad_(True)
    taps = VD.head_forward(hp, VD.backbone_taps(bl, clip, cfg), words, cfg)
    cost = VD.losses(hp, taps, tgt, cfg)["cost"] + cfg.weight_decay * sum(0.5 * (bl[n] ** 2).sum() for n in names_b)
    gb = dict(zip(names_b, torch.autograd.grad(cost, [bl[n] for n in names_b])))
    P = U.pkg()
    m = P.get_segmentation_model("CMPC_video_mm_tgraph_allvec", head_params=hp, backbone_params=bp, finetune=True, frames=cfg.frames, **U.model_kwargs(cfg, "f32"))
    we, sl, fr = m._video_feeds(words, vi, clip)
    feats = m.bb_trainer.forward(fr)
    m.loss_and_grads(feats, we, tgt, sl)
    h, w, Fr = cfg.vf_h, cfg.vf_w, cfg.sample_frames
    m.bb_trainer.backward({5: m.eng.tap("dc5").view(Fr, h, w, -1), 4: m.eng.tap("dc4").view(Fr, h, w, -1), 3: m.eng.tap("dc3").view(Fr, h, w, -1)})
    torch.cuda.synchronize()
    print("finetune (CMPC_video) worst backbone gradient error:", _backbone_grad_check(m, gb, bp, cfg.weight_decay))
    s, sc = m.train_step_video(words, None, tgt, vi, clip)
    assert s == 1 and np.isfinite(float(sc["loss_all"]))
