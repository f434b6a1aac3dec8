"""GPU parity tests of the CMPC_video path (BASELINE.json config 5; reference CMPC_video/CMPC_video_mm_tgraph_allvec.py): the HIP path
through the C ABI against the oracle (oracle/cmpc_video_torch.py) on identical seeded inputs -- one synthetic A2D-style clip (batch 1: the
reference graph is only valid for one sample), 5 sampled frames through the backbone, front-padded words + valid_idx as the driver feeds
them (trainval_video.py:93-101).  PARITY UNPINNED against TensorFlow itself (absent), like the rest of the path."""
import numpy as np
import pytest
import torch

from tests import util as U
from oracle import cmpc_torch as O
from oracle import cmpc_video_torch as VD

pytestmark = pytest.mark.gpu
NAME = "CMPC_video_mm_tgraph_allvec"


def make_case(seed=1):
    torch.set_num_threads(8)
    cfg = VD.tiny_cfg()
    hp, bp = VD.init_head_params(cfg), O.init_backbone_params(cfg)
    g = torch.Generator().manual_seed(3)
    for k in hp:
        if k.endswith(("biases", "beta")):
            hp[k] = (torch.randn(hp[k].shape, generator=g) * 0.05).float()
    words, clip, tgt = VD.synth_clip(cfg, seed=seed)
    feats = VD.backbone_taps(bp, clip, cfg)
    vi = torch.tensor([[int((words[0] == 0).sum())]], dtype=torch.int32)
    return dict(cfg=cfg, hp=hp, bp=bp, words=words, clip=clip, tgt=tgt, feats=feats, vi=vi)


def build(case, dtype, mode="train"):
    P = U.pkg()
    kw = U.model_kwargs(case["cfg"], dtype, mode)
    return P.get_segmentation_model(NAME, head_params=case["hp"], backbone_params=case["bp"], frames=case["cfg"].frames, **kw)


def taps_as_oracle(o, cfg, n):
    T, h, w, C, M, Fr = cfg.num_steps, cfg.vf_h, cfg.vf_w, cfg.v_emb_dim, cfg.mlp_dim, cfg.sample_frames
    f = lambda x: x.detach().float().cpu()
    out = {"words_feat": f(o["words_feat"])[:n, :C].reshape(1, 1, n, C), "words_parse": f(o["words_parse"])[:n].reshape(1, 1, n, 5),
           "nec_lang": f(o["nec_lang"])[:, :C].reshape(1, 1, 1, C)}
    for lv in VD.LEVELS:
        out[f"lat_{lv}"] = U.unpad_map(o[f"lat_{lv}"], Fr, h, w, C)
        out[f"mm_{lv}"] = U.unpad_map(o[f"mm_{lv}"], Fr, h, w, C)
        out[f"tg_pool_{lv}"] = f(o[f"tg_pool_{lv}"])[:Fr, :C].reshape(1, 1, Fr, C)
        out[f"tgraph_{lv}"] = f(o[f"tgraph_{lv}"])[:Fr, :C].reshape(1, 1, Fr, C)
        out[f"temp_ctx_{lv}"] = U.unpad_map(o[f"temp_ctx_{lv}"], 1, h, w, C)
        out[f"spa_graph_{lv}"] = U.unpad_map(o[f"spa_graph_{lv}"], 1, h, w, C)
        out[f"fusion_{lv}"] = U.unpad_map(o[f"fusion_{lv}"], 1, h, w, M)
        out[f"gw_w_{lv}"], out[f"gw_v_{lv}"] = f(o[f"gw_w_{lv}"])[:, :, :n], f(o[f"gw_v_{lv}"])[:, :, :n]
        out[f"score_{lv}"], out[f"up_{lv}"] = f(o[f"score_{lv}"]), f(o[f"up_{lv}"])
    for k in ("exg_c3_2", "exg_c4_2", "exg_c5_2", "fused"):
        out[k] = U.unpad_map(o[k], 1, h, w, M)
    for k in ("pred", "up", "sigm"):
        out[k] = f(o[k])
    return out


def ref_grad(case, grads, n):
    flags = {k: f for k, _, _, f in VD.head_param_specs(case["cfg"])}
    g = grads[n] / (2.0 if "x2" in flags[n] else 1.0)
    return g - case["cfg"].weight_decay * case["hp"][n] if "reg" in flags[n] else g


def run_product(m, case):
    we, sl, fr = m._video_feeds(case["words"], case["vi"], case["clip"])
    o = m.loss_and_grads([f.to(m.device) for f in case["feats"]], we, case["tgt"], sl)
    torch.cuda.synchronize()
    return o, int(sl[0])


def test_video_forward_backward_fp32_matches_oracle():
    case = make_case()
    cfg = case["cfg"]
    scal, grads, taps = VD.grads_of(case["hp"], case["feats"], case["words"], case["tgt"], cfg)
    m = build(case, "f32")
    o, n = run_product(m, case)
    pt = taps_as_oracle(o, cfg, n)
    for k, ref in taps.items():
        assert U.rel_err(pt[k], ref) < 3e-5, k
    for k in ("loss_c5", "loss_c4", "loss_c3", "loss_last", "loss_all"):
        assert abs(float(o[k]) - scal[k]) <= 2e-5 * abs(scal[k]), k
    g = m.store.grad_dict()
    assert set(g) == set(grads)
    worst = ("", 0.0)
    for name in grads:
        ref = ref_grad(case, grads, name)
        if any(t in name for t in ("spa_graph_key", "tg_vtrans", "tg_key", "ctx_trans")) and name.endswith("biases"):
            # each of these biases adds a constant to every logit of a softmax row (lt . b_v, q_i . b_k, mt_n . b_c): the exact gradient is 0
            # and both sides return rounding noise
            assert float(g[name].abs().max()) < 1e-5 and float(ref.abs().max()) < 1e-5, name
            continue
        tol = 3e-3 if (("spa_graph_trans2" in name or "mm_trans" in name) and name.endswith("biases")) else 5e-4
        err = U.rel_err(g[name], ref)
        worst = max(worst, (name, err), key=lambda kv: kv[1])
        assert err < tol, (name, err)
    print("video fp32 worst gradient error:", worst)


def test_video_driver_calls_and_f16():
    """forward_video / train_step_video with the reference driver's feeds (front-padded words, valid_idx, the 16-frame clip); f16 storage
    within its tolerance; two runs bit-identical."""
    case = make_case(seed=2)
    cfg = case["cfg"]
    scal, grads, taps = VD.grads_of(case["hp"], case["feats"], case["words"], case["tgt"], cfg)
    m = build(case, "f32", mode="eval")
    out = m.forward_video(case["words"], None, case["vi"], case["clip"])
    assert tuple(out["up"].shape) == (1, cfg.H, cfg.W, 1) and U.rel_err(out["up"].float().cpu(), taps["up"]) < 1e-4
    with pytest.raises(ValueError):
        m.forward_video(torch.flip(case["words"], dims=[1]), None, case["vi"], case["clip"])      # end-padded ids are not what the graph expects
    runs = []
    for _ in range(2):
        m = build(case, "f16")
        o, n = run_product(m, case)
        pt = taps_as_oracle(o, cfg, n)
        assert U.rel_err(pt["up"], taps["up"]) < 2e-2 and abs(float(o["loss_all"]) - scal["loss_all"]) <= 1e-2 * abs(scal["loss_all"])
        for step in range(2):
            s, sc = m.train_step_video(case["words"], None, case["tgt"], case["vi"], case["clip"])
        torch.cuda.synchronize()
        assert s == 2 and m.grad_nonfinite() == 0
        runs.append((m.eng.params.clone(), float(sc["loss_all"])))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    # TF-Adam on the fp32 path: one step against the oracle
    m = build(case, "f32")
    hp = {k: v.clone() for k, v in case["hp"].items()}
    opt = O.TFAdam(hp)
    ref = VD.train_step(hp, opt, 0, case["feats"], case["words"], case["tgt"], cfg)
    s, sc = m.train_step_video(case["words"], None, case["tgt"], case["vi"], case["clip"])
    torch.cuda.synchronize()
    assert abs(float(sc["loss_all"]) - ref["loss_all"]) <= 1e-4 * abs(ref["loss_all"]) and abs(sc["learning_rate"] - ref["lr"]) < 1e-12
    sd = m.state_dict()
    for name, r in hp.items():
        d = (sd[name] - r).abs().flatten()
        assert float(torch.quantile(d[:200000], 0.99)) <= 0.5 * cfg.start_lr, name


def test_config5_full_size_mean_iou_delta_vs_oracle():
    """BASELINE.json config 5 at the reference's sizes: 16-frame 320x320 clips, L = 20, C = 1000, M = 500, ResNet-101 on the 5 sampled
    frames, batch 1.  The reference reports mean IoU over the evaluated clips (trainval_video.py:268-283); here 6 synthetic clips:
    |mean-IoU(HIP) - mean-IoU(oracle)| <= 1e-4 in fp32 and f16 storage on identical inputs and weights (per-clip deltas printed)."""
    torch.set_num_threads(16)
    cfg = VD.Cfg(batch_size=1)
    hp, bp = VD.init_head_params(cfg), O.init_backbone_params(cfg)
    clips, refs = [], []
    for seed in range(4, 10):
        words, clip, tgt = VD.synth_clip(cfg, seed=seed)
        vi = torch.tensor([[int((words[0] == 0).sum())]], dtype=torch.int32)
        with torch.no_grad():
            taps = VD.head_forward(hp, VD.backbone_taps(bp, clip, cfg), words, cfg)
            refs.append((float(VD.losses(hp, taps, tgt, cfg)["mIoU"]), taps["up"]))
        clips.append((words, vi, clip, tgt))
    P = U.pkg()
    res = {}
    for dtype in ("f32", "f16"):
        m = P.get_segmentation_model(NAME, batch_size=1, mode="train", dtype=dtype, head_params=hp, backbone_params=bp)
        ious, flips, errs = [], 0, 0.0
        for (words, vi, clip, tgt), (riou, rup) in zip(clips, refs):
            we, sl, fr = m._video_feeds(words, vi, clip)
            with torch.no_grad():
                o = m.head(m.features(fr), we, sl, tgt)
            torch.cuda.synchronize()
            up = o["up"].float().cpu()
            ious.append(float(o["mIoU"]))
            flips += int(((up > 0) != (rup > 0)).sum())
            errs = max(errs, U.rel_err(up, rup))
        per_clip = [abs(a - r[0]) for a, r in zip(ious, refs)]
        res[dtype] = (abs(float(np.mean(ious)) - float(np.mean([r[0] for r in refs]))), max(per_clip), flips, errs)
        del m, o
        torch.cuda.empty_cache()
    print("config 5 parity:", {k: f"d(mean IoU)={v[0]:.2e} worst clip={v[1]:.2e} flipped_px={v[2]} up_rel_err={v[3]:.2e}" for k, v in res.items()},
          "oracle mean IoU", float(np.mean([r[0] for r in refs])))
    assert res["f32"][0] <= 1e-4 and res["f32"][1] <= 1e-4 and res["f32"][3] < 1e-3
    assert res["f16"][0] <= 1e-4
