"""ORACLE (test infrastructure only) -- torch-CPU restatement of CMPC_video/CMPC_video_mm_tgraph_allvec.py (BASELINE.json config 5):
build_graph() + train_op() op by op with TF1 semantics.  PARITY UNPINNED (TensorFlow absent, no golden vectors in the reference).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product package never does.

Citations "vid:" are file:line in /root/reference/CMPC_video/CMPC_video_mm_tgraph_allvec.py.

What the graph does (and only does for batch_size = 1: it indexes the merged batch*frame axis with `sample_frames // 2` (vid:379,383) and
feeds the ConvLSTM `[[feat_exg3_2[0], ...]]` (vid:323-324)):
  * 5 of the clip's 16 frames (indices 0, 4, 8, 12, 15; vid:69-73) go through the backbone; laterals + Mutan fusion run on all 5 (vid:151-157,
    352-366) with the entity+attribute language vector (vid:189-201);
  * a temporal graph over the 5 frames: per-frame language-attention pooling of the multimodal maps with the ACTION vector (vid:203-213), a
    5 x 5 attention adjacency and one graph_conv (vid:458-503);
  * the middle frame (index 2) gets a temporal context (attention of every pixel over the 5 graph nodes, vid:505-530) and the spatial
    word graph of CMPC_model (vid:435-456, no masks: the padded words are sliced away, vid:141-142);
  * fusion over [lateral | spatial graph | temporal context | language | grid] = 3C + R + 8 channels (vid:396-400); the rest is CMPC_model.
The text encoder is the old-style loop (vid:105-142): front-padded word ids, `tf.cond(words[0, n] == 0)` skipping the pad steps (zero output,
state unchanged) and a slice that drops the first valid_idx outputs -- for one sample: a BasicLSTMCell (gates i, j, f, o, forget_bias 1,
the same arithmetic as LSTMCell) run over the valid words only.  `lang_feat` (vid:144-146) is passed around but never consumed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from oracle import cmpc_torch as O

LEVELS = ("c5", "c4", "c3")
EXG = O.EXG
FRAME_IDX = (0, 4, 8, 12, 15)          # vid:70


@dataclass
class Cfg(O.Cfg):
    frames: int = 16                    # vid:36
    sample_frames: int = 5              # vid:69


def head_param_specs(cfg: Cfg):
    """Trainable variables under scope text_objseg in creation order; flags as in oracle/cmpc_torch.py ('reg': 'DW' in the name, vid:545;
    'x2': 'biases', vid:569)."""
    C, M, R = cfg.v_emb_dim, cfg.mlp_dim, cfg.rnn_size
    specs = []

    def conv(name, k, cin, cout):
        specs.append((f"text_objseg/{name}/DW", (k, k, cin, cout), "xavier", ("reg",)))
        specs.append((f"text_objseg/{name}/biases", (cout,), "zeros", ("x2",)))

    def ln(scope, dim):
        specs.append((f"text_objseg/{scope}/beta", (dim,), "zeros", ()))
        specs.append((f"text_objseg/{scope}/gamma", (dim,), "ones", ()))

    specs.append(("text_objseg/Variable", (cfg.vocab_size, cfg.glove_dim), "glove", ()))                                   # vid:101
    specs.append(("text_objseg/RNN/multi_rnn_cell/cell_0/basic_lstm_cell/kernel", (cfg.glove_dim + R, 4 * R), "glorot", ()))   # vid:106-133
    specs.append(("text_objseg/RNN/multi_rnn_cell/cell_0/basic_lstm_cell/bias", (4 * R,), "zeros", ()))
    conv("c5_lateral", 1, cfg.vf_dim, C)
    conv("c4_lateral", 1, cfg.c4_dim, C)
    conv("c3_lateral", 1, cfg.c3_dim, C)
    conv("words_parse_1", 1, R, cfg.parse_dim)
    conv("words_parse_2", 1, cfg.parse_dim, 5)                                                                             # vid:406
    for lv in LEVELS:                                                                                                      # build_lang2vis, vid:368-402
        for h in range(1, 6):
            conv(f"vis_trans_{lv}_head{h}", 1, C + 8, C)
            conv(f"lang_trans_{lv}_head{h}", 1, R, C)
        conv(f"tg_vtrans_{lv}", 1, C, C)                                                                                   # build_temp_graph, vid:458-503
        conv(f"tg_ltrans_{lv}", 1, R, R)
        conv(f"tg_query_{lv}", 1, C, C)
        conv(f"tg_key_{lv}", 1, C, C)
        ln(f"gconv_feat_ln_temp_graph_{lv}", C)
        conv(f"gconv_update_temp_graph_{lv}", 1, C, C)
        ln(f"gconv_update_ln_temp_graph_{lv}", C)
        conv(f"mm_trans_{lv}", 1, C, C)                                                                                    # build_temp_ctx, vid:505-530
        conv(f"ctx_trans_{lv}", 1, C, C)
        conv(f"words_trans_{lv}", 1, R, R)                                                                                 # build_spa_graph, vid:435-456
        conv(f"spa_graph_trans2_{lv}", 1, C, C)
        ln(f"gconv_feat_ln_spa_graph_{lv}", C)
        conv(f"gconv_update_spa_graph_{lv}", 1, C, C)
        ln(f"gconv_update_ln_spa_graph_{lv}", C)
        conv(f"fusion_{lv}", 1, 3 * C + R + 8, M)                                                                          # vid:398-400
    conv("score_c5", 3, M, 1)
    conv("score_c4", 3, M, 1)
    conv("score_c3", 3, M, 1)
    for lv in EXG:
        conv(f"spa_graph_key_{lv}gv_f1", 1, M, M)
        conv(f"lang_query_{lv}gv_f1", 1, R, M)
        conv(f"gv_lang_{lv}gv_f1", 1, M + R, M)
        conv(f"lang_feat_{lv}_f1", 1, M, M)
        conv(f"trans_feat_{lv}_f1", 1, M, M)
        conv(f"lang_feat_{lv}_f2", 1, M, M)
        conv(f"trans_feat_{lv}_f2", 1, M, M)
    pre = "rnn/conv_lstm_cell"
    specs.append((f"text_objseg/{pre}/kernel", (1, 1, 2 * M, 4 * M), "glorot", ()))
    specs.append((f"text_objseg/{pre}/W_ci", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    specs.append((f"text_objseg/{pre}/W_cf", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    ln(f"{pre}/LayerNorm", M); ln(f"{pre}/LayerNorm_1", M); ln(f"{pre}/LayerNorm_2", M)
    specs.append((f"text_objseg/{pre}/W_co", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    ln(f"{pre}/LayerNorm_3", M); ln(f"{pre}/LayerNorm_4", M)
    conv("score", 3, M, 1)
    return specs


def init_head_params(cfg: Cfg, seed: int = 1234, glove_seed: int = 7, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind, _ in head_param_specs(cfg):
        if kind in ("xavier", "glorot"):
            fi, fo = O._fans(shape)
            t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / (fi + fo))
        elif kind == "zeros":
            t = torch.zeros(shape, dtype=torch.float64)
        elif kind == "ones":
            t = torch.ones(shape, dtype=torch.float64)
        else:
            t = torch.randn(shape, generator=torch.Generator().manual_seed(glove_seed), dtype=torch.float64) * 0.4
        out[name] = t.to(dtype)
    return out


def text_encoder(p, words, cfg: Cfg):
    """vid:101-150.  words [1, T] FRONT-padded (util/text_processing.py:42-53); returns words_feat [1, 1, T', R] of the T' valid words."""
    assert words.shape[0] == 1, "the graph is only valid for batch_size = 1 (vid:125,379)"
    R = cfg.rnn_size
    K = p["text_objseg/RNN/multi_rnn_cell/cell_0/basic_lstm_cell/kernel"]
    bias = p["text_objseg/RNN/multi_rnn_cell/cell_0/basic_lstm_cell/bias"]
    emb = p["text_objseg/Variable"][words.long()]
    h = torch.zeros(1, R, dtype=emb.dtype)
    c = torch.zeros(1, R, dtype=emb.dtype)
    outs = []
    for n in range(cfg.num_steps):
        if int(words[0, n]) == 0:                                   # f1: zero output, state unchanged (vid:113-115,129)
            outs.append(torch.zeros(1, R, dtype=emb.dtype))
            continue
        z = torch.cat([emb[:, n], h], 1) @ K + bias
        i, j, f, o = z.split(R, 1)
        c = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    valid_idx = int((words[0] == 0).sum())                          # the driver feeds the number of pad words (front padding)
    wf = torch.stack(outs, 1)[:, valid_idx:]                        # vid:141-142
    return O.l2_normalize(wf, 2).unsqueeze(1)                       # vid:148-150


def lang_parser(p, words_feat):
    x = F.relu(O.conv1x1(p, "words_parse_1", words_feat))
    return torch.softmax(O.conv1x1(p, "words_parse_2", x), 3)      # [1,1,T',5]: entity, attribute, static relation, action, unnecessary (vid:404-412)


def pooled_lang(words_parse, words_feat, cols):
    """ea_lang (cols 0,1; vid:189-201), ac_lang (col 3 = [-2]; vid:203-213), valid_lang (all but 4; vid:215-227)."""
    B, _, T, R = words_feat.shape
    wts = sum(words_parse[:, :, :, c] for c in cols)
    return O.l2_normalize(wts @ words_feat.reshape(B, T, R), 2).reshape(B, 1, 1, R)


def graph_conv(p, graph_feat, adj, name, lv):
    """vid:414-433 (graph_feat [B,1,n,C], adj [B,n,n])."""
    B, _, n, C = graph_feat.shape
    g = (adj @ graph_feat.reshape(B, n, C)).reshape(B, 1, n, C)
    g = O.tf_layer_norm(g, p[f"text_objseg/gconv_feat_ln_{name}_{lv}/gamma"], p[f"text_objseg/gconv_feat_ln_{name}_{lv}/beta"])
    g = F.relu(graph_feat + g)
    u = O.conv1x1(p, f"gconv_update_{name}_{lv}", g)
    u = O.tf_layer_norm(u, p[f"text_objseg/gconv_update_ln_{name}_{lv}/gamma"], p[f"text_objseg/gconv_update_ln_{name}_{lv}/beta"])
    return F.relu(u)


def build_temp_graph(p, mm, ac_lang, lv, cfg: Cfg, taps):
    """vid:458-503.  mm [F, h, w, C] (batch 1 x F frames)."""
    Fr, N, C = cfg.sample_frames, cfg.N, cfg.v_emb_dim
    vt = O.conv1x1(p, f"tg_vtrans_{lv}", mm).reshape(Fr, N, C)
    lt = O.conv1x1(p, f"tg_ltrans_{lv}", ac_lang).reshape(1, 1, cfg.rnn_size).expand(Fr, 1, -1)
    attn = torch.softmax(lt @ vt.transpose(1, 2) / (C ** 0.5), 2)                    # [F,1,N]
    tg = (attn @ mm.reshape(Fr, N, C)).reshape(1, 1, Fr, C)
    taps[f"tg_pool_{lv}"] = tg
    q = O.conv1x1(p, f"tg_query_{lv}", tg).reshape(1, Fr, C)
    k = O.conv1x1(p, f"tg_key_{lv}", tg).reshape(1, Fr, C)
    adj = torch.softmax(q @ k.transpose(1, 2) / (C ** 0.5), 2)
    out = O.l2_normalize(graph_conv(p, tg, adj, "temp_graph", lv), 3)
    taps[f"tgraph_{lv}"] = out
    return out


def build_temp_ctx(p, mm_mid, ctx, lv, cfg: Cfg):
    """vid:505-530.  mm_mid [1,h,w,C], ctx [1,1,F,C]."""
    Fr, N, C = cfg.sample_frames, cfg.N, cfg.v_emb_dim
    mt = O.conv1x1(p, f"mm_trans_{lv}", mm_mid).reshape(1, N, C)
    ct = O.conv1x1(p, f"ctx_trans_{lv}", ctx).reshape(1, Fr, C)
    attn = torch.softmax(mt @ ct.transpose(1, 2) / (C ** 0.5), 2)                    # [1,N,F]
    glo = (attn @ ctx.reshape(1, Fr, C)).reshape(1, cfg.vf_h, cfg.vf_w, C)
    return O.l2_normalize(glo, 3)


def build_spa_graph(p, x, words_feat, words_parse, lv, cfg: Cfg, taps):
    """vid:435-456: CMPC_model's word graph without masks (every remaining word is valid)."""
    T = words_feat.shape[2]
    N, C = cfg.N, cfg.v_emb_dim
    wt = O.conv1x1(p, f"words_trans_{lv}", words_feat).reshape(1, T, cfg.rnn_size)
    t2 = O.conv1x1(p, f"spa_graph_trans2_{lv}", x).reshape(1, N, C)
    affi = words_parse[:, :, :, 2] * (t2 @ wt.transpose(1, 2) / (C ** 0.5))
    gw_w, gw_v = torch.softmax(affi, 2), torch.softmax(affi, 1)
    taps[f"gw_w_{lv}"], taps[f"gw_v_{lv}"] = gw_w, gw_v
    g = graph_conv(p, x.reshape(1, 1, N, C), gw_w @ gw_v.transpose(1, 2), "spa_graph", lv).reshape(1, cfg.vf_h, cfg.vf_w, C)
    return O.l2_normalize(g, 3)


def build_lang2vis(p, vis, words_feat, words_parse, spatial, lv, cfg: Cfg, taps):
    """vid:368-402.  vis [F, h, w, C]."""
    Fr = cfg.sample_frames
    ea = pooled_lang(words_parse, words_feat, (0, 1))
    mm = O.mutan_fusion(p, ea.expand(Fr, -1, -1, -1), spatial.expand(Fr, -1, -1, -1), vis, lv)        # vid:330-366 (tiles of lang and grid)
    taps[f"mm_{lv}"] = mm
    ac = pooled_lang(words_parse, words_feat, (3,))
    tg = build_temp_graph(p, mm, ac, lv, cfg, taps)
    mid = Fr // 2
    gtf_vis, gtf_mm = vis[mid:mid + 1], mm[mid:mid + 1]                                               # vid:379-386
    ctx = build_temp_ctx(p, gtf_mm, tg, lv, cfg)
    taps[f"temp_ctx_{lv}"] = ctx
    sg = build_spa_graph(p, gtf_mm, words_feat, words_parse, lv, cfg, taps)
    taps[f"spa_graph_{lv}"] = sg
    vl = pooled_lang(words_parse, words_feat, (0, 1, 2, 3))
    feat_all = torch.cat([gtf_vis, sg, ctx, vl.expand(-1, cfg.vf_h, cfg.vf_w, -1), spatial], 3)        # vid:396
    return F.relu(O.conv1x1(p, f"fusion_{lv}", feat_all))


def head_forward(p, feats, words, cfg: Cfg):
    """build_graph(), vid:91-187.  feats = (c3, c4, c5) of the 5 sampled frames, NHWC [5, h, w, .]; words [1, T] front-padded."""
    c3, c4, c5 = feats
    taps = {}
    words_feat = text_encoder(p, words, cfg)
    taps["words_feat"] = words_feat
    lat = {"c5": O.l2_normalize(O.conv1x1(p, "c5_lateral", c5), 3), "c4": O.l2_normalize(O.conv1x1(p, "c4_lateral", c4), 3),
           "c3": O.l2_normalize(O.conv1x1(p, "c3_lateral", c3), 3)}
    spatial = O.generate_spatial_batch(1, cfg.vf_h, cfg.vf_w, dtype=c5.dtype)
    words_parse = lang_parser(p, words_feat)
    taps["words_parse"] = words_parse
    fus = {}
    for lv in LEVELS:
        taps[f"lat_{lv}"] = lat[lv]
        fus[lv] = build_lang2vis(p, lat[lv], words_feat, words_parse, spatial, lv, cfg, taps)
        taps[f"fusion_{lv}"] = fus[lv]
    for lv in LEVELS:
        sc = O.conv3x3(p, f"score_{lv}", fus[lv])
        taps[f"score_{lv}"] = sc
        taps[f"up_{lv}"] = O.resize_bilinear(sc, cfg.H, cfg.W)
    vl = pooled_lang(words_parse, words_feat, (0, 1, 2, 3))                                            # vid:180
    taps["nec_lang"] = vl
    c1 = O.Cfg(**{k: getattr(cfg, k) for k in O.Cfg.__dataclass_fields__})
    c1.batch_size = 1
    f3, f4, f5 = fus["c3"], fus["c4"], fus["c5"]
    e3 = O.l2_normalize(O.gated_exchange_module(p, f3, f4, f5, vl, "c3", c1), 3)
    e4 = O.l2_normalize(O.gated_exchange_module(p, f4, f3, f5, vl, "c4", c1), 3)
    e5 = O.l2_normalize(O.gated_exchange_module(p, f5, f3, f4, vl, "c5", c1), 3)
    e32 = O.l2_normalize(O.gated_exchange_module(p, e3, e4, e5, vl, "c3_2", c1), 3)
    e42 = O.l2_normalize(O.gated_exchange_module(p, e4, e3, e5, vl, "c4_2", c1), 3)
    e52 = O.l2_normalize(O.gated_exchange_module(p, e5, e3, e4, vl, "c5_2", c1), 3)
    taps["exg_c3_2"], taps["exg_c4_2"], taps["exg_c5_2"] = e32, e42, e52
    fused = O.conv_lstm(p, (e32, e42, e52), c1)
    taps["fused"] = fused
    pred = O.conv3x3(p, "score", fused)
    taps["pred"] = pred
    taps["up"] = O.resize_bilinear(pred, cfg.H, cfg.W)
    taps["sigm"] = torch.sigmoid(taps["up"])
    return taps


def losses(p, taps, target_fine, cfg: Cfg):
    """vid:551-560: as CMPC_model (0.7 / 0.1 / 0.1 / 0.1, L2 on 'DW')."""
    def wll(scores):
        return O.sigmoid_xent(scores, target_fine).sum(dim=(1, 2, 3)).mean()
    out = {"loss_c5": wll(taps["up_c5"]), "loss_c4": wll(taps["up_c4"]), "loss_c3": wll(taps["up_c3"]), "loss_last": wll(taps["up"])}
    out["loss_all"] = 0.7 * out["loss_last"] + 0.1 * out["loss_c5"] + 0.1 * out["loss_c4"] + 0.1 * out["loss_c3"]
    reg = 0.0
    for name, _, _, flags in head_param_specs(cfg):
        if "reg" in flags:
            reg = reg + (p[name] ** 2).sum() / 2
    out["reg_loss"] = cfg.weight_decay * reg
    out["cost"] = out["loss_all"] + out["reg_loss"]
    pred, labl = taps["up"] > 0, target_fine != 0
    out["mIoU"] = ((pred & labl).sum(dim=(1, 2, 3)).double() / (pred | labl).sum(dim=(1, 2, 3)).double()).mean()
    return out


def grads_of(p, feats, words, target_fine, cfg: Cfg):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    taps = head_forward(leaves, feats, words, cfg)
    ls = losses(leaves, taps, target_fine, cfg)
    names = list(leaves)
    gs = torch.autograd.grad(ls["cost"], [leaves[n] for n in names], allow_unused=True)
    flags = {n: f for n, _, _, f in head_param_specs(cfg)}
    grads = {}
    for n, g in zip(names, gs):
        g = torch.zeros_like(leaves[n]) if g is None else g
        grads[n] = (g * 2.0 if "x2" in flags[n] else g).detach()
    return {k: float(v.detach()) for k, v in ls.items()}, grads, {k: v.detach() for k, v in taps.items()}


def train_step(p, opt: O.TFAdam, step, feats, words, target_fine, cfg: Cfg):
    scal, grads, _ = grads_of(p, feats, words, target_fine, cfg)
    lr = O.poly_lr(step, cfg)
    with torch.no_grad():
        opt.step(p, grads, lr)
    scal["lr"] = lr
    return scal


def synth_clip(cfg: Cfg, seed=0):
    """One synthetic A2D-style sample: clip uint8 [1, 16, H, W, 3] -> BGR minus mean; front-padded words; one rectangle mask (of the
    annotated middle frame)."""
    rng = np.random.default_rng(seed)
    T = cfg.num_steps
    clip = rng.integers(0, 256, size=(1, cfg.frames, cfg.H, cfg.W, 3), dtype=np.uint8).astype(np.float32)[..., ::-1] - O.MU
    n = int(rng.integers(min(3, T), T + 1))
    words = np.zeros((1, T), dtype=np.int32)
    words[0, T - n:] = rng.integers(min(4, cfg.vocab_size - 1), cfg.vocab_size, size=(n,))
    target = np.zeros((1, cfg.H, cfg.W, 1), dtype=np.float32)
    hh, ww = rng.integers(max(cfg.H // 8, 1), max(cfg.H * 5 // 8, 2) + 1, size=2)
    y0, x0 = rng.integers(0, cfg.H - hh + 1), rng.integers(0, cfg.W - ww + 1)
    target[0, y0:y0 + hh, x0:x0 + ww, 0] = 1.0
    return torch.from_numpy(words), torch.from_numpy(np.ascontiguousarray(clip)), torch.from_numpy(target)


def backbone_taps(bp, clip, cfg: Cfg):
    """The 5 sampled frames through the backbone (vid:69-77): (c3, c4, c5), each [5, h, w, .]."""
    frames = clip[0, list(FRAME_IDX)]
    bc = O.Cfg(**{k: getattr(cfg, k) for k in O.Cfg.__dataclass_fields__})
    return O.backbone_forward(bp, frames, bc)


def tiny_cfg(T=6, hw=8, C=40, M=24):
    return Cfg(batch_size=1, num_steps=T, vf_h=hw, vf_w=hw, H=hw * 8, W=hw * 8, vf_dim=256, c4_dim=128, c3_dim=64, vocab_size=50,
               v_emb_dim=C, mlp_dim=M, rnn_size=C, glove_dim=12, parse_dim=20, backbone_width=8, backbone_blocks=(1, 1, 2, 1))
