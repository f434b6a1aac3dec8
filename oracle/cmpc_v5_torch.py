"""ORACLE (test infrastructure only) -- torch-CPU restatement of CMPCv5_BiLSTM_model.LSTM_model (BASELINE.json config 4) and of
its HSV variant (CMPCv5_BiLSTM_HSV_model.py): build_graph() + train_op() op by op with TF1 / tf.contrib.slim semantics.

PARITY UNPINNED: TensorFlow 1.x (and tf.contrib.slim) cannot be installed here and the reference holds no golden vectors for this
path.  Cross-checked against an independent NumPy float64 restatement (oracle/cmpc_v5_numpy.py) in tests/test_oracle_v5.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product package never does.

Citations are file:line in /root/reference; "v5:" = CMPCv5_BiLSTM_model.py, "hsv:" = CMPCv5_BiLSTM_HSV_model.py.  Semantics of the
third-party ops the file leans on (TensorFlow 1.13-1.15, un-vendored; restated from their published behaviour):
  * tf.nn.bidirectional_dynamic_rnn(sequence_length): the backward cell runs on array_ops.reverse_sequence(inputs, seq_len) -- the first
    seq_len[b] steps of sample b reversed, the rest in place -- and its outputs are reversed back the same way; outputs past seq_len are
    zero, states frozen.  Variables: bidirectional_rnn/{fw,bw}/lstm_cell/{kernel,bias}.
  * slim conv2d under resnet_v2.resnet_arg_scope(batch_norm_decay): 'SAME' padding, no bias, batch_norm(decay, epsilon 1e-5, scale=True,
    fused) then relu; variables <scope>/weights, <scope>/BatchNorm/{beta,gamma,moving_mean,moving_variance}.  Training mode normalises
    with the batch mean and BIASED variance; the moving variance is updated with the UNBIASED one (fused_batch_norm), both as
    moving = decay * moving + (1 - decay) * batch.  Inference mode uses the moving statistics.
  * resnet_utils.conv2d_same(stride=1, rate=r) = slim conv2d(rate=r, padding='SAME').
  * tf.image.rgb_to_hsv (core/kernels/colorspace_op.h): v = max, range = v - min, s = v > 0 ? range / v : 0,
    h = (r == v ? (g - b) : g == v ? (b - r) + 2 range : (r - g) + 4 range) / (6 range), 0 when range == 0, +1 when negative.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from oracle import cmpc_torch as O

LEVELS = ("c5", "c4")                              # v5:134-137
EXG = ("c4", "c5", "c4_2", "c5_2")                 # v5:364-375
BN_EPS = 1e-5                                      # resnet_arg_scope(batch_norm_epsilon=1e-5)


@dataclass
class Cfg(O.Cfg):
    """Constructor arguments of CMPCv5_BiLSTM_model.LSTM_model (v5:21-49) that shape the graph."""
    batch_norm_decay: float = 0.9997               # v5:42
    hsv: bool = False                              # CMPCv5_BiLSTM_HSV_model: HSV of the image appended to the c5 / c4 taps (hsv:120-134)
    aspp_depth: int = 256                          # v5:208
    low_dim: int = 48                              # v5:196
    aspp_rates: Tuple[int, int, int] = (6, 12, 18)  # output_stride 16 (v5:153,225)
    train_mode: bool = True                        # mode == 'train': batch-norm uses batch statistics (v5:153-154)

    @property
    def c2_dim(self):
        return 4 * self.backbone_width             # res2b_relu channels (v5:88)

    @property
    def lat_extra(self):
        return 3 if self.hsv else 0


def bn_scopes(cfg: Cfg) -> List[Tuple[str, int, int, int]]:
    """(scope, kernel, cin, cout) of every slim conv2d + BatchNorm, in creation order (v5:234-249 then :196-204)."""
    M, D = cfg.mlp_dim, cfg.aspp_depth
    return [("aspp/conv_1x1", 1, M, D), ("aspp/conv_3x3_1", 3, M, D), ("aspp/conv_3x3_2", 3, M, D), ("aspp/conv_3x3_3", 3, M, D),
            ("aspp/image_level_features/conv_1x1", 1, M, D), ("aspp/conv_1x1_concat", 1, 5 * D, D),
            ("decoder/low_level_features/conv_1x1", 1, cfg.c2_dim, cfg.low_dim),
            ("decoder/upsampling_logits/conv_3x3_1", 3, D + cfg.low_dim, D), ("decoder/upsampling_logits/conv_3x3_2", 3, D, D)]


def head_param_specs(cfg: Cfg):
    """(name, shape, initialiser, flags) of the TRAINABLE variables under scope text_objseg, in creation order.
    flags: 'reg' = in reg_var_list ('DW' in the name or name ends in 'weights', v5:530), 'x2' = gradient x 2 ('biases', v5:561)."""
    C, M, R = cfg.v_emb_dim, cfg.mlp_dim, cfg.rnn_size
    specs = []

    def conv(name, k, cin, cout):
        specs.append((f"text_objseg/{name}/DW", (k, k, cin, cout), "xavier", ("reg",)))
        specs.append((f"text_objseg/{name}/biases", (cout,), "zeros", ("x2",)))

    def ln(scope, dim):
        specs.append((f"text_objseg/{scope}/beta", (dim,), "zeros", ()))
        specs.append((f"text_objseg/{scope}/gamma", (dim,), "ones", ()))

    specs.append(("text_objseg/Variable", (cfg.vocab_size, cfg.glove_dim), "glove", ()))                     # v5:160
    for d in ("fw", "bw"):                                                                                     # v5:164-174
        specs.append((f"text_objseg/bidirectional_rnn/{d}/lstm_cell/kernel", (cfg.glove_dim + R, 4 * R), "glorot", ()))
        specs.append((f"text_objseg/bidirectional_rnn/{d}/lstm_cell/bias", (4 * R,), "zeros", ()))
    conv("words_feat", 1, 2 * R, R)                                                                            # v5:182
    conv("c5_lateral", 1, cfg.vf_dim + cfg.lat_extra, C)                                                       # v5:120, hsv:129
    conv("c4_lateral", 1, cfg.c4_dim + cfg.lat_extra, C)                                                       # v5:123, hsv:134
    conv("words_parse_1", 1, R, cfg.parse_dim)                                                                 # v5:444-446
    conv("words_parse_2", 1, cfg.parse_dim, 4)
    for lv in LEVELS:                                                                                          # build_lang2vis v5:425-440
        for h in range(1, 6):
            conv(f"vis_trans_{lv}_head{h}", 1, C + 8, C)
            conv(f"lang_trans_{lv}_head{h}", 1, R, C)
        conv(f"words_trans_{lv}", 1, R, R)
        conv(f"spa_graph_trans2_{lv}", 1, C, C)
        ln(f"gconv_feat_ln_spa_graph_{lv}", C)
        conv(f"gconv_update_spa_graph_{lv}", 1, C, C)
        ln(f"gconv_update_ln_spa_graph_{lv}", C)
        conv(f"fusion_{lv}", 1, 2 * C + R + 8, M)
    conv("score_c5", 3, M, 1)                                                                                  # v5:142-145
    conv("score_c4", 3, M, 1)
    for lv in EXG:                                                                                             # v5:333-347
        conv(f"spa_graph_key_{lv}gv_f1", 1, M, M)
        conv(f"lang_query_{lv}gv_f1", 1, R, M)
        conv(f"gv_lang_{lv}gv_f1", 1, M + R, M)
        conv(f"lang_feat_{lv}_f1", 1, M, M)
        conv(f"trans_feat_{lv}_f1", 1, M, M)
    pre = "rnn/conv_lstm_cell"                                                                                 # util/cell.py:36-66
    specs.append((f"text_objseg/{pre}/kernel", (1, 1, 2 * M, 4 * M), "glorot", ()))
    specs.append((f"text_objseg/{pre}/W_ci", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    specs.append((f"text_objseg/{pre}/W_cf", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    ln(f"{pre}/LayerNorm", M)
    ln(f"{pre}/LayerNorm_1", M)
    ln(f"{pre}/LayerNorm_2", M)
    specs.append((f"text_objseg/{pre}/W_co", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    ln(f"{pre}/LayerNorm_3", M)
    ln(f"{pre}/LayerNorm_4", M)
    for scope, k, cin, cout in bn_scopes(cfg):                                                                 # v5:208-251,190-204
        specs.append((f"text_objseg/{scope}/weights", (k, k, cin, cout), "vscale", ("reg",)))
        specs.append((f"text_objseg/{scope}/BatchNorm/beta", (cout,), "zeros", ()))
        specs.append((f"text_objseg/{scope}/BatchNorm/gamma", (cout,), "ones", ()))
    specs.append(("text_objseg/decoder/upsampling_logits/conv_1x1/weights", (1, 1, cfg.aspp_depth, 1), "vscale", ("reg",)))   # v5:205
    specs.append(("text_objseg/decoder/upsampling_logits/conv_1x1/biases", (1,), "zeros", ("x2",)))
    return specs


def bn_state_specs(cfg: Cfg):
    """Non-trainable batch-norm statistics (updated by the UPDATE_OPS of v5:575-577)."""
    out = []
    for scope, _, _, cout in bn_scopes(cfg):
        out.append((f"text_objseg/{scope}/BatchNorm/moving_mean", (cout,), "zeros"))
        out.append((f"text_objseg/{scope}/BatchNorm/moving_variance", (cout,), "ones"))
    return out


def init_head_params(cfg: Cfg, seed: int = 1234, glove_seed: int = 7, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Reference initialisers: xavier / glorot uniform; slim's variance_scaling_initializer() (factor 2, FAN_IN) as a plain normal
    of std sqrt(2 / fan_in) (its truncation is an initialiser detail, the weights are synthetic); biases, beta 0; gamma 1."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind, _ in head_param_specs(cfg):
        if kind in ("xavier", "glorot"):
            fi, fo = O._fans(shape)
            t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / (fi + fo))
        elif kind == "vscale":
            fi, _ = O._fans(shape)
            t = torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / fi)
        elif kind == "zeros":
            t = torch.zeros(shape, dtype=torch.float64)
        elif kind == "ones":
            t = torch.ones(shape, dtype=torch.float64)
        elif kind == "glove":
            t = torch.randn(shape, generator=torch.Generator().manual_seed(glove_seed), dtype=torch.float64) * 0.4
        else:
            raise ValueError(kind)
        out[name] = t.to(dtype)
    return out


def init_bn_state(cfg: Cfg, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {n: (torch.zeros(s) if k == "zeros" else torch.ones(s)).to(dtype) for n, s, k in bn_state_specs(cfg)}


# ----------------------------------------------------------------------------------------
# text encoder: BiLSTM(), v5:159-187
# ----------------------------------------------------------------------------------------
def reverse_sequence(x, seq_len):
    """array_ops.reverse_sequence(x [B,T,...], seq_len, seq_axis=1, batch_axis=0)."""
    B, T = x.shape[:2]
    idx = torch.arange(T).view(1, T).expand(B, T).clone()
    for b in range(B):
        n = int(seq_len[b])
        idx[b, :n] = torch.arange(n - 1, -1, -1)
    return torch.gather(x, 1, idx.view(B, T, *([1] * (x.dim() - 2))).expand_as(x))


def _lstm_dir(emb, K, bias, seq_len, R):
    """tf LSTMCell under dynamic_rnn(sequence_length): gates i,j,f,o, forget_bias 1, zero outputs / frozen state past the length."""
    B, T, _ = emb.shape
    h = torch.zeros(B, R, dtype=emb.dtype)
    c = torch.zeros(B, R, dtype=emb.dtype)
    outs = []
    for t in range(T):
        z = torch.cat([emb[:, t], h], 1) @ K + bias
        i, j, f, o = z.split(R, 1)
        c_new = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        live = (t < seq_len).to(emb.dtype).view(B, 1)
        outs.append(h_new * live)
        h = live * h_new + (1 - live) * h
        c = live * c_new + (1 - live) * c
    return torch.stack(outs, 1)


def bilstm(p, words, seq_len, cfg: Cfg):
    emb = p["text_objseg/Variable"][words.long()]                                   # v5:160-163
    R = cfg.rnn_size
    pre = "text_objseg/bidirectional_rnn/"
    fw = _lstm_dir(emb, p[pre + "fw/lstm_cell/kernel"], p[pre + "fw/lstm_cell/bias"], seq_len, R)
    bw = _lstm_dir(reverse_sequence(emb, seq_len), p[pre + "bw/lstm_cell/kernel"], p[pre + "bw/lstm_cell/bias"], seq_len, R)
    bw = reverse_sequence(bw, seq_len)
    cat = torch.cat([fw, bw], -1).unsqueeze(1)                                      # [B,1,T,2R]  v5:178-180
    seq_mask = (cat.abs().sum(-1, keepdim=True) != 0).to(emb.dtype)                 # v5:181
    wf = torch.tanh(O.conv1x1(p, "words_feat", cat))                                # v5:182-183
    return O.l2_normalize(wf, -1), seq_mask, cat                                    # v5:185


# ----------------------------------------------------------------------------------------
# stages that differ from CMPC_model
# ----------------------------------------------------------------------------------------
def rgb_to_hsv(rgb):
    """tf.image.rgb_to_hsv on [...,3] floats (any range), see header."""
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    v = rgb.max(-1).values
    rng = v - rgb.min(-1).values
    s = torch.where(v > 0, rng / v, torch.zeros_like(v))
    norm = 1.0 / (6.0 * rng)
    h = torch.where(r == v, norm * (g - b), torch.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
    h = torch.where(rng > 0, h, torch.zeros_like(h))
    h = torch.where(h < 0, h + 1.0, h)
    return torch.stack([h, s, v], -1)


def hsv_map(im, cfg: Cfg):
    """hsv:120-126: im is BGR minus mean; + mean, reverse -> RGB, rgb_to_hsv, legacy bilinear to the feature size."""
    rgb = torch.flip(im.float() + torch.from_numpy(O.MU), dims=[-1]).to(im.dtype)      # the graph adds the mean back in float32
    return O.resize_bilinear(rgb_to_hsv(rgb), cfg.vf_h, cfg.vf_w)


def build_spa_graph(p, spa_graph, words_feat, words_parse, seq_mask, lv, cfg: Cfg, taps):
    """build_spa_graph, v5:470-504: as CMPC_model.py:376-410 but the word softmax runs over ALL T logits (a padded word enters with
    logit parse_R * affinity = 0) and is masked afterwards (v5:486-487)."""
    B, T, N, C = cfg.batch_size, cfg.num_steps, cfg.N, cfg.v_emb_dim
    words_trans = O.conv1x1(p, f"words_trans_{lv}", words_feat).reshape(B, T, cfg.rnn_size)
    t2 = O.conv1x1(p, f"spa_graph_trans2_{lv}", spa_graph).reshape(B, N, C)
    affi = t2 @ words_trans.transpose(1, 2) / (C ** 0.5)
    affi = words_parse[:, :, :, 2] * affi
    mask = seq_mask.reshape(B, 1, T)
    gw_w = mask * torch.softmax(affi, 2)
    gw_v = mask * torch.softmax(affi, 1)
    adj = gw_w @ gw_v.transpose(1, 2)
    taps[f"gw_w_{lv}"], taps[f"gw_v_{lv}"] = gw_w, gw_v
    g = O.graph_conv(p, spa_graph.reshape(B, 1, N, C), adj, lv).reshape(B, cfg.vf_h, cfg.vf_w, C)
    return O.l2_normalize(g, 3)


def build_lang2vis(p, vis, words_feat, words_parse, seq_mask, spatial, lv, cfg, taps):
    """v5:425-440 (same as CMPC_model.py:330-345 around the v5 build_spa_graph)."""
    vl = O.weighted_lang(words_parse, words_feat, "valid")
    vis_la_sp = O.mutan_fusion(p, vl, spatial, vis, lv)
    taps[f"vis_la_sp_{lv}"] = vis_la_sp
    spa = build_spa_graph(p, vis_la_sp, words_feat, words_parse, seq_mask, lv, cfg, taps)
    taps[f"spa_graph_{lv}"] = spa
    feat_all = torch.cat([vis_la_sp, spa, vl.expand(-1, cfg.vf_h, cfg.vf_w, -1), spatial], 3)
    return F.relu(O.conv1x1(p, f"fusion_{lv}", feat_all))


def global_vec(p, feat, lang, lv, cfg):
    """v5:299-331: as CMPC_model.py:212-243 plus tanh before the all-dims l2_normalize (v5:328-329)."""
    B, N, M = cfg.batch_size, cfg.N, cfg.mlp_dim
    key = O.conv1x1(p, f"spa_graph_key_{lv}", feat).reshape(B, N, M)
    q = O.conv1x1(p, f"lang_query_{lv}", lang).reshape(B, 1, M)
    attn = torch.softmax(key @ q.transpose(1, 2) / (M ** 0.5), 1)
    pooled = (attn.transpose(1, 2) @ feat.reshape(B, N, M)).reshape(B, 1, 1, M)
    gv = torch.tanh(O.conv1x1(p, f"gv_lang_{lv}", torch.cat([pooled, lang], 3)))
    return O.l2_normalize(gv, None)


def gated_exchange_module(p, feat, feat1, lang, lv, cfg):
    """v5:333-347: one gated branch."""
    gv = global_vec(p, feat, lang, lv + "gv_f1", cfg)
    return feat + O.lang_se(p, feat1, gv, lv + "_f1")


def conv_lstm(p, xs, cfg: Cfg):
    return O.conv_lstm(p, xs, cfg)                  # util/cell.py:36-79 over the 2 stacked maps (v5:378-385)


def slim_conv_bn(p, bn, scope, x_nhwc, cfg: Cfg, rate=1, new_state=None):
    """slim conv2d (no bias) + batch_norm + relu under resnet_arg_scope (see header).  bn: moving statistics; new_state (dict or None)
    receives the updated moving statistics in training mode."""
    w = p[f"text_objseg/{scope}/weights"]
    y = O.tf_conv2d(x_nhwc.permute(0, 3, 1, 2), w, dilation=rate).permute(0, 2, 3, 1)
    gamma, beta = p[f"text_objseg/{scope}/BatchNorm/gamma"], p[f"text_objseg/{scope}/BatchNorm/beta"]
    mm, mv = bn[f"text_objseg/{scope}/BatchNorm/moving_mean"], bn[f"text_objseg/{scope}/BatchNorm/moving_variance"]
    if cfg.train_mode:
        dims = (0, 1, 2)
        n = y.shape[0] * y.shape[1] * y.shape[2]
        mean = y.mean(dim=dims)
        var = ((y - mean) ** 2).mean(dim=dims)
        if new_state is not None:
            d = cfg.batch_norm_decay
            with torch.no_grad():
                new_state[f"text_objseg/{scope}/BatchNorm/moving_mean"] = mm * d + mean.detach() * (1 - d)
                new_state[f"text_objseg/{scope}/BatchNorm/moving_variance"] = mv * d + var.detach() * (n / max(n - 1, 1)) * (1 - d)
    else:
        mean, var = mm, mv
    return F.relu((y - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta)


def aspp(p, bn, x, cfg: Cfg, new_state=None, taps=None):
    """atrous_spatial_pyramid_pooling, v5:208-251 (output_stride 16 -> rates 6, 12, 18)."""
    B, h, w, _ = x.shape
    branches = [slim_conv_bn(p, bn, "aspp/conv_1x1", x, cfg, 1, new_state)]
    for i, r in enumerate(cfg.aspp_rates):
        branches.append(slim_conv_bn(p, bn, f"aspp/conv_3x3_{i + 1}", x, cfg, r, new_state))
    pooled = x.mean(dim=(1, 2), keepdim=True)                                                        # v5:242
    img = slim_conv_bn(p, bn, "aspp/image_level_features/conv_1x1", pooled, cfg, 1, new_state)      # v5:244
    img = O.resize_bilinear(img, h, w)                                                               # v5:246 (1x1 -> constant map)
    if taps is not None:
        taps["aspp_branches"] = torch.cat(branches, 3)
        taps["aspp_image"] = img[:, 0, 0]
    net = torch.cat(branches + [img], 3)                                                             # v5:248
    return slim_conv_bn(p, bn, "aspp/conv_1x1_concat", net, cfg, 1, new_state)                       # v5:249


def decoder(p, bn, enc, c2, cfg: Cfg, new_state=None, taps=None):
    """decoder, v5:190-206."""
    low = slim_conv_bn(p, bn, "decoder/low_level_features/conv_1x1", c2, cfg, 1, new_state)          # v5:196
    net = O.resize_bilinear(enc, low.shape[1], low.shape[2])                                         # v5:201
    net = torch.cat([net, low], 3)                                                                   # v5:202
    if taps is not None:
        taps["dec_cat"] = net
    net = slim_conv_bn(p, bn, "decoder/upsampling_logits/conv_3x3_1", net, cfg, 1, new_state)        # v5:203
    net = slim_conv_bn(p, bn, "decoder/upsampling_logits/conv_3x3_2", net, cfg, 1, new_state)        # v5:204
    if taps is not None:
        taps["dec_net2"] = net
    w = p["text_objseg/decoder/upsampling_logits/conv_1x1/weights"][0, 0]
    return net @ w + p["text_objseg/decoder/upsampling_logits/conv_1x1/biases"]                      # v5:205


def head_forward(p, bn, feats, words, seq_len, cfg: Cfg, im=None, new_state=None):
    """build_graph(), v5:101-157.  feats = (c2, c4, c5) NHWC = res2b_relu, res4b22_relu, res5c_relu (v5:86-88); im is needed by the
    HSV variant only.  Returns all taps."""
    c2, c4, c5 = feats
    taps = {}
    words_feat, seq_mask, cat = bilstm(p, words, seq_len, cfg)
    taps["words_feat"], taps["seq_mask"], taps["bilstm_out"] = words_feat, seq_mask, cat
    if cfg.hsv:
        hsv = hsv_map(im, cfg)
        taps["hsv"] = hsv
        c5, c4 = torch.cat([c5, hsv], -1), torch.cat([c4, hsv], -1)                  # hsv:128,133
    v5 = O.l2_normalize(torch.tanh(O.conv1x1(p, "c5_lateral", c5)), 3)               # v5:120-122
    v4 = O.l2_normalize(torch.tanh(O.conv1x1(p, "c4_lateral", c4)), 3)               # v5:123-125
    taps["lat_c5"], taps["lat_c4"] = v5, v4
    spatial = O.generate_spatial_batch(cfg.batch_size, cfg.vf_h, cfg.vf_w, dtype=c5.dtype)
    words_parse = O.lang_parser(p, words_feat, seq_mask)                             # v5:442-451
    taps["words_parse"] = words_parse
    fus = {}
    for lv, v in (("c5", v5), ("c4", v4)):
        fus[lv] = build_lang2vis(p, v, words_feat, words_parse, seq_mask, spatial, lv, cfg, taps)
        taps[f"fusion_{lv}"] = fus[lv]
    for lv in LEVELS:                                                                # v5:142-145
        sc = O.conv3x3(p, f"score_{lv}", fus[lv])
        taps[f"score_{lv}"] = sc
        taps[f"up_{lv}"] = O.resize_bilinear(sc, cfg.H, cfg.W)
    nec = O.weighted_lang(words_parse, words_feat, "nec")                            # v5:149
    taps["nec_lang"] = nec
    f4, f5 = fus["c4"], fus["c5"]                                                    # gated_exchange_fusion_lstm_2times v5:349-388
    e4 = O.l2_normalize(gated_exchange_module(p, f4, f5, nec, "c4", cfg), 3)
    e5 = O.l2_normalize(gated_exchange_module(p, f5, f4, nec, "c5", cfg), 3)
    e42 = O.l2_normalize(gated_exchange_module(p, e4, e5, nec, "c4_2", cfg), 3)
    e52 = O.l2_normalize(gated_exchange_module(p, e5, e4, nec, "c5_2", cfg), 3)
    taps["exg_c4"], taps["exg_c5"], taps["exg_c4_2"], taps["exg_c5_2"] = e4, e5, e42, e52
    fused = conv_lstm(p, (e42, e52), cfg)
    taps["fused"] = fused
    enc = aspp(p, bn, fused, cfg, new_state, taps)                                   # v5:153
    taps["aspp"] = enc
    pred = decoder(p, bn, enc, c2, cfg, new_state, taps)                             # v5:154
    taps["pred"] = pred
    taps["up"] = O.resize_bilinear(pred, cfg.H, cfg.W)                               # v5:156
    taps["sigm"] = torch.sigmoid(taps["up"])
    return taps


def losses(p, taps, target_fine, cfg: Cfg):
    """train_op() loss part, v5:535-544,585-589."""
    def wll(scores):
        return O.sigmoid_xent(scores, target_fine).sum(dim=(1, 2, 3)).mean()
    out = {"loss_c5": wll(taps["up_c5"]), "loss_c4": wll(taps["up_c4"]), "loss_last": wll(taps["up"])}
    out["loss_all"] = 0.8 * out["loss_last"] + 0.1 * out["loss_c5"] + 0.1 * out["loss_c4"]
    reg = 0.0
    for name, _, _, flags in head_param_specs(cfg):
        if "reg" in flags:
            reg = reg + (p[name] ** 2).sum() / 2
    out["reg_loss"] = cfg.weight_decay * reg
    out["cost"] = out["loss_all"] + out["reg_loss"]
    pred, labl = taps["up"] > 0, target_fine != 0
    inter = (pred & labl).sum(dim=(1, 2, 3)).to(torch.float64)
    union = (pred | labl).sum(dim=(1, 2, 3)).to(torch.float64)
    out["mIoU"] = (inter / union).mean()
    return out


def grads_of(p, bn, feats, words, seq_len, target_fine, cfg: Cfg, im=None):
    """d cost / d every trainable variable with the x2 multiplier on 'biases' (v5:558-572).  Returns (scalars, grads, taps, new bn state)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    new_state = {}
    taps = head_forward(leaves, bn, feats, words, seq_len, cfg, im=im, new_state=new_state)
    ls = losses(leaves, taps, target_fine, cfg)
    names = list(leaves)
    gs = torch.autograd.grad(ls["cost"], [leaves[n] for n in names], allow_unused=True)
    flags = {n: f for n, _, _, f in head_param_specs(cfg)}
    grads = {}
    for n, g in zip(names, gs):
        g = torch.zeros_like(leaves[n]) if g is None else g
        grads[n] = (g * 2.0 if "x2" in flags[n] else g).detach()
    return {k: float(v.detach()) for k, v in ls.items()}, grads, {k: v.detach() for k, v in taps.items()}, new_state


def train_step(p, bn, opt: O.TFAdam, step, feats, words, seq_len, target_fine, cfg: Cfg, im=None):
    """One sess.run([train, ...]): UPDATE_OPS (moving statistics) then apply_gradients (v5:575-577).  p and bn updated in place."""
    scal, grads, _, new_state = grads_of(p, bn, feats, words, seq_len, target_fine, cfg, im=im)
    lr = O.poly_lr(step, cfg)
    with torch.no_grad():
        opt.step(p, grads, lr)
        bn.update(new_state)
    scal["lr"] = lr
    return scal


STEM_GAMMA = 1.0 / 256.0


def init_backbone_params(cfg: Cfg, seed: int = 4321, dtype=torch.float32):
    """Synthetic backbone for the CMPCv5 graphs: O.init_backbone_params with bn_conv1/gamma = 1/256.  The frozen inference network is
    positively homogeneous (convolutions, ReLU, BN with beta = mean = 0), so the image's 0..255 scale otherwise reaches the taps as
    |x| ~ 1e3: harmless for CMPC_model, whose laterals are scale-invariant (l2_normalize right behind a linear conv), but it drives
    CMPCv5's tanh laterals (v5:121,124) into saturation with a few hypersensitive unsaturated units -- an artefact of random weights
    (trained ResNet taps are O(1)), which would make every parity number a statement about that artefact.  1/256 gives taps of rms ~1."""
    bp = O.init_backbone_params(cfg, seed=seed, dtype=dtype)
    bp["bn_conv1/gamma"] = bp["bn_conv1/gamma"] * STEM_GAMMA
    return bp


def backbone_taps(bp, im, cfg: Cfg):
    """(c2, c4, c5) = res2b_relu, res4b22_relu, res5c_relu (v5:86-88)."""
    _, c4, c5, c2 = O.backbone_forward(bp, im, cfg, extra_taps=("2b",))
    return c2, c4, c5


def tiny_cfg(B=2, T=6, hw=8, C=40, M=24, hsv=False, train_mode=True):
    """Shrunken graph: backbone width 16 -> res2b 64 ch, res4 256, res5 512; ASPP depth 16 (rates 1, 3, 6: the last leaves most taps
    outside the 8x8 map), 8 low-level channels."""
    return Cfg(batch_size=B, num_steps=T, vf_h=hw, vf_w=hw, H=hw * 8, W=hw * 8, vf_dim=512, c4_dim=256, c3_dim=128,
               vocab_size=50, v_emb_dim=C, mlp_dim=M, rnn_size=C, glove_dim=12, parse_dim=20,
               backbone_width=16, backbone_blocks=(2, 1, 2, 1), hsv=hsv, aspp_depth=16, low_dim=8, aspp_rates=(1, 3, 6), train_mode=train_mode)
