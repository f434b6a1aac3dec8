"""ORACLE (test infrastructure only) -- torch-CPU restatement of the CMPC hot path.

PARITY UNPINNED: the reference computes everything with TensorFlow 1.x, which is not
installable here, and the reference holds no golden vectors / known-answer tests for this
path.  This file restates the reference graph op by op with TF1
semantics; it is cross-checked against an independent NumPy float64 restatement
(oracle/cmpc_numpy.py) and against the known-answer properties derivable from the
reference source (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product package (cmpc-refseg_amd/) never does.

All citations are file:line in /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

F32_MIN = float(np.finfo(np.float32).min)  # tf.float32.min, CMPC_model.py:390


@dataclass
class Cfg:
    """Constructor arguments of LSTM_model (CMPC_model.py:15-40) that shape the graph."""
    batch_size: int = 1
    num_steps: int = 20
    vf_h: int = 40
    vf_w: int = 40
    H: int = 320
    W: int = 320
    vf_dim: int = 2048          # c5 channels (CMPC_model.py:108)
    c4_dim: int = 1024          # hard-coded at CMPC_model.py:110
    c3_dim: int = 512           # hard-coded at CMPC_model.py:112
    vocab_size: int = 12112
    v_emb_dim: int = 1000
    mlp_dim: int = 500
    rnn_size: int = 1000
    glove_dim: int = 300
    parse_dim: int = 500        # hard-coded 500 at CMPC_model.py:349
    start_lr: float = 0.00025
    lr_decay_step: int = 800000
    end_lr: float = 0.00001     # CMPC_model.py:451
    lr_power: float = 0.9
    weight_decay: float = 0.0005
    backbone_width: int = 64    # conv1 width; 64 = ResNet-101 as in deeplab_resnet/model.py:20
    backbone_blocks: Tuple[int, int, int, int] = (3, 4, 23, 3)

    @property
    def N(self):
        return self.vf_h * self.vf_w


LEVELS = ("c5", "c4", "c3")            # build order, CMPC_model.py:120-125
EXG = ("c3", "c4", "c5", "c3_2", "c4_2", "c5_2")  # CMPC_model.py:271-283


# ----------------------------------------------------------------------------------------
# Parameter manifest.  kind: how the reference initialises it.
#   xavier  = tf.contrib.layers.xavier_initializer_conv2d (uniform, fans include kh*kw)
#   glorot  = tf.get_variable default initializer (glorot_uniform)
#   zeros / ones / glove
# flags: 'reg'  -> in reg_var_list (name contains 'DW', CMPC_model.py:433)
#        'x2'   -> gradient multiplied by 2 (name contains 'biases', CMPC_model.py:464-465)
# ----------------------------------------------------------------------------------------
def head_param_specs(cfg: Cfg) -> List[Tuple[str, Tuple[int, ...], str, Tuple[str, ...]]]:
    C, M, R = cfg.v_emb_dim, cfg.mlp_dim, cfg.rnn_size
    specs = []

    def conv(name, k, cin, cout):
        specs.append((f"text_objseg/{name}/DW", (k, k, cin, cout), "xavier", ("reg",)))
        specs.append((f"text_objseg/{name}/biases", (cout,), "zeros", ("x2",)))

    def ln(scope):
        # tf.contrib.layers.layer_norm creates beta then gamma
        specs.append((f"text_objseg/{scope}/beta", (C,), "zeros", ()))
        specs.append((f"text_objseg/{scope}/gamma", (C,), "ones", ()))

    # lstm(), CMPC_model.py:144-156
    specs.append(("text_objseg/Variable", (cfg.vocab_size, cfg.glove_dim), "glove", ()))
    specs.append(("text_objseg/rnn/lstm_cell/kernel", (cfg.glove_dim + R, 4 * R), "glorot", ()))
    specs.append(("text_objseg/rnn/lstm_cell/bias", (4 * R,), "zeros", ()))
    # laterals :108-113
    conv("c5_lateral", 1, cfg.vf_dim, C)
    conv("c4_lateral", 1, cfg.c4_dim, C)
    conv("c3_lateral", 1, cfg.c3_dim, C)
    # lang parser :349-351
    conv("words_parse_1", 1, R, cfg.parse_dim)
    conv("words_parse_2", 1, cfg.parse_dim, 4)
    for lv in LEVELS:                      # build_lang2vis :330-345
        for h in range(1, 6):              # mutan_head :295-309
            conv(f"vis_trans_{lv}_head{h}", 1, C + 8, C)
            conv(f"lang_trans_{lv}_head{h}", 1, R, C)
        conv(f"words_trans_{lv}", 1, R, R)          # :378
        conv(f"spa_graph_trans2_{lv}", 1, C, C)     # :381
        ln(f"gconv_feat_ln_spa_graph_{lv}")         # :364
        conv(f"gconv_update_spa_graph_{lv}", 1, C, C)  # :368
        ln(f"gconv_update_ln_spa_graph_{lv}")       # :370
        conv(f"fusion_{lv}", 1, 2 * C + R + 8, M)   # :341
    conv("score_c5", 3, M, 1)              # :128-132
    conv("score_c4", 3, M, 1)
    conv("score_c3", 3, M, 1)
    for lv in EXG:                         # gated_exchange_module :245-259
        conv(f"spa_graph_key_{lv}gv_f1", 1, M, M)      # :221
        conv(f"lang_query_{lv}gv_f1", 1, R, M)         # :223
        conv(f"gv_lang_{lv}gv_f1", 1, M + R, M)        # :239
        conv(f"lang_feat_{lv}_f1", 1, M, M)            # :202
        conv(f"trans_feat_{lv}_f1", 1, M, M)           # :205
        conv(f"lang_feat_{lv}_f2", 1, M, M)
        conv(f"trans_feat_{lv}_f2", 1, M, M)
    # ConvLSTMCell, util/cell.py:36-66 (kernel [1,1]; normalize -> no bias)
    specs.append(("text_objseg/rnn/conv_lstm_cell/kernel", (1, 1, 2 * M, 4 * M), "glorot", ()))
    specs.append(("text_objseg/rnn/conv_lstm_cell/W_ci", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    specs.append(("text_objseg/rnn/conv_lstm_cell/W_cf", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    for i, _g in enumerate(("j", "i", "f")):
        s = "LayerNorm" if i == 0 else f"LayerNorm_{i}"
        specs.append((f"text_objseg/rnn/conv_lstm_cell/{s}/beta", (M,), "zeros", ()))
        specs.append((f"text_objseg/rnn/conv_lstm_cell/{s}/gamma", (M,), "ones", ()))
    specs.append(("text_objseg/rnn/conv_lstm_cell/W_co", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    for i in (3, 4):                       # o, c  (util/cell.py:64-66)
        specs.append((f"text_objseg/rnn/conv_lstm_cell/LayerNorm_{i}/beta", (M,), "zeros", ()))
        specs.append((f"text_objseg/rnn/conv_lstm_cell/LayerNorm_{i}/gamma", (M,), "ones", ()))
    conv("score", 3, M, 1)                 # :138
    return specs


def _fans(shape):
    # TF _compute_fans: receptive field = prod(shape[:-2]); fan_in = shape[-2]*rf, fan_out = shape[-1]*rf
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def init_head_params(cfg: Cfg, seed: int = 1234, glove_seed: int = 7, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Reference initialisers: xavier/glorot uniform +-sqrt(6/(fan_in+fan_out)),
    biases/beta 0, gamma 1; the GloVe table (missing blob data/Gref_emb.npy) ~ N(0, 0.4^2)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind, _ in head_param_specs(cfg):
        if kind in ("xavier", "glorot"):
            fi, fo = _fans(shape)
            lim = math.sqrt(6.0 / (fi + fo))
            t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim
        elif kind == "zeros":
            t = torch.zeros(shape, dtype=torch.float64)
        elif kind == "ones":
            t = torch.ones(shape, dtype=torch.float64)
        elif kind == "glove":
            gg = torch.Generator().manual_seed(glove_seed)
            t = torch.randn(shape, generator=gg, dtype=torch.float64) * 0.4
        else:
            raise ValueError(kind)
        out[name] = t.to(dtype)
    return out


# ----------------------------------------------------------------------------------------
# Backbone: DeepLab-ResNet-101, output stride 8 (deeplab_resnet/model.py:19-401,
# kaffe/tensorflow/network.py:105-270).  Frozen, inference BN.
# ----------------------------------------------------------------------------------------
def backbone_layout(cfg: Cfg):
    """[(conv_name, bn_name, k, cin, cout, stride, dilation)] in graph order + block wiring."""
    w = cfg.backbone_width
    convs = [("conv1", "bn_conv1", 7, 3, w, 2, 1)]
    blocks = []  # (stage_prefix, block_suffix, has_branch1, cin, mid, cout, stride, dilation)
    nb = cfg.backbone_blocks
    stage_cfg = [  # (stage, n_blocks, mid, cout, stride_first, dilation)
        (2, nb[0], w, 4 * w, 1, 1),          # model.py:23-55
        (3, nb[1], 2 * w, 8 * w, 2, 1),      # stride 2 on res3a 1x1s, model.py:60,64
        (4, nb[2], 4 * w, 16 * w, 1, 2),     # atrous rate 2, model.py:114
        (5, nb[3], 8 * w, 32 * w, 1, 4),     # atrous rate 4, model.py:371
    ]
    cin = w
    for stage, n, mid, cout, stride, dil in stage_cfg:
        for b in range(n):
            # naming: res2a/b/c, res3a/b1..b3, res4a/b1..b22, res5a/b/c (model.py:23-401)
            suf = "abc"[b] if stage in (2, 5) else ("a" if b == 0 else f"b{b}")
            blocks.append((stage, suf, b == 0, cin, mid, cout, stride if b == 0 else 1, dil))
            cin = cout
    return convs, blocks


def backbone_param_specs(cfg: Cfg):
    convs, blocks = backbone_layout(cfg)
    specs = []

    def add(conv_name, bn_name, k, cin, cout):
        specs.append((f"{conv_name}/weights", (k, k, cin, cout)))
        for s in ("gamma", "beta", "moving_mean", "moving_variance"):
            specs.append((f"{bn_name}/{s}", (cout,)))

    add("conv1", "bn_conv1", 7, 3, cfg.backbone_width)
    for stage, suf, has_b1, cin, mid, cout, stride, dil in blocks:
        p = f"{stage}{suf}"
        if has_b1:
            add(f"res{p}_branch1", f"bn{p}_branch1", 1, cin, cout)
        add(f"res{p}_branch2a", f"bn{p}_branch2a", 1, cin, mid)
        add(f"res{p}_branch2b", f"bn{p}_branch2b", 3, mid, mid)
        add(f"res{p}_branch2c", f"bn{p}_branch2c", 1, mid, cout)
    return specs


def init_backbone_params(cfg: Cfg, seed: int = 4321, dtype=torch.float32):
    """Synthetic backbone weights (the checkpoint deeplab_resnet_init.ckpt is not in the tree,
    trainval_model.py:50): He-normal convs; BN gamma=1 (0.2 on each block's last BN to keep
    activations bounded), beta=0, mean=0, var=1."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in backbone_param_specs(cfg):
        if name.endswith("/weights"):
            fan_in = shape[0] * shape[1] * shape[2]
            t = torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / fan_in)
        elif name.endswith("/gamma"):
            t = torch.full(shape, 0.2 if "branch2c" in name else 1.0, dtype=torch.float64)
        elif name.endswith("/moving_variance"):
            t = torch.ones(shape, dtype=torch.float64)
        else:
            t = torch.zeros(shape, dtype=torch.float64)
        out[name] = t.to(dtype)
    return out


def _same_pad(size, k, stride, dil):
    """TF 'SAME': out = ceil(in/stride); total = max((out-1)*stride + (k-1)*dil + 1 - in, 0);
    before = total // 2 (the extra pixel goes to the bottom/right)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - size, 0)
    return total // 2, total - total // 2


def tf_conv2d(x_nchw, w_hwio, stride=1, dilation=1):
    """tf.nn.conv2d / atrous_conv2d, padding='SAME' (network.py:105-188)."""
    k = w_hwio.shape[0]
    pt, pb = _same_pad(x_nchw.shape[2], k, stride, dilation)
    pl, pr = _same_pad(x_nchw.shape[3], k, stride, dilation)
    if pt or pb or pl or pr:
        x_nchw = F.pad(x_nchw, (pl, pr, pt, pb))
    return F.conv2d(x_nchw.contiguous(), w_hwio.permute(3, 2, 0, 1).contiguous(), stride=stride, dilation=dilation)


def _bn(x, p, name, relu):
    # slim.batch_norm(is_training=False, scale=True), epsilon 1e-3 (network.py:260-270)
    sc = p[f"{name}/gamma"] / torch.sqrt(p[f"{name}/moving_variance"] + 1e-3)
    sh = p[f"{name}/beta"] - p[f"{name}/moving_mean"] * sc
    y = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    return F.relu(y) if relu else y


def backbone_forward(bp: Dict[str, torch.Tensor], im_nhwc: torch.Tensor, cfg: Cfg, extra_taps=()):
    """im [B,H,W,3] (BGR, mean-subtracted) -> (c3, c4, c5) NHWC.
    Taps res3b3_relu / res4b22_relu / res5c_relu (CMPC_model.py:73-76) = outputs of stages 3/4/5.
    extra_taps: block names ("2b" = res2b_relu, CMPCv5_BiLSTM_model.py:88) appended to the result in the order given."""
    _, blocks = backbone_layout(cfg)
    x = im_nhwc.permute(0, 3, 1, 2)
    x = _bn(tf_conv2d(x, bp["conv1/weights"], stride=2), bp, "bn_conv1", True)   # model.py:20-21
    # max_pool 3x3 s2 SAME: padding ignored -> pad with -inf (model.py:22)
    pt, pb = _same_pad(x.shape[2], 3, 2, 1)
    pl, pr = _same_pad(x.shape[3], 3, 2, 1)
    x = F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
    taps = {}
    for stage, suf, has_b1, cin, mid, cout, stride, dil in blocks:
        p = f"{stage}{suf}"
        if has_b1:
            sc = _bn(tf_conv2d(x, bp[f"res{p}_branch1/weights"], stride=stride), bp, f"bn{p}_branch1", False)
        else:
            sc = x
        y = _bn(tf_conv2d(x, bp[f"res{p}_branch2a/weights"], stride=stride), bp, f"bn{p}_branch2a", True)
        y = _bn(tf_conv2d(y, bp[f"res{p}_branch2b/weights"], dilation=dil), bp, f"bn{p}_branch2b", True)
        y = _bn(tf_conv2d(y, bp[f"res{p}_branch2c/weights"]), bp, f"bn{p}_branch2c", False)
        x = F.relu(sc + y)
        taps[stage] = x
        taps[p] = x
    return tuple(taps[s].permute(0, 2, 3, 1).contiguous() for s in (3, 4, 5) + tuple(extra_taps))


# ----------------------------------------------------------------------------------------
# TF1 op semantics
# ----------------------------------------------------------------------------------------
def l2_normalize(x, dim=None, eps=1e-12):
    """tf.nn.l2_normalize: x * rsqrt(max(sum(x^2, axis), eps)); axis=None -> all dims."""
    if dim is None:
        ss = (x * x).sum()
    else:
        ss = (x * x).sum(dim=dim, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=eps))


def tf_layer_norm(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm defaults: begin_norm_axis=1 (moments over ALL non-batch
    axes), params over the last axis, variance_epsilon 1e-12, via tf.nn.batch_normalization."""
    dims = tuple(range(1, x.dim()))
    mean = x.mean(dim=dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def conv1x1(p, name, x):
    """_conv with filter_size 1 (CMPC_model.py:412-417): y = x . DW[0,0] + biases."""
    w = p[f"text_objseg/{name}/DW"]
    return x @ w[0, 0] + p[f"text_objseg/{name}/biases"]


def conv3x3(p, name, x_nhwc):
    w = p[f"text_objseg/{name}/DW"]
    y = tf_conv2d(x_nhwc.permute(0, 3, 1, 2), w)
    return y.permute(0, 2, 3, 1) + p[f"text_objseg/{name}/biases"]


def generate_spatial_batch(N, h, w, dtype=torch.float32):
    """util/processing_tools.py:5-17 (values computed in float64, stored float32)."""
    a = np.zeros((N, h, w, 8), dtype=np.float32)
    for y in range(h):
        for x in range(w):
            xmin = x / w * 2 - 1
            xmax = (x + 1) / w * 2 - 1
            xctr = (xmin + xmax) / 2
            ymin = y / h * 2 - 1
            ymax = (y + 1) / h * 2 - 1
            yctr = (ymin + ymax) / 2
            a[:, y, x, :] = [xmin, ymin, xmax, ymax, xctr, yctr, 1 / w, 1 / h]
    return torch.from_numpy(a).to(dtype)


def resize_bilinear(x_nhwc, H, W):
    """tf.image.resize_bilinear, align_corners=False, legacy (no half-pixel centres):
    src = dst * (in/out); lo = floor(src); hi = min(lo+1, in-1); lerp order as in TF's kernel."""
    B, h, w, C = x_nhwc.shape
    dt = x_nhwc.dtype

    def axis(n_in, n_out):
        scale = n_in / n_out
        src = torch.arange(n_out, dtype=torch.float64) * scale
        lo = torch.floor(src).long()
        hi = torch.clamp(lo + 1, max=n_in - 1)
        # TF computes the lerp weight in float32: in = out_idx * scale (float), lerp = in - floor(in)
        srcf = (torch.arange(n_out, dtype=torch.float32) * np.float32(scale))
        lerp = (srcf - torch.floor(srcf)).to(dt) if dt != torch.float64 else (src - torch.floor(src))
        return lo, hi, lerp

    ylo, yhi, yl = axis(h, H)
    xlo, xhi, xl = axis(w, W)
    top = x_nhwc[:, ylo]          # [B,H,w,C]
    bot = x_nhwc[:, yhi]
    xl_ = xl.view(1, 1, W, 1)
    yl_ = yl.view(1, H, 1, 1)
    t = top[:, :, xlo] + (top[:, :, xhi] - top[:, :, xlo]) * xl_
    b = bot[:, :, xlo] + (bot[:, :, xhi] - bot[:, :, xlo]) * xl_
    return t + (b - t) * yl_


def sigmoid_xent(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


# ----------------------------------------------------------------------------------------
# The CMPC head (CMPC_model.py:89-410)
# ----------------------------------------------------------------------------------------
def text_lstm(p, words, seq_len, cfg: Cfg):
    """lstm(), CMPC_model.py:144-164.  tf LSTMCell: gates i,j,f,o on [x|h].kernel + bias,
    forget_bias 1.0; dynamic_rnn(sequence_length): zero outputs / frozen state past length."""
    emb = p["text_objseg/Variable"][words.long()]          # [B,T,glove]
    B, T, _ = emb.shape
    R = cfg.rnn_size
    K = p["text_objseg/rnn/lstm_cell/kernel"]
    bias = p["text_objseg/rnn/lstm_cell/bias"]
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
    outputs = torch.stack(outs, 1)                          # [B,T,R]
    words_feat = l2_normalize(outputs, -1).unsqueeze(1)     # [B,1,T,R]  :159-160
    seq_mask = (words_feat.abs().sum(-1, keepdim=True) != 0).to(emb.dtype)   # :163
    return words_feat, seq_mask


def lang_parser(p, words_feat, seq_mask):
    """build_lang_parser, CMPC_model.py:347-357."""
    x = F.relu(conv1x1(p, "words_parse_1", words_feat))
    x = conv1x1(p, "words_parse_2", x)
    return torch.softmax(x, 3) * seq_mask                   # [B,1,T,4]


def weighted_lang(words_parse, words_feat, cols):
    """valid_lang (:166-178, cols=(0,1)) / nec_lang (:180-192, cols=(0,1,2) == sum - [3])."""
    B, _, T, Rr = words_feat.shape
    if cols == "valid":
        wts = words_parse[:, :, :, 0] + words_parse[:, :, :, 1]
    else:
        wts = words_parse.sum(3) - words_parse[:, :, :, 3]
    v = wts @ words_feat.reshape(B, T, Rr)                  # [B,1,R]
    return l2_normalize(v, 2).reshape(B, 1, 1, Rr)


def mutan_fusion(p, lang, spatial, vis, lv):
    """mutan_head x5 + mutan_fusion, CMPC_model.py:295-328."""
    heads = []
    xs = torch.cat([vis, spatial], 3)
    for h in range(1, 6):
        vt = torch.tanh(conv1x1(p, f"vis_trans_{lv}_head{h}", xs))
        lt = torch.tanh(conv1x1(p, f"lang_trans_{lv}_head{h}", lang))
        heads.append(vt * lt)
    fused = torch.stack(heads, 4).sum(4)
    return l2_normalize(torch.tanh(fused), 3)


def graph_conv(p, graph_feat, adj, lv, taps=None):
    """graph_conv, CMPC_model.py:359-374.  graph_feat [B,1,N,C], adj [B,N,N]."""
    B, _, N, C = graph_feat.shape
    g = (adj @ graph_feat.reshape(B, N, C)).reshape(B, 1, N, C)
    g = tf_layer_norm(g, p[f"text_objseg/gconv_feat_ln_spa_graph_{lv}/gamma"],
                      p[f"text_objseg/gconv_feat_ln_spa_graph_{lv}/beta"])
    g = F.relu(graph_feat + g)
    u = conv1x1(p, f"gconv_update_spa_graph_{lv}", g)
    u = tf_layer_norm(u, p[f"text_objseg/gconv_update_ln_spa_graph_{lv}/gamma"],
                      p[f"text_objseg/gconv_update_ln_spa_graph_{lv}/beta"])
    return F.relu(u)


def build_spa_graph(p, spa_graph, words_feat, words_parse, seq_mask, lv, cfg: Cfg, taps):
    """build_spa_graph, CMPC_model.py:376-410 (adjacency materialised exactly as written)."""
    B, T, N, C = cfg.batch_size, cfg.num_steps, cfg.N, cfg.v_emb_dim
    words_trans = conv1x1(p, f"words_trans_{lv}", words_feat).reshape(B, T, cfg.rnn_size)
    t2 = conv1x1(p, f"spa_graph_trans2_{lv}", spa_graph).reshape(B, N, C)
    affi = t2 @ words_trans.transpose(1, 2)                 # [B,N,T]
    affi = affi / (C ** 0.5)
    affi = words_parse[:, :, :, 2] * affi                   # [B,1,T] * [B,N,T]
    mask = seq_mask.reshape(B, 1, T)
    mask_softmax = (1 - mask) * F32_MIN
    gw_w = torch.softmax(mask * affi + mask_softmax, 2)
    gw_v = torch.softmax(affi, 1) * mask
    adj = gw_w @ gw_v.transpose(1, 2)                       # [B,N,N]
    taps[f"gw_w_{lv}"] = gw_w
    taps[f"gw_v_{lv}"] = gw_v
    g = graph_conv(p, spa_graph.reshape(B, 1, N, C), adj, lv)
    g = g.reshape(B, cfg.vf_h, cfg.vf_w, C)
    return l2_normalize(g, 3)


def build_lang2vis(p, vis, words_feat, words_parse, seq_mask, spatial, lv, cfg, taps):
    """build_lang2vis, CMPC_model.py:330-345."""
    vl = weighted_lang(words_parse, words_feat, "valid")
    vis_la_sp = mutan_fusion(p, vl, spatial, vis, lv)
    taps[f"vis_la_sp_{lv}"] = vis_la_sp
    spa = build_spa_graph(p, vis_la_sp, words_feat, words_parse, seq_mask, lv, cfg, taps)
    taps[f"spa_graph_{lv}"] = spa
    lang_tile = vl.expand(-1, cfg.vf_h, cfg.vf_w, -1)
    feat_all = torch.cat([vis_la_sp, spa, lang_tile, spatial], 3)
    return F.relu(conv1x1(p, f"fusion_{lv}", feat_all))


def global_vec(p, feat, lang, lv, cfg):
    """global_vec, CMPC_model.py:212-243.  NOTE l2_normalize with no axis -> whole [B,1,1,M]."""
    B, N, M = cfg.batch_size, cfg.N, cfg.mlp_dim
    key = conv1x1(p, f"spa_graph_key_{lv}", feat).reshape(B, N, M)
    q = conv1x1(p, f"lang_query_{lv}", lang).reshape(B, 1, M)
    attn = key @ q.transpose(1, 2) / (M ** 0.5)             # [B,N,1]
    attn = torch.softmax(attn, 1)
    pooled = (attn.transpose(1, 2) @ feat.reshape(B, N, M)).reshape(B, 1, 1, M)
    gv = conv1x1(p, f"gv_lang_{lv}", torch.cat([pooled, lang], 3))
    return l2_normalize(gv, None)


def lang_se(p, feat, gv, lv):
    """lang_se, CMPC_model.py:194-210."""
    g = torch.sigmoid(conv1x1(p, f"lang_feat_{lv}", gv))
    return F.relu(conv1x1(p, f"trans_feat_{lv}", feat)) * g


def gated_exchange_module(p, feat, feat1, feat2, lang, lv, cfg):
    gv = global_vec(p, feat, lang, lv + "gv_f1", cfg)
    return feat + lang_se(p, feat1, gv, lv + "_f1") + lang_se(p, feat2, gv, lv + "_f2")


def conv_lstm(p, xs, cfg: Cfg, taps=None):
    """ConvLSTMCell.call over the 3 stacked maps, util/cell.py:36-79 via CMPC_model.py:287-290."""
    pre = "text_objseg/rnn/conv_lstm_cell/"
    M = cfg.mlp_dim
    W = p[pre + "kernel"][0, 0]
    B = xs[0].shape[0]
    c = torch.zeros(B, cfg.vf_h, cfg.vf_w, M, dtype=xs[0].dtype)
    h = torch.zeros_like(c)

    def ln(x, i):
        s = "LayerNorm" if i == 0 else f"LayerNorm_{i}"
        return tf_layer_norm(x, p[pre + s + "/gamma"], p[pre + s + "/beta"])

    for x in xs:
        y = torch.cat([x, h], 3) @ W
        j, i, f, o = y.split(M, 3)                          # util/cell.py:46
        i = i + p[pre + "W_ci"] * c
        f = f + p[pre + "W_cf"] * c
        j, i, f = ln(j, 0), ln(i, 1), ln(f, 2)
        f = torch.sigmoid(f + 1.0)
        i = torch.sigmoid(i)
        c = c * f + i * torch.tanh(j)
        o = o + p[pre + "W_co"] * c
        o, c = ln(o, 3), ln(c, 4)
        o = torch.sigmoid(o)
        h = o * torch.tanh(c)
    return h


def head_forward(p, feats, words, seq_len, cfg: Cfg):
    """build_graph(), CMPC_model.py:89-142.  feats = (c3, c4, c5) NHWC.  Returns all taps."""
    c3, c4, c5 = feats
    taps = {}
    words_feat, seq_mask = text_lstm(p, words, seq_len, cfg)
    taps["words_feat"], taps["seq_mask"] = words_feat, seq_mask
    v5 = l2_normalize(conv1x1(p, "c5_lateral", c5), 3)
    v4 = l2_normalize(conv1x1(p, "c4_lateral", c4), 3)
    v3 = l2_normalize(conv1x1(p, "c3_lateral", c3), 3)
    taps["lat_c5"], taps["lat_c4"], taps["lat_c3"] = v5, v4, v3
    spatial = generate_spatial_batch(cfg.batch_size, cfg.vf_h, cfg.vf_w, dtype=c5.dtype)
    words_parse = lang_parser(p, words_feat, seq_mask)
    taps["words_parse"] = words_parse
    fus = {}
    for lv, v in (("c5", v5), ("c4", v4), ("c3", v3)):
        fus[lv] = build_lang2vis(p, v, words_feat, words_parse, seq_mask, spatial, lv, cfg, taps)
        taps[f"fusion_{lv}"] = fus[lv]
    for lv in LEVELS:
        sc = conv3x3(p, f"score_{lv}", fus[lv])
        taps[f"score_{lv}"] = sc
        taps[f"up_{lv}"] = resize_bilinear(sc, cfg.H, cfg.W)
    nec = weighted_lang(words_parse, words_feat, "nec")
    taps["nec_lang"] = nec
    # gated_exchange_fusion_lstm_2times, :261-293
    f3, f4, f5 = fus["c3"], fus["c4"], fus["c5"]
    e3 = l2_normalize(gated_exchange_module(p, f3, f4, f5, nec, "c3", cfg), 3)
    e4 = l2_normalize(gated_exchange_module(p, f4, f3, f5, nec, "c4", cfg), 3)
    e5 = l2_normalize(gated_exchange_module(p, f5, f3, f4, nec, "c5", cfg), 3)
    taps["exg_c3"], taps["exg_c4"], taps["exg_c5"] = e3, e4, e5
    e32 = l2_normalize(gated_exchange_module(p, e3, e4, e5, nec, "c3_2", cfg), 3)
    e42 = l2_normalize(gated_exchange_module(p, e4, e3, e5, nec, "c4_2", cfg), 3)
    e52 = l2_normalize(gated_exchange_module(p, e5, e3, e4, nec, "c5_2", cfg), 3)
    taps["exg_c3_2"], taps["exg_c4_2"], taps["exg_c5_2"] = e32, e42, e52
    fused = conv_lstm(p, (e32, e42, e52), cfg)
    taps["fused"] = fused
    pred = conv3x3(p, "score", fused)
    taps["pred"] = pred
    taps["up"] = resize_bilinear(pred, cfg.H, cfg.W)
    taps["sigm"] = torch.sigmoid(taps["up"])
    return taps


def losses(p, taps, target_fine, cfg: Cfg):
    """train_op() loss part, CMPC_model.py:438-447,486-490; util/loss.py:6-16,28-32."""
    def wll(scores):
        return sigmoid_xent(scores, target_fine).sum(dim=(1, 2, 3)).mean()
    out = {
        "loss_c5": wll(taps["up_c5"]), "loss_c4": wll(taps["up_c4"]),
        "loss_c3": wll(taps["up_c3"]), "loss_last": wll(taps["up"]),
    }
    out["loss_all"] = 0.7 * out["loss_last"] + 0.1 * out["loss_c5"] + 0.1 * out["loss_c4"] + 0.1 * out["loss_c3"]
    reg = 0.0
    for name, _, _, flags in head_param_specs(cfg):
        if "reg" in flags:
            reg = reg + (p[name] ** 2).sum() / 2
    out["reg_loss"] = cfg.weight_decay * reg
    out["cost"] = out["loss_all"] + out["reg_loss"]
    pred = taps["up"] > 0
    labl = target_fine != 0
    inter = (pred & labl).sum(dim=(1, 2, 3)).to(torch.float64)
    union = (pred | labl).sum(dim=(1, 2, 3)).to(torch.float64)
    out["mIoU"] = (inter / union).mean()
    return out


def poly_lr(step, cfg: Cfg):
    """tf.train.polynomial_decay (CMPC_model.py:451-452), cycle=False."""
    gs = min(step, cfg.lr_decay_step)
    return (cfg.start_lr - cfg.end_lr) * (1 - gs / cfg.lr_decay_step) ** cfg.lr_power + cfg.end_lr


class TFAdam:
    """tf.train.AdamOptimizer defaults (CMPC_model.py:456,478): beta 0.9/0.999, eps 1e-8,
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, params):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, params, grads, lr):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = lr * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
        for k in params:
            g = grads[k]
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            params[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))


def grads_of(p, feats, words, seq_len, target_fine, cfg: Cfg):
    """d cost / d every head parameter (optimizer.compute_gradients, CMPC_model.py:461) with the
    x2 multiplier on 'biases' applied (:462-475).  Returns (scalars, grads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    taps = head_forward(leaves, feats, words, seq_len, cfg)
    ls = losses(leaves, taps, target_fine, cfg)
    names = list(leaves)
    gs = torch.autograd.grad(ls["cost"], [leaves[n] for n in names], allow_unused=True)
    flags = {n: f for n, _, _, f in head_param_specs(cfg)}
    grads = {}
    for n, g in zip(names, gs):
        g = torch.zeros_like(leaves[n]) if g is None else g
        if "x2" in flags[n]:
            g = g * 2.0
        grads[n] = g.detach()
    scal = {k: float(v.detach()) for k, v in ls.items()}
    return scal, grads, {k: v.detach() for k, v in taps.items()}


def train_step(p, opt: TFAdam, step, feats, words, seq_len, target_fine, cfg: Cfg):
    """One sess.run([train, ...]) (trainval_model.py:98-107): params updated in place."""
    scal, grads, _ = grads_of(p, feats, words, seq_len, target_fine, cfg)
    lr = poly_lr(step, cfg)
    with torch.no_grad():
        opt.step(p, grads, lr)
    scal["lr"] = lr
    return scal


def conv5_trainable(bp):
    """Backbone variables train_op() adds with conv5=True (CMPC_model.py:427-430): names starting with res3 / res4 / res5, i.e. the
    convolution weights only (the batch-norm variables are named bn...)."""
    return [k for k in bp if k.startswith(("res3", "res4", "res5"))]


def grads_of_conv5(p, bp, im, words, seq_len, target_fine, cfg: Cfg):
    """compute_gradients with conv5=True: d cost / d (head parameters, res3-res5 convolution weights); the L2 term also covers the trained
    backbone weights (reg_var_list: name[-9:-2] == 'weights', CMPC_model.py:433).  Returns (scalars, head grads, backbone grads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    names_b = conv5_trainable(bp)
    bl = dict(bp)
    for n in names_b:
        bl[n] = bp[n].detach().clone().requires_grad_(True)
    feats = backbone_forward(bl, im, cfg)
    taps = head_forward(leaves, feats, words, seq_len, cfg)
    ls = losses(leaves, taps, target_fine, cfg)
    cost = ls["cost"] + cfg.weight_decay * sum(0.5 * (bl[n] ** 2).sum() for n in names_b)
    names = list(leaves)
    gs = torch.autograd.grad(cost, [leaves[n] for n in names] + [bl[n] for n in names_b], allow_unused=True)
    flags = {n: f for n, _, _, f in head_param_specs(cfg)}
    grads = {}
    for n, g in zip(names, gs[:len(names)]):
        g = torch.zeros_like(leaves[n]) if g is None else g
        grads[n] = (g * 2.0 if "x2" in flags[n] else g).detach()
    gb = {n: g.detach() for n, g in zip(names_b, gs[len(names):])}
    return {k: float(v.detach()) for k, v in ls.items()}, grads, gb


def train_step_conv5(p, bp, opt: TFAdam, opt_b: TFAdam, step, im, words, seq_len, target_fine, cfg: Cfg):
    """One train step with conv5=True: head parameters and res3-res5 weights updated in place (two TFAdam objects sharing the step)."""
    scal, grads, gb = grads_of_conv5(p, bp, im, words, seq_len, target_fine, cfg)
    lr = poly_lr(step, cfg)
    with torch.no_grad():
        opt.step(p, grads, lr)
        opt_b.step({n: bp[n] for n in gb}, gb, lr)
    scal["lr"] = lr
    return scal


# ----------------------------------------------------------------------------------------
# Synthetic inputs
# ----------------------------------------------------------------------------------------
MU = np.array((104.00698793, 116.66876762, 122.67891434), dtype=np.float32)  # trainval_model.py:371


def synth_batch(cfg: Cfg, seed=0):
    rng = np.random.default_rng(seed)
    B, T = cfg.batch_size, cfg.num_steps
    im = rng.integers(0, 256, size=(B, cfg.H, cfg.W, 3), dtype=np.uint8).astype(np.float32)
    im = im[:, :, :, ::-1] - MU                      # RGB->BGR, minus mean (trainval_model.py:90-91)
    seq_len = rng.integers(min(3, T), T + 1, size=(B,)).astype(np.int32)
    seq_len[0] = T
    words = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        words[b, :seq_len[b]] = rng.integers(min(4, cfg.vocab_size - 1), cfg.vocab_size, size=(seq_len[b],))
    target = np.zeros((B, cfg.H, cfg.W, 1), dtype=np.float32)
    lo, hi = max(cfg.H // 8, 1), max(cfg.H * 5 // 8, 2)
    for b in range(B):
        hh, ww = rng.integers(lo, hi + 1, size=2)
        y0 = rng.integers(0, cfg.H - hh + 1)
        x0 = rng.integers(0, cfg.W - ww + 1)
        target[b, y0:y0 + hh, x0:x0 + ww, 0] = 1.0
    return (torch.from_numpy(words), torch.from_numpy(np.ascontiguousarray(im)),
            torch.from_numpy(seq_len), torch.from_numpy(target))
