"""ORACLE (test infrastructure only) -- independent NumPy float64 restatement of the CMPC
forward path, written against the reference source without sharing code with
oracle/cmpc_torch.py, so that a mis-used torch op in one shows up as a disagreement.

PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference).
Only tests/ may import this file.  Citations are file:line in /root/reference.
"""
import numpy as np

F32_MIN = float(np.finfo(np.float32).min)


def _l2n(x, axis, eps=1e-12):
    # tf.nn.l2_normalize
    ss = np.sum(x * x, axis=axis, keepdims=axis is not None)
    return x / np.sqrt(np.maximum(ss, eps))


def _ln(x, gamma, beta, eps=1e-12):
    # tf.contrib.layers.layer_norm: moments over all non-batch axes, params on last axis
    ax = tuple(range(1, x.ndim))
    mu = x.mean(axis=ax, keepdims=True)
    var = x.var(axis=ax, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _c1(p, name, x):
    # _conv 1x1, CMPC_model.py:412-417
    return x @ p["text_objseg/%s/DW" % name][0, 0] + p["text_objseg/%s/biases" % name]


def conv_same(x, w, stride=1, dil=1):
    """NHWC x HWIO 'SAME' convolution (tf.nn.conv2d / atrous_conv2d) by explicit loops over taps."""
    B, H, W, Ci = x.shape
    k = w.shape[0]
    oh, ow = -(-H // stride), -(-W // stride)
    th = max((oh - 1) * stride + (k - 1) * dil + 1 - H, 0)
    tw = max((ow - 1) * stride + (k - 1) * dil + 1 - W, 0)
    xp = np.pad(x, ((0, 0), (th // 2, th - th // 2), (tw // 2, tw - tw // 2), (0, 0)))
    out = np.zeros((B, oh, ow, w.shape[3]), dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            patch = xp[:, i * dil: i * dil + (oh - 1) * stride + 1: stride,
                       j * dil: j * dil + (ow - 1) * stride + 1: stride, :]
            out += patch @ w[i, j]
    return out


def spatial_grid(B, h, w):
    # util/processing_tools.py:5-17
    a = np.zeros((B, h, w, 8), dtype=np.float32)
    for y in range(h):
        for x in range(w):
            xmin, xmax = x / w * 2 - 1, (x + 1) / w * 2 - 1
            ymin, ymax = y / h * 2 - 1, (y + 1) / h * 2 - 1
            a[:, y, x] = [xmin, ymin, xmax, ymax, (xmin + xmax) / 2, (ymin + ymax) / 2, 1 / w, 1 / h]
    return a.astype(np.float64)


def resize_bilinear(x, H, W):
    # tf.image.resize_bilinear legacy (align_corners=False)
    B, h, w, C = x.shape
    out = np.zeros((B, H, W, C), dtype=x.dtype)
    for Y in range(H):
        sy = Y * (h / H)
        y0 = int(np.floor(sy)); y1 = min(y0 + 1, h - 1); fy = sy - y0
        for X in range(W):
            sx = X * (w / W)
            x0 = int(np.floor(sx)); x1 = min(x0 + 1, w - 1); fx = sx - x0
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * fx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * fx
            out[:, Y, X] = top + (bot - top) * fy
    return out


def head_forward(p, feats, words, seq_len, dims):
    """dims: dict(B,T,h,w,H,W,C,M,R).  p: name -> float64 ndarray.  Follows CMPC_model.py:89-142."""
    B, T, h, w, H, W, C, M, R = (dims[k] for k in "B T h w H W C M R".split())
    N = h * w
    c3, c4, c5 = feats
    taps = {}
    # lstm(): CMPC_model.py:144-164
    emb = p["text_objseg/Variable"][words]
    K, bias = p["text_objseg/rnn/lstm_cell/kernel"], p["text_objseg/rnn/lstm_cell/bias"]
    hs, cs = np.zeros((B, R)), np.zeros((B, R))
    outs = np.zeros((B, T, R))
    for t in range(T):
        z = np.concatenate([emb[:, t], hs], 1) @ K + bias
        i, j, f, o = np.split(z, 4, axis=1)
        cn = _sig(f + 1.0) * cs + _sig(i) * np.tanh(j)
        hn = _sig(o) * np.tanh(cn)
        for b in range(B):
            if t < seq_len[b]:
                outs[b, t] = hn[b]; hs[b] = hn[b]; cs[b] = cn[b]
    wf = _l2n(outs, -1)[:, None]                                   # [B,1,T,R]
    mask = (np.abs(wf).sum(-1, keepdims=True) != 0).astype(np.float64)
    taps["words_feat"], taps["seq_mask"] = wf, mask
    lat = {"c5": _l2n(_c1(p, "c5_lateral", c5), 3), "c4": _l2n(_c1(p, "c4_lateral", c4), 3),
           "c3": _l2n(_c1(p, "c3_lateral", c3), 3)}
    sp = spatial_grid(B, h, w)
    # build_lang_parser :347-357
    wp = _softmax(_c1(p, "words_parse_2", np.maximum(_c1(p, "words_parse_1", wf), 0)), 3) * mask
    taps["words_parse"] = wp
    wfr = wf.reshape(B, T, R)

    def lang(weights):                                               # :166-192
        v = weights @ wfr
        return _l2n(v, 2).reshape(B, 1, 1, R)

    vl = lang(wp[..., 0] + wp[..., 1])
    fus = {}
    for lv in ("c5", "c4", "c3"):
        vis = lat[lv]
        xs = np.concatenate([vis, sp], 3)
        acc = 0
        for hd in range(1, 6):                                       # :295-322
            acc = acc + np.tanh(_c1(p, "vis_trans_%s_head%d" % (lv, hd), xs)) * \
                np.tanh(_c1(p, "lang_trans_%s_head%d" % (lv, hd), vl))
        vls = _l2n(np.tanh(acc), 3)
        taps["vis_la_sp_" + lv] = vls
        # build_spa_graph :376-410
        wt = _c1(p, "words_trans_" + lv, wf).reshape(B, T, R)
        t2 = _c1(p, "spa_graph_trans2_" + lv, vls).reshape(B, N, C)
        affi = np.einsum("bnc,btc->bnt", t2, wt) / C ** 0.5
        affi = wp[:, :, :, 2] * affi
        gm = mask.reshape(B, 1, T)
        gw_w = _softmax(gm * affi + (1 - gm) * F32_MIN, 2)
        gw_v = _softmax(affi, 1) * gm
        taps["gw_w_" + lv], taps["gw_v_" + lv] = gw_w, gw_v
        adj = gw_w @ gw_v.transpose(0, 2, 1)
        X = vls.reshape(B, 1, N, C)
        g = (adj @ vls.reshape(B, N, C)).reshape(B, 1, N, C)        # graph_conv :359-374
        g = _ln(g, p["text_objseg/gconv_feat_ln_spa_graph_%s/gamma" % lv], p["text_objseg/gconv_feat_ln_spa_graph_%s/beta" % lv])
        g = np.maximum(X + g, 0)
        u = _c1(p, "gconv_update_spa_graph_" + lv, g)
        u = _ln(u, p["text_objseg/gconv_update_ln_spa_graph_%s/gamma" % lv], p["text_objseg/gconv_update_ln_spa_graph_%s/beta" % lv])
        spa = _l2n(np.maximum(u, 0).reshape(B, h, w, C), 3)
        taps["spa_graph_" + lv] = spa
        allf = np.concatenate([vls, spa, np.broadcast_to(vl, (B, h, w, R)), sp], 3)
        fus[lv] = np.maximum(_c1(p, "fusion_" + lv, allf), 0)
        taps["fusion_" + lv] = fus[lv]
        sc = conv_same(fus[lv], p["text_objseg/score_%s/DW" % lv]) + p["text_objseg/score_%s/biases" % lv]
        taps["up_" + lv] = resize_bilinear(sc, H, W)
    nec = lang(wp.sum(3) - wp[..., 3])

    def exch(feat, f1, f2, lv):                                      # :212-259
        key = _c1(p, "spa_graph_key_%sgv_f1" % lv, feat).reshape(B, N, M)
        q = _c1(p, "lang_query_%sgv_f1" % lv, nec).reshape(B, 1, M)
        attn = _softmax(key @ q.transpose(0, 2, 1) / M ** 0.5, 1)
        pooled = (attn.transpose(0, 2, 1) @ feat.reshape(B, N, M)).reshape(B, 1, 1, M)
        gv = _l2n(_c1(p, "gv_lang_%sgv_f1" % lv, np.concatenate([pooled, nec], 3)), None)
        s1 = np.maximum(_c1(p, "trans_feat_%s_f1" % lv, f1), 0) * _sig(_c1(p, "lang_feat_%s_f1" % lv, gv))
        s2 = np.maximum(_c1(p, "trans_feat_%s_f2" % lv, f2), 0) * _sig(_c1(p, "lang_feat_%s_f2" % lv, gv))
        return _l2n(feat + s1 + s2, 3)

    f3, f4, f5 = fus["c3"], fus["c4"], fus["c5"]
    e3, e4, e5 = exch(f3, f4, f5, "c3"), exch(f4, f3, f5, "c4"), exch(f5, f3, f4, "c5")
    e32, e42, e52 = exch(e3, e4, e5, "c3_2"), exch(e4, e3, e5, "c4_2"), exch(e5, e3, e4, "c5_2")
    taps["exg_c5_2"] = e52
    # ConvLSTMCell util/cell.py:36-79
    pre = "text_objseg/rnn/conv_lstm_cell/"
    Wk = p[pre + "kernel"][0, 0]
    c = np.zeros((B, h, w, M)); hh = np.zeros((B, h, w, M))
    lnn = lambda x, i: _ln(x, p[pre + ("LayerNorm" if i == 0 else "LayerNorm_%d" % i) + "/gamma"],
                           p[pre + ("LayerNorm" if i == 0 else "LayerNorm_%d" % i) + "/beta"])
    for x in (e32, e42, e52):
        y = np.concatenate([x, hh], 3) @ Wk
        j, i, f, o = np.split(y, 4, axis=3)
        i = i + p[pre + "W_ci"] * c
        f = f + p[pre + "W_cf"] * c
        j, i, f = lnn(j, 0), lnn(i, 1), lnn(f, 2)
        c = c * _sig(f + 1.0) + _sig(i) * np.tanh(j)
        o = o + p[pre + "W_co"] * c
        o, c = lnn(o, 3), lnn(c, 4)
        hh = _sig(o) * np.tanh(c)
    taps["fused"] = hh
    pred = conv_same(hh, p["text_objseg/score/DW"]) + p["text_objseg/score/biases"]
    taps["pred"] = pred
    taps["up"] = resize_bilinear(pred, H, W)
    taps["sigm"] = _sig(taps["up"])
    return taps
