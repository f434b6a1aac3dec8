"""ORACLE (test infrastructure only) -- independent NumPy float64 restatement of the CMPCv5_BiLSTM forward path (and its HSV
variant), written against the reference source without sharing code with oracle/cmpc_v5_torch.py (only the elementary helpers of
oracle/cmpc_numpy.py: l2-normalise, layer-norm, softmax, SAME convolution by explicit tap loops, legacy bilinear by explicit pixel
loops), so that a mis-used torch op in one shows up as a disagreement.

PARITY UNPINNED (no TensorFlow / slim here, no golden vectors in the reference).  Only tests/ may import this file.
Citations: v5 = /root/reference/CMPCv5_BiLSTM_model.py, hsv = CMPCv5_BiLSTM_HSV_model.py.
"""
import numpy as np

from oracle.cmpc_numpy import _c1, _l2n, _ln, _sig, _softmax, conv_same, resize_bilinear, spatial_grid

MU = np.array((104.00698793, 116.66876762, 122.67891434), dtype=np.float32)
BN_EPS = 1e-5


def _lstm(emb, K, bias, seq_len, R):
    B, T, _ = emb.shape
    hs, cs, outs = np.zeros((B, R)), np.zeros((B, R)), np.zeros((B, T, R))
    for t in range(T):
        z = np.concatenate([emb[:, t], hs], 1) @ K + bias
        i, j, f, o = np.split(z, 4, axis=1)
        cn = _sig(f + 1.0) * cs + _sig(i) * np.tanh(j)
        hn = _sig(o) * np.tanh(cn)
        for b in range(B):
            if t < seq_len[b]:
                outs[b, t] = hn[b]; hs[b] = hn[b]; cs[b] = cn[b]
    return outs


def _revseq(x, seq_len):
    out = x.copy()
    for b in range(x.shape[0]):
        n = int(seq_len[b])
        out[b, :n] = x[b, :n][::-1]
    return out


def _hsv(rgb):
    # tf.image.rgb_to_hsv, pixel by pixel
    out = np.zeros_like(rgb)
    flat_in, flat_out = rgb.reshape(-1, 3), out.reshape(-1, 3)
    for i in range(flat_in.shape[0]):
        r, g, b = flat_in[i]
        v = max(r, g, b); rng = v - min(r, g, b)
        s = rng / v if v > 0 else 0.0
        if rng > 0:
            if r == v:
                h = (g - b) / (6 * rng)
            elif g == v:
                h = (b - r) / (6 * rng) + 2.0 / 6.0
            else:
                h = (r - g) / (6 * rng) + 4.0 / 6.0
            if h < 0:
                h += 1.0
        else:
            h = 0.0
        flat_out[i] = (h, s, v)
    return out


def _conv_bn_relu(p, bn, scope, x, train, rate=1):
    # slim conv2d (no bias) + batch_norm (eps 1e-5) + relu; training: batch mean / biased variance
    y = conv_same(x, p["text_objseg/%s/weights" % scope], 1, rate)
    if train:
        mean = y.reshape(-1, y.shape[-1]).mean(0)
        var = y.reshape(-1, y.shape[-1]).var(0)
    else:
        mean, var = bn["text_objseg/%s/BatchNorm/moving_mean" % scope], bn["text_objseg/%s/BatchNorm/moving_variance" % scope]
    y = (y - mean) / np.sqrt(var + BN_EPS) * p["text_objseg/%s/BatchNorm/gamma" % scope] + p["text_objseg/%s/BatchNorm/beta" % scope]
    return np.maximum(y, 0)


def head_forward(p, bn, feats, words, seq_len, dims, im=None):
    """dims: dict(B,T,h,w,H,W,C,M,R, hsv, train, rates).  feats = (c2, c4, c5).  Follows v5:101-157."""
    B, T, h, w, H, W, C, M, R = (dims[k] for k in "B T h w H W C M R".split())
    N = h * w
    c2, c4, c5 = feats
    taps = {}
    # BiLSTM(): v5:159-187
    emb = p["text_objseg/Variable"][words]
    pre = "text_objseg/bidirectional_rnn/"
    fw = _lstm(emb, p[pre + "fw/lstm_cell/kernel"], p[pre + "fw/lstm_cell/bias"], seq_len, R)
    bw = _revseq(_lstm(_revseq(emb, seq_len), p[pre + "bw/lstm_cell/kernel"], p[pre + "bw/lstm_cell/bias"], seq_len, R), seq_len)
    cat = np.concatenate([fw, bw], -1)[:, None]
    mask = (np.abs(cat).sum(-1, keepdims=True) != 0).astype(np.float64)
    wf = _l2n(np.tanh(_c1(p, "words_feat", cat)), -1)
    taps["words_feat"], taps["seq_mask"] = wf, mask
    if dims.get("hsv"):
        rgb = (im.astype(np.float32) + MU)[..., ::-1].astype(np.float64)          # hsv:122-123 (float32 add, like the graph)
        hsv = resize_bilinear(_hsv(rgb), h, w)
        taps["hsv"] = hsv
        c5, c4 = np.concatenate([c5, hsv], -1), np.concatenate([c4, hsv], -1)
    lat = {"c5": _l2n(np.tanh(_c1(p, "c5_lateral", c5)), 3), "c4": _l2n(np.tanh(_c1(p, "c4_lateral", c4)), 3)}
    taps["lat_c5"], taps["lat_c4"] = lat["c5"], lat["c4"]
    sp = spatial_grid(B, h, w)
    wp = _softmax(_c1(p, "words_parse_2", np.maximum(_c1(p, "words_parse_1", wf), 0)), 3) * mask
    taps["words_parse"] = wp
    wfr = wf.reshape(B, T, R)

    def lang(weights):
        return _l2n(weights @ wfr, 2).reshape(B, 1, 1, R)

    vl = lang(wp[..., 0] + wp[..., 1])
    fus = {}
    for lv in ("c5", "c4"):
        xs = np.concatenate([lat[lv], sp], 3)
        acc = 0
        for hd in range(1, 6):
            acc = acc + np.tanh(_c1(p, "vis_trans_%s_head%d" % (lv, hd), xs)) * np.tanh(_c1(p, "lang_trans_%s_head%d" % (lv, hd), vl))
        vls = _l2n(np.tanh(acc), 3)
        taps["vis_la_sp_" + lv] = vls
        wt = _c1(p, "words_trans_" + lv, wf).reshape(B, T, R)
        t2 = _c1(p, "spa_graph_trans2_" + lv, vls).reshape(B, N, C)
        affi = wp[:, :, :, 2] * (np.einsum("bnc,btc->bnt", t2, wt) / C ** 0.5)
        gm = mask.reshape(B, 1, T)
        gw_w = gm * _softmax(affi, 2)                                              # v5:486-487: mask AFTER the softmax over words
        gw_v = gm * _softmax(affi, 1)
        taps["gw_w_" + lv], taps["gw_v_" + lv] = gw_w, gw_v
        adj = gw_w @ gw_v.transpose(0, 2, 1)
        X = vls.reshape(B, 1, N, C)
        g = (adj @ vls.reshape(B, N, C)).reshape(B, 1, N, C)
        g = _ln(g, p["text_objseg/gconv_feat_ln_spa_graph_%s/gamma" % lv], p["text_objseg/gconv_feat_ln_spa_graph_%s/beta" % lv])
        g = np.maximum(X + g, 0)
        u = _ln(_c1(p, "gconv_update_spa_graph_" + lv, g), p["text_objseg/gconv_update_ln_spa_graph_%s/gamma" % lv],
                p["text_objseg/gconv_update_ln_spa_graph_%s/beta" % lv])
        spa = _l2n(np.maximum(u, 0).reshape(B, h, w, C), 3)
        taps["spa_graph_" + lv] = spa
        allf = np.concatenate([vls, spa, np.broadcast_to(vl, (B, h, w, R)), sp], 3)
        fus[lv] = np.maximum(_c1(p, "fusion_" + lv, allf), 0)
        taps["fusion_" + lv] = fus[lv]
        sc = conv_same(fus[lv], p["text_objseg/score_%s/DW" % lv]) + p["text_objseg/score_%s/biases" % lv]
        taps["up_" + lv] = resize_bilinear(sc, H, W)
    nec = lang(wp.sum(3) - wp[..., 3])

    def exch(feat, f1, lv):                                                        # v5:299-347
        key = _c1(p, "spa_graph_key_%sgv_f1" % lv, feat).reshape(B, N, M)
        q = _c1(p, "lang_query_%sgv_f1" % lv, nec).reshape(B, 1, M)
        attn = _softmax(key @ q.transpose(0, 2, 1) / M ** 0.5, 1)
        pooled = (attn.transpose(0, 2, 1) @ feat.reshape(B, N, M)).reshape(B, 1, 1, M)
        gv = _l2n(np.tanh(_c1(p, "gv_lang_%sgv_f1" % lv, np.concatenate([pooled, nec], 3))), None)
        s1 = np.maximum(_c1(p, "trans_feat_%s_f1" % lv, f1), 0) * _sig(_c1(p, "lang_feat_%s_f1" % lv, gv))
        return _l2n(feat + s1, 3)

    f4, f5 = fus["c4"], fus["c5"]
    e4, e5 = exch(f4, f5, "c4"), exch(f5, f4, "c5")
    e42, e52 = exch(e4, e5, "c4_2"), exch(e5, e4, "c5_2")
    taps["exg_c5_2"] = e52
    pre = "text_objseg/rnn/conv_lstm_cell/"
    Wk = p[pre + "kernel"][0, 0]
    c = np.zeros((B, h, w, M)); hh = np.zeros((B, h, w, M))
    lnn = lambda x, i: _ln(x, p[pre + ("LayerNorm" if i == 0 else "LayerNorm_%d" % i) + "/gamma"],
                           p[pre + ("LayerNorm" if i == 0 else "LayerNorm_%d" % i) + "/beta"])
    for x in (e42, e52):
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
    # atrous_spatial_pyramid_pooling v5:208-251
    train, rates = dims["train"], dims["rates"]
    br = [_conv_bn_relu(p, bn, "aspp/conv_1x1", hh, train)]
    for k, r in enumerate(rates):
        br.append(_conv_bn_relu(p, bn, "aspp/conv_3x3_%d" % (k + 1), hh, train, r))
    img = _conv_bn_relu(p, bn, "aspp/image_level_features/conv_1x1", hh.mean(axis=(1, 2), keepdims=True), train)
    br.append(np.broadcast_to(img, (B, h, w, img.shape[-1])))
    enc = _conv_bn_relu(p, bn, "aspp/conv_1x1_concat", np.concatenate(br, 3), train)
    taps["aspp"] = enc
    # decoder v5:190-206
    low = _conv_bn_relu(p, bn, "decoder/low_level_features/conv_1x1", c2, train)
    net = np.concatenate([resize_bilinear(enc, low.shape[1], low.shape[2]), low], 3)
    net = _conv_bn_relu(p, bn, "decoder/upsampling_logits/conv_3x3_1", net, train)
    net = _conv_bn_relu(p, bn, "decoder/upsampling_logits/conv_3x3_2", net, train)
    pred = net @ p["text_objseg/decoder/upsampling_logits/conv_1x1/weights"][0, 0] + p["text_objseg/decoder/upsampling_logits/conv_1x1/biases"]
    taps["pred"] = pred
    taps["up"] = resize_bilinear(pred, H, W)
    taps["sigm"] = _sig(taps["up"])
    return taps
