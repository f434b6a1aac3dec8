"""Stage operators of the CMPC head: each is a torch.autograd.Function whose forward AND backward
are chains of HIP kernels launched through the C ABI (include/cmpc.h).  PyTorch does the autograd
bookkeeping (graph, fan-out accumulation) and owns the device memory; it computes nothing on
this path except trivial reshapes of [B, <=64] language-side vectors.

Parameter gradients are NOT returned through autograd: the kernels accumulate them straight into
the flat fp32 gradient buffer of ParamStore (one RCCL all-reduce, one fused Adam launch).

Reference citations are file:line under /root/reference.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, DT_BF16, DT_F16, DT_F32, ConvLstmDln, ConvLstmLn,
                   GemmNtArgs, GemmTnArgs)
from .params import HeadCfg, ParamStore

F32 = DT_F32


def tdt(dt: int):
    return {DT_F32: torch.float32, DT_BF16: torch.bfloat16, DT_F16: torch.float16}[dt]


def esz(dt: int) -> int:
    return 4 if dt == DT_F32 else 2


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _st():
    # torch.cuda.current_stream() costs ~8 us of host time per call (x ~850 launches per step)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def empty(shape, dt, dev):
    return torch.empty(shape, dtype=tdt(dt), device=dev)


def zeros(shape, dt, dev):
    return torch.zeros(shape, dtype=tdt(dt), device=dev)


# ---------------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------------
def gemm_nt(dt, segs: Sequence[Tuple], C, ldc, M, N, n_valid=None, batch=1, sC=0, c_f32=False, bias=None,
            sbias=None, ld_sbias=0, pbias=None, ld_pbias=0, rows_per_sample=0, act=ACT_NONE, alpha=1.0,
            accumulate=False):
    """segs: (A, lda, Bt, ldb, K[, sA, sB]) with A / Bt tensors or raw device addresses."""
    a = GemmNtArgs()
    a.dtype, a.nseg = dt, len(segs)
    for i, s in enumerate(segs):
        A, lda, Bt, ldb, K = s[:5]
        a.A[i], a.lda[i], a.Bt[i], a.ldb[i], a.K[i] = _p(A), lda, _p(Bt), ldb, K
        a.sA[i] = s[5] if len(s) > 5 else 0
        a.sB[i] = s[6] if len(s) > 6 else 0
    a.C, a.ldc, a.sC, a.c_f32 = _p(C), ldc, sC, int(c_f32)
    a.M, a.N, a.n_valid, a.batch = M, N, (N if n_valid is None else n_valid), batch
    a.bias = _p(bias)
    a.sbias, a.ld_sbias = _p(sbias), ld_sbias
    a.pbias, a.ld_pbias = _p(pbias), ld_pbias
    a.rows_per_sample = rows_per_sample
    a.act, a.alpha, a.accumulate = act, alpha, int(accumulate)
    _lib.call("cmpc_gemm_nt", ctypes.byref(a), _st())


def on_wgrad_stream(cx, tensors, fn):
    """Run fn() (a weight-gradient launch: it only feeds the optimizer) on the weight-gradient stream so
    that it overlaps the dX chain of the backward pass.  `tensors` are its device inputs: they are
    recorded on that stream so the caching allocator does not recycle them early."""
    ws = getattr(cx, "wg", None) if cx is not None else None
    if ws is None:
        return fn()
    ws.wait_stream(torch.cuda.current_stream())
    for t in tensors:
        if torch.is_tensor(t):
            t.record_stream(ws)
    with torch.cuda.stream(ws):
        fn()


def gemm_tn(dt, A, lda, Ka, D, ldd, Nd, out, ldo, R, Kv, Nv, offs=((0, 0, 0),), nb2=1, a_bs=0, d_bs=0, o_bs=0,
            alpha=1.0, rsplit=None, wg=None):
    """out[k, n] += alpha * sum_r A[r, k] D[r, n]; offs: (a_off, d_off, o_off) per inner batch (elements).
    wg=cx: weight gradient -> deferred to cx.flush_wgrad() (one grouped launch at the end of the backward
    pass), or launched on cx's weight-gradient stream when that is enabled."""
    if wg is not None and getattr(wg, "defer", False):
        wg = wg.deferred
    elif wg is not None and getattr(wg, "wg", None) is not None:
        return on_wgrad_stream(wg, (A, D), lambda: gemm_tn(dt, A, lda, Ka, D, ldd, Nd, out, ldo, R, Kv, Nv, offs, nb2, a_bs, d_bs,
                                                             o_bs, alpha, rsplit))
    a = GemmTnArgs()
    a.dtype = dt
    a.A, a.lda, a.Ka = _p(A), lda, Ka
    a.D, a.ldd, a.Nd = _p(D), ldd, Nd
    a.out, a.ldo = _p(out), ldo
    a.R, a.Kv, a.Nv = R, Kv, Nv
    a.nb = len(offs)
    for i, (ao, do, oo) in enumerate(offs):
        a.a_off[i], a.d_off[i], a.o_off[i] = ao, do, oo
    a.nb2, a.a_bs, a.d_bs, a.o_bs = nb2, a_bs, d_bs, o_bs
    if rsplit is None:
        tiles = ((Kv + 127) // 128) * ((Nv + 127) // 128) * len(offs) * nb2
        br = 64 if dt != DT_F32 else 32
        rsplit = max(1, min((R + 4 * br - 1) // (4 * br), (512 + tiles - 1) // tiles))
    a.rsplit, a.alpha = rsplit, alpha
    a.zeros = _zero_page(A.device if torch.is_tensor(A) else torch.device('cuda', torch.cuda.current_device()))
    if isinstance(wg, list):
        wg.append((a, A, D))         # operands stay referenced (hence allocated and unmodified) until the flush
        return
    _lib.call("cmpc_gemm_tn", ctypes.byref(a), _st())


_ZERO_PAGES = {}


def _zero_page(dev):
    k = str(dev)
    if k not in _ZERO_PAGES:
        _ZERO_PAGES[k] = torch.zeros(256, dtype=torch.uint8, device=dev)
    return _ZERO_PAGES[k].data_ptr()


def cast(src, src_dt, dst, dst_dt, n):
    _lib.call("cmpc_cast", src_dt, _p(src), dst_dt, _p(dst), n, _st())


def colsum(dt, dy, R, stride, ld, C, db=None, y=None, dpre=None, act=ACT_NONE, dsb=None, ld_dsb=0, rows_per_sample=0):
    """dpre = dy * act'(y) (optional), db[c] += column sums, dsb[b][c] += per-sample sums.
    Columns are processed in windows of <= 2048 (the kernel keeps per-column registers)."""
    e = esz(dt)
    for c0 in range(0, ld, 2048):
        w = min(2048, ld - c0)
        cv = max(0, min(C - c0, w))
        if cv == 0:
            continue
        _lib.call("cmpc_act_bwd", dt, _p(dy) + c0 * e, (_p(y) + c0 * e) if y is not None else None,
                  (_p(dpre) + c0 * e) if dpre is not None else None, act, R, stride, w, cv,
                  (_p(db) + 4 * c0) if db is not None else None, (_p(dsb) + 4 * c0) if dsb is not None else None,
                  ld_dsb, rows_per_sample, _st())


def _detach_tree(x):
    if torch.is_tensor(x):
        return x.detach()
    if isinstance(x, (tuple, list)):
        return type(x)(_detach_tree(t) for t in x)
    return x


def _nocycle(fwd):
    """forward decorator of the stage operators.  They keep what backward needs in `ctx.saved`; an OUTPUT tensor kept
    there as the same Python object forms the cycle node -> ctx -> tensor -> grad_fn -> node that neither reference
    counting nor the cyclic GC frees (it crosses into C++): 1.1 GiB leaked per train step.  Saving detached aliases
    (same storage, no grad_fn) keeps the data alive exactly as long as the graph and breaks the cycle."""
    def wrapper(ctx, *args):
        out = fwd(ctx, *args)
        if hasattr(ctx, "saved"):
            ctx.saved = _detach_tree(ctx.saved)
        return out
    wrapper.__doc__ = fwd.__doc__
    return staticmethod(wrapper)


def spatial_grid_padded(h: int, w: int) -> torch.Tensor:
    """generate_spatial_batch (util/processing_tools.py:5-17) for one image as an [h*w, 64] fp32 matrix: the 8 coordinate
    channels [xmin, ymin, xmax, ymax, xctr, yctr, 1/w, 1/h] (computed in float64, stored float32 like the reference's
    np.float32 array), zero padded to the 64-wide K tile of the GEMMs that consume it."""
    sp = torch.zeros(h * w, 64, dtype=torch.float32)
    ys = torch.arange(h, dtype=torch.float64).view(h, 1).expand(h, w)
    xs = torch.arange(w, dtype=torch.float64).view(1, w).expand(h, w)
    xmin, xmax = xs / w * 2 - 1, (xs + 1) / w * 2 - 1
    ymin, ymax = ys / h * 2 - 1, (ys + 1) / h * 2 - 1
    grid = torch.stack([xmin, ymin, xmax, ymax, (xmin + xmax) / 2, (ymin + ymax) / 2,
                        torch.full_like(xs, 1 / w), torch.full_like(xs, 1 / h)], -1)
    sp[:, :8] = grid.reshape(h * w, 8).float()
    return sp


class Ctx:
    """Per-model handles shared by every stage operator."""

    def __init__(self, cfg: HeadCfg, store: ParamStore, vis_dt: int):
        self.cfg, self.ps, self.dt = cfg, store, vis_dt
        self.dev = store.device
        B, N = cfg.batch_size, cfg.N
        sp = spatial_grid_padded(cfg.vf_h, cfg.vf_w)
        self.spatial = sp.repeat(B, 1).to(self.dev).to(tdt(vis_dt)).contiguous()
        self.anchor = torch.zeros((), device=self.dev, requires_grad=True)
        self.wg = None              # weight-gradient stream (set by LSTM_model.set_streams)
        # weight-gradient products have no consumer before the optimizer: they are collected during the
        # backward pass and issued as a few grouped launches by flush_wgrad() (CMPC_WGRAD_DEFER=0: in place)
        self.defer = os.environ.get("CMPC_WGRAD_DEFER", "1") != "0"
        self.deferred = []
        self.flush_stream, self.lanes = None, ()      # set by LSTM_model.set_streams

    def flush_wgrad(self, early=False):
        """Launch the deferred weight-gradient products.  early=False: on the current stream (which must be
        ordered after every stream that produced their operands).  early=True (called when the text encoder's
        backward starts -- the last stage of the backward pass, a serial chain of ~100 small kernels): on the
        flush stream, after everything queued so far on the current stream and on the lanes, so that the
        products run beside that chain (the grouped kernel is persistent, so it leaves the dispatcher free)."""
        st = self.flush_stream if early else None
        if early and st is None:
            return
        items, self.deferred = self.deferred, []
        if not items:
            return
        arr = (GemmTnArgs * len(items))()
        for i, (a, _A, _D) in enumerate(items):
            ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(a), ctypes.sizeof(GemmTnArgs))
        cur = torch.cuda.current_stream(self.dev)
        if st is not None:
            st.wait_stream(cur)
            for ln in self.lanes:
                if ln is not st:
                    st.wait_stream(ln)
        run = st if st is not None else cur
        with torch.cuda.stream(run):
            _lib.call("cmpc_gemm_tn_grouped", arr, len(items), _st())
        for _a, A, D in items:        # the caching allocator must not recycle the operands before the launch has run
            for t in (A, D):
                if torch.is_tensor(t):
                    t.record_stream(run)

    def op(self, key):
        return self.ps.ops[key]

    def opp(self, key, row=0, col=0):
        return self.ps.ops[key].ptr(self.ps.arena, row, col)


# ---------------------------------------------------------------------------------------------
# S0: text encoder -- lstm(), CMPC_model.py:144-164
# ---------------------------------------------------------------------------------------------
class TextEncoder(torch.autograd.Function):
    @_nocycle
    def forward(ctx, anchor, words, seq_len, cx: Ctx):
        cfg, ps, dev = cx.cfg, cx.ps, cx.dev
        B, T, R, G = cfg.batch_size, cfg.num_steps, cfg.rnn_size, cfg.glove_dim
        Cp, Gp = cfg.Cp, cfg.Gp
        words_tb = words.view(B, T).t().contiguous().view(-1).to(torch.int32)
        emb = empty((T * B, Gp), F32, dev)
        _lib.call("cmpc_embed_gather", ps.pptr("Variable"), _p(words_tb), _p(emb), T * B, G, Gp, cfg.vocab_size, _st())
        xg = empty((T * B, 4 * Cp), F32, dev)
        ldk = Gp + Cp
        gemm_nt(F32, [(emb, Gp, cx.opp("lstm.t"), ldk, Gp)], xg, 4 * Cp, T * B, 4 * Cp, bias=cx.opp("lstm.b"))
        gates = empty((T, B, 4 * Cp), F32, dev)
        h_all = zeros((T + 1, B, Cp), F32, dev)
        c_all = zeros((T + 1, B, Cp), F32, dev)
        outs = empty((B * T, Cp), F32, dev)
        for t in range(T):
            gemm_nt(F32, [(h_all[t], Cp, cx.opp("lstm.t", 0, Gp), ldk, Cp)], gates[t], 4 * Cp, B, 4 * Cp,
                    sbias=xg[t * B:], ld_sbias=4 * Cp, rows_per_sample=1)
            _lib.call("cmpc_lstm_cell_fwd", _p(gates[t]), _p(c_all[t]), _p(h_all[t]), _p(seq_len), t,
                      _p(c_all[t + 1]), _p(h_all[t + 1]), _p(outs) + 4 * t * Cp, T * Cp, B, Cp, R, _st())
        wf = empty((B * T, Cp), F32, dev)
        rstd = empty((B * T,), F32, dev)
        mask = empty((B * T,), F32, dev)
        _lib.call("cmpc_l2norm_rows_fwd", F32, _p(outs), _p(wf), _p(rstd), _p(mask), B * T, Cp, R, _st())
        ctx.cx, ctx.saved = cx, (words_tb, seq_len, emb, gates, h_all, c_all, wf, rstd)
        ctx.mark_non_differentiable(mask)
        return wf, mask

    @staticmethod
    def backward(ctx, dwf, _dmask):
        cx = ctx.cx
        cfg, ps, dev = cx.cfg, cx.ps, cx.dev
        words_tb, seq_len, emb, gates, h_all, c_all, wf, rstd = ctx.saved
        B, T, R, G = cfg.batch_size, cfg.num_steps, cfg.rnn_size, cfg.glove_dim
        Cp, Gp = cfg.Cp, cfg.Gp
        cx.flush_wgrad(early=True)       # every other stage's weight gradients run beside this serial chain
        dwf = dwf.contiguous()
        douts = empty((B * T, Cp), F32, dev)
        _lib.call("cmpc_l2norm_rows_bwd", F32, _p(dwf), _p(wf), _p(rstd), _p(douts), B * T, Cp, R, 0, _st())
        dh = zeros((B, Cp), F32, dev)
        dc = zeros((B, Cp), F32, dev)
        dg = empty((T, B, 4 * Cp), F32, dev)
        if B <= 8 and os.environ.get("CMPC_LSTM_FUSED", "1") != "0":
            # one launch per step: dh += dg[t] . W_h^T fused with the cell backward of step t-1 (the product of step 0
            # would only feed the unused gradient of the initial state)
            _lib.call("cmpc_lstm_cell_bwd", _p(gates[T - 1]), _p(c_all[T - 1]), _p(c_all[T]), _p(seq_len), T - 1,
                      _p(douts) + 4 * (T - 1) * Cp, T * Cp, _p(dh), _p(dc), _p(dg[T - 1]), B, Cp, R, _st())
            wn = cx.opp("lstm.n", Gp, 0)
            for t in range(T - 1, 0, -1):
                _lib.call("cmpc_lstm_bwd_step", _p(dg[t]), wn, 4 * Cp, _p(gates[t - 1]), _p(c_all[t - 1]), _p(c_all[t]), _p(seq_len), t - 1,
                          _p(douts) + 4 * (t - 1) * Cp, T * Cp, _p(dh), _p(dc), _p(dg[t - 1]), B, Cp, R, _st())
        else:
            for t in reversed(range(T)):
                _lib.call("cmpc_lstm_cell_bwd", _p(gates[t]), _p(c_all[t]), _p(c_all[t + 1]), _p(seq_len), t,
                          _p(douts) + 4 * t * Cp, T * Cp, _p(dh), _p(dc), _p(dg[t]), B, Cp, R, _st())
                gemm_nt(F32, [(dg[t], 4 * Cp, cx.opp("lstm.n", Gp, 0), 4 * Cp, 4 * Cp)], dh, Cp, B, Cp, n_valid=R,
                        accumulate=True)
        gk = ps.gptr("rnn/lstm_cell/kernel")
        gate_offs = lambda row0: tuple((0, g * Cp, row0 * 4 * R + g * R) for g in range(4))
        gemm_tn(F32, emb, Gp, Gp, dg, 4 * Cp, Cp, gk, 4 * R, T * B, G, R, offs=gate_offs(0), wg=cx)
        gemm_tn(F32, h_all, Cp, Cp, dg, 4 * Cp, Cp, gk, 4 * R, T * B, R, R, offs=gate_offs(G), wg=cx)
        gb = ps.gptr("rnn/lstm_cell/bias")
        for g in range(4):
            colsum(F32, _p(dg) + 4 * g * Cp, T * B, 4 * Cp, Cp, R, db=gb + 4 * g * R)
        demb = empty((T * B, Gp), F32, dev)
        gemm_nt(F32, [(dg, 4 * Cp, cx.opp("lstm.n"), 4 * Cp, 4 * Cp)], demb, Gp, T * B, Gp, n_valid=G)
        _lib.call("cmpc_embed_scatter", _p(demb), Gp, _p(words_tb), ps.gptr("Variable"), T * B, G, cfg.vocab_size, _st())
        return None, None, None, None


# ---------------------------------------------------------------------------------------------
# S1: build_lang_parser, CMPC_model.py:347-357
# ---------------------------------------------------------------------------------------------
class LangParser(torch.autograd.Function):
    @_nocycle
    def forward(ctx, wf, mask, cx: Ctx):
        cfg, ps, dev = cx.cfg, cx.ps, cx.dev
        BT, R, P = cfg.batch_size * cfg.num_steps, cfg.rnn_size, cfg.parse_dim
        Cp, Pp = cfg.Cp, cfg.Pp
        wf = wf.contiguous()
        h1 = empty((BT, Pp), F32, dev)
        gemm_nt(F32, [(wf, Cp, cx.opp("parse1.t"), Cp, Cp)], h1, Pp, BT, Pp, n_valid=P,
                bias=ps.pptr("words_parse_1/biases"), act=ACT_RELU)
        lg = empty((BT, 64), F32, dev)
        gemm_nt(F32, [(h1, Pp, cx.opp("parse2.t"), Pp, Pp)], lg, 64, BT, 64, n_valid=4, bias=ps.pptr("words_parse_2/biases"))
        parse = empty((BT, 4), F32, dev)
        _lib.call("cmpc_parse_softmax_fwd", _p(lg), 64, _p(mask), _p(parse), BT, _st())
        ctx.cx, ctx.saved = cx, (wf, mask, h1, parse)
        return parse

    @staticmethod
    def backward(ctx, dparse):
        cx = ctx.cx
        cfg, ps, dev = cx.cfg, cx.ps, cx.dev
        wf, mask, h1, parse = ctx.saved
        BT, R, P = cfg.batch_size * cfg.num_steps, cfg.rnn_size, cfg.parse_dim
        Cp, Pp = cfg.Cp, cfg.Pp
        dparse = dparse.contiguous()
        dlg = empty((BT, 64), F32, dev)
        _lib.call("cmpc_parse_softmax_bwd", _p(dparse), _p(parse), _p(mask), _p(dlg), 64, BT, _st())
        colsum(F32, dlg, BT, 64, 64, 4, db=ps.gptr("words_parse_2/biases"))
        gemm_tn(F32, h1, Pp, Pp, dlg, 64, 64, ps.gptr("words_parse_2/DW"), 4, BT, P, 4, wg=cx)
        dh1 = empty((BT, Pp), F32, dev)
        gemm_nt(F32, [(dlg, 64, cx.opp("parse2.n"), 64, 64)], dh1, Pp, BT, Pp, n_valid=P)
        colsum(F32, dh1, BT, Pp, Pp, P, db=ps.gptr("words_parse_1/biases"), y=h1, dpre=dh1, act=ACT_RELU)
        gemm_tn(F32, wf, Cp, Cp, dh1, Pp, Pp, ps.gptr("words_parse_1/DW"), P, BT, R, P, wg=cx)
        dwf = empty((BT, Cp), F32, dev)
        gemm_nt(F32, [(dh1, Pp, cx.opp("parse1.n"), Pp, Pp)], dwf, Cp, BT, Cp, n_valid=R)
        return dwf, None, None


# ---------------------------------------------------------------------------------------------
# S2: valid_lang / nec_lang, CMPC_model.py:166-192
# ---------------------------------------------------------------------------------------------
class LangPool(torch.autograd.Function):
    @_nocycle
    def forward(ctx, parse, wf, ncls: int, cx: Ctx):
        cfg, dev = cx.cfg, cx.dev
        B, T, R, Cp = cfg.batch_size, cfg.num_steps, cfg.rnn_size, cfg.Cp
        parse, wf = parse.contiguous(), wf.contiguous()
        v = empty((B, Cp), F32, dev)
        rstd = empty((B,), F32, dev)
        _lib.call("cmpc_lang_pool_fwd", _p(parse), _p(wf), _p(v), _p(rstd), B, T, Cp, R, ncls, _st())
        ctx.cx, ctx.ncls, ctx.saved = cx, ncls, (parse, wf, v, rstd)
        return v

    @staticmethod
    def backward(ctx, dv):
        cx = ctx.cx
        cfg, dev = cx.cfg, cx.dev
        parse, wf, v, rstd = ctx.saved
        B, T, R, Cp = cfg.batch_size, cfg.num_steps, cfg.rnn_size, cfg.Cp
        dparse = zeros((B * T, 4), F32, dev)
        dwf = zeros((B * T, Cp), F32, dev)
        _lib.call("cmpc_lang_pool_bwd", _p(dv.contiguous()), _p(v), _p(rstd), _p(parse), _p(wf), _p(dparse), _p(dwf),
                  B, T, Cp, R, ctx.ncls, _st())
        return dparse, dwf, None, None


# ---------------------------------------------------------------------------------------------
# S3: lateral 1x1 conv + l2_normalize, CMPC_model.py:108-113
# ---------------------------------------------------------------------------------------------
class Lateral(torch.autograd.Function):
    @_nocycle
    def forward(ctx, anchor, feat, lv: str, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        R, C, Cp = cfg.batch_size * cfg.N, cfg.v_emb_dim, cfg.Cp
        cin = feat.shape[-1]
        feat = feat.reshape(R, cin).contiguous()
        X0 = empty((R, Cp), dt, dev)
        gemm_nt(dt, [(feat, cin, cx.opp(f"lat_{lv}.t"), cin, cin)], X0, Cp, R, Cp, n_valid=C, bias=ps.pptr(f"{lv}_lateral/biases"))
        rstd = empty((R,), F32, dev)
        _lib.call("cmpc_l2norm_rows_fwd", dt, _p(X0), _p(X0), _p(rstd), None, R, Cp, C, _st())
        ctx.cx, ctx.lv, ctx.saved = cx, lv, (feat, X0, rstd)
        return X0

    @staticmethod
    def backward(ctx, dX0):
        cx, lv = ctx.cx, ctx.lv
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        feat, X0, rstd = ctx.saved
        R, C, Cp = cfg.batch_size * cfg.N, cfg.v_emb_dim, cfg.Cp
        cin = feat.shape[-1]
        dV = empty((R, Cp), dt, dev)
        _lib.call("cmpc_l2norm_rows_bwd", dt, _p(dX0.contiguous()), _p(X0), _p(rstd), _p(dV), R, Cp, C, 0, _st())
        colsum(dt, dV, R, Cp, Cp, C, db=ps.gptr(f"{lv}_lateral/biases"))
        gemm_tn(dt, feat, cin, cin, dV, Cp, Cp, ps.gptr(f"{lv}_lateral/DW"), C, R, cin, C, wg=cx)
        return None, None, None, None


# ---------------------------------------------------------------------------------------------
# S4: mutan_fusion, CMPC_model.py:295-328
# ---------------------------------------------------------------------------------------------
class Mutan(torch.autograd.Function):
    @_nocycle
    def forward(ctx, X0, vl, lv: str, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, N, C, Cp = cfg.batch_size, cfg.N, cfg.v_emb_dim, cfg.Cp
        R = B * N
        X0, vl = X0.contiguous(), vl.contiguous()
        g = empty((B, 5 * Cp), F32, dev)
        gemm_nt(F32, [(vl, Cp, cx.opp(f"mlang_{lv}.t"), Cp, Cp)], g, 5 * Cp, B, 5 * Cp, bias=cx.opp(f"mlang_{lv}.b"), act=ACT_TANH)
        P = empty((R, 5 * Cp), dt, dev)
        ldk = Cp + 64
        gemm_nt(dt, [(X0, Cp, cx.opp(f"mutan_{lv}.t"), ldk, Cp), (cx.spatial, 64, cx.opp(f"mutan_{lv}.t", 0, Cp), ldk, 64)],
                P, 5 * Cp, R, 5 * Cp, bias=cx.opp(f"mutan_{lv}.b"))
        X1 = empty((R, Cp), dt, dev)
        rstd = empty((R,), F32, dev)
        _lib.call("cmpc_mutan_fwd", dt, _p(P), _p(g), _p(X1), _p(rstd), B, N, Cp, C, _st())
        ctx.cx, ctx.lv, ctx.saved = cx, lv, (X0, vl, g, P, X1, rstd)
        return X1

    @staticmethod
    def backward(ctx, dX1):
        cx, lv = ctx.cx, ctx.lv
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        X0, vl, g, Th, X1, rstd = ctx.saved
        B, N, C, Cp, Rr = cfg.batch_size, cfg.N, cfg.v_emb_dim, cfg.Cp, cfg.rnn_size
        R = B * N
        dg = zeros((B, 5 * Cp), F32, dev)
        _lib.call("cmpc_mutan_bwd", dt, _p(Th), _p(g), _p(X1), _p(rstd), _p(dX1.contiguous()), _p(dg), B, N, Cp, C, _st())
        dP = Th     # overwritten in place
        e = esz(dt)
        base_w = ps.poff(f"vis_trans_{lv}_head1/DW")
        offs_v, offs_s = [], []
        for h in range(5):
            rel = ps.poff(f"vis_trans_{lv}_head{h + 1}/DW") - base_w
            offs_v.append((0, h * Cp, rel))
            offs_s.append((0, h * Cp, rel + C * C))
            colsum(dt, _p(dP) + h * Cp * e, R, 5 * Cp, Cp, C, db=ps.gptr(f"vis_trans_{lv}_head{h + 1}/biases"))
        gw = ps.gptr(f"vis_trans_{lv}_head1/DW")
        gemm_tn(dt, X0, Cp, Cp, dP, 5 * Cp, Cp, gw, C, R, C, C, offs=offs_v, wg=cx)
        gemm_tn(dt, cx.spatial, 64, 64, dP, 5 * Cp, Cp, gw, C, R, 8, C, offs=offs_s, wg=cx)
        dX0 = empty((R, Cp), dt, dev)
        gemm_nt(dt, [(dP, 5 * Cp, cx.opp(f"mutan_{lv}.n"), 5 * Cp, 5 * Cp)], dX0, Cp, R, Cp, n_valid=C)
        # language gates
        base_l = ps.poff(f"lang_trans_{lv}_head1/DW")
        offs_l = []
        for h in range(5):
            colsum(F32, _p(dg) + 4 * h * Cp, B, 5 * Cp, Cp, C, db=ps.gptr(f"lang_trans_{lv}_head{h + 1}/biases"),
                   y=_p(g) + 4 * h * Cp, dpre=_p(dg) + 4 * h * Cp, act=ACT_TANH)
            offs_l.append((0, h * Cp, ps.poff(f"lang_trans_{lv}_head{h + 1}/DW") - base_l))
        gemm_tn(F32, vl, Cp, Cp, dg, 5 * Cp, Cp, ps.gptr(f"lang_trans_{lv}_head1/DW"), C, B, Rr, C, offs=offs_l, wg=cx)
        dvl = empty((B, Cp), F32, dev)
        gemm_nt(F32, [(dg, 5 * Cp, cx.opp(f"mlang_{lv}.n"), 5 * Cp, 5 * Cp)], dvl, Cp, B, Cp, n_valid=Rr)
        return dX0, dvl, None, None


# ---------------------------------------------------------------------------------------------
# S5: build_spa_graph + graph_conv, CMPC_model.py:359-410.  The N x N adjacency gw_w.gw_v^T is
# never formed: adj.X = gw_w.(gw_v^T.X); spa_graph_trans2 is folded into the word side
# (X.W2 + b).Wd^T = X.(W2.Wd^T) + b.Wd^T.
# ---------------------------------------------------------------------------------------------
class SpaGraph(torch.autograd.Function):
    @_nocycle
    def forward(ctx, X1, wf, parse, mask, lv: str, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, N, T, C, Cp, Tp = cfg.batch_size, cfg.N, cfg.num_steps, cfg.v_emb_dim, cfg.Cp, cfg.Tp
        R = B * N
        X1, wf = X1.contiguous(), wf.contiguous()
        scale = 1.0 / math.sqrt(C)
        Wd = zeros((B * Tp, Cp), F32, dev)
        gemm_nt(F32, [(wf, Cp, cx.opp(f"wtrans_{lv}.t"), Cp, Cp, T * Cp, 0)], Wd, Cp, T, Cp, n_valid=C, batch=B, sC=Tp * Cp,
                bias=ps.pptr(f"words_trans_{lv}/biases"))
        PTf = empty((B * Tp, Cp), F32, dev)
        gemm_nt(F32, [(Wd, Cp, cx.opp(f"t2_{lv}.n"), Cp, Cp)], PTf, Cp, B * Tp, Cp, n_valid=C)
        PT = empty((B * Tp, Cp), dt, dev)
        cast(PTf, F32, PT, dt, PT.numel())
        PTtf = empty((Cp, B * Tp), F32, dev)
        gemm_nt(F32, [(cx.opp(f"t2_{lv}.n"), Cp, Wd, Cp, Cp)], PTtf, B * Tp, Cp, B * Tp)
        PTt = empty((Cp, B * Tp), dt, dev)
        cast(PTtf, F32, PTt, dt, PTt.numel())
        k0s = empty((B * Tp,), F32, dev)
        _lib.call("cmpc_rowdot1", F32, _p(Wd), ps.pptr(f"spa_graph_trans2_{lv}/biases"), 0, _p(k0s), 1, B * Tp, Cp, C, scale, _st())
        A0 = empty((B, N, Tp), F32, dev)
        gemm_nt(dt, [(X1, Cp, PT, Cp, Cp, N * Cp, Tp * Cp)], A0, Tp, N, Tp, batch=B, sC=N * Tp, c_f32=True, alpha=scale,
                sbias=k0s, ld_sbias=Tp, rows_per_sample=N)
        pr = parse.view(B * T, 4)[:, 2].contiguous()
        gw_w = empty((B, N, Tp), F32, dev)
        gw_v = empty((B, N, Tp), F32, dev)
        gw_w_t = empty((B, N, Tp), dt, dev)
        gw_v_t = empty((B, N, Tp), dt, dev)
        gsc = empty((B * ((N + 63) // 64) * 128,), F32, dev)
        _lib.call("cmpc_graph_softmax_fwd", dt, _p(A0), _p(pr), _p(mask), _p(gw_w), _p(gw_v), _p(gw_w_t), _p(gw_v_t), _p(gsc), B, N, T, Tp, _st())
        Ztf = zeros((B, Cp, Tp), F32, dev)        # Z^T = X1^T . gw_v   [C, T] per sample
        gemm_tn(dt, X1, Cp, Cp, gw_v_t, Tp, Tp, Ztf, Tp, N, C, T, nb2=B, a_bs=N * Cp, d_bs=N * Tp, o_bs=Cp * Tp)
        Zt = empty((B, Cp, Tp), dt, dev)
        cast(Ztf, F32, Zt, dt, Zt.numel())
        Y = empty((R, Cp), dt, dev)
        gemm_nt(dt, [(gw_w_t, Tp, Zt, Tp, Tp, N * Tp, Cp * Tp)], Y, Cp, N, Cp, n_valid=C, batch=B, sC=N * Cp)
        sums1 = torch.empty((B, 2), dtype=torch.float64, device=dev)
        _lib.call("cmpc_sample_stats", dt, _p(Y), _p(sums1), B, N, Cp, C, _st())
        G = empty((R, Cp), dt, dev)
        ln1, ln2 = f"gconv_feat_ln_spa_graph_{lv}", f"gconv_update_ln_spa_graph_{lv}"
        _lib.call("cmpc_gconv_pre_fwd", dt, _p(Y), _p(X1), _p(sums1), ps.pptr(ln1 + "/gamma"), ps.pptr(ln1 + "/beta"), _p(G), B, N, Cp, C, _st())
        U = empty((R, Cp), dt, dev)
        gemm_nt(dt, [(G, Cp, cx.opp(f"gupd_{lv}.t"), Cp, Cp)], U, Cp, R, Cp, n_valid=C, bias=ps.pptr(f"gconv_update_spa_graph_{lv}/biases"))
        sums2 = torch.empty((B, 2), dtype=torch.float64, device=dev)
        _lib.call("cmpc_sample_stats", dt, _p(U), _p(sums2), B, N, Cp, C, _st())
        X2 = empty((R, Cp), dt, dev)
        rrow = empty((R,), F32, dev)
        _lib.call("cmpc_gconv_post_fwd", dt, _p(U), _p(sums2), ps.pptr(ln2 + "/gamma"), ps.pptr(ln2 + "/beta"), _p(X2), _p(rrow), B, N, Cp, C, _st())
        ctx.cx, ctx.lv = cx, lv
        ctx.saved = (X1, wf, Wd, PT, PTt, A0, pr, mask, gw_w, gw_v, gw_w_t, gw_v_t, Y, sums1, G, U, sums2, X2, rrow)
        ctx.mark_non_differentiable(gw_w, gw_v)
        return X2, gw_w, gw_v

    @staticmethod
    def backward(ctx, dX2, _d1, _d2):
        cx, lv = ctx.cx, ctx.lv
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        (X1, wf, Wd, PT, PTt, A0, pr, mask, gw_w, gw_v, gw_w_t, gw_v_t, Y, sums1, G, U, sums2, X2, rrow) = ctx.saved
        B, N, T, C, Cp, Tp = cfg.batch_size, cfg.N, cfg.num_steps, cfg.v_emb_dim, cfg.Cp, cfg.Tp
        R = B * N
        scale = 1.0 / math.sqrt(C)
        ln1, ln2 = f"gconv_feat_ln_spa_graph_{lv}", f"gconv_update_ln_spa_graph_{lv}"
        bs = torch.empty((B, 2), dtype=torch.float64, device=dev)
        dU = empty((R, Cp), dt, dev)
        _lib.call("cmpc_gconv_post_bwd", dt, _p(dX2.contiguous()), _p(X2), _p(rrow), _p(U), _p(sums2), ps.pptr(ln2 + "/gamma"),
                  _p(dU), ps.gptr(ln2 + "/gamma"), ps.gptr(ln2 + "/beta"), _p(bs), B, N, Cp, C, _st())
        colsum(dt, dU, R, Cp, Cp, C, db=ps.gptr(f"gconv_update_spa_graph_{lv}/biases"))
        gemm_tn(dt, G, Cp, Cp, dU, Cp, Cp, ps.gptr(f"gconv_update_spa_graph_{lv}/DW"), C, R, C, C, wg=cx)
        dG = empty((R, Cp), dt, dev)
        gemm_nt(dt, [(dU, Cp, cx.opp(f"gupd_{lv}.n"), Cp, Cp)], dG, Cp, R, Cp, n_valid=C)
        dX1 = empty((R, Cp), dt, dev)
        dY = empty((R, Cp), dt, dev)      # (dU is still being read by the weight-gradient stream)
        _lib.call("cmpc_gconv_pre_bwd", dt, _p(dG), _p(G), _p(Y), _p(sums1), ps.pptr(ln1 + "/gamma"), _p(dX1), 0, _p(dY),
                  ps.gptr(ln1 + "/gamma"), ps.gptr(ln1 + "/beta"), _p(bs), B, N, Cp, C, _st())
        # Y = gw_w . Z,  Z = gw_v^T . X1
        Zf = zeros((B, Tp, Cp), F32, dev)
        gemm_tn(dt, gw_v_t, Tp, Tp, X1, Cp, Cp, Zf, Cp, N, T, C, nb2=B, a_bs=N * Tp, d_bs=N * Cp, o_bs=Tp * Cp)
        Z = empty((B, Tp, Cp), dt, dev)
        cast(Zf, F32, Z, dt, Z.numel())
        dgw_w = empty((B, N, Tp), F32, dev)
        gemm_nt(dt, [(dY, Cp, Z, Cp, Cp, N * Cp, Tp * Cp)], dgw_w, Tp, N, Tp, batch=B, sC=N * Tp, c_f32=True)
        dZf = zeros((B, Tp, Cp), F32, dev)
        gemm_tn(dt, gw_w_t, Tp, Tp, dY, Cp, Cp, dZf, Cp, N, T, C, nb2=B, a_bs=N * Tp, d_bs=N * Cp, o_bs=Tp * Cp)
        dZ = empty((B, Tp, Cp), dt, dev)
        cast(dZf, F32, dZ, dt, dZ.numel())
        dZtf = zeros((B, Cp, Tp), F32, dev)
        gemm_tn(dt, dY, Cp, Cp, gw_w_t, Tp, Tp, dZtf, Tp, N, C, T, nb2=B, a_bs=N * Cp, d_bs=N * Tp, o_bs=Cp * Tp)
        dZt = empty((B, Cp, Tp), dt, dev)
        cast(dZtf, F32, dZt, dt, dZt.numel())
        dgw_v = empty((B, N, Tp), F32, dev)
        gemm_nt(dt, [(X1, Cp, dZ, Cp, Cp, N * Cp, Tp * Cp)], dgw_v, Tp, N, Tp, batch=B, sC=N * Tp, c_f32=True)
        gemm_nt(dt, [(gw_v_t, Tp, dZt, Tp, Tp, N * Tp, Cp * Tp)], dX1, Cp, N, Cp, n_valid=C, batch=B, sC=N * Cp, accumulate=True)
        dA0 = empty((B, N, Tp), F32, dev)
        dA0_t = empty((B, N, Tp), dt, dev)
        dpr = empty((B * T,), F32, dev)
        _lib.call("cmpc_graph_softmax_bwd", dt, _p(dgw_w), _p(dgw_v), _p(gw_w), _p(gw_v), _p(A0), _p(pr), _p(mask),
                  _p(dA0), _p(dA0_t), _p(dpr), _p(empty((B * ((N + 63) // 64) * 128,), F32, dev)), B, N, T, Tp, _st())
        # A0 = scale * (X1 . PT^T) + k0s
        gemm_nt(dt, [(dA0_t, Tp, PTt, B * Tp, Tp, N * Tp, Tp)], dX1, Cp, N, Cp, n_valid=C, batch=B, sC=N * Cp, alpha=scale, accumulate=True)
        dPT = zeros((B * Tp, Cp), F32, dev)
        gemm_tn(dt, dA0_t, Tp, Tp, X1, Cp, Cp, dPT, Cp, N, T, C, nb2=B, a_bs=N * Tp, d_bs=N * Cp, o_bs=Tp * Cp, alpha=scale)
        dk0s = zeros((B, Tp), F32, dev)
        colsum(F32, dA0, R, Tp, Tp, T, dsb=dk0s, ld_dsb=Tp, rows_per_sample=N)
        # k0s = scale * Wd . b_t2 ; PT = Wd . W_t2^T
        _lib.call("cmpc_wcolsum", F32, _p(Wd), _p(dk0s), ps.gptr(f"spa_graph_trans2_{lv}/biases"), 0, 1, B * Tp, Cp, C, scale, _st())
        dWd = empty((B * Tp, Cp), F32, dev)
        gemm_nt(F32, [(dPT, Cp, cx.opp(f"t2_{lv}.t"), Cp, Cp)], dWd, Cp, B * Tp, Cp, n_valid=C)
        gemm_tn(F32, dPT, Cp, Cp, Wd, Cp, Cp, ps.gptr(f"spa_graph_trans2_{lv}/DW"), C, B * Tp, C, C, wg=cx)
        _lib.call("cmpc_rank1_update", F32, _p(dWd), _p(dk0s), ps.pptr(f"spa_graph_trans2_{lv}/biases"), None, None, 0,
                  scale, 0.0, 1, B * Tp, Cp, C, _st())
        # Wd = wf . W_w + b_w   (rows t < T of every sample; pad rows of dWd are zero)
        colsum(F32, dWd, B * Tp, Cp, Cp, C, db=ps.gptr(f"words_trans_{lv}/biases"))
        gemm_tn(F32, wf, Cp, Cp, dWd, Cp, Cp, ps.gptr(f"words_trans_{lv}/DW"), C, T, C, C, nb2=B, a_bs=T * Cp, d_bs=Tp * Cp, o_bs=0, wg=cx)
        dwf = empty((B * T, Cp), F32, dev)
        gemm_nt(F32, [(dWd, Cp, cx.opp(f"wtrans_{lv}.n"), Cp, Cp, Tp * Cp, 0)], dwf, Cp, T, Cp, n_valid=C, batch=B, sC=T * Cp)
        dparse = zeros((B * T, 4), F32, dev)
        dparse[:, 2] = dpr
        return dX1, dwf, dparse, None, None, None


# ---------------------------------------------------------------------------------------------
# S6: fusion 1x1 conv over [vis_la_sp | spa_graph | tile(valid_lang) | spatial], CMPC_model.py:338-344
# ---------------------------------------------------------------------------------------------
class Fusion(torch.autograd.Function):
    @_nocycle
    def forward(ctx, X1, X2, vl, lv: str, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, N, C, Cp, M, Mp = cfg.batch_size, cfg.N, cfg.v_emb_dim, cfg.Cp, cfg.mlp_dim, cfg.Mp
        R = B * N
        X1, X2, vl = X1.contiguous(), X2.contiguous(), vl.contiguous()
        sb = empty((B, Mp), F32, dev)
        gemm_nt(F32, [(vl, Cp, cx.opp(f"fusl_{lv}.t"), Cp, Cp)], sb, Mp, B, Mp, n_valid=M)
        F = empty((R, Mp), dt, dev)
        ldk = 2 * Cp + 64
        gemm_nt(dt, [(X1, Cp, cx.opp(f"fus_{lv}.t"), ldk, Cp), (X2, Cp, cx.opp(f"fus_{lv}.t", 0, Cp), ldk, Cp),
                     (cx.spatial, 64, cx.opp(f"fus_{lv}.t", 0, 2 * Cp), ldk, 64)],
                F, Mp, R, Mp, n_valid=M, bias=ps.pptr(f"fusion_{lv}/biases"), sbias=sb, ld_sbias=Mp, rows_per_sample=N, act=ACT_RELU)
        ctx.cx, ctx.lv, ctx.saved = cx, lv, (X1, X2, vl, F)
        return F

    @staticmethod
    def backward(ctx, dF):
        cx, lv = ctx.cx, ctx.lv
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        X1, X2, vl, F = ctx.saved
        B, N, C, Cp, M, Mp, Rr = cfg.batch_size, cfg.N, cfg.v_emb_dim, cfg.Cp, cfg.mlp_dim, cfg.Mp, cfg.rnn_size
        R = B * N
        dpre = empty((R, Mp), dt, dev)
        dsb = zeros((B, Mp), F32, dev)
        colsum(dt, dF.contiguous(), R, Mp, Mp, M, db=ps.gptr(f"fusion_{lv}/biases"), y=F, dpre=dpre, act=ACT_RELU,
               dsb=dsb, ld_dsb=Mp, rows_per_sample=N)
        gw = ps.gptr(f"fusion_{lv}/DW")
        gemm_tn(dt, X1, Cp, Cp, dpre, Mp, Mp, gw, M, R, C, M, wg=cx)
        gemm_tn(dt, X2, Cp, Cp, dpre, Mp, Mp, gw + 4 * C * M, M, R, C, M, wg=cx)
        gemm_tn(dt, cx.spatial, 64, 64, dpre, Mp, Mp, gw + 4 * (2 * C + Rr) * M, M, R, 8, M, wg=cx)
        gemm_tn(F32, vl, Cp, Cp, dsb, Mp, Mp, gw + 4 * 2 * C * M, M, B, Rr, M, wg=cx)
        dX1 = empty((R, Cp), dt, dev)
        dX2 = empty((R, Cp), dt, dev)
        gemm_nt(dt, [(dpre, Mp, cx.opp(f"fus_{lv}.n"), Mp, Mp)], dX1, Cp, R, Cp, n_valid=C)
        gemm_nt(dt, [(dpre, Mp, cx.opp(f"fus_{lv}.n", Cp, 0), Mp, Mp)], dX2, Cp, R, Cp, n_valid=C)
        dvl = empty((B, Cp), F32, dev)
        gemm_nt(F32, [(dsb, Mp, cx.opp(f"fusl_{lv}.n"), Mp, Mp)], dvl, Cp, B, Cp, n_valid=Rr)
        return dX1, dX2, dvl, None, None


# ---------------------------------------------------------------------------------------------
# S7: score head: _conv 3x3 M->1, resize_bilinear, sigmoid, weighed_logistic_loss, mIoU counters
#     (CMPC_model.py:128-142,440-447,486-490).  Returns (up, loss_term) with
#     loss_term = weight * mean_b sum_{H,W} BCE; backward assumes d(cost)/d(loss_term) = 1.
# ---------------------------------------------------------------------------------------------
class ScoreHead(torch.autograd.Function):
    @_nocycle
    def forward(ctx, feat, name: str, target, weight: float, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, h, w, H, W, M, Mp = cfg.batch_size, cfg.vf_h, cfg.vf_w, cfg.H, cfg.W, cfg.mlp_dim, cfg.Mp
        feat = feat.contiguous()
        score = empty((B, h, w, 1), F32, dev)
        _lib.call("cmpc_score_conv_fwd", dt, _p(feat), ps.pptr(f"{name}/DW"), ps.pptr(f"{name}/biases"), _p(score), B, h, w, Mp, M, _st())
        up = empty((B, H, W, 1), F32, dev)
        sigm = empty((B, H, W, 1), F32, dev)
        loss = zeros((B,), F32, dev)
        iu = torch.zeros((2, B), dtype=torch.int32, device=dev)
        _lib.call("cmpc_upsample_fwd", _p(score), _p(up), _p(sigm), _p(target) if target is not None else None,
                  _p(loss), _p(iu[0]), _p(iu[1]), B, h, w, H, W, _st())
        ctx.cx, ctx.name, ctx.weight, ctx.saved = cx, name, weight, (feat, up, target)
        ctx.mark_non_differentiable(score, up, sigm, iu)
        return loss, score, up, sigm, iu

    @staticmethod
    def backward(ctx, _dl, *_):
        cx, name = ctx.cx, ctx.name
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        feat, up, target = ctx.saved
        if target is None:
            raise RuntimeError("ScoreHead.backward needs target_fine")
        B, h, w, H, W, M, Mp = cfg.batch_size, cfg.vf_h, cfg.vf_w, cfg.H, cfg.W, cfg.mlp_dim, cfg.Mp
        dscore = empty((B, h, w), F32, dev)
        _lib.call("cmpc_upsample_loss_bwd", _p(up), _p(target), _p(dscore), ctx.weight / B, B, h, w, H, W, _st())
        dfeat = empty((B * h * w, Mp), dt, dev)
        _lib.call("cmpc_score_conv_bwd", dt, _p(dscore), _p(feat), ps.pptr(f"{name}/DW"), _p(dfeat), 0,
                  ps.gptr(f"{name}/DW"), ps.gptr(f"{name}/biases"), B, h, w, Mp, M, _st())
        return dfeat, None, None, None, None


# ---------------------------------------------------------------------------------------------
# S8: gated_exchange_module + l2_normalize, CMPC_model.py:194-259,271-284.  The key convolution is
# folded into the query: key.q = feat.(W_k q) + b_k.q, and softmax_N ignores the constant.
# ---------------------------------------------------------------------------------------------
class Exchange(torch.autograd.Function):
    @_nocycle
    def forward(ctx, feat, f1, f2, nec, lv: str, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, N, Cp, M, Mp = cfg.batch_size, cfg.N, cfg.Cp, cfg.mlp_dim, cfg.Mp
        R = B * N
        feat, f1, f2, nec = feat.contiguous(), f1.contiguous(), f2.contiguous(), nec.contiguous()
        s = 1.0 / math.sqrt(M)
        q = empty((B, Mp), F32, dev)
        gemm_nt(F32, [(nec, Cp, cx.opp(f"query_{lv}.t"), Cp, Cp)], q, Mp, B, Mp, n_valid=M, bias=ps.pptr(f"lang_query_{lv}gv_f1/biases"))
        kq = empty((B, Mp), F32, dev)
        gemm_nt(F32, [(q, Mp, cx.opp(f"key_{lv}.n"), Mp, Mp)], kq, Mp, B, Mp, n_valid=M)
        logits = empty((B, N), F32, dev)
        _lib.call("cmpc_rowdot1", dt, _p(feat), _p(kq), Mp, _p(logits), B, N, Mp, M, s, _st())
        attn = empty((B, N), F32, dev)
        _lib.call("cmpc_softmax_n_fwd", _p(logits), _p(attn), B, N, _st())
        pooled = zeros((B, Mp), F32, dev)
        _lib.call("cmpc_wcolsum", dt, _p(feat), _p(attn), _p(pooled), Mp, B, N, Mp, M, 1.0, _st())
        gvpre = empty((B, Mp), F32, dev)
        ldk = Mp + Cp
        gemm_nt(F32, [(pooled, Mp, cx.opp(f"gv_{lv}.t"), ldk, Mp), (nec, Cp, cx.opp(f"gv_{lv}.t", 0, Mp), ldk, Cp)],
                gvpre, Mp, B, Mp, n_valid=M, bias=ps.pptr(f"gv_lang_{lv}gv_f1/biases"))
        gv = empty((B, Mp), F32, dev)
        rs1 = empty((1,), F32, dev)
        _lib.call("cmpc_l2norm_all_fwd", _p(gvpre), _p(gv), _p(rs1), B * Mp, _st())
        g, r = [], []
        for k, fx in (("f1", f1), ("f2", f2)):
            gk = empty((B, Mp), F32, dev)
            gemm_nt(F32, [(gv, Mp, cx.opp(f"lfeat_{lv}_{k}.t"), Mp, Mp)], gk, Mp, B, Mp, n_valid=M,
                    bias=ps.pptr(f"lang_feat_{lv}_{k}/biases"), act=ACT_SIGMOID)
            rk = empty((R, Mp), dt, dev)
            gemm_nt(dt, [(fx, Mp, cx.opp(f"tfeat_{lv}_{k}.t"), Mp, Mp)], rk, Mp, R, Mp, n_valid=M,
                    bias=ps.pptr(f"trans_feat_{lv}_{k}/biases"), act=ACT_RELU)
            g.append(gk)
            r.append(rk)
        out = empty((R, Mp), dt, dev)
        rstd = empty((R,), F32, dev)
        _lib.call("cmpc_exchange_combine_fwd", dt, _p(feat), _p(r[0]), _p(r[1]), _p(g[0]), _p(g[1]), Mp, _p(out), _p(rstd), B, N, Mp, M, _st())
        ctx.cx, ctx.lv = cx, lv
        ctx.saved = (feat, f1, f2, nec, q, kq, attn, pooled, gv, rs1, g, r, out, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        cx, lv = ctx.cx, ctx.lv
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        feat, f1, f2, nec, q, kq, attn, pooled, gv, rs1, g, r, out, rstd = ctx.saved
        B, N, Cp, M, Mp, Rr = cfg.batch_size, cfg.N, cfg.Cp, cfg.mlp_dim, cfg.Mp, cfg.rnn_size
        R = B * N
        s = 1.0 / math.sqrt(M)
        dfeat = empty((R, Mp), dt, dev)
        dp = [empty((R, Mp), dt, dev), empty((R, Mp), dt, dev)]
        dg = [zeros((B, Mp), F32, dev), zeros((B, Mp), F32, dev)]
        _lib.call("cmpc_exchange_combine_bwd", dt, _p(dout.contiguous()), _p(out), _p(rstd), _p(r[0]), _p(r[1]), _p(g[0]), _p(g[1]), Mp,
                  _p(dfeat), 0, _p(dp[0]), _p(dp[1]), _p(dg[0]), _p(dg[1]), B, N, Mp, M, _st())
        dfs = []
        dgv = empty((B, Mp), F32, dev)
        for i, (k, fx) in enumerate((("f1", f1), ("f2", f2))):
            colsum(dt, dp[i], R, Mp, Mp, M, db=ps.gptr(f"trans_feat_{lv}_{k}/biases"))
            gemm_tn(dt, fx, Mp, Mp, dp[i], Mp, Mp, ps.gptr(f"trans_feat_{lv}_{k}/DW"), M, R, M, M, wg=cx)
            dfx = empty((R, Mp), dt, dev)
            gemm_nt(dt, [(dp[i], Mp, cx.opp(f"tfeat_{lv}_{k}.n"), Mp, Mp)], dfx, Mp, R, Mp, n_valid=M)
            dfs.append(dfx)
            colsum(F32, dg[i], B, Mp, Mp, M, db=ps.gptr(f"lang_feat_{lv}_{k}/biases"), y=g[i], dpre=dg[i], act=ACT_SIGMOID)
            gemm_tn(F32, gv, Mp, Mp, dg[i], Mp, Mp, ps.gptr(f"lang_feat_{lv}_{k}/DW"), M, B, M, M, wg=cx)
            gemm_nt(F32, [(dg[i], Mp, cx.opp(f"lfeat_{lv}_{k}.n"), Mp, Mp)], dgv, Mp, B, Mp, n_valid=M, accumulate=(i == 1))
        dgvpre = empty((B, Mp), F32, dev)
        _lib.call("cmpc_l2norm_all_bwd", _p(dgv), _p(gv), _p(rs1), _p(dgvpre), B * Mp, _st())
        colsum(F32, dgvpre, B, Mp, Mp, M, db=ps.gptr(f"gv_lang_{lv}gv_f1/biases"))
        gwg = ps.gptr(f"gv_lang_{lv}gv_f1/DW")
        gemm_tn(F32, pooled, Mp, Mp, dgvpre, Mp, Mp, gwg, M, B, M, M, wg=cx)
        gemm_tn(F32, nec, Cp, Cp, dgvpre, Mp, Mp, gwg + 4 * M * M, M, B, Rr, M, wg=cx)
        dpooled = empty((B, Mp), F32, dev)
        gemm_nt(F32, [(dgvpre, Mp, cx.opp(f"gv_{lv}.n"), Mp, Mp)], dpooled, Mp, B, Mp, n_valid=M)
        dnec = empty((B, Cp), F32, dev)
        gemm_nt(F32, [(dgvpre, Mp, cx.opp(f"gv_{lv}.n", Mp, 0), Mp, Mp)], dnec, Cp, B, Cp, n_valid=Rr)
        dattn = empty((B, N), F32, dev)
        _lib.call("cmpc_rowdot1", dt, _p(feat), _p(dpooled), Mp, _p(dattn), B, N, Mp, M, 1.0, _st())
        dlog = empty((B, N), F32, dev)
        _lib.call("cmpc_softmax_n_bwd", _p(dattn), _p(attn), _p(dlog), B, N, _st())
        _lib.call("cmpc_rank1_update", dt, _p(dfeat), _p(attn), _p(dpooled), _p(dlog), _p(kq), Mp, 1.0, s, B, N, Mp, M, _st())
        dkq = zeros((B, Mp), F32, dev)
        _lib.call("cmpc_wcolsum", dt, _p(feat), _p(dlog), _p(dkq), Mp, B, N, Mp, M, s, _st())
        dq = empty((B, Mp), F32, dev)
        gemm_nt(F32, [(dkq, Mp, cx.opp(f"key_{lv}.t"), Mp, Mp)], dq, Mp, B, Mp, n_valid=M)
        gemm_tn(F32, dkq, Mp, Mp, q, Mp, Mp, ps.gptr(f"spa_graph_key_{lv}gv_f1/DW"), M, B, M, M, wg=cx)
        colsum(F32, dq, B, Mp, Mp, M, db=ps.gptr(f"lang_query_{lv}gv_f1/biases"))
        gemm_tn(F32, nec, Cp, Cp, dq, Mp, Mp, ps.gptr(f"lang_query_{lv}gv_f1/DW"), M, B, Rr, M, wg=cx)
        gemm_nt(F32, [(dq, Mp, cx.opp(f"query_{lv}.n"), Mp, Mp)], dnec, Cp, B, Cp, n_valid=Rr, accumulate=True)
        return dfeat, dfs[0], dfs[1], dnec, None, None


# ---------------------------------------------------------------------------------------------
# S9: ConvLSTM over (exg3_2, exg4_2, exg5_2), util/cell.py:36-79 via CMPC_model.py:287-290
# ---------------------------------------------------------------------------------------------
_LN_NAMES = ("LayerNorm", "LayerNorm_1", "LayerNorm_2", "LayerNorm_3", "LayerNorm_4")   # j, i, f, o, c


def _clstm_ln(ps: ParamStore):
    pre = "rnn/conv_lstm_cell/"
    ln, dln = ConvLstmLn(), ConvLstmDln()
    for i, s in enumerate(_LN_NAMES):
        ln.beta[i], ln.gamma[i] = ps.pptr(pre + s + "/beta"), ps.pptr(pre + s + "/gamma")
        dln.dbeta[i], dln.dgamma[i] = ps.gptr(pre + s + "/beta"), ps.gptr(pre + s + "/gamma")
    return ln, dln


class ConvLSTM(torch.autograd.Function):
    @_nocycle
    def forward(ctx, x1, x2, x3, cx: Ctx):
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, N, M, Mp = cfg.batch_size, cfg.N, cfg.mlp_dim, cfg.Mp
        R = B * N
        pre = "rnn/conv_lstm_cell/"
        ln, _ = _clstm_ln(ps)
        xs = [x.contiguous() for x in (x1, x2, x3)]
        st = []
        h = c = None
        for s, x in enumerate(xs):
            Yg = empty((R, 4 * Mp), dt, dev)
            segs = [(x, Mp, cx.opp("clstm.t"), 2 * Mp, Mp)]
            if s > 0:
                segs.append((h, Mp, cx.opp("clstm.t", 0, Mp), 2 * Mp, Mp))
            gemm_nt(dt, segs, Yg, 4 * Mp, R, 4 * Mp)
            sums = torch.empty((5, B, 2), dtype=torch.float64, device=dev)
            _lib.call("cmpc_convlstm_a", dt, _p(Yg), _p(c), ps.pptr(pre + "W_ci"), ps.pptr(pre + "W_cf"), _p(sums), B, N, Mp, M, _st())
            c_pre = empty((R, Mp), dt, dev)
            _lib.call("cmpc_convlstm_b", dt, _p(Yg), _p(c), ps.pptr(pre + "W_co"), ctypes.byref(ln), _p(sums), _p(c_pre), B, N, Mp, M, _st())
            c_new = empty((R, Mp), dt, dev)
            h_new = empty((R, Mp), dt, dev)
            _lib.call("cmpc_convlstm_c", dt, _p(Yg), _p(c_pre), ctypes.byref(ln), _p(sums), _p(c_new), _p(h_new), B, N, Mp, M, _st())
            st.append((x, h, c, Yg, sums, c_pre))
            h, c = h_new, c_new
        ctx.cx, ctx.saved = cx, st
        return h

    @staticmethod
    def backward(ctx, dh):
        cx = ctx.cx
        cfg, ps, dev, dt = cx.cfg, cx.ps, cx.dev, cx.dt
        B, N, M, Mp = cfg.batch_size, cfg.N, cfg.mlp_dim, cfg.Mp
        R = B * N
        pre = "rnn/conv_lstm_cell/"
        ln, dln = _clstm_ln(ps)
        dh = dh.contiguous()
        dc = None
        dxs = [None, None, None]
        gk = ps.gptr(pre + "kernel")
        scr = empty((R, Mp), dt, dev)
        bs = torch.empty((5, B, 2), dtype=torch.float64, device=dev)
        for s in reversed(range(3)):
            x, h_prev, c_prev, Yg, sums, c_pre = ctx.saved[s]
            dYg = empty((R, 4 * Mp), dt, dev)
            dc_prev = empty((R, Mp), dt, dev) if s > 0 else None
            _lib.call("cmpc_convlstm_bwd", dt, _p(dh), _p(dc), _p(Yg), _p(c_prev), _p(c_pre),
                      ps.pptr(pre + "W_ci"), ps.pptr(pre + "W_cf"), ps.pptr(pre + "W_co"), ctypes.byref(ln), _p(sums),
                      _p(dYg), _p(dc_prev), ps.gptr(pre + "W_ci"), ps.gptr(pre + "W_cf"), ps.gptr(pre + "W_co"),
                      ctypes.byref(dln), _p(scr), _p(bs), B, N, Mp, M, _st())
            gemm_tn(dt, x, Mp, Mp, dYg, 4 * Mp, Mp, gk, 4 * M, R, M, M, offs=tuple((0, g * Mp, g * M) for g in range(4)), wg=cx)
            dx = empty((R, Mp), dt, dev)
            gemm_nt(dt, [(dYg, 4 * Mp, cx.opp("clstm.n"), 4 * Mp, 4 * Mp)], dx, Mp, R, Mp, n_valid=M)
            dxs[s] = dx
            if s > 0:
                gemm_tn(dt, h_prev, Mp, Mp, dYg, 4 * Mp, Mp, gk, 4 * M, R, M, M,
                        offs=tuple((0, g * Mp, M * 4 * M + g * M) for g in range(4)), wg=cx)
                dh = empty((R, Mp), dt, dev)
                gemm_nt(dt, [(dYg, 4 * Mp, cx.opp("clstm.n", Mp, 0), 4 * Mp, 4 * Mp)], dh, Mp, R, Mp, n_valid=M)
                dc = dc_prev
        return dxs[0], dxs[1], dxs[2], None
