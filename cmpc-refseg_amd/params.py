"""Parameter manifest, flat fp32 master buffers and packed GEMM operands of the CMPC head.

Names, shapes, creation order and initialisers follow the variables the reference graph creates
under scope "text_objseg" (CMPC_model.py:84; _conv :412-417; lstm :144-156; layer_norm :364,370;
util/cell.py:42-66).  76,055,608 trainable scalars at the default sizes.

Layout decisions (MI355X-first):
  * ONE flat fp32 master buffer + ONE flat fp32 gradient buffer (+ Adam m, v): a single fused Adam
    launch and a single RCCL all-reduce over the gradient buffer.
  * GEMM operands are packed copies (bf16 or fp32) of the masters, zero-padded to multiples of 64
    so every MFMA tile is full; each weight is packed output-major (forward) and input-major
    (dX) so that both products read K-contiguous rows.  Packing is one launch per step.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, AdamSeg, PackDesc

LEVELS = ("c5", "c4", "c3")                              # CMPC_model.py:120-125
EXG = ("c3", "c4", "c5", "c3_2", "c4_2", "c5_2")         # CMPC_model.py:271-283


def pad64(x: int) -> int:
    return (x + 63) // 64 * 64


@dataclass
class HeadCfg:
    """Graph-shaping constructor arguments of LSTM_model (CMPC_model.py:15-40)."""
    batch_size: int = 1
    num_steps: int = 20
    vf_h: int = 40
    vf_w: int = 40
    H: int = 320
    W: int = 320
    vf_dim: int = 2048
    c4_dim: int = 1024          # hard-coded in the reference, CMPC_model.py:110
    c3_dim: int = 512           # CMPC_model.py:112
    vocab_size: int = 12112
    v_emb_dim: int = 1000
    mlp_dim: int = 500
    rnn_size: int = 1000
    glove_dim: int = 300
    parse_dim: int = 500        # CMPC_model.py:349
    start_lr: float = 0.00025
    lr_decay_step: int = 800000
    end_lr: float = 0.00001
    lr_power: float = 0.9
    weight_decay: float = 0.0005
    # which graph (get_model.py:15-17): "CMPC_model" or "CMPCv5_BiLSTM_model" (+ hsv = CMPCv5_BiLSTM_HSV_model), and the latter's extra
    # constructor arguments / hard-coded sizes (CMPCv5_BiLSTM_model.py:42,88,153,196,208,225)
    model: int = 0
    hsv: int = 0
    bn_train: int = 0
    bn_decay: float = 0.9997
    c2_dim: int = 256
    c2_h: int = 80
    c2_w: int = 80
    aspp_depth: int = 256
    low_dim: int = 48
    aspp_rates: tuple = (6, 12, 18)
    sample_frames: int = 5      # CMPC_video_mm_tgraph_allvec.py:69
    freeze_bn: int = 0          # CMPCv5_BiLSTM_model.py:528-529
    conv5: int = 0              # CMPC_model.py:427-430: res3-res5 convolution weights trained (the handle then also returns d cost / d taps)

    @property
    def N(self):
        return self.vf_h * self.vf_w

    @property
    def Cp(self):
        return pad64(self.v_emb_dim)

    @property
    def Mp(self):
        return pad64(self.mlp_dim)

    @property
    def Gp(self):
        return pad64(self.glove_dim)

    @property
    def Pp(self):
        return pad64(self.parse_dim)

    @property
    def Tp(self):
        return 64


def head_param_specs(cfg: HeadCfg) -> List[Tuple[str, Tuple[int, ...], str, Tuple[str, ...]]]:
    """(name, shape, initialiser, flags); flags: 'reg' = L2-regularised ('DW' in the name,
    CMPC_model.py:433), 'x2' = gradient doubled ('biases' in the name, :464-465)."""
    C, M, R = cfg.v_emb_dim, cfg.mlp_dim, cfg.rnn_size
    out = []

    def conv(name, k, cin, cout):
        out.append((f"text_objseg/{name}/DW", (k, k, cin, cout), "xavier", ("reg",)))
        out.append((f"text_objseg/{name}/biases", (cout,), "zeros", ("x2",)))

    def ln(scope, dim):
        out.append((f"text_objseg/{scope}/beta", (dim,), "zeros", ()))
        out.append((f"text_objseg/{scope}/gamma", (dim,), "ones", ()))

    out.append(("text_objseg/Variable", (cfg.vocab_size, cfg.glove_dim), "glove", ()))
    out.append(("text_objseg/rnn/lstm_cell/kernel", (cfg.glove_dim + R, 4 * R), "glorot", ()))
    out.append(("text_objseg/rnn/lstm_cell/bias", (4 * R,), "zeros", ()))
    conv("c5_lateral", 1, cfg.vf_dim, C)
    conv("c4_lateral", 1, cfg.c4_dim, C)
    conv("c3_lateral", 1, cfg.c3_dim, C)
    conv("words_parse_1", 1, R, cfg.parse_dim)
    conv("words_parse_2", 1, cfg.parse_dim, 4)
    for lv in LEVELS:
        for h in range(1, 6):
            conv(f"vis_trans_{lv}_head{h}", 1, C + 8, C)
            conv(f"lang_trans_{lv}_head{h}", 1, R, C)
        conv(f"words_trans_{lv}", 1, R, R)
        conv(f"spa_graph_trans2_{lv}", 1, C, C)
        ln(f"gconv_feat_ln_spa_graph_{lv}", C)
        conv(f"gconv_update_spa_graph_{lv}", 1, C, C)
        ln(f"gconv_update_ln_spa_graph_{lv}", C)
        conv(f"fusion_{lv}", 1, 2 * C + R + 8, M)
    for lv in LEVELS:
        conv(f"score_{lv}", 3, M, 1)
    for lv in EXG:
        conv(f"spa_graph_key_{lv}gv_f1", 1, M, M)
        conv(f"lang_query_{lv}gv_f1", 1, R, M)
        conv(f"gv_lang_{lv}gv_f1", 1, M + R, M)
        conv(f"lang_feat_{lv}_f1", 1, M, M)
        conv(f"trans_feat_{lv}_f1", 1, M, M)
        conv(f"lang_feat_{lv}_f2", 1, M, M)
        conv(f"trans_feat_{lv}_f2", 1, M, M)
    pre = "rnn/conv_lstm_cell"
    out.append((f"text_objseg/{pre}/kernel", (1, 1, 2 * M, 4 * M), "glorot", ()))
    out.append((f"text_objseg/{pre}/W_ci", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    out.append((f"text_objseg/{pre}/W_cf", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    ln(f"{pre}/LayerNorm", M)
    ln(f"{pre}/LayerNorm_1", M)
    ln(f"{pre}/LayerNorm_2", M)
    out.append((f"text_objseg/{pre}/W_co", (cfg.vf_h, cfg.vf_w, M), "glorot", ()))
    ln(f"{pre}/LayerNorm_3", M)
    ln(f"{pre}/LayerNorm_4", M)
    conv("score", 3, M, 1)
    return out


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def init_head_params(cfg: HeadCfg, seed: int = 1234, glove_seed: int = 7) -> Dict[str, torch.Tensor]:
    """Reference initialisers: xavier_initializer_conv2d / glorot_uniform = U(+-sqrt(6/(fi+fo)));
    biases, beta 0; gamma 1; the GloVe table (blob missing from the reference tree) ~ N(0, 0.4^2)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind, _ in head_param_specs(cfg):
        if kind in ("xavier", "glorot"):
            fi, fo = _fans(shape)
            t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / (fi + fo))
        elif kind == "zeros":
            t = torch.zeros(shape, dtype=torch.float64)
        elif kind == "ones":
            t = torch.ones(shape, dtype=torch.float64)
        else:
            gg = torch.Generator().manual_seed(glove_seed)
            t = torch.randn(shape, generator=gg, dtype=torch.float64) * 0.4
        out[name] = t.float()
    return out


@dataclass
class Operand:
    """A packed, zero-padded copy of (a slice of) one or more weights inside the operand arena."""
    off: int        # byte offset
    dt: int
    rows: int
    ld: int

    def ptr(self, arena: torch.Tensor, row: int = 0, col: int = 0) -> int:
        esz = 4 if self.dt == DT_F32 else 2
        return arena.data_ptr() + self.off + (row * self.ld + col) * esz


class ParamStore:
    """Flat fp32 masters / gradients / Adam state + the packed-operand arena."""

    def __init__(self, cfg: HeadCfg, device, vis_dt: int):
        self.cfg, self.device, self.vis_dt = cfg, torch.device(device), vis_dt
        self.specs = head_param_specs(cfg)
        self.index: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape, _, _ in self.specs:
            self.index[name] = (off, shape)
            off += (int(np.prod(shape)) + 3) // 4 * 4          # keep every parameter 16-B aligned
        self.total = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.m = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.v = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.step = 0
        self._descs: List[PackDesc] = []
        self._arena_bytes = 0
        self.ops: Dict[str, Operand] = {}
        self._plan_operands()
        self.arena = torch.zeros(self._arena_bytes, dtype=torch.uint8, device=self.device)
        self._upload_tables()

    # ---- views ------------------------------------------------------------------------------
    def p(self, name: str) -> torch.Tensor:
        off, shape = self.index["text_objseg/" + name]
        return self.params[off: off + int(np.prod(shape))].view(shape)

    def g(self, name: str) -> torch.Tensor:
        off, shape = self.index["text_objseg/" + name]
        return self.grads[off: off + int(np.prod(shape))].view(shape)

    def poff(self, name: str) -> int:
        return self.index["text_objseg/" + name][0]

    def pptr(self, name: str, elem: int = 0) -> int:
        return self.params.data_ptr() + 4 * (self.poff(name) + elem)

    def gptr(self, name: str, elem: int = 0) -> int:
        return self.grads.data_ptr() + 4 * (self.poff(name) + elem)

    def load_state(self, named: Dict[str, torch.Tensor]):
        """Set weights by reference variable name (the counterpart of tf.train.Saver.restore)."""
        for name, (off, shape) in self.index.items():
            if name not in named:
                raise KeyError(f"missing parameter {name}")
            t = named[name]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
            self.params[off: off + t.numel()].copy_(t.reshape(-1).to(torch.float32))
        self.pack()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: self.params[o: o + int(np.prod(s))].view(s).detach().cpu().clone() for n, (o, s) in self.index.items()}

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        return {n: self.grads[o: o + int(np.prod(s))].view(s).detach().cpu().clone() for n, (o, s) in self.index.items()}

    # ---- operand planning -----------------------------------------------------------------------
    def _new_operand(self, key: str, dt: int, rows: int, ld: int) -> Operand:
        esz = 4 if dt == DT_F32 else 2
        op = Operand(self._arena_bytes, dt, rows, ld)
        self._arena_bytes += (rows * ld * esz + 255) // 256 * 256
        self.ops[key] = op
        return op

    def _add_desc(self, op: Operand, pname: str, ld_src: int, transpose: int, row0: int, col0: int, rows: int, cols: int,
                  ks, ns, src_elem_off: int = 0):
        """Fill the sub-block [row0:row0+rows, col0:col0+cols] of operand `op` from parameter pname.
        ks / ns: lists of (src, len, dst) relative to the sub-block (k = reduction index of the
        MASTER matrix rows, n = its columns)."""
        esz = 4 if op.dt == DT_F32 else 2
        d = PackDesc()
        d.src_off = self.poff(pname) + src_elem_off
        d.ld_src = ld_src
        d.dst_off = op.off + (row0 * op.ld + col0) * esz
        d.dst_dt, d.transpose = op.dt, transpose
        d.rows, d.cols, d.ld_dst = rows, cols, op.ld
        d.nks = len(ks)
        for i, (s, l, t) in enumerate(ks):
            d.ks_src[i], d.ks_len[i], d.ks_dst[i] = s, l, t
        d.nns = len(ns)
        for i, (s, l, t) in enumerate(ns):
            d.ns_src[i], d.ns_len[i], d.ns_dst[i] = s, l, t
        self._descs.append(d)

    def _linear(self, key: str, pname: str, dt: int, K: int, N: int, Kp: int, Np: int, fwd=True, bwd=True,
                ks=None, ns=None):
        """Standard pair: '<key>.t' = [Np][Kp] (output-major, forward) and '<key>.n' = [Kp][Np]."""
        ks = ks or [(0, K, 0)]
        ns = ns or [(0, N, 0)]
        ld_src = self.index["text_objseg/" + pname][1][-1]
        if fwd:
            op = self._new_operand(key + ".t", dt, Np, Kp)
            self._add_desc(op, pname, ld_src, 1, 0, 0, Np, Kp, ks, ns)
        if bwd:
            op = self._new_operand(key + ".n", dt, Kp, Np)
            self._add_desc(op, pname, ld_src, 0, 0, 0, Kp, Np, ks, ns)

    def _plan_operands(self):
        c = self.cfg
        C, M, R, G, P = c.v_emb_dim, c.mlp_dim, c.rnn_size, c.glove_dim, c.parse_dim
        Cp, Mp, Gp, Pp = c.Cp, c.Mp, c.Gp, c.Pp
        V, L = self.vis_dt, DT_F32
        if R != C:
            raise ValueError("rnn_size must equal v_emb_dim (the affinity contracts them, CMPC_model.py:384)")
        # text LSTM: kernel [G+R, 4R], gates i,j,f,o -> padded gate blocks of Cp
        gate_ns = [(g * R, R, g * Cp) for g in range(4)]
        self._linear("lstm", "rnn/lstm_cell/kernel", L, G + R, 4 * R, Gp + Cp, 4 * Cp,
                     ks=[(0, G, 0), (G, R, Gp)], ns=gate_ns)
        ob = self._new_operand("lstm.b", L, 1, 4 * Cp)       # bias re-blocked to the padded gate layout
        self._add_desc(ob, "rnn/lstm_cell/bias", 4 * R, 0, 0, 0, 1, 4 * Cp, [(0, 1, 0)], gate_ns)
        self._linear("parse1", "words_parse_1/DW", L, R, P, Cp, Pp)
        self._linear("parse2", "words_parse_2/DW", L, P, 4, Pp, 64)
        # everything above is what the text encoder + parser read: packed (and published) first, see pack()
        self.stage0_ndesc = len(self._descs)
        for lv, cin in (("c5", c.vf_dim), ("c4", c.c4_dim), ("c3", c.c3_dim)):
            self._linear(f"lat_{lv}", f"{lv}_lateral/DW", V, cin, C, pad64(cin), Cp, bwd=False)
        for lv in LEVELS:
            # mutan: five heads side by side.  forward operand [5Cp][Cp+64] (k: C visual rows then 8 spatial rows)
            opt = self._new_operand(f"mutan_{lv}.t", V, 5 * Cp, Cp + 64)
            opn = self._new_operand(f"mutan_{lv}.n", V, Cp, 5 * Cp)
            lgt = self._new_operand(f"mlang_{lv}.t", L, 5 * Cp, Cp)
            lgn = self._new_operand(f"mlang_{lv}.n", L, Cp, 5 * Cp)
            opb = self._new_operand(f"mutan_{lv}.b", L, 1, 5 * Cp)
            lgb = self._new_operand(f"mlang_{lv}.b", L, 1, 5 * Cp)
            for h in range(5):
                pn = f"vis_trans_{lv}_head{h + 1}/DW"
                self._add_desc(opt, pn, C, 1, h * Cp, 0, Cp, Cp + 64, [(0, C, 0), (C, 8, Cp)], [(0, C, 0)])
                self._add_desc(opn, pn, C, 0, 0, h * Cp, Cp, Cp, [(0, C, 0)], [(0, C, 0)])
                pl = f"lang_trans_{lv}_head{h + 1}/DW"
                self._add_desc(lgt, pl, C, 1, h * Cp, 0, Cp, Cp, [(0, R, 0)], [(0, C, 0)])
                self._add_desc(lgn, pl, C, 0, 0, h * Cp, Cp, Cp, [(0, R, 0)], [(0, C, 0)])
                self._add_desc(opb, f"vis_trans_{lv}_head{h + 1}/biases", C, 0, 0, h * Cp, 1, Cp, [(0, 1, 0)], [(0, C, 0)])
                self._add_desc(lgb, f"lang_trans_{lv}_head{h + 1}/biases", C, 0, 0, h * Cp, 1, Cp, [(0, 1, 0)], [(0, C, 0)])
            self._linear(f"wtrans_{lv}", f"words_trans_{lv}/DW", L, R, R, Cp, Cp)
            self._linear(f"t2_{lv}", f"spa_graph_trans2_{lv}/DW", L, C, C, Cp, Cp)
            self._linear(f"gupd_{lv}", f"gconv_update_spa_graph_{lv}/DW", V, C, C, Cp, Cp)
            fus = f"fusion_{lv}/DW"
            # visual part: K = [vis_la_sp | spa_graph | spatial(8->64)]
            self._linear(f"fus_{lv}", fus, V, 2 * C + R + 8, M, 2 * Cp + 64, Mp,
                         ks=[(0, C, 0), (C, C, Cp), (2 * C + R, 8, 2 * Cp)])
            self._linear(f"fusl_{lv}", fus, L, R, M, Cp, Mp, ks=[(2 * C, R, 0)])
        for lv in EXG:
            self._linear(f"key_{lv}", f"spa_graph_key_{lv}gv_f1/DW", L, M, M, Mp, Mp)
            self._linear(f"query_{lv}", f"lang_query_{lv}gv_f1/DW", L, R, M, Cp, Mp)
            self._linear(f"gv_{lv}", f"gv_lang_{lv}gv_f1/DW", L, M + R, M, Mp + Cp, Mp, ks=[(0, M, 0), (M, R, Mp)])
            for f in ("f1", "f2"):
                self._linear(f"lfeat_{lv}_{f}", f"lang_feat_{lv}_{f}/DW", L, M, M, Mp, Mp)
                self._linear(f"tfeat_{lv}_{f}", f"trans_feat_{lv}_{f}/DW", V, M, M, Mp, Mp)
        self._linear("clstm", "rnn/conv_lstm_cell/kernel", V, 2 * M, 4 * M, 2 * Mp, 4 * Mp,
                     ks=[(0, M, 0), (M, M, Mp)], ns=[(g * M, M, g * Mp) for g in range(4)])

    def _upload_tables(self):
        n = len(self._descs)
        arr = (PackDesc * n)(*self._descs)
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.descs_dev = torch.from_numpy(raw).to(self.device)
        self.ndesc = n
        pref = np.zeros(n + 1, dtype=np.int32)
        for i, d in enumerate(self._descs):
            kp, np_ = (d.cols, d.rows) if d.transpose else (d.rows, d.cols)
            pref[i + 1] = pref[i] + ((kp + 63) // 64) * ((np_ + 127) // 128)
        self.tile_prefix = torch.from_numpy(pref).to(self.device)
        self.total_tiles = int(pref[-1])
        self.tile_desc = torch.from_numpy(np.repeat(np.arange(n, dtype=np.int32), np.diff(pref))).to(self.device)
        self.stage0_tiles = int(pref[self.stage0_ndesc])
        # Adam segments: <= 8192 elements each, inside one parameter
        segs = []
        for name, shape, _, flags in self.specs:
            off, _ = self.index[name]
            cnt = int(np.prod(shape))
            wd = self.cfg.weight_decay if "reg" in flags else 0.0
            gm = 2.0 if "x2" in flags else 1.0
            for s in range(0, cnt, 8192):
                segs.append(AdamSeg(off + s, min(8192, cnt - s), wd, gm))
        sarr = (AdamSeg * len(segs))(*segs)
        self.segs_dev = torch.from_numpy(np.frombuffer(bytes(sarr), dtype=np.uint8).copy()).to(self.device)
        self.nseg = len(segs)

    # ---- device ops -----------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def pack(self, part=None):
        """master -> packed operands.  part=0: the text encoder's and parser's operands only; part=1: the rest;
        None: everything (one launch)."""
        lo = 0 if part in (None, 0) else self.stage0_tiles
        hi = self.total_tiles if part in (None, 1) else self.stage0_tiles
        _lib.call("cmpc_pack_weights_range", self.params.data_ptr(), self.arena.data_ptr(), self.descs_dev.data_ptr(),
                  self.tile_prefix.data_ptr(), self.tile_desc.data_ptr(), self.ndesc, lo, hi, self._stream())

    def zero_grads(self):
        self.grads.zero_()

    def lr(self) -> float:
        """tf.train.polynomial_decay(start_lr, step, decay_steps, end 1e-5, power 0.9) (CMPC_model.py:451)."""
        c = self.cfg
        gs = min(self.step, c.lr_decay_step)
        return (c.start_lr - c.end_lr) * (1 - gs / c.lr_decay_step) ** c.lr_power + c.end_lr

    def adam_step(self, gscale: float = 1.0, on_stage0=None):
        """TF AdamOptimizer.apply_gradients (CMPC_model.py:456,478) over the whole flat buffer, then repack.
        on_stage0: called between the two pack launches (the text encoder's operands are final there)."""
        lr = self.lr()
        t = self.step + 1
        b1, b2 = 0.9, 0.999
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        _lib.call("cmpc_adam_step", self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                  self.segs_dev.data_ptr(), self.nseg, lr_t, b1, b2, 1e-8, gscale, None, self._stream())
        self.step = t
        if on_stage0 is None:
            self.pack()
        else:
            self.pack(0)
            on_stage0()
            self.pack(1)
        return lr
