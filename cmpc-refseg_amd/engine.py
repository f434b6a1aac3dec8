"""Host-side wrapper of the whole-path C ABI (include/cmpc.h: cmpc_create / cmpc_forward / cmpc_backward /
cmpc_optimizer_step / cmpc_tap ...).  The handle owns parameters, packed operands, workspaces and lane streams; this
module only (a) fills cmpc_cfg from the LSTM_model keyword arguments (CMPC_model.py:15-40), (b) exposes the handle's
device buffers as torch tensors (zero-copy views through __cuda_array_interface__) so that state_dict / load_weights /
the RCCL all-reduce can reach them, and (c) forwards the three per-step calls.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, EngineCfg, Feeds, Fetches
from .params import HeadCfg

_TYPESTR = {0: "<f4", 1: "<i2", 2: "<f2", 3: "<i4", 4: "<f8"}      # cmpc_tap dtype codes; bf16 travels as int16 and is re-viewed


class _DevMem:
    """A span of device memory owned by the handle, described for torch.as_tensor."""

    def __init__(self, ptr: int, shape: Tuple[int, ...], typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def dev_tensor(ptr: int, shape, dt: int, device) -> torch.Tensor:
    t = torch.as_tensor(_DevMem(ptr, tuple(shape), _TYPESTR[dt]), device=device)
    return t.view(torch.bfloat16) if dt == 1 else t


class Engine:
    """One cmpc_handle.  `store`-style accessors (params / grads views, state_dict, grad_dict, load_state) mirror what
    tf.train.Saver and tf.gradients give the reference's driver (trainval_model.py:46-63)."""

    def __init__(self, cfg: HeadCfg, dt: int, device: torch.device, n_lanes: int = 3, loss_w=None):
        lib = _lib.load()
        self.lib, self.cfg, self.dt, self.device = lib, cfg, dt, torch.device(device)
        c = EngineCfg()
        _lib.call("cmpc_default_cfg", C.byref(c))
        for k in ("batch_size", "num_steps", "vf_h", "vf_w", "H", "W", "vf_dim", "c4_dim", "c3_dim", "vocab_size", "v_emb_dim",
                  "mlp_dim", "rnn_size", "glove_dim", "parse_dim", "start_lr", "end_lr", "lr_power", "lr_decay_step", "weight_decay"):
            setattr(c, k, getattr(cfg, k))
        _lib.call("cmpc_default_cfg_model", C.byref(c), int(cfg.model), int(cfg.hsv))
        if cfg.model == _lib.MODEL_V5_BILSTM:
            for k in ("bn_train", "bn_decay", "c2_dim", "c2_h", "c2_w", "aspp_depth", "low_dim"):
                setattr(c, k, getattr(cfg, k))
            for i, r in enumerate(cfg.aspp_rates):
                c.aspp_rates[i] = int(r)
        if cfg.model == _lib.MODEL_VIDEO:
            c.sample_frames = int(cfg.sample_frames)
        c.conv5 = int(getattr(cfg, "conv5", 0))
        c.freeze_bn = int(getattr(cfg, "freeze_bn", 0))
        if loss_w is not None:
            for i, w in enumerate(loss_w):
                c.loss_w[i] = w
        c.dtype, c.n_lanes = dt, n_lanes
        c.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.ccfg = c
        h = C.c_void_p()
        _lib.call("cmpc_create", C.byref(c), C.byref(h))
        self.h = h
        eff = EngineCfg()
        _lib.call("cmpc_get_cfg", h, C.byref(eff))
        self.loss_scale = float(eff.loss_scale)           # the gradient buffer carries this factor (f16 storage: 256)
        # manifest
        self.index: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.order: List[str] = []
        name, off, rank, shape = C.c_char_p(), C.c_int64(), C.c_int(), (C.c_int64 * 4)()
        for i in range(lib.cmpc_param_count(h)):
            _lib.call("cmpc_param_info", h, i, C.byref(name), C.byref(off), C.byref(rank), C.byref(shape))
            n = name.value.decode()
            self.index[n] = (off.value, tuple(int(shape[k]) for k in range(rank.value)))
            self.order.append(n)
        p, g, m, v, tot = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        _lib.call("cmpc_buffers", h, C.byref(p), C.byref(g), C.byref(m), C.byref(v), C.byref(tot))
        self.total = tot.value
        self.params, self.grads, self.m, self.v = (dev_tensor(x.value, (self.total,), 0, self.device) for x in (p, g, m, v))
        self._taps: Dict[str, torch.Tensor] = {}
        # non-trainable variables (CMPCv5_BiLSTM: batch-norm moving statistics)
        self.state_index: Dict[str, int] = {}
        cnt = C.c_int64()
        for i in range(lib.cmpc_state_count(h)):
            _lib.call("cmpc_state_info", h, i, C.byref(name), C.byref(cnt))
            self.state_index[name.value.decode()] = cnt.value

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.cmpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def step(self) -> int:
        s = C.c_int64()
        _lib.call("cmpc_get_step", self.h, C.byref(s))
        return s.value

    @step.setter
    def step(self, v: int):
        _lib.call("cmpc_set_step", self.h, int(v))

    def p(self, name: str) -> torch.Tensor:
        off, shape = self.index["text_objseg/" + name]
        return self.params[off: off + int(np.prod(shape))].view(shape)

    def g(self, name: str) -> torch.Tensor:
        off, shape = self.index["text_objseg/" + name]
        return self.grads[off: off + int(np.prod(shape))].view(shape)

    def load_state(self, named: Dict[str, torch.Tensor]):
        """Set every variable by its reference name (tf.train.Saver.restore), then repack the GEMM operands."""
        for name, (off, shape) in self.index.items():
            if name not in named:
                raise KeyError(f"missing parameter {name}")
            t = named[name]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
            self.params[off: off + t.numel()].copy_(t.reshape(-1).to(torch.float32))
        self.pack()

    def pack(self):
        _lib.call("cmpc_pack", self.h, self._stream())

    def get_state(self) -> Dict[str, np.ndarray]:
        out = {}
        for n, cnt in self.state_index.items():
            a = np.empty(cnt, dtype=np.float32)
            _lib.call("cmpc_get_state", self.h, n.encode(), a.ctypes.data_as(C.c_void_p), cnt)
            out[n] = a
        return out

    def set_state(self, named: Dict[str, np.ndarray]):
        for n, v in named.items():
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1))
            if n not in self.state_index or self.state_index[n] != a.size:
                raise KeyError(f"no state variable {n} of {a.size} elements")
            _lib.call("cmpc_set_state", self.h, n.encode(), a.ctypes.data_as(C.c_void_p), a.size)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        torch.cuda.synchronize(self.device)
        return {n: self.params[o: o + int(np.prod(s))].view(s).detach().cpu().clone() for n, (o, s) in self.index.items()}

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        """d cls_loss_all / d theta by variable name (the loss scale of f16 storage divided out)."""
        torch.cuda.synchronize(self.device)
        inv = 1.0 / self.loss_scale
        return {n: self.grads[o: o + int(np.prod(s))].view(s).detach().cpu() * inv for n, (o, s) in self.index.items()}

    # ---- the three per-step calls --------------------------------------------------------------
    def forward(self, words, seq_len, c3, c4, c5, target=None, feats_ready: Optional[torch.cuda.Event] = None, fetches=None, c2=None, im=None,
                levels_done: Optional[torch.cuda.Event] = None, feats_ready_lv=None):
        f = Feeds()
        f.words, f.seq_len = words.data_ptr(), seq_len.data_ptr()
        f.c3, f.c4, f.c5 = (c3.data_ptr() if c3 is not None else None), c4.data_ptr(), c5.data_ptr()
        f.c2 = c2.data_ptr() if c2 is not None else None
        f.im = im.data_ptr() if im is not None else None
        f.target_fine = target.data_ptr() if target is not None else None
        f.feats_ready = feats_ready.cuda_event if feats_ready is not None else None
        f.levels_done = levels_done.cuda_event if levels_done is not None else None
        if feats_ready_lv is not None:
            for i, ev in enumerate(feats_ready_lv):
                f.feats_ready_lv[i] = ev.cuda_event if ev is not None else None
        fe = None
        if fetches is not None:
            fe = Fetches()
            fe.pred, fe.up, fe.sigm = (t.data_ptr() if t is not None else None for t in fetches)
        _lib.call("cmpc_forward", self.h, C.byref(f), C.byref(fe) if fe is not None else None, self._stream())

    def backward(self):
        _lib.call("cmpc_backward", self.h, self._stream())

    def optimizer_step(self, gscale: float = 1.0) -> float:
        lr = C.c_double()
        _lib.call("cmpc_optimizer_step", self.h, float(gscale), self._stream(), C.byref(lr))
        return lr.value

    def optimizer_bucket(self, b: int, gscale: float = 1.0) -> float:
        """Adam + repack for gradient bucket b on the current stream (which first waits, on the device, for the bucket)."""
        lr = C.c_double()
        _lib.call("cmpc_optimizer_bucket", self.h, int(b), float(gscale), self._stream(), C.byref(lr))
        return lr.value

    @property
    def n_buckets(self) -> int:
        return self.lib.cmpc_grad_bucket_count(self.h)

    # ---- intermediates ------------------------------------------------------------------------
    def tap_names(self) -> List[str]:
        out, name = [], C.c_char_p()
        for i in range(self.lib.cmpc_tap_count(self.h)):
            _lib.call("cmpc_tap_name", self.h, i, C.byref(name))
            out.append(name.value.decode())
        return out

    def tap(self, name: str) -> torch.Tensor:
        """Zero-copy view of a named intermediate (valid for the handle's lifetime; contents = the last step's)."""
        t = self._taps.get(name)
        if t is None:
            ptr, dt, rank, shape = C.c_void_p(), C.c_int(), C.c_int(), (C.c_int64 * 4)()
            _lib.call("cmpc_tap", self.h, name.encode(), C.byref(ptr), C.byref(dt), C.byref(rank), C.byref(shape))
            t = dev_tensor(ptr.value, tuple(int(shape[k]) for k in range(rank.value)), dt.value, self.device)
            self._taps[name] = t
        return t

    def launch_count(self) -> int:
        n = C.c_int64()
        _lib.call("cmpc_launch_count", self.h, C.byref(n))
        return n.value

    def set_lanes(self, n: int):
        _lib.call("cmpc_set_lanes", self.h, int(n))

    def grad_buckets(self):
        """[[(offset, count), ...], ...]: the ranges of the flat gradient buffer in the order they become final during backward."""
        out, n, offs, cnts = [], C.c_int(), (C.c_int64 * 4)(), (C.c_int64 * 4)()
        for b in range(self.lib.cmpc_grad_bucket_count(self.h)):
            _lib.call("cmpc_grad_bucket", self.h, b, C.byref(n), C.byref(offs), C.byref(cnts))
            out.append([(int(offs[i]), int(cnts[i])) for i in range(n.value)])
        return out

    def bucket_wait(self, b: int, stream: torch.cuda.Stream):
        """`stream` waits (on the device) until gradient bucket b of the last backward() is final."""
        _lib.call("cmpc_grad_bucket_wait", self.h, b, C.c_void_p(stream.cuda_stream))

    def kernel_timing(self, enable: bool):
        _lib.call("cmpc_kernel_timing", self.h, int(enable))

    def kernel_timing_read(self):
        """(seconds, algorithmic flops, algorithmic bytes, launches) of the bf16 MFMA gemm_nt launches since enabled."""
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        _lib.call("cmpc_kernel_timing_read", self.h, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n))
        return ms.value * 1e-3, fl.value, by.value, n.value

    def phase_marks(self, enable: bool):
        _lib.call("cmpc_phase_marks", self.h, int(enable))

    def phase_marks_read(self):
        """[(name, ms since the first mark)] of the marks recorded since phase_marks(True)."""
        out, name, ms, i = [], C.c_char_p(), C.c_float(), 0
        while self.lib.cmpc_phase_marks_read(self.h, i, C.byref(name), C.byref(ms)) == 0:
            out.append((name.value.decode(), ms.value))
            i += 1
        return out
