"""TEST / BENCH SCAFFOLDING (not product code): op-level launch helpers over the C ABI (include/cmpc.h) -- thin ctypes wrappers of
cmpc_gemm_nt / cmpc_gemm_tn / cmpc_cast / cmpc_act_bwd on torch device tensors, for the kernel-level parity tests and the
micro-benchmarks under scripts/.  The product path does not go through them:
LSTM_model drives the whole head through ONE handle (cmpc_create / cmpc_forward / cmpc_backward / cmpc_optimizer_step,
csrc/engine.hip), which chains the stage kernels in C++.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

import importlib

_lib = importlib.import_module("cmpc-refseg_amd")._lib
ACT_NONE, DT_BF16, DT_F16, DT_F32, GemmNtArgs, GemmTnArgs = (_lib.ACT_NONE, _lib.DT_BF16, _lib.DT_F16, _lib.DT_F32, _lib.GemmNtArgs,
                                                             _lib.GemmTnArgs)

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


def _nt_args(dt, segs, C, ldc, M, N, n_valid=None, bias=None, act=ACT_NONE, alpha=1.0, accumulate=False):
    a = GemmNtArgs()
    a.dtype, a.nseg = dt, len(segs)
    for i, s in enumerate(segs):
        A, lda, Bt, ldb, K = s[:5]
        a.A[i], a.lda[i], a.Bt[i], a.ldb[i], a.K[i] = _p(A), lda, _p(Bt), ldb, K
    a.C, a.ldc = _p(C), ldc
    a.M, a.N, a.n_valid, a.batch = M, N, (N if n_valid is None else n_valid), 1
    a.bias = _p(bias)
    a.act, a.alpha, a.accumulate = act, alpha, int(accumulate)
    return a


def gemm_nt_pair(dt, j1, j2, M, N):
    """Two independent products (dicts: segs, C, ldc, n_valid, bias, act) in one launch where possible: cmpc_gemm_nt_pair."""
    a, b = (_nt_args(dt, j["segs"], j["C"], j["ldc"], M, N, j.get("n_valid"), j.get("bias"), j.get("act", ACT_NONE)) for j in (j1, j2))
    _lib.call("cmpc_gemm_nt_pair", ctypes.byref(a), ctypes.byref(b), _st())


def lowrank_nn(dt, A, lda, sA, Bk, ldb, sB, C, ldc, sC, M, N, Kv, n_valid=None, batch=1, alpha=1.0, accumulate=False):
    """C[b][m, n] (+)= alpha * sum_{k < Kv} A[b][m, k] Bk[b][k, n] (16-bit storage, Kv <= 24, Bk k-major): cmpc_lowrank_nn."""
    _lib.call("cmpc_lowrank_nn", dt, _p(A), lda, sA, _p(Bk), ldb, sB, _p(C), ldc, sC, M, N, N if n_valid is None else n_valid, Kv, batch,
              float(alpha), int(accumulate), _st())


def gemm_tn(dt, A, lda, Ka, D, ldd, Nd, out, ldo, R, Kv, Nv, offs=((0, 0, 0),), nb2=1, a_bs=0, d_bs=0, o_bs=0,
            alpha=1.0, rsplit=None, wg=None):
    """out[k, n] += alpha * sum_r A[r, k] D[r, n]; offs: (a_off, d_off, o_off) per inner batch (elements).
    wg: an object with a `deferred` list -> the product is appended there (args, A, D) for one cmpc_gemm_tn_grouped launch."""
    if wg is not None:
        wg = wg.deferred
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
