"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/cmpc.h
declares (with matching arity), the product's parameter manifest equals the oracle's, and the
host-side planning (operand packing tables, Adam segments) is self-consistent."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

from tests import util as U
from tests.util import O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "cmpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(cmpc_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",")])
    return protos


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    protos = _header_prototypes()
    assert len(protos) >= 40
    for name, nargs in protos.items():
        fn = getattr(lib, name)                      # raises if not exported
        if name == "cmpc_abi_version":
            continue
        sig = pkg._lib.SIGNATURES.get(name, pkg._lib.COUNTS.get(name))
        assert sig is not None, f"{name} missing from the ctypes binding"
        assert len(sig) == nargs, f"{name}: header has {nargs} args, binding {len(sig)}"
    assert set(pkg._lib.SIGNATURES) | set(pkg._lib.COUNTS) <= set(protos)
    assert lib.cmpc_abi_version() == pkg._lib.ABI_VERSION == 3
    assert isinstance(lib.cmpc_last_error(), bytes)
    hdr = open(os.path.join(ROOT, "include", "cmpc.h")).read()
    assert int(re.search(r"#define CMPC_ABI_VERSION (\d+)", hdr).group(1)) == pkg._lib.ABI_VERSION     # the binding refuses any other library


def test_bad_arguments_are_rejected_without_a_gpu(pkg):
    """Argument validation happens before any launch: a NULL / malformed GEMM returns CMPC_EINVAL."""
    lib = pkg._lib.load()
    a = pkg._lib.GemmNtArgs()
    assert lib.cmpc_gemm_nt(ctypes.byref(a), None) == -1
    assert b"gemm_nt" in lib.cmpc_last_error()
    t = pkg._lib.GemmTnArgs()
    assert lib.cmpc_gemm_tn(ctypes.byref(t), None) == -1


def test_manifest_matches_oracle(pkg):
    for cfg in (U.tiny_cfg(), O.Cfg()):
        hc = pkg.HeadCfg(batch_size=cfg.batch_size, num_steps=cfg.num_steps, vf_h=cfg.vf_h, vf_w=cfg.vf_w, H=cfg.H, W=cfg.W,
                         vf_dim=cfg.vf_dim, c4_dim=cfg.c4_dim, c3_dim=cfg.c3_dim, vocab_size=cfg.vocab_size,
                         v_emb_dim=cfg.v_emb_dim, mlp_dim=cfg.mlp_dim, rnn_size=cfg.rnn_size, glove_dim=cfg.glove_dim,
                         parse_dim=cfg.parse_dim)
        a = {n: (tuple(s), k, tuple(f)) for n, s, k, f in pkg.head_param_specs(hc)}
        b = {n: (tuple(s), k, tuple(f)) for n, s, k, f in O.head_param_specs(cfg)}
        assert a == b
    assert sum(int(np.prod(s)) for _, s, _, _ in pkg.head_param_specs(pkg.HeadCfg())) == 76055608


def test_product_init_equals_oracle_init(pkg):
    cfg = U.tiny_cfg()
    hc = pkg.HeadCfg(batch_size=2, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=256, c4_dim=128, c3_dim=64, vocab_size=50,
                     v_emb_dim=40, mlp_dim=24, rnn_size=40, glove_dim=12, parse_dim=20)
    a, b = pkg.init_head_params(hc), O.init_head_params(cfg)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    bb = importlib.import_module("cmpc-refseg_amd.backbone")
    pa, pb = bb.init_params(8, (1, 1, 2, 1)), O.init_backbone_params(cfg)
    assert set(pa) == set(pb)
    for k in pb:
        assert torch.equal(pa[k], pb[k]), k


def _emulate_pack(store, d, master):
    """numpy restatement of pack_kernel for one descriptor."""
    Kp, Np = (d.cols, d.rows) if d.transpose else (d.rows, d.cols)
    blk = np.zeros((Kp, Np), dtype=np.float32)
    for i in range(d.nks):
        for j in range(d.nns):
            ks, kl, kd = d.ks_src[i], d.ks_len[i], d.ks_dst[i]
            ns, nl, nd = d.ns_src[j], d.ns_len[j], d.ns_dst[j]
            rows = np.arange(ks, ks + kl)[:, None] * d.ld_src + np.arange(ns, ns + nl)[None, :]
            blk[kd:kd + kl, nd:nd + nl] = master[d.src_off + rows]
    return blk.T if d.transpose else blk


def test_pack_tables_cover_operands_without_overlap(pkg):
    cfg = pkg.HeadCfg(batch_size=2, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=256, c4_dim=128, c3_dim=64, vocab_size=50,
                      v_emb_dim=40, mlp_dim=24, rnn_size=40, glove_dim=12, parse_dim=20)
    st = pkg.ParamStore(cfg, "cpu", 1)
    used = np.zeros(st._arena_bytes, dtype=np.int32)
    for d in st._descs:
        esz = 4 if d.dst_dt == 0 else 2
        for r in range(d.rows):
            o = d.dst_off + r * d.ld_dst * esz
            used[o:o + d.cols * esz] += 1
        # every source index stays inside its parameter
        assert d.src_off >= 0 and d.nks >= 1 and d.nns >= 1
    assert used.max() == 1                          # no two descriptors write the same bytes
    # every operand is fully covered by its descriptors
    for key, op in st.ops.items():
        esz = 4 if op.dt == 0 else 2
        assert used[op.off: op.off + op.rows * op.ld * esz].min() == 1, key
    # forward operand of the fusion layer: K segments skip the language rows (folded into a per-sample bias)
    master = np.arange(st.total, dtype=np.float32)
    d = [x for x in st._descs if x.dst_off == st.ops["fus_c5.t"].off][0]
    blk = _emulate_pack(st, d, master)
    C, M = cfg.v_emb_dim, cfg.mlp_dim
    off = st.poff("fusion_c5/DW")
    assert blk.shape == (cfg.Mp, 2 * cfg.Cp + 64)
    assert blk[3, 5] == master[off + 5 * M + 3]                        # vis_la_sp rows
    assert blk[3, cfg.Cp + 5] == master[off + (C + 5) * M + 3]         # spa_graph rows
    assert blk[3, 2 * cfg.Cp + 2] == master[off + (3 * C + 2) * M + 3]  # spatial rows (2C + R + i)
    assert blk[M:, :].sum() == 0 and blk[:, C:cfg.Cp].sum() == 0      # padding is zero
    segs = np.frombuffer(st.segs_dev.numpy().tobytes(), dtype=np.dtype({"names": ["off", "count", "wd", "gm"], "formats": ["<i8", "<i4", "<f4", "<f4"],
                                         "offsets": [0, 8, 12, 16], "itemsize": ctypes.sizeof(pkg._lib.AdamSeg)}))
    assert int(segs["count"].sum()) == sum(int(np.prod(s)) for _, s, _, _ in st.specs)
    assert set(np.unique(segs["gm"])) == {1.0, 2.0} and set(np.unique(segs["wd"])) == {0.0, np.float32(0.0005)}


def _plan_only_handle(pkg, hc, dt):
    lib = pkg._lib.load()
    c = pkg._lib.EngineCfg()
    assert lib.cmpc_default_cfg(ctypes.byref(c)) == 0
    for k in ("batch_size", "num_steps", "vf_h", "vf_w", "H", "W", "vf_dim", "c4_dim", "c3_dim", "vocab_size", "v_emb_dim",
              "mlp_dim", "rnn_size", "glove_dim", "parse_dim"):
        setattr(c, k, getattr(hc, k))
    c.dtype, c.device = dt, -1                      # planning only: no GPU is touched
    h = ctypes.c_void_p()
    assert lib.cmpc_create(ctypes.byref(c), ctypes.byref(h)) == 0, lib.cmpc_last_error()
    return lib, h


@pytest.mark.parametrize("full", [False, True])
def test_engine_plan_equals_python_plan(pkg, full):
    """The whole-path handle (cmpc_create, C++) plans the SAME parameter manifest, packed operands and pack descriptors
    as the Python ParamStore the op-level tests use -- checked without a GPU through a planning-only handle."""
    hc = pkg.HeadCfg() if full else pkg.HeadCfg(batch_size=2, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=256, c4_dim=128,
                                                 c3_dim=64, vocab_size=50, v_emb_dim=40, mlp_dim=24, rnn_size=40, glove_dim=12, parse_dim=20)
    for dt in (0, 1):
        lib, h = _plan_only_handle(pkg, hc, dt)
        st = pkg.ParamStore(hc, "cpu", dt)
        n = lib.cmpc_param_count(h)
        assert n == len(st.specs)
        name, off, rank, shape = ctypes.c_char_p(), ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i, (pn, pshape, _, _) in enumerate(st.specs):
            assert lib.cmpc_param_info(h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(rank), ctypes.byref(shape)) == 0
            assert name.value.decode() == pn and off.value == st.index[pn][0]
            assert tuple(shape[k] for k in range(rank.value)) == tuple(pshape)
        descs, nd, ab, wb, s0 = ctypes.POINTER(pkg._lib.PackDesc)(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        assert lib.cmpc_plan_info(h, ctypes.byref(descs), ctypes.byref(nd), ctypes.byref(ab), ctypes.byref(wb), ctypes.byref(s0)) == 0
        assert nd.value == len(st._descs) and ab.value == st._arena_bytes and s0.value == st.stage0_ndesc and wb.value > 0
        for i, d in enumerate(st._descs):
            assert bytes(descs[i]) == bytes(d), i
        bo, odt, rows, ld = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for key, op in st.ops.items():
            assert lib.cmpc_operand_info(h, key.encode(), ctypes.byref(bo), ctypes.byref(odt), ctypes.byref(rows), ctypes.byref(ld)) == 0, key
            assert (bo.value, odt.value, rows.value, ld.value) == (op.off, op.dt, op.rows, op.ld), key
        # compute entry points refuse a planning-only handle, and nothing crashes
        assert lib.cmpc_backward(h, None) == -1
        assert lib.cmpc_destroy(h) == 0


def test_engine_rejects_bad_configs(pkg):
    lib = pkg._lib.load()
    c = pkg._lib.EngineCfg()
    lib.cmpc_default_cfg(ctypes.byref(c))
    h = ctypes.c_void_p()
    c.device = -1
    for field, bad in (("vf_dim", 2000), ("rnn_size", 900), ("num_steps", 0), ("num_steps", 65), ("dtype", 7), ("n_lanes", 4), ("v_emb_dim", 4000)):
        old = getattr(c, field)
        setattr(c, field, bad)
        assert lib.cmpc_create(ctypes.byref(c), ctypes.byref(h)) == -1, field
        assert lib.cmpc_last_error()
        setattr(c, field, old)
    assert lib.cmpc_create(None, ctypes.byref(h)) == -1
    assert lib.cmpc_tap(None, b"up", None, None, None, None) == -1
