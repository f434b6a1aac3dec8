"""The whole-path C ABI (include/cmpc.h: cmpc_create / cmpc_set_weights / cmpc_pack / cmpc_forward / cmpc_backward /
cmpc_optimizer_step / cmpc_tap / cmpc_destroy) driven directly through ctypes -- no LSTM_model, no ops.py, no engine.py:
what a non-Python host (the cgo / JNI stub of INTEGRATION.md) would do.  torch is used for device memory only.
Checked against the oracle on the tiny seeded case: fetches, loss scalars, gradients, one optimizer step."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import util as U
from tests.util import O

pytestmark = pytest.mark.gpu


def _cfg_struct(L, cfg, dt, lanes):
    c = L.EngineCfg()
    assert L.load().cmpc_default_cfg(C.byref(c)) == 0
    for k in ("batch_size", "num_steps", "vf_h", "vf_w", "H", "W", "vf_dim", "c4_dim", "c3_dim", "vocab_size", "v_emb_dim",
              "mlp_dim", "rnn_size", "glove_dim", "parse_dim", "start_lr", "lr_decay_step", "weight_decay"):
        setattr(c, k, getattr(cfg, k))
    c.dtype, c.n_lanes, c.device = dt, lanes, 0
    return c


def _tap(lib, h, name, dev):
    from importlib import import_module
    E = import_module("cmpc-refseg_amd.engine")
    ptr, dt, rank, shape = C.c_void_p(), C.c_int(), C.c_int(), (C.c_int64 * 4)()
    assert lib.cmpc_tap(h, name.encode(), C.byref(ptr), C.byref(dt), C.byref(rank), C.byref(shape)) == 0, lib.cmpc_last_error()
    return E.dev_tensor(ptr.value, tuple(shape[k] for k in range(rank.value)), dt.value, dev)      # zero-copy view


@pytest.mark.parametrize("lanes", [3, 1])
def test_c_abi_forward_backward_optimizer(lanes):
    L = U.pkg()._lib
    lib = L.load()
    dev = torch.device("cuda:0")
    torch.set_num_threads(8)
    cfg = U.tiny_cfg()
    hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
    words, im, sl, tgt = O.synth_batch(cfg)
    feats = O.backbone_forward(bp, im, cfg)
    scal, grads, taps = O.grads_of(hp, feats, words, sl, tgt, cfg)

    h = C.c_void_p()
    c = _cfg_struct(L, cfg, 0, lanes)
    assert lib.cmpc_create(C.byref(c), C.byref(h)) == 0, lib.cmpc_last_error()
    try:
        # tf.train.Saver.restore by variable name, from host memory
        n = lib.cmpc_param_count(h)
        assert n == len(hp)
        name, off, rank, shape = C.c_char_p(), C.c_int64(), C.c_int(), (C.c_int64 * 4)()
        for i in range(n):
            assert lib.cmpc_param_info(h, i, C.byref(name), C.byref(off), C.byref(rank), C.byref(shape)) == 0
            w = np.ascontiguousarray(hp[name.value.decode()].numpy(), dtype=np.float32)
            assert tuple(shape[k] for k in range(rank.value)) == w.shape
            assert lib.cmpc_set_weights(h, name.value, w.ctypes.data_as(C.c_void_p), w.size) == 0, lib.cmpc_last_error()
        assert lib.cmpc_set_weights(h, b"text_objseg/nope", None, 1) == -1
        assert lib.cmpc_set_weights(h, b"text_objseg/score/biases", None, 7) == -1          # wrong element count
        st = torch.cuda.current_stream(dev).cuda_stream
        assert lib.cmpc_pack(h, C.c_void_p(st)) == 0
        back = np.empty(4, dtype=np.float32)
        assert lib.cmpc_get_weights(h, b"text_objseg/words_parse_2/biases", back.ctypes.data_as(C.c_void_p), 4) == 0
        assert np.array_equal(back, hp["text_objseg/words_parse_2/biases"].numpy())

        d = {k: v.to(dev).contiguous() for k, v in dict(words=words.int(), sl=sl.int(), tgt=tgt.float(), c3=feats[0], c4=feats[1], c5=feats[2]).items()}
        f = L.Feeds()
        f.words, f.seq_len, f.c3, f.c4, f.c5 = (d[k].data_ptr() for k in ("words", "sl", "c3", "c4", "c5"))
        f.target_fine = d["tgt"].data_ptr()
        B, H, W, hh, ww = cfg.batch_size, cfg.H, cfg.W, cfg.vf_h, cfg.vf_w
        pred, up, sigm = torch.empty(B, hh, ww, 1, device=dev), torch.empty(B, H, W, 1, device=dev), torch.empty(B, H, W, 1, device=dev)
        fe = L.Fetches()
        fe.pred, fe.up, fe.sigm = pred.data_ptr(), up.data_ptr(), sigm.data_ptr()
        assert lib.cmpc_forward(h, C.byref(f), C.byref(fe), C.c_void_p(st)) == 0, lib.cmpc_last_error()
        assert lib.cmpc_backward(h, C.c_void_p(st)) == 0, lib.cmpc_last_error()
        torch.cuda.synchronize()
        assert U.rel_err(up.cpu(), taps["up"]) < 2e-5 and U.rel_err(pred.cpu(), taps["pred"]) < 2e-5
        assert torch.allclose(sigm.cpu(), torch.sigmoid(taps["up"]), atol=1e-6)
        s = _tap(lib, h, "scalars", dev).cpu()
        for i, k in enumerate(("loss_all", "loss_c3", "loss_c4", "loss_c5", "loss_last")):
            assert abs(float(s[i]) - scal[k]) <= 1e-5 * abs(scal[k]), k
        assert abs(float(s[5]) - scal["mIoU"]) <= 1e-6
        wp = _tap(lib, h, "words_parse", dev).cpu().view(B, 1, cfg.num_steps, 4)
        assert U.rel_err(wp, taps["words_parse"]) < 2e-5
        # tf.gradients(cost): the flat gradient buffer holds d cls_loss_all / d theta
        p_, g_, tot = C.c_void_p(), C.c_void_p(), C.c_int64()
        assert lib.cmpc_buffers(h, C.byref(p_), C.byref(g_), None, None, C.byref(tot)) == 0
        from importlib import import_module
        E = import_module("cmpc-refseg_amd.engine")
        gflat = E.dev_tensor(g_.value, (tot.value,), 0, dev).cpu()
        flags = {k: fl for k, _, _, fl in O.head_param_specs(cfg)}
        for i in range(n):
            lib.cmpc_param_info(h, i, C.byref(name), C.byref(off), C.byref(rank), C.byref(shape))
            nm = name.value.decode()
            ref = grads[nm] / (2.0 if "x2" in flags[nm] else 1.0)
            if "reg" in flags[nm]:
                ref = ref - cfg.weight_decay * hp[nm]
            got = gflat[off.value: off.value + ref.numel()].view(ref.shape)
            if "spa_graph_key" in nm and nm.endswith("biases"):
                assert float(got.abs().max()) == 0.0
                continue
            tol = 2e-3 if ("spa_graph_trans2" in nm and nm.endswith("biases")) else 2e-4
            assert U.rel_err(got, ref) < tol, nm
        # one TF-Adam step; the second forward runs on the repacked operands
        lr = C.c_double()
        assert lib.cmpc_optimizer_step(h, 1.0, C.c_void_p(st), C.byref(lr)) == 0, lib.cmpc_last_error()
        assert abs(lr.value - O.poly_lr(0, cfg)) < 1e-12
        step = C.c_int64()
        assert lib.cmpc_get_step(h, C.byref(step)) == 0 and step.value == 1
        hp2 = {k: v.clone() for k, v in hp.items()}
        opt = O.TFAdam(hp2)
        ref = O.train_step(hp2, opt, 0, feats, words, sl, tgt, cfg)
        assert lib.cmpc_forward(h, C.byref(f), None, C.c_void_p(st)) == 0, lib.cmpc_last_error()
        torch.cuda.synchronize()
        with torch.no_grad():
            t2 = O.head_forward(hp2, feats, words, sl, cfg)
        assert U.rel_err(_tap(lib, h, "up", dev).cpu(), t2["up"]) < 5e-3       # Adam's sign-like first step amplifies fp32 noise
        nl = C.c_int64()
        assert lib.cmpc_launch_count(h, C.byref(nl)) == 0 and nl.value > 0
        # inference call without a target, then backward must refuse
        f.target_fine = None
        assert lib.cmpc_forward(h, C.byref(f), None, C.c_void_p(st)) == 0
        assert lib.cmpc_backward(h, C.c_void_p(st)) == -1
        torch.cuda.synchronize()
    finally:
        assert lib.cmpc_destroy(h) == 0
