"""The one part of the hot path the reference can still execute here -- its pure-NumPy helpers
(util/processing_tools.py:5-17,24-42, util/eval_tools.py:31-35) -- pins the oracle AND the product:
tests/golden/ref_processing_tools.npz holds inputs and outputs of those functions as run from
/root/reference by tests/golden/make_ref_fixtures.py.  Pinned by this file: SURVEY 8a row D (spatial grid) and
the metric half of row S (mean IoU, I/U counters).  Everything else on the path needs TensorFlow and stays
"parity unpinned" (DESIGN.md section 2)."""
import importlib
import os

import numpy as np
import pytest
import torch

from tests import util as U
from tests.util import O
from oracle import cmpc_numpy as NP

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_processing_tools.npz"))
GRIDS = [k[5:] for k in G.files if k.startswith("grid/")]


def _nhw(key):
    return tuple(int(v) for v in key.split("x"))


@pytest.mark.parametrize("key", GRIDS)
def test_oracle_spatial_grid_is_the_references(key):
    n, h, w = _nhw(key)
    ref = G["grid/" + key]
    assert ref.dtype == np.float32 and ref.shape == (n, h, w, 8)
    assert np.array_equal(O.generate_spatial_batch(n, h, w).numpy(), ref)              # bit-exact (float32)
    assert np.array_equal(NP.spatial_grid(n, h, w).astype(np.float32), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("key", GRIDS)
def test_engine_spatial_grid_is_the_references(key):
    """The grid the PRODUCT uses: the K-segment every Mutan / fusion GEMM reads is built inside cmpc_create (csrc/engine.hip) and exposed as
    cmpc_tap("spatial") -- f32 mode, a handle per fixture size, compared bit for bit with the reference's generate_spatial_batch."""
    n, h, w = _nhw(key)
    P = U.pkg()
    E = importlib.import_module("cmpc-refseg_amd.engine")
    cfg = P.HeadCfg(batch_size=n, num_steps=4, vf_h=h, vf_w=w, H=h * 8, W=w * 8, vf_dim=64, c4_dim=64, c3_dim=64, vocab_size=10,
                    v_emb_dim=16, mlp_dim=8, rnn_size=16, glove_dim=4, parse_dim=4)
    eng = E.Engine(cfg, P._lib.DT_F32, torch.device("cuda:0"))
    sp = eng.tap("spatial").cpu()
    assert tuple(sp.shape) == (n * h * w, 64) and sp.dtype == torch.float32 and torch.all(sp[:, 8:] == 0)
    assert np.array_equal(sp[:, :8].reshape(n, h, w, 8).numpy(), G["grid/" + key])      # bit-exact, every sample of the batch
    eng.close()


def test_oracle_and_host_metrics_are_the_references():
    H = importlib.import_module("cmpc-refseg_amd.hostutil")
    for i in range(int(G["iou/n"])):
        scores, labels = G[f"iou/{i}/scores"], G[f"iou/{i}/labels"]
        I, U_ = (int(v) for v in G[f"iou/{i}/IU"])
        assert H.compute_mask_IU(scores > 0, labels != 0) == (I, U_)
        ev = H.SegEval()
        ev.add(scores > 0, labels != 0)
        assert ev.result()["mean_IoU"] == pytest.approx(float(G[f"iou/{i}/meanIoU"]), abs=1e-15)
        # the oracle's in-graph metric (CMPC_model.py:486-490) on a batch of one
        up = torch.from_numpy(scores).view(1, *scores.shape, 1)
        tg = torch.from_numpy(labels).view(1, *labels.shape, 1)
        taps = {k: up for k in ("up", "up_c3", "up_c4", "up_c5")}
        cfg = O.Cfg(batch_size=1)
        pred, lab = taps["up"] > 0, tg != 0
        miou = float(((pred & lab).sum().double() / (pred | lab).sum().double()))
        assert miou == pytest.approx(float(G[f"iou/{i}/meanIoU"]), abs=1e-15)


def test_oracle_batch_miou_is_mean_of_reference_per_image():
    up, tg = torch.from_numpy(G["batch/up"]), torch.from_numpy(G["batch/target"])
    cfg = U.tiny_cfg(B=up.shape[0])
    hp = O.init_head_params(cfg)
    out = O.losses(hp, {k: up for k in ("up", "up_c3", "up_c4", "up_c5")}, tg, cfg)
    assert float(out["mIoU"]) == pytest.approx(float(G["batch/meanIoU_per_image"].mean()), abs=1e-12)


@pytest.mark.gpu
def test_product_iu_counters_are_the_references():
    """cmpc_upsample_fwd's intersection / union counters (what LSTM_model reports as mean_IOU) on the reference's
    inputs: with h = H the legacy bilinear resize is the identity, so `up` is the fixture's score map itself."""
    P = U.pkg()
    P._lib.load()
    ops = importlib.import_module("tests.opwrap")
    dev = torch.device("cuda:0")
    up_in, tg = torch.from_numpy(G["batch/up"]).to(dev), torch.from_numpy(G["batch/target"]).to(dev)
    B, H, W, _ = up_in.shape
    up = torch.empty_like(up_in); sigm = torch.empty_like(up_in)
    loss = torch.zeros(B, device=dev); iu = torch.zeros(2, B, dtype=torch.int32, device=dev)
    P._lib.call("cmpc_upsample_fwd", up_in.data_ptr(), up.data_ptr(), sigm.data_ptr(), tg.data_ptr(), loss.data_ptr(),
                iu[0].data_ptr(), iu[1].data_ptr(), B, H, W, H, W, ops._st())
    torch.cuda.synchronize()
    assert torch.equal(up, up_in)
    got = (iu[0].double() / iu[1].double()).cpu().numpy()
    assert np.array_equal(got, G["batch/meanIoU_per_image"])
    for i in range(int(G["iou/n"])):
        s, l = torch.from_numpy(G[f"iou/{i}/scores"]).to(dev), torch.from_numpy(G[f"iou/{i}/labels"]).to(dev)
        h, w = s.shape
        up = torch.empty_like(s); iu = torch.zeros(2, 1, dtype=torch.int32, device=dev); loss = torch.zeros(1, device=dev)
        P._lib.call("cmpc_upsample_fwd", s.data_ptr(), up.data_ptr(), None, l.data_ptr(), loss.data_ptr(),
                    iu[0].data_ptr(), iu[1].data_ptr(), 1, h, w, h, w, ops._st())
        torch.cuda.synchronize()
        assert [int(iu[0, 0]), int(iu[1, 0])] == [int(v) for v in G[f"iou/{i}/IU"]], i
