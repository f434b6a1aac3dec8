"""world_size-2 data-parallel test on CPU (gloo): the product's gradient exchange (one all-reduce of
the flat buffer + 1/world folded into Adam) leaves both ranks with identical parameters, equal to a
single-process emulation that averages the two shards' gradients."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tests import util as U
    from tests.util import O
    D = importlib.import_module("cmpc-refseg_amd.dist")
    w, r, lr_ = D.init_from_env("gloo")
    assert (w, r) == (world, rank)
    cfg = U.tiny_cfg()
    hp = O.init_head_params(cfg, seed=100 + rank)         # deliberately different: rank 0's weights must win
    names = list(hp)
    flat = torch.cat([hp[n].reshape(-1) for n in names])
    D.broadcast_params_(flat, 0)
    off = 0
    for n in names:
        hp[n] = flat[off: off + hp[n].numel()].view(hp[n].shape).clone(); off += hp[n].numel()
    bp = O.init_backbone_params(cfg)
    words, im, sl, tgt = O.synth_batch(cfg, seed=rank)     # each rank its own shard
    feats = O.backbone_forward(bp, im, cfg)
    _, grads, _ = O.grads_of(hp, feats, words, sl, tgt, cfg)
    gflat = torch.cat([grads[n].reshape(-1) for n in names])
    local = gflat.clone()
    dist.all_reduce(gflat, op=dist.ReduceOp.SUM)
    scale = 1.0 / D.world_size()
    assert scale == 1.0 / world
    opt = O.TFAdam(hp)
    off = 0
    g = {}
    for n in names:
        g[n] = (gflat[off: off + hp[n].numel()] * scale).view(hp[n].shape); off += hp[n].numel()
    with torch.no_grad():
        opt.step(hp, g, O.poly_lr(0, cfg))
    # numpy arrays travel by value: a torch tensor would travel as a shared-memory handle that dies with this process
    q.put((rank, local.numpy(), torch.cat([hp[n].reshape(-1) for n in names]).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, local, params = q.get(timeout=300)
        res[r] = (torch.from_numpy(local), torch.from_numpy(params))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1])                       # replicas stay identical
    # single-process emulation
    sys.path.insert(0, ROOT)
    from tests import util as U
    from tests.util import O
    cfg = U.tiny_cfg()
    hp = O.init_head_params(cfg, seed=100)
    names = list(hp)
    avg = (res[0][0] + res[1][0]) / 2
    g, off = {}, 0
    for n in names:
        g[n] = avg[off: off + hp[n].numel()].view(hp[n].shape); off += hp[n].numel()
    opt = O.TFAdam(hp)
    with torch.no_grad():
        opt.step(hp, g, O.poly_lr(0, cfg))
    ref = torch.cat([hp[n].reshape(-1) for n in names])
    assert torch.allclose(res[0][1], ref, rtol=0, atol=1e-7)


@pytest.mark.gpu
def test_product_data_parallel_two_ranks_one_gpu(tmp_path):
    """The PRODUCT's data-parallel path under world_size 2: LSTM_model.enable_data_parallel (broadcast of rank 0's weights + repack)
    and train_step's bucketed gradient exchange (cmpc_grad_bucket_wait + all-reduce per bucket on the communication stream, 1/world in
    the Adam kernel), three steps on different shards.  Two fresh child processes (started before this process touches the GPU) share
    the test box's one GPU and exchange over gloo; RCCL over xGMI needs one GPU per rank and is what bench.py uses.  Checks: both
    replicas end with bit-identical parameters, and they equal the oracle's emulation (average of the two shards' gradients, TF-Adam)."""
    import subprocess
    import numpy as np
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path / f"rank{r}.npz"), "gloo", "tiny_f32"], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    for k in a.files:
        if k != "losses":
            assert np.array_equal(a[k], b[k]), k                     # replicas stay identical
    sys.path.insert(0, ROOT)
    from tests import util as U
    from tests.util import O
    torch.set_num_threads(8)
    cfg = U.tiny_cfg()
    hp, bp = O.init_head_params(cfg, seed=100), O.init_backbone_params(cfg)
    opt = O.TFAdam(hp)
    shards = []
    for r in range(2):
        words, im, sl, tgt = O.synth_batch(cfg, seed=r)
        shards.append((O.backbone_forward(bp, im, cfg), words, sl, tgt))
    for step in range(3):
        gs = []
        for r, (feats, words, sl, tgt) in enumerate(shards):
            scal, grads, _ = O.grads_of(hp, feats, words, sl, tgt, cfg)
            gs.append(grads)
            assert abs(scal["loss_all"] - float((a, b)[r]["losses"][step])) <= 2e-4 * abs(scal["loss_all"]), (step, r)
        avg = {n: (gs[0][n] + gs[1][n]) / 2 for n in hp}
        with torch.no_grad():
            opt.step(hp, avg, O.poly_lr(step, cfg))
    lr = cfg.start_lr
    for n, ref in hp.items():
        if "spa_graph_key" in n and n.endswith("biases"):
            continue          # exact-zero gradient here vs Adam-amplified rounding noise in the oracle
        d = float(np.abs(a[n.replace("/", "|")] - ref.numpy()).max())
        assert d <= 0.35 * lr, (n, d)


def _run_workers(tmp_path, world, backend, case, timeout=1200):
    import subprocess
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", CMPC_DP_SINGLE="1")
        out = tmp_path / f"{backend}_{case}_rank{r}.npz"
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(out), backend, case], env=env, cwd=ROOT), out))
    for p, _ in procs:
        assert p.wait(timeout=timeout) == 0
    import numpy as np
    return [np.load(o) for _, o in procs]


@pytest.mark.gpu
def test_rccl_world_of_one_equals_no_process_group(tmp_path):
    """RCCL readiness that one GPU can prove: a `nccl` (= RCCL) process group of ONE rank in a fresh child process, enable_data_parallel()
    + 3 train steps with every gradient bucket all-reduced IN PLACE on the engine-owned hipMalloc buffer (views through
    __cuda_array_interface__), on the communication stream, behind cmpc_grad_bucket_wait -- the memory, stream and event pattern RCCL
    meets at N = 8.  Parameters and losses must equal the run without any process group bit for bit (sum over one rank, gscale = 1).
    Scaling itself stays UNMEASURED until an 8-GPU record exists."""
    import numpy as np
    (a,) = _run_workers(tmp_path, 1, "nccl", "tiny_f32")
    (b,) = _run_workers(tmp_path, 1, "none", "tiny_f32")
    assert np.array_equal(a["losses"], b["losses"])
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    (c,) = _run_workers(tmp_path, 1, "nccl", "full_f16")
    (d,) = _run_workers(tmp_path, 1, "none", "full_f16")
    assert np.array_equal(c["losses"], d["losses"]) and np.array_equal(c["params"], d["params"]) and int(c["nonfinite"].sum()) == 0


@pytest.mark.gpu
def test_product_data_parallel_two_ranks_full_size_f16(tmp_path):
    """The 2-rank product path at BASELINE config 2's per-GPU shard (B = 8, 320x320, L = 20, f16 storage: 304 MB of fp32 gradients in
    five buckets, chunked all-reduces) -- two processes on the test box's one GPU over gloo.  Replicas end bit-identical; the per-step
    losses of each rank follow the oracle's emulation (both shards' gradients averaged, TF-Adam) -- step 2 and 3 only match if the
    exchanged update was the right one."""
    import numpy as np
    a, b = _run_workers(tmp_path, 2, "gloo", "full_f16", timeout=1500)
    assert np.array_equal(a["params"], b["params"]) and int(a["nonfinite"].sum()) == 0 and int(b["nonfinite"].sum()) == 0
    sys.path.insert(0, ROOT)
    from tests.util import O
    from bench import synth_batch
    torch.set_num_threads(16)
    cfg = O.Cfg(batch_size=8)
    hp, bp = O.init_head_params(cfg, seed=100), O.init_backbone_params(cfg)
    opt = O.TFAdam(hp)
    shards = []
    for r in range(2):
        w, im, sl, tg = map(torch.from_numpy, synth_batch(8, 20, 320, 320, cfg.vocab_size, 40 + r))
        with torch.no_grad():
            shards.append((O.backbone_forward(bp, im, cfg), w, sl, tg))
    for step in range(3):
        gs = []
        for r, (feats, w, sl, tg) in enumerate(shards):
            scal, grads, _ = O.grads_of(hp, feats, w, sl, tg, cfg)
            gs.append(grads)
            got = float((a, b)[r]["losses"][step])
            assert abs(scal["loss_all"] - got) <= 1e-2 * abs(scal["loss_all"]), (step, r, scal["loss_all"], got)
        with torch.no_grad():
            opt.step(hp, {n: (gs[0][n] + gs[1][n]) / 2 for n in hp}, O.poly_lr(step, cfg))


@pytest.mark.gpu
def test_conv5_data_parallel_two_ranks_one_gpu(tmp_path):
    """conv5=True under world_size 2 (gloo, both ranks on the test box's GPU): rank 0's head AND backbone weights are broadcast, the head's
    buckets and the backbone's gradient buffer are all-reduced, 1/world in both Adam calls.  Replicas bit-identical after 3 steps on
    different shards and equal to the oracle's emulation (average of the shards' gradients, TF-Adam over head + res3-res5 weights)."""
    import subprocess
    import numpy as np
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path / f"rank{r}.npz"), "gloo", "conv5_f32"], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    for k in a.files:
        if k != "losses":
            assert np.array_equal(a[k], b[k]), k
    sys.path.insert(0, ROOT)
    from tests.util import O
    torch.set_num_threads(8)
    cfg = O.Cfg(batch_size=2, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=1024, c4_dim=512, c3_dim=256, vocab_size=50, v_emb_dim=40, mlp_dim=24,
                rnn_size=40, glove_dim=12, parse_dim=20, backbone_width=32, backbone_blocks=(1, 2, 2, 1))
    hp, bp = O.init_head_params(cfg, seed=100), O.init_backbone_params(cfg, seed=4321)
    names_b = O.conv5_trainable(bp)
    opt, opt_b = O.TFAdam(hp), O.TFAdam({n: bp[n] for n in names_b})
    shards = [O.synth_batch(cfg, seed=r) for r in range(2)]
    for step in range(3):
        gh, gbb = [], []
        for r, (words, im, sl, tgt) in enumerate(shards):
            scal, g1, g2 = O.grads_of_conv5(hp, bp, torch.as_tensor(im), words, sl, tgt, cfg)
            gh.append(g1); gbb.append(g2)
            assert abs(scal["loss_all"] - float((a, b)[r]["losses"][step])) <= 3e-4 * abs(scal["loss_all"]), (step, r)
        with torch.no_grad():
            lr = O.poly_lr(step, cfg)
            opt.step(hp, {n: (gh[0][n] + gh[1][n]) / 2 for n in hp}, lr)
            opt_b.step({n: bp[n] for n in names_b}, {n: (gbb[0][n] + gbb[1][n]) / 2 for n in names_b}, lr)
    for n in names_b:
        d = np.abs(a["bb|" + n.replace("/", "|")] - bp[n].numpy()).reshape(-1)
        assert np.quantile(d[:200000], 0.99) <= 0.6 * cfg.start_lr and d.max() <= 6.5 * cfg.start_lr, (n, float(d.max()))

