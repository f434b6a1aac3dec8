"""One rank of the 2-rank data-parallel test (tests/test_dp_gloo.py::test_product_data_parallel_two_ranks_one_gpu): builds the product
model on cuda:0, joins a gloo group (both ranks share the one GPU of the test box; the measured configuration is one rank per GPU over
RCCL), runs 3 train steps on its own shard through LSTM_model.train_step (bucketed all-reduce included) and writes its parameters."""
import importlib, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    from tests import util as U
    from tests.util import O
    D = importlib.import_module("cmpc-refseg_amd.dist")
    world, rank, _ = D.init_from_env("gloo")
    P = U.pkg()
    cfg = U.tiny_cfg()
    hp, bp = O.init_head_params(cfg, seed=100 + rank), O.init_backbone_params(cfg)     # different weights: rank 0's must win
    m = P.LSTM_model(head_params=hp, backbone_params=bp, **U.model_kwargs(cfg, "f32"))
    assert m.enable_data_parallel() == world
    words, im, sl, tgt = O.synth_batch(cfg, seed=rank)                                  # each rank its own shard
    losses = []
    for _ in range(3):
        _, scal = m.train_step(words, im, tgt, sl)
        losses.append(float(scal["loss_all"]))
    sd = m.state_dict()
    np.savez(out, losses=np.asarray(losses), **{k.replace("/", "|"): v.numpy() for k, v in sd.items()})
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
