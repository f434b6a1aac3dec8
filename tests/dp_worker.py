"""One rank of the data-parallel GPU tests (tests/test_dp_gloo.py): builds the product model on cuda:0, joins a process group, runs
3 train steps on its own shard through LSTM_model.train_step (bucketed all-reduce included) and writes its parameters.
    dp_worker.py <out.npz> [backend=gloo|nccl|none] [case=tiny_f32|full_f16]
gloo: both ranks share the one GPU of the test box (the measured configuration is one rank per GPU over RCCL); nccl: a world of ONE rank
(CMPC_DP_SINGLE=1) -- RCCL then meets the engine-owned buffers, the communication stream and the bucket events it will see at N = 8;
none: no process group at all (the reference run the nccl world-1 result must equal bit for bit)."""
import importlib, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    case = sys.argv[3] if len(sys.argv) > 3 else "tiny_f32"
    from tests import util as U
    from tests.util import O
    D = importlib.import_module("cmpc-refseg_amd.dist")
    world, rank = 1, 0
    if backend != "none":
        world, rank, _ = D.init_from_env(backend, device=torch.device("cuda:0"))
        assert D.is_initialized()
    P = U.pkg()
    if case == "conv5_f32":                        # conv5=True: backbone gradients are exchanged too (width-32 backbone: the library's conv path)
        cfg = O.Cfg(batch_size=2, num_steps=6, vf_h=8, vf_w=8, H=64, W=64, vf_dim=1024, c4_dim=512, c3_dim=256, vocab_size=50, v_emb_dim=40, mlp_dim=24,
                    rnn_size=40, glove_dim=12, parse_dim=20, backbone_width=32, backbone_blocks=(1, 2, 2, 1))
        hp, bp = O.init_head_params(cfg, seed=100 + rank), O.init_backbone_params(cfg, seed=4321)
        for n in O.conv5_trainable(bp):                # same frozen part on every rank (one checkpoint), different TRAINED weights: rank 0's must win
            bp[n] = bp[n] * (1.0 + 0.05 * rank)
        m = P.LSTM_model(head_params=hp, backbone_params=bp, conv5=True, **U.model_kwargs(cfg, "f32"))
        words, im, sl, tgt = O.synth_batch(cfg, seed=rank)
    elif case == "tiny_f32":
        cfg = U.tiny_cfg()
        hp, bp = O.init_head_params(cfg, seed=100 + rank), O.init_backbone_params(cfg)     # different weights: rank 0's must win
        m = P.LSTM_model(head_params=hp, backbone_params=bp, **U.model_kwargs(cfg, "f32"))
        words, im, sl, tgt = O.synth_batch(cfg, seed=rank)                                  # each rank its own shard
    else:                                           # BASELINE config 2's per-GPU shard: B = 8, 320x320, L = 20, f16 storage
        from bench import synth_batch
        cfg = O.Cfg(batch_size=8)
        hp, bp = O.init_head_params(cfg, seed=100 + rank), O.init_backbone_params(cfg)
        m = P.LSTM_model(batch_size=8, mode="train", dtype="f16", head_params=hp, backbone_params=bp)
        words, im, sl, tgt = map(torch.from_numpy, synth_batch(8, 20, 320, 320, cfg.vocab_size, 40 + rank))
    assert m.enable_data_parallel() == world and m.dp_on == (backend != "none")
    losses = []
    for _ in range(3):
        _, scal = m.train_step(words, im, tgt, sl)
        losses.append(float(scal["loss_all"]))
    torch.cuda.synchronize()
    if case in ("tiny_f32", "conv5_f32"):
        sd = dict(m.state_dict())
        if case == "conv5_f32":
            sd.update({"bb/" + k: v for k, v in m.bb_trainer.named_weights().items()})
        np.savez(out, losses=np.asarray(losses), **{k.replace("/", "|"): v.numpy() for k, v in sd.items()})
    else:
        np.savez(out, losses=np.asarray(losses), params=m.eng.params.cpu().numpy(), nonfinite=m.eng.tap("grad_nonfinite").cpu().numpy())
    if backend != "none":
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
