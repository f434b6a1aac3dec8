"""GPU debugging aid: run the tiny config through the HIP head and the oracle, print per-tap and
per-parameter-gradient errors.  usage: python scripts/debug_parity.py [f32|bf16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from tests.util import O

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = U.tiny_cfg()
P = U.pkg()
hp = O.init_head_params(cfg)
bp = O.init_backbone_params(cfg)
words, im, sl, tgt = O.synth_batch(cfg)
feats = O.backbone_forward(bp, im, cfg)
scal, grads, taps = O.grads_of(hp, feats, words, sl, tgt, cfg)
m = P.LSTM_model(head_params=hp, backbone_params=bp, **U.model_kwargs(cfg, dtype))
feats_dev = [f.to(m.device) for f in feats]
o = m.loss_and_grads(feats_dev, words, tgt, sl)
torch.cuda.synchronize()
pt = U.product_taps_as_oracle(o, cfg)
print("== taps (rel err vs oracle)")
for k in taps:
    if k in pt:
        print(f"{k:18s} {U.rel_err(pt[k], taps[k]):.3e}")
print("== scalars")
for k in ("loss_c5", "loss_c4", "loss_c3", "loss_last", "loss_all", "mIoU"):
    print(k, float(o[k]), scal[k])
print("== grads (rel err), x2 on biases applied in the oracle; product applies it in Adam")
g = m.store.grad_dict()
flags = {n: f for n, _, _, f in O.head_param_specs(cfg)}
bad = []
for n in grads:
    ref = grads[n] / (2.0 if "x2" in flags[n] else 1.0)
    if "reg" in flags[n]:
        ref = ref - cfg.weight_decay * hp[n]
    e = U.rel_err(g[n], ref)
    bad.append((e, n, float(ref.abs().max())))
for e, n, mag in sorted(bad, reverse=True)[:60]:
    print(f"{e:.3e}  {n:60s} |ref|max={mag:.3e}")
