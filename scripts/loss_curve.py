"""Loss curve of N train steps, product (fp32 / f16 storage) against the oracle's TF-Adam loop, tiny configuration.  usage: loss_curve.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from tests.util import O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.set_num_threads(8)
cfg = U.tiny_cfg(B=2)
hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
batches = [O.synth_batch(cfg, seed=s) for s in range(4)]
feats = [O.backbone_forward(bp, b[1], cfg) for b in batches]
hp_o = {k: v.clone() for k, v in hp.items()}
opt = O.TFAdam(hp_o)
ref = []
for step in range(n):
    w, im, sl, tg = batches[step % 4]
    ref.append(O.train_step(hp_o, opt, step, feats[step % 4], w, sl, tg, cfg)["loss_all"])
P = U.pkg()
for dtype in ("f32", "f16"):
    m = P.LSTM_model(head_params=hp, backbone_params=bp, **U.model_kwargs(cfg, dtype))
    got = []
    for step in range(n):
        w, im, sl, tg = batches[step % 4]
        got.append(float(m.train_step(w, im, tg, sl)[1]["loss_all"]))
    torch.cuda.synchronize()
    rel = [abs(a - b) / abs(b) for a, b in zip(got, ref)]
    print(dtype, "loss", ["%.1f" % x for x in got[::10]], "oracle", ["%.1f" % x for x in ref[::10]])
    print(dtype, "max rel diff over steps: first 10 %.2e, all %.2e; last-step %.2e" % (max(rel[:10]), max(rel), rel[-1]), flush=True)
