import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from tests.util import O
cfg = U.tiny_cfg()
P = U.pkg()
hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
words, im, sl, tgt = O.synth_batch(cfg)
m = P.LSTM_model(head_params=hp, backbone_params=bp, **U.model_kwargs(cfg, sys.argv[1] if len(sys.argv) > 1 else "f32"))
torch.cuda.synchronize(); print("model ok", flush=True)
for step in range(2):
    f = m.features(im); torch.cuda.synchronize(); print("features ok", [tuple(x.shape) for x in f], flush=True)
    o = m.loss_and_grads(f, words, tgt, sl); torch.cuda.synchronize(); print("loss_and_grads ok", float(o["loss_all"].detach()), flush=True)
    m.store.adam_step(1.0); torch.cuda.synchronize(); print("adam ok", flush=True)
print("done")
