"""f16-storage mean-IoU delta of BASELINE config 4 (CMPCv5_BiLSTM_HSV, 512x512, L=25, B=2) against the oracle over several seeds.  usage: v5_seeds.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from oracle import cmpc_v5_torch as V
from bench import synth_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.set_num_threads(16)
B = 2
cfg = V.Cfg(batch_size=B, num_steps=25, vf_h=64, vf_w=64, H=512, W=512, hsv=True)
hp, bp, bn = V.init_head_params(cfg), V.init_backbone_params(cfg), V.init_bn_state(cfg)
P = U.pkg()
m = P.get_segmentation_model("CMPCv5_BiLSTM_HSV_model", batch_size=B, num_steps=25, vf_h=64, vf_w=64, H=512, W=512, mode="train", dtype="f16", head_params=hp, backbone_params=bp)
out = []
for seed in range(30, 30 + n):
    w, im, sl, tg = map(torch.from_numpy, synth_batch(B, 25, 512, 512, cfg.vocab_size, seed))
    with torch.no_grad():
        taps = V.head_forward(hp, bn, V.backbone_taps(bp, im, cfg), w, sl, cfg, im=im)
        ref = V.losses(hp, taps, tg, cfg)
        o = m.head(m.features(im), w, sl, tg, im=im)
    torch.cuda.synchronize()
    up = o["up"].float().cpu()
    d = abs(float(o["mIoU"]) - float(ref["mIoU"]))
    out.append(d)
    print(f"seed {seed}: dIoU {d:.2e} flipped {int(((up > 0) != (taps['up'] > 0)).sum())} of {up.numel()} oracle mIoU {float(ref['mIoU']):.4f}", flush=True)
print("max", max(out), "mean", sum(out) / len(out))
