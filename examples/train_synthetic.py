"""The reference's train / snapshot / evaluate loop (trainval_model.py:83-142, test.py:267-330) on synthetic data, with this package in
place of the TensorFlow model: get_segmentation_model -> train_step -> Saver.save -> latest_checkpoint / restore -> forward -> IoU (+ DenseCRF).
usage: python examples/train_synthetic.py [--model CMPC_model] [--iters 40] [--batch 4] [--out /tmp/cmpc_ckpt]"""
import argparse, importlib, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="CMPC_model")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--out", default="/tmp/cmpc_ckpt")
args = ap.parse_args()

pkg = importlib.import_module("cmpc-refseg_amd")
CK = importlib.import_module("cmpc-refseg_amd.checkpoint")
HU = importlib.import_module("cmpc-refseg_amd.hostutil")
v5 = args.model.startswith("CMPCv5")
H = W = 512 if v5 else 320
T = 25 if v5 else 20
kw = dict(vf_h=H // 8, vf_w=W // 8) if v5 else {}
model = pkg.get_segmentation_model(args.model, mode="train", batch_size=args.batch, num_steps=T, H=H, W=W, **kw)
batches = [synth_batch(args.batch, T, H, W, model.cfg.vocab_size, seed) for seed in range(4)]          # a 4-batch "data set"
saver = CK.Saver(max_to_keep=2, fmt="tf")
os.makedirs(args.out, exist_ok=True)
for it in range(args.iters):
    words, im, seq_len, mask = batches[it % 4]
    step, scal = model.train_step(words, im, mask, seq_len)
    if (it + 1) % 10 == 0:
        print("iter %d  loss %.1f  lr %.2e  mean IoU %.4f" % (step, float(scal["loss_all"]), scal["learning_rate"], float(scal["mean_IOU"])), flush=True)
path = saver.save(model, os.path.join(args.out, args.model))
print("snapshot:", path)

# evaluation: a fresh model in eval mode restored from the snapshot (test.py:252-264)
ev = pkg.get_segmentation_model(args.model, mode="eval", batch_size=1, num_steps=T, H=H, W=W, **kw)
CK.Saver().restore(ev, CK.latest_checkpoint(os.path.join(args.out, args.model)))
seg = HU.SegEval()
for words, im, seq_len, mask in batches:
    for b in range(args.batch):
        out = ev.forward(words[b:b + 1], im[b:b + 1], seq_len[b:b + 1])
        pred = (out["up"][0, :, :, 0] >= 1e-9).cpu().numpy()
        seg.add(pred, mask[b, :, :, 0] > 0)
print({k: round(v, 4) for k, v in seg.result().items()})
