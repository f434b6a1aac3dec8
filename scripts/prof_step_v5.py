"""Profiling driver for BASELINE config 4 / 5: N train steps with every launch on ONE stream.
  cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 scripts/prof_step_v5.py [steps] [lanes] [v5|video]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
which = sys.argv[3] if len(sys.argv) > 3 else "v5"
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
if which == "v5":
    B, T, H, W = 8, 25, 512, 512
    m = pkg.get_segmentation_model("CMPCv5_BiLSTM_HSV_model", batch_size=B, num_steps=T, vf_h=64, vf_w=64, H=H, W=W, mode="train", dtype="f16", n_lanes=3)
    m.set_lanes(lanes)
    w, im, sl, tg = (torch.from_numpy(x).to(dev) for x in synth_batch(B, T, H, W, m.cfg.vocab_size, 4))
    for i in range(steps):
        m.train_step(w, im, tg, sl)
else:
    m = pkg.get_segmentation_model("CMPC_video_mm_tgraph_allvec", batch_size=1, mode="train", dtype="f16", n_lanes=3)
    m.set_lanes(lanes)
    g = torch.Generator().manual_seed(5)
    words = torch.zeros(1, 20, dtype=torch.int64); words[0, 11:] = torch.randint(1, m.cfg.vocab_size, (9,), generator=g)
    vi = torch.tensor([[11]], dtype=torch.int32)
    clip = (torch.rand(1, 16, 320, 320, 3, generator=g) * 255 - 120).to(dev)
    tg = (torch.rand(1, 320, 320, 1, generator=g) < 0.2).float().to(dev)
    for i in range(steps):
        m.train_step_video(words, None, tg, vi, clip)
torch.cuda.synchronize()
print("launches/step (library calls checked):", m.eng.launch_count(), flush=True)
