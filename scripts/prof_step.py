"""Profiling driver: N train steps of the benchmark configuration (B=8, 320x320, L=20, bf16) with every launch on ONE stream
(set_lanes(1)) so that rocprofv3's per-kernel durations are not inflated by overlap.
  cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 scripts/prof_step.py [steps] [lanes] [batch]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
m = pkg.LSTM_model(batch_size=B, mode="train", dtype=os.environ.get("CMPC_DTYPE", "f16"), n_lanes=3)
m.set_lanes(lanes)
w, im, sl, tg = (torch.from_numpy(x).to(dev) for x in synth_batch(B, 20, 320, 320, m.cfg.vocab_size, 0))
for i in range(steps):
    m.train_step(w, im, tg, sl)
torch.cuda.synchronize()
print("launches/step (library calls checked):", m.eng.launch_count(), flush=True)
