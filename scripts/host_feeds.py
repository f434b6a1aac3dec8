"""The train step fed from HOST numpy arrays every step (what the reference's feed_dict does), against device-resident feeds.  usage: host_feeds.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
m = pkg.LSTM_model(batch_size=8, mode="train")
w, im, sl, tg = synth_batch(8, 20, 320, 320, m.cfg.vocab_size, 0)
dfeeds = [torch.from_numpy(x).to(dev) for x in (w, im, tg, sl)]
pinned = [torch.from_numpy(x).pin_memory() for x in (w, im, tg, sl)]
for name, feeds in (("device-resident", dfeeds), ("host numpy (pageable)", (w, im, tg, sl)), ("host pinned tensors", pinned)):
    for _ in range(8):
        m.train_step(*feeds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        m.train_step(*feeds)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"{name:24s} {dt*1e3:7.3f} ms per step  {8/dt:7.1f} images/s", flush=True)
