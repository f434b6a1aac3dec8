"""Host-side cost of the three library calls of a train step (wall clock around each call; the GPU runs behind)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = pkg.LSTM_model(batch_size=8, mode="train", dtype="f16")
m.set_lanes(lanes)
w, im, sl, tg = [torch.from_numpy(x).to(dev) for x in synth_batch(8, 20, 320, 320, m.cfg.vocab_size, 0)]
torch.cuda.synchronize(); ready = torch.cuda.Event(); ready.record()
for _ in range(8): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
E = m.eng
acc = {"forward": 0.0, "backward": 0.0, "optimizer_step": 0.0, "features_async": 0.0}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(obj, name, g)
for n in ("forward", "backward", "optimizer_step"): wrap(E, n)
wrap(m, "features_async")
N = 20
for sync in (False, True):
    for k in acc: acc[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(N):
        m.train_step(w, im, tg, sl, ready=ready)
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"lanes={lanes} sync_each_step={sync}: step {1e3*tot/N:.2f} ms; host ms/step: " + ", ".join(f"{k} {1e3*v/N:.2f}" for k, v in acc.items()), flush=True)
