"""Phase boundaries of the multi-stream train step on the GPU clock (cmpc_phase_marks: hipEvents at the stage boundaries, no
profiler, so the overlap between the lanes is the real one).  usage: python scripts/phase_timeline.py [dtype] [steps]"""
import collections, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = pkg.LSTM_model(batch_size=8, mode="train", dtype=dtype)
w, im, sl, tg = [torch.from_numpy(x).to(dev) for x in synth_batch(8, 20, 320, 320, m.cfg.vocab_size, 0)]
torch.cuda.synchronize(); ready = torch.cuda.Event(); ready.record()
for _ in range(8): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
m.eng.phase_marks(True)
for _ in range(N): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
marks = m.eng.phase_marks_read()
steps, cur = [], None
for name, t in marks:
    if name == "fwd:start":
        cur = []; steps.append(cur)
    cur.append((name, t))
acc = collections.OrderedDict()
for i in range(1, len(steps) - 1):
    t0 = steps[i][0][1]
    for name, t in steps[i]:
        acc.setdefault(name, []).append(t - t0)
    acc.setdefault("next fwd:start", []).append(steps[i + 1][0][1] - t0)
print("phase boundary, ms after fwd:start on the GPU clock (mean over %d steps)" % (len(steps) - 2))
for name, v in sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print(f"  {name:22s} {sum(v)/len(v):8.3f}")
