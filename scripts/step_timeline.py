"""Phase boundaries of the multi-stream train step on the GPU clock (CMPC_MARKS=1 events; no profiler, so the
overlap is the real one).  usage: CMPC_MARKS=1 python scripts/step_timeline.py"""
import sys, os
os.environ["CMPC_MARKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections, torch, importlib
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
m = pkg.LSTM_model(batch_size=8, mode="train", dtype="bf16")
w, im, sl, tg = synth_batch(8, 20, 320, 320, m.cfg.vocab_size, 0)
w, im, sl, tg = [torch.from_numpy(x).to(dev) for x in (w, im, sl, tg)]
torch.cuda.synchronize(); ready = torch.cuda.Event(); ready.record()
for _ in range(5): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
m.marks.clear()
N = 10
for _ in range(N): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
# split into steps at "step_start"
steps, cur = [], None
for name, ev in m.marks:
    if name == "step_start":
        cur = []; steps.append(cur)
    cur.append((name, ev))
acc = collections.OrderedDict()
for i in range(1, len(steps) - 1):
    t0 = steps[i][0][1]
    for name, ev in steps[i]:
        acc.setdefault(name, []).append(t0.elapsed_time(ev))
    acc.setdefault("next_step_start", []).append(t0.elapsed_time(steps[i + 1][0][1]))
print("phase boundary, ms after step_start on the GPU clock (mean over %d steps); backbone / adam are on their own streams" % (len(steps) - 2))
for name, v in sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print(f"  {name:18s} {sum(v)/len(v):8.3f}")
