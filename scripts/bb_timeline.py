"""When does the backbone of step k run relative to the head of step k?  torch timing events on the backbone stream and on main."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
m = pkg.LSTM_model(batch_size=8, mode="train", dtype="f16")
w, im, sl, tg = [torch.from_numpy(x).to(dev) for x in synth_batch(8, 20, 320, 320, m.cfg.vocab_size, 0)]
torch.cuda.synchronize(); ready = torch.cuda.Event(); ready.record()
for _ in range(8): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
rec = []
orig = m.features_async
def traced(im_, ready_=None):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(m.bb_stream)
    out = orig(im_, ready_)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(m.bb_stream)
    em = torch.cuda.Event(enable_timing=True); em.record(torch.cuda.current_stream())
    rec.append((e0, e1, em))
    return out
m.features_async = traced
base = torch.cuda.Event(enable_timing=True); base.record()
for _ in range(10): m.train_step(w, im, tg, sl, ready=ready)
torch.cuda.synchronize()
for i, (e0, e1, em) in enumerate(rec):
    print(f"step {i}: backbone start {base.elapsed_time(e0):8.3f}  end {base.elapsed_time(e1):8.3f}  (dur {e0.elapsed_time(e1):6.3f})   main reaches forward at {base.elapsed_time(em):8.3f}")
