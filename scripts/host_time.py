import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 320      # 64: GPU work is negligible, the step time is the host enqueue time
m = pkg.LSTM_model(batch_size=8, mode="train", dtype="bf16", H=HW, W=HW, vf_h=HW // 8, vf_w=HW // 8)
w, im, sl, tg = synth_batch(8, 20, HW, HW, m.cfg.vocab_size, 0)
w, im, sl, tg = [torch.from_numpy(x).to(dev) for x in (w, im, sl, tg)]
for _ in range(5): m.train_step(w, im, tg, sl)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): m.train_step(w, im, tg, sl)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
# phases
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.train_step(w, im, tg, sl)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
