"""Per-step GPU-clock durations of consecutive train steps (events at every step start, no host sync inside).
usage: python scripts/step_jitter.py [n_steps]"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib, time
from bench import synth_batch
pkg = importlib.import_module("cmpc-refseg_amd")
dev = torch.device("cuda:0")
m = pkg.LSTM_model(batch_size=8, mode="train", dtype="bf16")
w, im, sl, tg = synth_batch(8, 20, 320, 320, m.cfg.vocab_size, 0)
w, im, sl, tg = [torch.from_numpy(x).to(dev) for x in (w, im, sl, tg)]
torch.cuda.synchronize(); ready = torch.cuda.Event(); ready.record()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for mode in ("gc on", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.disable()
    for _ in range(5): m.train_step(w, im, tg, sl, ready=ready)
    torch.cuda.synchronize()
    evs, host = [], []
    t0 = time.perf_counter()
    for _ in range(N):
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        h0 = time.perf_counter()
        m.train_step(w, im, tg, sl, ready=ready)
        host.append(1e3 * (time.perf_counter() - h0))
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / N
    d = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
    ds = sorted(d); hs = sorted(host)
    print(f"{mode}: wall {wall:.2f} ms/step | GPU step: min {ds[0]:.2f} median {ds[N//2]:.2f} p90 {ds[int(N*0.9)]:.2f} max {ds[-1]:.2f} | host enqueue: min {hs[0]:.2f} median {hs[N//2]:.2f} p90 {hs[int(N*0.9)]:.2f} max {hs[-1]:.2f}")
    print(f"   memory: allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB, hipMalloc calls {torch.cuda.memory_stats().get('num_device_alloc', -1)}")
    print("   first 20 GPU step times:", " ".join(f"{x:.1f}" for x in d[:20]))
