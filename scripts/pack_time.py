"""Time of one repack (all operands) and of one Adam step.  usage: python scripts/pack_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
pkg = importlib.import_module("cmpc-refseg_amd")
m = pkg.LSTM_model(batch_size=8, mode="train", dtype="bf16")
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"pack all {bench(m.store.pack):.1f} us   pack stage0 {bench(lambda: m.store.pack(0)):.1f} us   tiles {m.store.total_tiles} (stage0 {m.store.stage0_tiles})")
arena = m.store.arena.clone()
m.store.pack()
torch.cuda.synchronize()
print("repack reproduces the arena:", bool(torch.equal(arena, m.store.arena)))
