"""gemm_tn on the head's weight-gradient shapes, reduction split 1/2/4/8 ways (split parts go through slabs + a fold launch).
With one workgroup per CU and no split, a 64-row step of a 128 x 128 tile takes ~0.95 us for 32 KB of operands = 34 GB/s per CU:
the Infinity-Cache / HBM ingest rate of a CU (MI355X_MICROARCH.md, gather table), not MFMA (0.25 us) or LDS."""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("tests.opwrap"); importlib.import_module("cmpc-refseg_amd")._lib.load()
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (R, K, N) in ((12800, 1024, 1024), (12800, 1024, 5120), (12800, 512, 512), (12800, 2048, 1024)):
    A = torch.randn(R, K, device=dev).bfloat16(); D = torch.randn(R, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
    res = {}
    for rs in (1, 2, 4, 8):
        us = t(lambda: ops.gemm_tn(1, A, K, K, D, N, N, out, N, R, K, N, rsplit=rs)); res[rs] = us
    print(f"R={R} K={K} N={N}: " + "  ".join(f"rsplit={rs}: {us:7.1f} us {2*R*K*N/us/1e6:6.0f} TF" for rs, us in res.items()), flush=True)
