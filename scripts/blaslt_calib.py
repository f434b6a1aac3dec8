"""Calibration only (not a product path): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on this box for the
head's dominant bf16 shapes, next to this library's gemm_nt on the same operands."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("tests.opwrap")
importlib.import_module("cmpc-refseg_amd")._lib.load()
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, N, K) in ((12800, 5120, 1088), (12800, 1024, 5120), (12800, 1024, 1024), (12800, 1024, 2048), (12800, 512, 2112), (12800, 2048, 1024), (12800, 512, 512), (12800, 1024, 512), (25600, 1024, 1024)):
    A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    tb = t(lambda: torch.matmul(A, Bt.t(), out=C))
    to = t(lambda: ops.gemm_nt(1, [(A, K, Bt, K, K)], C, N, M, N))
    print(f"M={M} N={N} K={K}: hipBLASLt {fl/tb/1e12:7.1f} TF ({tb*1e6:6.1f} us)   gemm_nt {fl/to/1e12:7.1f} TF ({to*1e6:6.1f} us)", flush=True)
