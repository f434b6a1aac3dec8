"""Does a power-of-two row stride (2 KiB for 1024 bf16 columns) cost L2 channel balance?  gemm_nt with the row
strides of A / Bt / C padded by `pad` elements (same K, N: the extra columns are never touched).
usage: python scripts/gemm_stride.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("cmpc-refseg_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((12800, 1024, 1024), (12800, 1024, 2048), (12800, 512, 512), (12800, 2048, 1024), (12800, 1024, 4096)):
    line = f"M={M} N={N} K={K}:"
    for (pa, pb, pc) in ((0, 0, 0), (64, 0, 0), (64, 64, 0), (64, 64, 64), (32, 32, 32), (8, 8, 8), (192, 192, 192)):
        A = torch.randn(M, K + pa, device=dev).bfloat16(); Bt = torch.randn(N, K + pb, device=dev).bfloat16()
        C = torch.empty(M, N + pc, device=dev, dtype=torch.bfloat16)
        us = bench(lambda: ops.gemm_nt(1, [(A, K + pa, Bt, K + pb, K)], C, N + pc, M, N))
        line += f"  pad({pa},{pb},{pc}) {us:6.1f}us {2*M*N*K/us/1e6:5.0f}TF"
    print(line, flush=True)
print("gemm_tn (unsplit):")
for (R, K, N) in ((12800, 2048, 2048), (12800, 1024, 1024)):
    line = f"R={R} K={K} N={N}:"
    for pad in (0, 64, 8):
        A = torch.randn(R, K + pad, device=dev).bfloat16(); D = torch.randn(R, N + pad, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
        us = bench(lambda: ops.gemm_tn(1, A, K + pad, K, D, N + pad, N, out, N, R, K, N, rsplit=2), n=10)
        line += f"  pad {pad} {us:7.1f}us {2*R*K*N/us/1e6:5.0f}TF"
    print(line, flush=True)
