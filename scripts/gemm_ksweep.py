"""K / M sweep of gemm_nt and gemm_tn (bf16): separates the fixed cost of a launch (prologue, epilogue, wave
quantisation) from the per-k-tile cost.  usage: python scripts/gemm_ksweep.py"""
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
print("gemm_nt bf16: M N K us TF")
for M in (8192, 12800, 16384):
    for N in (512, 1024):
        for K in (256, 512, 1024, 2048, 4096, 8192):
            A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            us = bench(lambda: ops.gemm_nt(1, [(A, K, Bt, K, K)], C, N, M, N))
            print(f"  nt {M:6d} {N:5d} {K:5d} {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TF", flush=True)
print("gemm_tn bf16: R K N us TF")
for R in (3200, 12800, 51200):
    for (K, N) in ((512, 512), (1024, 1024), (2048, 1024)):
        A = torch.randn(R, K, device=dev).bfloat16(); D = torch.randn(R, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
        us = bench(lambda: ops.gemm_tn(1, A, K, K, D, N, N, out, N, R, K, N))
        print(f"  tn {R:6d} {K:5d} {N:5d} {us:8.1f} us {2*R*N*K/us/1e6:7.1f} TF", flush=True)
