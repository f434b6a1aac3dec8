"""Calibration of the backbone's 1x1 shapes (res4 at 40x40, B=8 and 64x64, B=8): cmpc_gemm_nt against the vendor GEMM (torch.matmul = hipBLASLt).
f16, bias + relu omitted on both sides.  usage: bb_shapes.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("tests.opwrap")
dev = torch.device("cuda:0")
torch.manual_seed(0)
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, what) in ((12800, 256, 1024, "res4 2a"), (12800, 1024, 256, "res4 2c"), (12800, 512, 2048, "res5 2a"), (12800, 2048, 512, "res5 2c"), (12800, 128, 512, "res3 2a"),
                        (12800, 512, 128, "res3 2c"), (51200, 64, 256, "res2 2a"), (51200, 256, 64, "res2 2c"), (32768, 256, 1024, "res4 2a @512"), (32768, 1024, 256, "res4 2c @512")):
    A = torch.randn(M, K, device=dev).half(); Bt = torch.randn(N, K, device=dev).half(); C = torch.empty(M, N, device=dev, dtype=torch.float16)
    us1 = bench(lambda: torch.matmul(A, Bt.t(), out=C))
    us = bench(lambda: ops.gemm_nt(2, [(A, K, Bt, K, K)], C, N, M, N))
    by = 2.0 * (M * K + M * N + N * K)
    print(f"{what:14s} {M:6d} {N:5d} {K:5d}  gemm_nt {us:7.1f} us {2*M*N*K/us/1e6:7.1f} TF {by/us/1e6:5.2f} TB/s | vendor {us1:7.1f} us {2*M*N*K/us1/1e6:7.1f} TF", flush=True)
