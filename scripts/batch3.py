"""Do the three pyramid levels' identical products run faster as ONE batched launch than as three launches on three streams?
usage: batch3.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("tests.opwrap")
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, what) in ((12800, 5120, 1088, "mutan fwd"), (12800, 1024, 5120, "mutan dX"), (12800, 1024, 1024, "gconv"), (12800, 512, 2048, "fusion")):
    A = torch.randn(3, M, K, device=dev).half(); Bt = torch.randn(3, N, K, device=dev).half(); C = torch.empty(3, M, N, device=dev, dtype=torch.float16)
    sts = [torch.cuda.Stream() for _ in range(3)]
    def three():
        for i in range(3):
            with torch.cuda.stream(sts[i]):
                ops.gemm_nt(2, [(A[i], K, Bt[i], K, K)], C[i], N, M, N)
    def serial():
        for i in range(3):
            ops.gemm_nt(2, [(A[i], K, Bt[i], K, K)], C[i], N, M, N)
    def batched():
        ops.gemm_nt(2, [(A, K, Bt, K, K, M * K, N * K)], C, N, M, N, batch=3, sC=M * N)
    res = {}
    for name, fn in (("3 streams", three), ("serial", serial), ("batched", batched)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        for s in sts: torch.cuda.current_stream().wait_stream(s)
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{what:10s} {M}x{N}x{K}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in res.items()), flush=True)
