"""gemm_tn with an UNSPLIT reduction (the regime of the grouped weight-gradient launch): v1 (register-staged) vs v2 (LDS-DMA).
usage: python scripts/tn_longloop.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("cmpc-refseg_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (R, K, N) in ((12800, 2048, 2048), (12800, 4096, 2048), (12800, 1024, 1024), (51200, 2048, 2048)):
    A = torch.randn(R, K, device=dev).bfloat16(); D = torch.randn(R, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
    res = {}
    for name, env in (("v1", None), ("v2", "CMPC_TN_V2")):
        if env: os.environ[env] = "1"
        for rs in (1, 2, 4):
            us = bench(lambda: ops.gemm_tn(1, A, K, K, D, N, N, out, N, R, K, N, rsplit=rs))
            res[(name, rs)] = us
        if env: del os.environ[env]
    tiles = (K // 128) * (N // 128)
    print(f"R={R} K={K} N={N} tiles={tiles}: " + "  ".join(f"{k[0]}/rs{k[1]} {v:7.1f}us {2*R*K*N/v/1e6:6.0f}TF" for k, v in res.items()), flush=True)
