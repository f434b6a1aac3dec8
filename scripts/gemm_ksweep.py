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
VAR = sys.argv[1] if len(sys.argv) > 1 else "CMPC_GEMM_V5"      # variant under test (first column) vs v2/v3
print(f"gemm_nt bf16: M N K | {VAR}=1 | v2/v3")
for M in (12800,):
    for N in (256, 512, 1024, 2048, 5120):
        for K in (256, 512, 1024, 2048, 4096):
            A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16()
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); C2 = torch.empty_like(C)
            bias = torch.randn(N, device=dev)
            os.environ[VAR] = "1"
            us5 = bench(lambda: ops.gemm_nt(1, [(A, K, Bt, K, K)], C, N, M, N, bias=bias, act=1))
            del os.environ[VAR]
            os.environ["CMPC_GEMM_V5"] = "0"
            us2 = bench(lambda: ops.gemm_nt(1, [(A, K, Bt, K, K)], C2, N, M, N, bias=bias, act=1))
            del os.environ["CMPC_GEMM_V5"]
            ref = torch.relu(A[:300].float() @ Bt.float().t() + bias)
            e5 = float((C[:300].float() - ref).abs().max() / ref.abs().max()); e2 = float((C2[:300].float() - ref).abs().max() / ref.abs().max())
            same = bool((C == C2).all())
            print(f"  nt {M:6d} {N:5d} {K:5d} | new {us5:8.1f} us {2*M*N*K/us5/1e6:7.1f} TF err {e5:.1e} | v2 {us2:8.1f} us {2*M*N*K/us2/1e6:7.1f} TF err {e2:.1e} | identical={same}", flush=True)
print("gemm_tn bf16: R K N us TF")
for R in (12800,):
    for (K, N) in ((512, 512), (1024, 1024), (2048, 1024)):
        A = torch.randn(R, K, device=dev).bfloat16(); D = torch.randn(R, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
        us = bench(lambda: ops.gemm_tn(1, A, K, K, D, N, N, out, N, R, K, N))
        print(f"  tn {R:6d} {K:5d} {N:5d} {us:8.1f} us {2*R*N*K/us/1e6:7.1f} TF", flush=True)
