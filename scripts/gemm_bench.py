"""GPU micro-benchmark of gemm_nt / gemm_tn on the head's shapes (bf16). usage: python scripts/gemm_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("tests.opwrap")
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
print("gemm_nt bf16  M N K  us  TFLOP/s")
tot = 0; tot1 = 0
for (M, N, K, cnt) in ((12800, 1024, 1024, 12), (12800, 1024, 2048, 1), (12800, 512, 512, 24), (12800, 512, 2112, 3), (12800, 5120, 1088, 3),
                  (12800, 2048, 1024, 2), (12800, 1024, 5120, 3), (12800, 512, 2048, 3), (12800, 1024, 512, 7)):
    A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us1 = bench(lambda: torch.matmul(A, Bt.t(), out=C))        # vendor GEMM (hipBLASLt), calibration only
    us = bench(lambda: ops.gemm_nt(1, [(A, K, Bt, K, K)], C, N, M, N))
    ref = (A[:256].float() @ Bt.float().t())
    err = float((C[:256].float() - ref).abs().max() / ref.abs().max())
    tot += us * cnt; tot1 += us1 * cnt
    print(f"  {M:6d} {N:5d} {K:5d}  gemm_nt {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TF | hipBLASLt {us1:8.1f} us {2*M*N*K/us1/1e6:7.1f} TF   err={err:.1e}")
print(f"  weighted total gemm_nt {tot/1e3:.2f} ms  hipBLASLt {tot1/1e3:.2f} ms")
print("gemm_tn bf16  R K N  us  TFLOP/s")
tot = 0; tot1 = 0
for (R, K, N, cnt) in ((12800, 1024, 1024, 9), (12800, 512, 512, 24), (12800, 1024, 512, 6), (12800, 2048, 1024, 1), (12800, 512, 2048, 5)):
    A = torch.randn(R, K, device=dev).bfloat16(); D = torch.randn(R, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
    Af = A.float(); Df = D.float()
    us1 = bench(lambda: torch.matmul(A.t(), D))                # vendor GEMM, calibration only
    out.zero_(); ops.gemm_tn(1, A, K, K, D, N, N, out, N, R, K, N)
    ref = A[:, :64].float().t() @ D.float()
    err = float((out[:64] - ref).abs().max() / ref.abs().max())
    us = bench(lambda: ops.gemm_tn(1, A, K, K, D, N, N, out, N, R, K, N))
    tot += us * cnt; tot1 += us1 * cnt
    print(f"  {R:6d} {K:5d} {N:5d}  gemm_tn {us:8.1f} us {2*R*N*K/us/1e6:7.1f} TF | hipBLASLt {us1:8.1f} us {2*R*N*K/us1/1e6:7.1f} TF  err={err:.1e}")
print(f"  weighted total gemm_tn {tot/1e3:.2f} ms hipBLASLt {tot1/1e3:.2f} ms")
