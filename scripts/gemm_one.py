"""Run one gemm_nt / gemm_tn shape a few times (for rocprofv3 --pmc). usage: gemm_one.py nt|tn M N K [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("tests.opwrap")
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
if kind == "nt":
    A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(reps): ops.gemm_nt(1, [(A, K, Bt, K, K)], C, N, M, N)
else:
    A = torch.randn(M, K, device=dev).bfloat16(); D = torch.randn(M, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
    for _ in range(reps): ops.gemm_tn(1, A, K, K, D, N, N, out, N, M, K, N)
torch.cuda.synchronize()
