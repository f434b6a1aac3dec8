"""GPU check of the raw GEMM entry points against torch (fp32 reference)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
P = U.pkg()
from importlib import import_module
ops = import_module("tests.opwrap")
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
def chk(name, got, ref, tol):
    global ok
    e = float((got.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))
    flag = "OK " if e < tol else "BAD"
    if e >= tol: ok = False
    print(f"{flag} {name:50s} rel_err={e:.3e}")

for dt, tdt, tol in ((0, torch.float32, 2e-6), (1, torch.bfloat16, 2e-2)):
    for (M, N, K) in ((300, 128, 64), (128, 256, 192), (37, 64, 128), (1000, 1024, 512), (8, 192, 64)):
        A = torch.randn(M, K, device=dev).to(tdt); Bt = torch.randn(N, K, device=dev).to(tdt)
        C = torch.empty(M, N, device=dev, dtype=tdt)
        ops.gemm_nt(dt, [(A, K, Bt, K, K)], C, N, M, N)
        chk(f"nt dt{dt} {M}x{N}x{K}", C.float(), A.float() @ Bt.float().t(), tol)
    # epilogue: bias, sbias, pbias, relu, alpha, n_valid, f32 out, accumulate, two segments
    M, N, K1, K2, rps = 96, 128, 64, 128, 24
    A1 = torch.randn(M, K1, device=dev).to(tdt); A2 = torch.randn(M, K2, device=dev).to(tdt)
    B1 = torch.randn(N, K1 + K2, device=dev).to(tdt)
    bias = torch.randn(N, device=dev); sb = torch.randn(M // rps, N, device=dev); pb = torch.randn(rps, N, device=dev)
    C = torch.randn(M, N, device=dev)
    C0 = C.clone()
    ops.gemm_nt(dt, [(A1, K1, B1, K1 + K2, K1), (A2, K2, B1.data_ptr() + K1 * (4 if dt == 0 else 2), K1 + K2, K2)], C, N, M, N,
                n_valid=100, c_f32=True, bias=bias, sbias=sb, ld_sbias=N, pbias=pb, ld_pbias=N, rows_per_sample=rps,
                act=1, alpha=0.5, accumulate=True)
    ref = 0.5 * (torch.cat([A1, A2], 1).float() @ B1.float().t()) + bias + sb.repeat_interleave(rps, 0) + pb.repeat(M // rps, 1)
    ref = torch.relu(ref); ref[:, 100:] = 0; ref = ref + C0
    chk(f"nt dt{dt} epilogue/2seg", C, ref, tol)
    # batched
    Bn, M, N, K = 3, 70, 64, 64
    A = torch.randn(Bn, M, K, device=dev).to(tdt); Bt = torch.randn(Bn, N, K, device=dev).to(tdt)
    C = torch.empty(Bn, M, N, device=dev, dtype=tdt)
    ops.gemm_nt(dt, [(A, K, Bt, K, K, M * K, N * K)], C, N, M, N, batch=Bn, sC=M * N)
    chk(f"nt dt{dt} batched", C.float(), torch.bmm(A.float(), Bt.float().transpose(1, 2)), tol)
    # tn
    for (R, K, N) in ((500, 128, 128), (1000, 200, 72), (64, 8, 40), (3000, 256, 384), (7, 40, 24)):
        Kp, Np = (K + 7) // 8 * 8, (N + 7) // 8 * 8
        A = torch.randn(R, Kp, device=dev).to(tdt); D = torch.randn(R, Np, device=dev).to(tdt)
        out = torch.zeros(K, N, device=dev)
        ops.gemm_tn(dt, A, Kp, Kp, D, Np, Np, out, N, R, K, N)
        chk(f"tn dt{dt} R{R} {K}x{N}", out, (A.float().t() @ D.float())[:K, :N], tol)
    # tn with inner/outer batches
    R, K, N, nb2 = 200, 64, 64, 3
    A = torch.randn(nb2, R, K, device=dev).to(tdt); D = torch.randn(nb2, R, 2 * N, device=dev).to(tdt)
    out = torch.zeros(nb2, K, 2 * N, device=dev)
    ops.gemm_tn(dt, A, K, K, D, 2 * N, N, out, 2 * N, R, K, N, offs=((0, 0, 0), (0, N, N)), nb2=nb2, a_bs=R * K, d_bs=R * 2 * N, o_bs=K * 2 * N, alpha=2.0)
    chk(f"tn dt{dt} batched", out, 2.0 * torch.bmm(A.float().transpose(1, 2), D.float()), tol)
torch.cuda.synchronize()
print("ALL OK" if ok else "SOME BAD")
sys.exit(0 if ok else 1)
