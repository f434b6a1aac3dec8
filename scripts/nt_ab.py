"""gemm_nt on the head's shapes (f16), one library build per process: run it once per build (CMPC_LIB_PATH) on the same box.
Prints per shape: us, TFLOP/s, and the launch-weighted total per step.  usage: python scripts/nt_ab.py [tag]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("tests.opwrap"); importlib.import_module("cmpc-refseg_amd")._lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "cur"
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (M, N, K, launches per step, batch)
SHAPES = ((12800, 512, 512, 24, 1), (12800, 512, 2048, 5, 1), (12800, 512, 2112, 3, 1), (1600, 1024, 64, 9, 8), (12800, 1024, 512, 7, 1),
          (12800, 1024, 1024, 7, 1), (12800, 1024, 2048, 1, 1), (12800, 1024, 5120, 3, 1), (12800, 2048, 512, 1, 1), (12800, 2048, 1024, 2, 1),
          (12800, 5120, 1088, 3, 1))
tot = 0.0; fl = 0.0
for (M, N, K, cnt, nb) in SHAPES:
    A = torch.randn(nb * M, K, device=dev).half(); Bt = (0.1 * torch.randn(nb * N, K, device=dev)).half()
    # outputs rotate through > 256 MB of buffers: inside the step an output is not cache-resident when it is written
    Cs = [torch.empty(nb * M, N, device=dev, dtype=torch.float16) for _ in range(max(2, min(24, int(4e8 / (nb * M * N * 2)))))]
    bias = torch.randn(N, device=dev); it = [0]
    def run():
        it[0] += 1
        ops.gemm_nt(2, [(A, K, Bt, K, K, M * K, N * K)], Cs[it[0] % len(Cs)], N, M, N, batch=nb, sC=M * N, bias=bias, act=1)
    us = bench(run)
    f = 2.0 * nb * M * N * K
    tot += us * cnt; fl += f * cnt
    print(f"[{tag}] {nb}x{M:6d} {N:5d} {K:5d}  {us:8.1f} us {f/us/1e6:7.1f} TF", flush=True)
print(f"[{tag}] launch-weighted total {tot/1e3:.3f} ms per step -> {fl/tot/1e6:.0f} TFLOP/s")
# the Mutan product with each epilogue activation (the heads' tanh is its epilogue in the engine)
M, N, K = 12800, 5120, 1088
A = torch.randn(M, K, device=dev).half(); Bt = (0.1 * torch.randn(N, K, device=dev)).half()
Cs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(3)]
bias = torch.randn(N, device=dev)
for act, name in ((0, "none"), (1, "relu"), (2, "tanh")):
    def run():
        it[0] += 1
        ops.gemm_nt(2, [(A, K, Bt, K, K)], Cs[it[0] % 3], N, M, N, bias=bias, act=act)
    us = bench(run)
    print(f"[{tag}] 12800 x 5120 x 1088 act={name}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF")
