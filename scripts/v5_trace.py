"""Phase timeline of the 256 x 256 gemm_nt kernel's workgroups (diagnostic build with -DCMPC_V5_TRACE, see gemm.hip):
100 MHz timestamps at entry / loads issued / first tile landed / main loop end / stores issued / stores drained.
usage: CMPC_LIB_PATH=build/libcmpc_trace.so python scripts/v5_trace.py M N K"""
import ctypes, importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("tests.opwrap"); lib = importlib.import_module("cmpc-refseg_amd")._lib.load()
dev = torch.device("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device=dev).half(); Bt = (0.1 * torch.randn(N, K, device=dev)).half(); C = torch.empty(M, N, device=dev, dtype=torch.float16)
bias = torch.randn(N, device=dev)
run = lambda: ops.gemm_nt(2, [(A, K, Bt, K, K)], C, N, M, N, bias=bias, act=1)
for _ in range(5): run()
torch.cuda.synchronize()
run(); torch.cuda.synchronize()
nb = ((M + 255) // 256) * ((N + 255) // 256)
buf = (ctypes.c_ulonglong * (8 * nb))()
fn = lib.cmpc_debug_v5_trace; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; fn.restype = ctypes.c_int
assert fn(buf, 8 * nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)[:, :6]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["entry", "loads issued", "first tile landed", "main loop end", "stores issued", "stores drained"]
print(f"M={M} N={N} K={K}: {nb} workgroups; kernel span {us.max():.1f} us")
order = np.argsort(us[:, 0])
for q in (0, nb // 4, nb // 2, 3 * nb // 4, nb - 1):
    r = us[order[q]]
    print(f"  workgroup #{q:4d} by start: " + "  ".join(f"{n} {v:7.2f}" for n, v in zip(names, r)))
d = np.diff(us, axis=1)
print("  median phase lengths (us): " + "  ".join(f"{names[i]}->{names[i+1]} {np.median(d[:, i]):.2f}" for i in range(5)))
print(f"  workgroup start times: first wave of {min(nb, 256)} within {np.sort(us[:, 0])[min(nb, 256) - 1]:.2f} us; median workgroup lifetime {np.median(us[:, 5] - us[:, 0]):.2f} us")
