"""Phase timeline of one gemm_nt_v2 launch (needs `make -C cmpc-refseg_amd/csrc libcmpc_hip_trace.so`).
usage: CMPC_LIB_PATH=cmpc-refseg_amd/csrc/libcmpc_hip_trace.so python scripts/gemm_trace.py M N K"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import util as U
from importlib import import_module
U.pkg()
ops = import_module("cmpc-refseg_amd.ops"); L = import_module("cmpc-refseg_amd._lib")
M, N, K = (int(x) for x in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
dev = torch.device("cuda:0")
if kind == "nt":
    A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16(); Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm_nt(1, [(A, K, Bt, K, K)], Cc, N, M, N)
else:           # tn: M = R (rows reduced over), out [K, N]
    A = torch.randn(M, K, device=dev).bfloat16(); D = torch.randn(M, N, device=dev).bfloat16(); out = torch.zeros(K, N, device=dev)
    for _ in range(5):
        ops.gemm_tn(1, A, K, K, D, N, N, out, N, M, K, N)
torch.cuda.synchronize()
lib = L.load()
n = 8 * 4096
buf = (C.c_longlong * n)()
lib.cmpc_debug_gemm_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.cmpc_debug_gemm_trace(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8)
nwg = int((t[:, 0] != 0).sum())
t = t[:nwg]
t0 = t[:, 0].min()
us = (t[:, :6] - t0) / 100.0
print(f"{kind} M={M} N={N} K={K}: {nwg} workgroups; kernel span {us[:,5].max():.2f} us")
names = ["start", "first tile landed", "main loop done", "slab written", "stores issued", "stores done"]
for i, nm in enumerate(names):
    print(f"  {nm:18s} min {us[:,i].min():7.2f}  median {np.median(us[:,i]):7.2f}  max {us[:,i].max():7.2f}")
d = np.diff(us, axis=1)
for i, nm in enumerate(["prologue", "main loop", "slab", "epilogue issue", "store drain"]):
    print(f"  d {nm:16s} median {np.median(d[:,i]):7.2f}  p90 {np.percentile(d[:,i],90):7.2f} max {d[:,i].max():7.2f}")
if (t[:, 7] > t[:, 6]).all() and (t[:, 6] > 1 << 20).all():      # v4 trace build: shader-clock stamps around the main loop
    ghz = (t[:, 7] - t[:, 6]) / ((t[:, 2] - t[:, 1]) * 10.0)
    print(f"  shader clock during the main loop: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f})")
order = np.argsort(us[:, 0])
print("  start times of workgroups (sorted, every 32nd):", np.round(us[order, 0][::32], 2).tolist())
late = us[:, 0] > np.median(us[:, 5]) * 0.5
print(f"  second-round workgroups: {int(late.sum())}; their median start {np.median(us[late,0]) if late.any() else 0:.2f}")
