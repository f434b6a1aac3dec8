"""Experiment: the frozen backbone as S independent batch chunks on S streams inside ONE captured graph (frozen BN: no coupling across
images).  res3-res5 at B=8 40x40 are 12 800-row products: 100-400 tiles of fixed cost ~15 us per launch on a 256-CU chip.
usage: bb_split.py [B] [H]"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("cmpc-refseg_amd")
bbm = importlib.import_module("cmpc-refseg_amd.backbone")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 320
dev = torch.device("cuda:0")
net = bbm.DeepLabResNet().to(dev)
net.load_tf(bbm.init_params())
net = net.to(torch.float16)
im = (torch.rand(B, H, H, 3, device=dev) * 255 - 120)
ref = net(im)


def run(split):
    net.split = split
    for _ in range(2):
        out = net(im)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        out = net(im)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print(f"B={B} {H}x{H} split={split}: {dt*1e3:.3f} ms per pass, bit-identical to unsplit: {same}", flush=True)


for s in (1, 2, 4, 1, 2, 4):
    if B % s == 0:
        run(s)


def run_eager(split):
    net.split = split
    for _ in range(3):
        out = net(im)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = net(im)
    th = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"eager B={B} split={split}: {dt*1e3:.3f} ms per pass (host enqueue {th*1e3:.3f} ms)", flush=True)


for s in (1, 2, 4):
    if B % s == 0:
        run_eager(s)
