"""Bisect HIP-graph capture problems: python scripts/graph_probe.py <case>  (run each case in its own process)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("cmpc-refseg_amd")
case = sys.argv[1]
B = 2
kw = dict(batch_size=B, mode="train", dtype="bf16")
if case.startswith("s1"):
    os.environ["CMPC_STREAMS"] = "1"
m = P.LSTM_model(**kw)
T, H, W = m.num_steps, m.H, m.W
g = torch.Generator().manual_seed(0)
words = torch.randint(1, 1000, (B, T), generator=g, dtype=torch.int32).cuda()
im = torch.randn(B, H, W, 3, generator=g).cuda()
tgt = (torch.rand(B, H, W, 1, generator=g) > 0.5).float().cuda()
sl = torch.full((B,), 7, dtype=torch.int32).cuda()
st = torch.cuda.Stream()

side = [torch.cuda.Stream() for _ in range(3)]
xl = torch.randn(256, 256, device="cuda", requires_grad=True)
ws = [torch.randn(256, 256, device="cuda", requires_grad=True) for _ in range(3)]

def torch_only():
    main = torch.cuda.current_stream()
    a = xl * 2
    bs = []
    for i in range(3):
        side[i].wait_stream(main)
        with torch.cuda.stream(side[i]):
            bs.append((a @ ws[i]).tanh())
    for s_ in side:
        main.wait_stream(s_)
    loss = (bs[0] + bs[1] + bs[2]).sum()
    loss.backward()
    for s_ in side:
        main.wait_stream(s_)
    return loss.detach()

def body():
    if "torchonly" in case:
        return torch_only()
    if "backbone" in case:
        with torch.no_grad():
            return m.features(im)[2].float().sum()
    if "fwd" in case:
        with torch.no_grad():
            feats, s2 = m.features_async(im)
            return m.head(feats, words, sl, after=s2)["up"].sum()
    if "headbwd" in case:
        o = m.loss_and_grads(FE, words, tgt, sl)
        return o["loss_all"].detach()
    return m._fwd_bwd(words, im, tgt, sl)

if "headbwd" in case:
    with torch.no_grad():
        FE = [f.clone() for f in m.features(im)]
for _ in range(2):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        r = body()
    torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
print(case, "eager", float(r.float().sum()), flush=True)
if os.environ.get("PROBE_MT", "1") == "0":
    torch.autograd.set_multithreading_enabled(False)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=st, capture_error_mode=os.environ.get("PROBE_MODE", "global")):
    r = body()
print(case, "captured", flush=True)
gr.replay()
torch.cuda.synchronize()
print(case, "replayed", float(r.float().sum()), flush=True)
