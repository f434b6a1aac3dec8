"""bf16 mask error budget: (a) bf16 head + bf16 backbone, (b) bf16 head + exact feats, (c) f32 head + bf16 backbone feats."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util as U
from tests.util import O
from bench import synth_batch
torch.set_num_threads(16)
P = U.pkg()
for B, seed in ((2, 11), (2, 12), (2, 13), (4, 7), (8, 0), (8, 1)):
    cfg = O.Cfg(batch_size=B)
    hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
    w, im, sl, tg = map(torch.from_numpy, synth_batch(B, 20, 320, 320, cfg.vocab_size, seed))
    with torch.no_grad():
        feats = O.backbone_forward(bp, im, cfg)
        taps = O.head_forward(hp, feats, w, sl, cfg)
        ref = O.losses(hp, taps, tg, cfg)
    mb = P.LSTM_model(batch_size=B, mode="train", dtype="bf16", head_params=hp, backbone_params=bp)
    mh = P.LSTM_model(batch_size=B, mode="train", dtype="f16", head_params=hp, backbone_params=bp)
    with torch.no_grad():
        fb, fh = mb.features(im), mh.features(im)
        res = {}
        for name, m, f in (("bf16", mb, fb), ("f16", mh, fh), ("f16 head, exact feats", mh, [x.to(mh.device) for x in feats])):
            o = m.head(f, w, sl, tg)
            torch.cuda.synchronize()
            up = o["up"].float().cpu()
            res[name] = (abs(float(o["mIoU"]) - float(ref["mIoU"])), int(((up > 0) != (taps["up"] > 0)).sum()))
    print(f"B={B} seed={seed}: " + "; ".join(f"{k}: dIoU={v[0]:.2e} flips={v[1]}" for k, v in res.items()), flush=True)
    del mb, mh
    torch.cuda.empty_cache()
