"""Per-tap error of the CMPCv5_BiLSTM product path against the oracle at config 4's sizes (diagnostic). usage: v5_probe.py [hsv=1] [B=2] [dtypes=f32,f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util as U
from oracle import cmpc_torch as O, cmpc_v5_torch as V
import tests.test_gpu_v5 as T
from bench import synth_batch
hsv = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtypes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "f16"]
torch.set_num_threads(16)
cfg = V.Cfg(batch_size=B, num_steps=25, vf_h=64, vf_w=64, H=512, W=512, hsv=hsv)
hp, bp, bn = V.init_head_params(cfg), O.init_backbone_params(cfg), V.init_bn_state(cfg)
bp["bn_conv1/gamma"] = bp["bn_conv1/gamma"] * float(os.environ.get("STEM_GAMMA", str(V.STEM_GAMMA)))
w, im, sl, tg = map(torch.from_numpy, synth_batch(B, 25, 512, 512, cfg.vocab_size, 21))
with torch.no_grad():
    feats = V.backbone_taps(bp, im, cfg)
    taps = V.head_forward(hp, bn, feats, w, sl, cfg, im=im)
    ref = V.losses(hp, taps, tg, cfg)
print("feat max", [float(f.abs().max()) for f in feats])
P = U.pkg()
for dtype in dtypes:
    m = P.get_segmentation_model(T._name(cfg), batch_size=B, num_steps=25, vf_h=64, vf_w=64, H=512, W=512, mode="train", dtype=dtype, head_params=hp, backbone_params=bp)
    with torch.no_grad():
        o = m.head(m.features(im), w, sl, tg, im=im)
    torch.cuda.synchronize()
    pt = T.taps_as_oracle(o, cfg)
    print(dtype, "dIoU %.2e" % abs(float(o["mIoU"]) - float(ref["mIoU"])), {k: "%.1e" % U.rel_err(pt[k], taps[k]) for k in taps if k in pt})
    # same head on the ORACLE's taps (isolates the backbone's share)
    with torch.no_grad():
        o = m.head([f.to(m.device) for f in feats], w, sl, tg, im=im)
    torch.cuda.synchronize()
    pt = T.taps_as_oracle(o, cfg)
    print(dtype, "oracle feats: dIoU %.2e" % abs(float(o["mIoU"]) - float(ref["mIoU"])), {k: "%.1e" % U.rel_err(pt[k], taps[k]) for k in ("lat_c5", "vis_la_sp_c5", "fusion_c5", "fused", "aspp", "dec_cat", "pred", "up")})
    del m, o
    torch.cuda.empty_cache()
