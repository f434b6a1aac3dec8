"""Where does the bf16 mode's mask error come from?  Full-size B=2 head vs the oracle: per-tap relative error and flipped pixels."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util as U
from tests.util import O
from bench import synth_batch
torch.set_num_threads(16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
cfg = O.Cfg(batch_size=B)
hp, bp = O.init_head_params(cfg), O.init_backbone_params(cfg)
w, im, sl, tg = map(torch.from_numpy, synth_batch(B, 20, 320, 320, cfg.vocab_size, seed))
with torch.no_grad():
    feats = O.backbone_forward(bp, im, cfg)
    taps = O.head_forward(hp, feats, w, sl, cfg)
    ref = O.losses(hp, taps, tg, cfg)
P = U.pkg()
for dtype in ("f32", "bf16"):
    m = P.LSTM_model(batch_size=B, mode="train", dtype=dtype, head_params=hp, backbone_params=bp)
    for src in ("own_backbone", "oracle_feats"):
        with torch.no_grad():
            f = m.features(im) if src == "own_backbone" else [x.to(m.device) for x in feats]
            o = m.head(f, w, sl, tg)
        torch.cuda.synchronize()
        pt = U.product_taps_as_oracle(o, cfg)
        up = pt["up"]
        flips = int(((up > 0) != (taps["up"] > 0)).sum())
        print(f"== {dtype} {src}: dIoU={abs(float(o['mIoU'])-float(ref['mIoU'])):.2e} flips={flips} |up| median={float(taps['up'].abs().median()):.3f}")
        if dtype == "bf16":
            for k in ("lat_c5", "vis_la_sp_c5", "spa_graph_c5", "fusion_c5", "fusion_c3", "exg_c5", "exg_c5_2", "exg_c3_2", "fused", "pred", "up", "up_c5"):
                d = (pt[k].double() - taps[k].double())
                print(f"   {k:14s} max-rel {U.rel_err(pt[k], taps[k]):.2e}  rms-rel {float(d.pow(2).mean().sqrt() / taps[k].double().pow(2).mean().sqrt()):.2e}")
    del m
    torch.cuda.empty_cache()
