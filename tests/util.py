"""Shared helpers for the parity tests: tiny / full configurations, oracle <-> product plumbing."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cmpc_torch as O   # noqa: E402  (tests may import the oracle)


def pkg():
    return importlib.import_module("cmpc-refseg_amd")


def tiny_cfg(B=2, T=6, hw=8, C=40, M=24):
    """Shrunken graph: backbone width 8 -> taps of 64 / 128 / 256 channels."""
    return O.Cfg(batch_size=B, num_steps=T, vf_h=hw, vf_w=hw, H=hw * 8, W=hw * 8, vf_dim=256, c4_dim=128, c3_dim=64,
                 vocab_size=50, v_emb_dim=C, mlp_dim=M, rnn_size=C, glove_dim=12, parse_dim=20,
                 backbone_width=8, backbone_blocks=(1, 1, 2, 1))


def model_kwargs(cfg, dtype="f32", mode="train"):
    return dict(batch_size=cfg.batch_size, num_steps=cfg.num_steps, vf_h=cfg.vf_h, vf_w=cfg.vf_w, H=cfg.H, W=cfg.W,
                vf_dim=cfg.vf_dim, c4_dim=cfg.c4_dim, c3_dim=cfg.c3_dim, vocab_size=cfg.vocab_size,
                v_emb_dim=cfg.v_emb_dim, mlp_dim=cfg.mlp_dim, rnn_size=cfg.rnn_size, glove_dim=cfg.glove_dim,
                parse_dim=cfg.parse_dim, backbone_width=cfg.backbone_width, backbone_blocks=cfg.backbone_blocks,
                start_lr=cfg.start_lr, lr_decay_step=cfg.lr_decay_step, weight_decay=cfg.weight_decay,
                mode=mode, dtype=dtype)


def unpad_map(x, B, h, w, C):
    """product map [B*N, ld] -> oracle NHWC [B,h,w,C] float32 on CPU"""
    return x.detach().float().cpu().view(B, h, w, -1)[..., :C]


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def product_taps_as_oracle(o, cfg):
    """Convert the product's fetch dict to the oracle's tap shapes."""
    B, T, h, w, C, M = cfg.batch_size, cfg.num_steps, cfg.vf_h, cfg.vf_w, cfg.v_emb_dim, cfg.mlp_dim
    N = h * w
    out = {}
    out["words_feat"] = o["words_feat"].detach().float().cpu().view(B, T, -1)[..., :C].reshape(B, 1, T, C)
    out["seq_mask"] = o["seq_mask"].detach().float().cpu().view(B, 1, T, 1)
    out["words_parse"] = o["words_parse"].detach().float().cpu().view(B, 1, T, 4)
    out["nec_lang"] = o["nec_lang"].detach().float().cpu()[:, :C].reshape(B, 1, 1, C)
    for lv in ("c5", "c4", "c3"):
        out[f"lat_{lv}"] = unpad_map(o[f"lat_{lv}"], B, h, w, C)
        out[f"vis_la_sp_{lv}"] = unpad_map(o[f"vis_la_sp_{lv}"], B, h, w, C)
        out[f"spa_graph_{lv}"] = unpad_map(o[f"spa_graph_{lv}"], B, h, w, C)
        out[f"fusion_{lv}"] = unpad_map(o[f"fusion_{lv}"], B, h, w, M)
        out[f"gw_w_{lv}"] = o[f"gw_w_{lv}"].detach().float().cpu()[:, :, :T]
        out[f"gw_v_{lv}"] = o[f"gw_v_{lv}"].detach().float().cpu()[:, :, :T]
        out[f"score_{lv}"] = o[f"score_{lv}"].detach().float().cpu()
        out[f"up_{lv}"] = o[f"up_{lv}"].detach().float().cpu()
    for k in ("exg_c3", "exg_c4", "exg_c5", "exg_c3_2", "exg_c4_2", "exg_c5_2", "fused"):
        out[k] = unpad_map(o[k], B, h, w, M)
    for k in ("pred", "up", "sigm"):
        out[k] = o[k].detach().float().cpu()
    return out
