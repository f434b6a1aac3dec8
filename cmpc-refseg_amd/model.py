"""LSTM_model -- drop-in counterpart of the reference's CMPC_model.LSTM_model (CMPC_model.py:13-492).

Same constructor keywords (CMPC_model.py:15-40), same feeds (words int32[B,T], im f32[B,H,W,3] BGR
minus mean, target_fine f32[B,H,W,1], seq_len int32[B]; :67-71) and fetches (pred, up, sigm :140-142;
up_c3/4/5 :129-133; words_parse :354; gw_w/gw_v :395,399; loss scalars and mIoU :481-491), driven as
    sess.run([train, train_step, merged], feed)   ->  model.train_step(words, im, target_fine, seq_len)
    sess.run([pred, up, sigm], feed)              ->  model.forward(words, im, seq_len)
(trainval_model.py:98-107, test.py:286-296).

The head is ONE C-ABI handle (include/cmpc.h: cmpc_create / cmpc_forward / cmpc_backward / cmpc_optimizer_step):
this class issues three library calls per train step and computes nothing itself; there is no autograd graph and
no Python in the launch path.  The frozen DeepLab-ResNet-101 backbone (HIP implicit-GEMM convolutions driven from
backbone.py, replayed from a captured HIP graph) runs on a side stream beside the text encoder.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, backbone as bb, dist
from ._lib import DT_BF16, DT_F16, DT_F32
from .engine import Engine
from .params import HeadCfg, init_head_params

MU = (104.00698793, 116.66876762, 122.67891434)       # trainval_model.py:371

_LEVELS = ("c5", "c4", "c3")
_EXG = ("c3", "c4", "c5", "c3_2", "c4_2", "c5_2")
_LEVELS_V5 = ("c5", "c4")                              # CMPCv5_BiLSTM_model.py:134-137
_EXG_V5 = ("c4", "c5", "c4_2", "c5_2")                 # :364-375
MODELS = {"CMPC_model": (_lib.MODEL_CMPC, 0), "CMPCv5_BiLSTM_model": (_lib.MODEL_V5_BILSTM, 0), "CMPCv5_BiLSTM_HSV_model": (_lib.MODEL_V5_BILSTM, 1),
          "CMPC_video_mm_tgraph_allvec": (_lib.MODEL_VIDEO, 0)}        # CMPC_video/CMPC_video_mm_tgraph_allvec.py (trainval_video.py:10,35)
FRAME_IDX = (0, 4, 8, 12, 15)                          # CMPC_video_mm_tgraph_allvec.py:70


def tdt(dt: int):
    return {DT_F32: torch.float32, DT_BF16: torch.bfloat16, DT_F16: torch.float16}[dt]


class LSTM_model(object):
    def __init__(self, batch_size=1, num_steps=20, vf_h=40, vf_w=40, H=320, W=320, vf_dim=2048,
                 vocab_size=12112, w_emb_dim=1000, v_emb_dim=1000, mlp_dim=500, start_lr=0.00025,
                 lr_decay_step=800000, lr_decay_rate=1.0, rnn_size=1000, keep_prob_rnn=1.0, keep_prob_emb=1.0,
                 keep_prob_mlp=1.0, num_rnn_layers=1, optimizer='adam', weight_decay=0.0005, mode='eval',
                 conv5=False, glove_dim=300, emb_name='Gref', emb_dir='data',
                 batch_norm_decay=0.9997, freeze_bn=False, is_aug=False,      # CMPCv5_BiLSTM_model.py:42,47,49
                 finetune=False, frames=16,                                   # CMPC_video_mm_tgraph_allvec.py:33,36
                 # --- extensions (not in the reference signature) ---
                 model="CMPC_model",         # which reference module this LSTM_model stands for (get_segmentation_model sets it)
                 aspp_depth=256, low_dim=48, aspp_rates=(6, 12, 18),          # hard-coded in CMPCv5_BiLSTM_model.py:196,208,225
                 device="cuda:0", dtype="f16", c4_dim=1024, c3_dim=512, parse_dim=500,
                 backbone_width=64, backbone_blocks=(3, 4, 23, 3), head_params: Optional[Dict] = None,
                 backbone_params: Optional[Dict] = None, seed=1234, n_lanes: Optional[int] = None, **ignored):
        # `ignored` swallows kwargs the reference driver passes but CMPC_model does not accept
        # (freeze_bn, is_aug: trainval_model.py:40).
        if optimizer != 'adam':
            raise ValueError("Unknown optimizer type %s!" % optimizer)          # CMPC_model.py:458
        self.conv5 = bool(conv5) or bool(finetune)        # the video model calls the same option `finetune` (vid:33,554-557)
        if keep_prob_rnn != 1.0 or keep_prob_emb != 1.0 or keep_prob_mlp != 1.0 or num_rnn_layers != 1:
            raise NotImplementedError("dropout / stacked LSTM are unused by the reference graph")
        if dtype not in ("bf16", "f16", "f32"):
            raise ValueError("dtype must be 'bf16', 'f16' or 'f32'")
        if model not in MODELS:
            raise ValueError("model must be one of %s" % (sorted(MODELS),))
        self.model_name = model
        model_id, hsv = MODELS[model]
        self.v5 = model_id == _lib.MODEL_V5_BILSTM
        self.video = model_id == _lib.MODEL_VIDEO
        self.frames = frames
        if self.video and (batch_size != 1 or frames <= max(FRAME_IDX)):
            raise ValueError("CMPC_video_mm_tgraph_allvec: the graph is only valid for batch_size = 1 (vid:323-324,379) and needs frames > 15 "
                             "(sample indices 0, 4, 8, 12, 15; vid:70)")
        if (freeze_bn or is_aug) and not self.v5:
            freeze_bn = is_aug = False                 # CMPC_model never accepted them (trainval_model.py:40 passes them to every model)
        # is_aug (v5:83-84): tf.image.random_brightness(im, 0.2, seed=42) in train mode -- ONE uniform delta in [-0.2, 0.2) per step, added to
        # the whole image batch.  TensorFlow's random stream cannot be reproduced without TensorFlow: a seeded NumPy generator stands in
        # (same distribution, same determinism; parity-unpinned)
        self.is_aug = bool(is_aug) and mode == 'train'
        self._aug_rng = np.random.default_rng(42)
        if dtype == "bf16":
            # diagnostic mode: same kernels and rate as f16, but 8-bit significands miss BASELINE's 1e-4 mean-IoU bar on some inputs
            # (measured up to 1.6e-4).  The default, f16 storage, meets it.
            warnings.warn("dtype='bf16' is a diagnostic mode: mean-IoU delta vs the fp32 reference path up to 1.6e-4 (bar 1e-4); "
                          "use the default dtype='f16' (same MFMA rate) for parity", stacklevel=2)
        if not torch.cuda.is_available():
            raise RuntimeError("LSTM_model needs an MI355X (gfx950): the CMPC head has no CPU path")
        _lib.load()
        self.mode, self.device = mode, torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dt = {"bf16": DT_BF16, "f16": DT_F16, "f32": DT_F32}[dtype]
        self.batch_size, self.num_steps, self.H, self.W = batch_size, num_steps, H, W
        self.cfg = HeadCfg(batch_size=batch_size, num_steps=num_steps, vf_h=vf_h, vf_w=vf_w, H=H, W=W, vf_dim=vf_dim,
                           c4_dim=c4_dim, c3_dim=c3_dim, vocab_size=vocab_size, v_emb_dim=v_emb_dim, mlp_dim=mlp_dim,
                           rnn_size=rnn_size, glove_dim=glove_dim, parse_dim=parse_dim, start_lr=start_lr,
                           lr_decay_step=lr_decay_step, weight_decay=weight_decay,
                           model=model_id, hsv=hsv, bn_train=int(self.v5 and mode == 'train'), bn_decay=batch_norm_decay,
                           c2_dim=4 * backbone_width, c2_h=-(-H // 4), c2_w=-(-W // 4), aspp_depth=aspp_depth, low_dim=low_dim,
                           aspp_rates=tuple(aspp_rates), sample_frames=len(FRAME_IDX), conv5=int(self.conv5), freeze_bn=int(bool(freeze_bn)))
        for name, v in (("vf_dim", vf_dim), ("c4_dim", c4_dim), ("c3_dim", c3_dim)):
            if v % 64:
                raise ValueError(f"{name}={v} must be a multiple of 64 (MFMA K tile)")
        if n_lanes is None:
            n_lanes = min(3, max(1, int(os.environ.get("CMPC_STREAMS", "3"))))
        with torch.cuda.device(self.device):
            self.eng = Engine(self.cfg, self.dt, self.device, n_lanes=n_lanes)
            self.store = self.eng                         # parameter / gradient accessors (state_dict, grad_dict, p, g, step)
            if head_params is None:
                head_params = self._init_params(seed)
                path = '{}/{}_emb.npy'.format(emb_dir, emb_name)                     # CMPC_model.py:79
                if os.path.exists(path):
                    glove = np.load(path)                                             # allow_pickle=False
                    head_params["text_objseg/Variable"] = torch.from_numpy(np.asarray(glove, dtype=np.float32))
            self.eng.load_state(head_params)
            # backbone on side stream 0 (beside the text encoder); the optimizer on a stream of its own (beside the next
            # step's backbone)
            self._side = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
            self.comm_stream = torch.cuda.Stream(device=self.device)       # gradient all-reduce (data-parallel runs)
            self.bb_stream, self.opt_stream = self._side if n_lanes > 1 else (None, None)
            self.backbone = bb.DeepLabResNet(backbone_width, backbone_blocks)
            if self.v5:
                self.backbone.taps_wanted = ("2b", 4, 5)      # res2b_relu, res4b22_relu, res5c_relu (CMPCv5_BiLSTM_model.py:86-88)
            # the frozen backbone's variables under their TensorFlow names (deeplab_resnet/model.py): kept for checkpoints
            self._backbone_vars = dict(backbone_params if backbone_params is not None else
                                      bb.init_params(backbone_width, backbone_blocks, stem_gamma=1.0 / 256.0 if self.v5 else 1.0))
            self.backbone.load_tf(self._backbone_vars)
            self.backbone = self.backbone.to(self.device).to(tdt(self.dt)).to(memory_format=torch.channels_last).eval()
            self.bb_trainer = None
            if self.conv5:
                # conv5=True (CMPC_model.py:427-430): the res3 / res4 / res5 convolution weights train with the head
                from .backbone_train import BackboneTrainer
                self.bb_trainer = BackboneTrainer(self.backbone, self._backbone_vars, self.device, weight_decay)
        self.world, self.dp_on = 1, False
        self.last = {}
        self._inflight = []
        self._prefetched = None            # (im tensor, feats, done event) of the batch whose backbone pass was enqueued during the previous step
        self._levels_done = None
        self._bb_graph_on = os.environ.get("CMPC_BACKBONE_GRAPH", "1") != "0"
        self._bb = {"calls": 0, "next": 0, "graph": [None, None], "inp": [None, None], "out": [None, None]}
        self._bb_staged = os.environ.get("CMPC_BB_STAGED", "0") != "0"     # three graphs (..res3 | res4 | res5) with an event behind each
        self._tap_events = None
        self._keep = []                # feeds / taps of the steps in flight (the handle reads them asynchronously)

    def _init_params(self, seed):
        """Reference initialisers by variable name (CMPC_model.py:412-417; slim's variance_scaling_initializer for `weights`), driven by the
        handle's own manifest so that both models share it."""
        if not (self.v5 or self.video):
            return init_head_params(self.cfg, seed=seed)
        import math
        g = torch.Generator().manual_seed(seed)
        out = {}
        for name in self.eng.order:
            shape = self.eng.index[name][1]
            leaf = name.rsplit("/", 1)[-1]
            if leaf in ("DW", "kernel", "W_ci", "W_cf", "W_co"):
                rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                fi, fo = (shape[-2] * rf, shape[-1] * rf) if len(shape) >= 2 else (shape[0], shape[0])
                t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / (fi + fo))
            elif leaf == "weights":
                t = torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))
            elif leaf == "gamma":
                t = torch.ones(shape, dtype=torch.float64)
            elif leaf == "Variable":
                t = torch.randn(shape, generator=torch.Generator().manual_seed(7), dtype=torch.float64) * 0.4
            else:
                t = torch.zeros(shape, dtype=torch.float64)
            out[name] = t.float()
        return out

    @property
    def backbone_vars(self):
        """The backbone's variables under their TensorFlow names (checkpoints); with conv5=True the trained res3-res5 weights are current."""
        if getattr(self, "bb_trainer", None) is not None:
            self._backbone_vars.update(self.bb_trainer.named_weights())
        return self._backbone_vars

    @backbone_vars.setter
    def backbone_vars(self, named):
        self._backbone_vars = dict(named)

    # variables for checkpoints besides the head's (checkpoint.py): the batch-norm moving statistics of the v5 graph; with conv5=True the
    # Adam slots of the trained backbone weights (slot names by the same TF1 rule as the head's: parity-unpinned)
    def extra_vars(self):
        out = dict(self.eng.get_state())
        if getattr(self, "bb_trainer", None) is not None:
            for k, v in self.bb_trainer.named_slots().items():
                out["text_objseg/" + k] = v
        return out

    def extra_var_names(self):
        names = tuple(self.eng.state_index)
        if getattr(self, "bb_trainer", None) is not None:
            names += tuple("text_objseg/" + k for k in self.bb_trainer.named_slots())
        return names

    def load_extra_vars(self, named):
        state = {k: v for k, v in named.items() if k in self.eng.state_index}
        if state:
            self.eng.set_state(state)
        if getattr(self, "bb_trainer", None) is not None:
            self.bb_trainer.load_slots({k[len("text_objseg/"):]: v for k, v in named.items() if k.startswith("text_objseg/res")})

    def set_lanes(self, n: int):
        """n = 3: levels / exchange modules on the handle's three lane streams, backbone and optimizer on side streams;
        n = 1: every launch of a step on the caller's stream, in program order (profiling, per-kernel timing)."""
        torch.cuda.synchronize(self.device)
        self.eng.set_lanes(n)
        self.bb_stream, self.opt_stream = self._side if n > 1 else (None, None)

    # ------------------------------------------------------------------------------------------
    def _check_feeds(self, words, im, seq_len, target=None):
        B, T, H, W = self.batch_size, self.num_steps, self.H, self.W
        if tuple(words.shape) != (B, T):
            raise ValueError(f"words must be [{B},{T}], got {tuple(words.shape)}")
        if im is not None and tuple(im.shape) != (B, H, W, 3):
            raise ValueError(f"im must be [{B},{H},{W},3], got {tuple(im.shape)}")
        if tuple(seq_len.shape) != (B,):
            raise ValueError(f"seq_len must be [{B}], got {tuple(seq_len.shape)}")
        if target is not None and tuple(target.shape) != (B, H, W, 1):
            raise ValueError(f"target_fine must be [{B},{H},{W},1], got {tuple(target.shape)}")

    def _dev(self, x, dtype):
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        return x.to(self.device, dtype=dtype, non_blocking=True).contiguous()

    def features(self, im):
        """backbone taps (c3, c4, c5), NHWC, head dtype (CMPC_model.py:73-76)."""
        with torch.cuda.device(self.device):
            return self.backbone(self._dev(im, torch.float32))

    def features_async(self, im, ready=None):
        """Backbone on the side stream so that it overlaps the (sequential, latency-bound) text LSTM that cmpc_forward
        enqueues first.  Returns (feats, event recorded once they are complete, or None when on the caller's stream).
        ready: optional torch.cuda.Event recorded after `im` (a device tensor) was produced; the backbone then waits
        for that event only, not for the tail of the previous train step on the caller's stream."""
        if self.bb_stream is None:
            return self.features(im), None
        main = torch.cuda.current_stream(self.device)
        if ready is not None and not torch.is_tensor(im):
            ready = None
        im = self._dev(im, torch.float32)
        st = self.bb_stream
        if ready is not None:
            st.wait_event(ready)
        else:
            st.wait_stream(main)
        feats = None
        with torch.cuda.stream(st):
            self._tap_events = None
            if self._bb_graph_on and not torch.cuda.is_current_stream_capturing():
                feats = self._backbone_graphed(im, st)
            if feats is None:
                feats = self.backbone(im)
                for f in feats:
                    f.record_stream(main)            # consumed by the handle on `main` and on lanes that join into it
            done = torch.cuda.Event()
            done.record(st)
        im.record_stream(st)
        return feats, done

    def _backbone_graphed(self, im, st):
        """The frozen backbone is a static single-stream chain of ~105 launches: after two eager passes it is replayed
        from a captured HIP graph.  Two graphs with their own static input / output buffers alternate: the taps of
        step n are read until the end of step n (lateral weight gradients) while step n+1's pass may already be
        running.  Returns None while still warming up (caller runs eagerly)."""
        s = self._bb
        if s["calls"] < 2:
            s["calls"] += 1
            return None
        k = s["next"]
        s["next"] = 1 - k
        if s["graph"][k] is None:
            inp = torch.empty_like(im)
            inp.copy_(im)
            torch.cuda.synchronize(self.device)
            if self._bb_staged:
                gs, x, taps = [], inp, {}
                for seg in range(3):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        x, t = self.backbone.forward_segment(x, seg)
                    taps.update(t)
                    gs.append(g)
                    torch.cuda.synchronize(self.device)
                s["graph"][k], s["inp"][k], s["out"][k] = gs, inp, tuple(taps[t] for t in self.backbone.taps_wanted)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    out = self.backbone(inp)
                s["graph"][k], s["inp"][k], s["out"][k] = g, inp, out
        s["inp"][k].copy_(im, non_blocking=True)
        if isinstance(s["graph"][k], list):          # staged: an event behind every segment, so that a level starts when ITS tap is complete
            evs = []
            for g in s["graph"][k]:
                g.replay()
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            want = self.backbone.taps_wanted
            seg_of = lambda t: 0 if (t == 3 or (isinstance(t, str) and t[0] in "23")) else (1 if t == 4 else 2)
            # per pyramid level (c5, c4, third slot = c3 or the res2b tap): the event of the segment that produces it
            self._tap_events = [evs[seg_of(want[2])], evs[seg_of(want[1])], evs[seg_of(want[0])]]
        else:
            s["graph"][k].replay()
        return s["out"][k]

    # ------------------------------------------------------------------------------------------
    def _fetch_dict(self, with_loss: bool):
        """Zero-copy views of the handle's intermediates under the names the parity tests and the reference's
        visualisers use (test_visualize_graph.py:243-253)."""
        t = self.eng.tap
        out = {k: t(k) for k in ("words_feat", "seq_mask", "words_parse", "nec_lang", "fused", "pred", "up", "sigm", "iu")}
        for lv in (_LEVELS_V5 if self.v5 else _LEVELS):
            for k in ("lat", "vis_la_sp", "spa_graph", "fusion", "gw_w", "gw_v", "score", "up"):
                out[f"{k}_{lv}"] = t(f"{k}_{lv}")
        for x in (_EXG_V5 if self.v5 else _EXG):
            out[f"exg_{x}"] = t(f"exg_{x}")
        if self.v5:
            for k in ("bilstm_fw", "bilstm_bw", "aspp_branches", "aspp_image", "aspp", "dec_cat", "dec_net2") + (("hsv",) if self.cfg.hsv else ()):
                out[k] = t(k)
        if self.video:
            out["ac_lang"] = t("ac_lang")
            for lv in _LEVELS:
                for k in ("mm", "tg_pool", "tgraph", "temp_ctx"):
                    out[f"{k}_{lv}"] = t(f"{k}_{lv}")
        if with_loss:
            s = t("scalars")
            out.update(loss_all=s[0], loss_c3=s[1], loss_c4=s[2], loss_c5=s[3], loss_last=s[4], mIoU=s[5])
        return out

    def head(self, feats, words, seq_len, target=None, after=None, im=None, levels_done=None):
        """build_graph() on given backbone taps (CMPC_model.py:89-142): one cmpc_forward call.  Returns the fetch dict
        (views into the handle's workspace: valid until the next call).  `after`: event that marks `feats` complete.
        feats = (c3, c4, c5), or (c2, c4, c5) for the CMPCv5_BiLSTM models, whose HSV variant also needs the image feed `im`."""
        with torch.cuda.device(self.device):
            f0, c4, c5 = [f.to(tdt(self.dt)).contiguous() for f in feats] if after is None else feats
            w = self._dev(words, torch.int32)
            sl = self._dev(seq_len, torch.int32)
            tg = self._dev(target, torch.float32) if target is not None else None
            imd = None
            if self.v5 and self.cfg.hsv:
                if im is None:
                    raise ValueError("CMPCv5_BiLSTM_HSV_model needs the image feed (CMPCv5_BiLSTM_HSV_model.py:120-126)")
                imd = self._dev(im, torch.float32)
            if self.v5:
                self.eng.forward(w, sl, None, c4, c5, tg, feats_ready=after, c2=f0, im=imd, levels_done=levels_done, feats_ready_lv=self._tap_events if after is not None else None)
            else:
                self.eng.forward(w, sl, f0, c4, c5, tg, feats_ready=after, levels_done=levels_done, feats_ready_lv=self._tap_events if after is not None else None)
            self._keep.append((w, sl, tg, f0, c4, c5, imd))
            del self._keep[:-3]
            return self._fetch_dict(tg is not None)

    @torch.no_grad()
    def forward(self, words, im, seq_len):
        """sess.run([pred, up, sigm, ...], {words, im, seq_len}) (test.py:286-296)."""
        self._check_feeds(words, im, seq_len)
        with torch.cuda.device(self.device):
            imd = self._dev(im, torch.float32)
            feats, ev = self.features_async(imd)
            o = self.head(feats, words, seq_len, after=ev, im=imd)
            B, T = self.batch_size, self.num_steps
            last = "c4" if self.v5 else "c3"            # the reference keeps the attributes of the LAST level built (CMPC_model.py:395,399)
            out = {"pred": o["pred"].clone(), "up": o["up"].clone(), "sigm": o["sigm"].clone(),
                   "up_c4": o["up_c4"].clone(), "up_c5": o["up_c5"].clone(),
                   "words_parse": o["words_parse"].view(B, 1, T, 4).clone(),
                   "gw_w": o[f"gw_w_{last}"][:, :, :T].clone(), "gw_v": o[f"gw_v_{last}"][:, :, :T].clone()}
            if not self.v5:
                out["up_c3"] = o["up_c3"].clone()
            return out

    # ---- CMPC_video_mm_tgraph_allvec: feeds words (FRONT-padded), im (unused by the graph), target_fine, valid_idx, clip (vid:62-66) -------
    def _video_feeds(self, words, valid_idx, clip):
        """The graph drops the pad steps (vid:125-142): the same LSTM over the valid words only.  words [1, T] front-padded with
        valid_idx[0, 0] pad words -> END-padded ids + seq_len for the handle; clip [1, frames, H, W, 3] -> the 5 sampled frames."""
        T, H, W = self.num_steps, self.H, self.W
        w = np.asarray(words.cpu() if torch.is_tensor(words) else words).astype(np.int64).reshape(1, T)
        vi = int(np.asarray(valid_idx.cpu() if torch.is_tensor(valid_idx) else valid_idx).reshape(-1)[0])
        if vi < 0 or vi >= T or np.any(w[0, :vi] != 0) or np.any(w[0, vi:] == 0):
            raise ValueError("words must be front-padded with exactly valid_idx[0, 0] pad ids (util/text_processing.py:42-53)")
        n = T - vi
        we = np.zeros((1, T), dtype=np.int32)
        we[0, :n] = w[0, vi:]
        if tuple(clip.shape) != (1, self.frames, H, W, 3):
            raise ValueError(f"clip must be [1,{self.frames},{H},{W},3], got {tuple(clip.shape)}")
        fr = self._dev(clip, torch.float32)[0, list(FRAME_IDX)].contiguous()           # tf.gather(clip, [0, 4, 8, 12, 15], axis=1) (vid:70)
        return torch.from_numpy(we), torch.tensor([n], dtype=torch.int32), fr

    def head_video(self, feats, words_end, seq_len, target=None, after=None):
        """build_graph() (vid:91-187) on the taps of the 5 sampled frames, feats = (c3, c4, c5) each [5, h, w, .]."""
        return self.head(feats, words_end, seq_len, target, after=after)

    @torch.no_grad()
    def forward_video(self, words, im, valid_idx, clip):
        """sess.run([pred, up, sigm], {words, im, valid_idx, clip}) (trainval_video.py:214-220)."""
        with torch.cuda.device(self.device):
            we, sl, fr = self._video_feeds(words, valid_idx, clip)
            feats, ev = self.features_async(fr)
            o = self.head(feats, we, sl, after=ev)
            return {"pred": o["pred"].clone(), "up": o["up"].clone(), "sigm": o["sigm"].clone()}

    def train_step_video(self, words, im, target_fine, valid_idx, clip):
        """sess.run([train_step, cls_loss, learning_rate, pred, target], feed) (trainval_video.py:93-101)."""
        if self.mode != 'train':
            raise RuntimeError("model was built with mode='eval'")
        with torch.cuda.device(self.device):
            if len(self._inflight) >= self.MAX_STEPS_IN_FLIGHT:
                self._inflight.pop(0).synchronize()
            we, sl, fr = self._video_feeds(words, valid_idx, clip)
            if self.bb_trainer is not None:
                return self._train_step_conv5(we, fr, target_fine, sl)
            feats, ev = self.features_async(fr)
            self.loss_and_grads(feats, we, target_fine, sl, after=ev)
            sv = self.eng.tap("scalars").clone()
            ost = self.opt_stream if self.opt_stream is not None else torch.cuda.current_stream(self.device)
            for b in range(self.eng.n_buckets):
                with torch.cuda.stream(ost):
                    lr = self.eng.optimizer_bucket(b, 1.0)
            done = torch.cuda.Event()
            done.record(ost)
            self._inflight.append(done)
        scal = {k: sv[i] for i, k in enumerate(self._SCALARS)}
        scal["mean_IOU"] = scal.pop("mIoU")
        scal["learning_rate"] = lr
        return self.eng.step, scal

    def predict(self, images, sentences, sequence_lenghts):
        """TF-serving signature of export_model_serving.py:57-71: images, sentences, sequence_lenghts -> masks."""
        return self.forward(sentences, images, sequence_lenghts)["sigm"]

    def loss_and_grads(self, feats, words, target_fine, seq_len, after=None, im=None, between=None):
        """forward + backward of `cost` (CMPC_model.py:447) into the flat gradient buffer: cmpc_forward + cmpc_backward
        (L2 and the x2 bias multiplier are applied inside the Adam kernel)."""
        ld = None
        if between is not None:                  # work the caller wants enqueued behind the levels' forward (the next batch's backbone)
            if self._levels_done is None:
                self._levels_done = torch.cuda.Event()
                self._levels_done.record()       # materialises the hipEvent_t the handle records
            ld = self._levels_done
        o = self.head(feats, words, seq_len, target_fine, after=after, im=im, levels_done=ld)
        with torch.cuda.device(self.device):
            if between is not None:
                between(ld)
            self.eng.backward()
        return o

    _SCALARS = ("loss_all", "loss_c3", "loss_c4", "loss_c5", "loss_last", "mIoU")
    MAX_STEPS_IN_FLIGHT = int(os.environ.get("CMPC_STEPS_IN_FLIGHT", "2"))

    def train_step(self, words, im, target_fine, seq_len, ready=None, next_im=None, next_ready=None, next_gate="fwd"):
        """sess.run([train, train_step, merged], feed) (trainval_model.py:98-107).
        ready: optional torch.cuda.Event recorded once the (device-resident, prefetched) feeds were complete.
        next_im: the NEXT step's image batch, already resident on the device (a prefetching loader has it: util/data_reader_refvos.py:38-46),
        next_ready its event.  The frozen backbone of that batch is then enqueued behind THIS step's levels' forward, where the step is a
        serial chain of small launches (exchange modules, ConvLSTM, scores) until the levels' backward; the next call must pass the same
        tensor as `im` and finds its taps ready.  Same arithmetic, same results: only the order on the device changes."""
        if self.mode != 'train':
            raise RuntimeError("model was built with mode='eval' (CMPC_model.py:85-86)")
        self._check_feeds(words, im, seq_len, target_fine)
        with torch.cuda.device(self.device):
            # bound the host's lead: the handle's workspace is static, so a third step must not be enqueued while the
            # first one's feeds / backbone taps (two alternating buffer sets) may still be read
            if len(self._inflight) >= self.MAX_STEPS_IN_FLIGHT:
                self._inflight.pop(0).synchronize()
            if not torch.is_tensor(im):
                ready = None                    # host feeds: the copy below is ordered on the caller's stream, which the backbone then waits for
            imd = self._dev(im, torch.float32)
            if self.is_aug:                       # v5:83-84: the augmented image replaces self.im for the backbone AND the HSV branch
                imd = imd + float(self._aug_rng.uniform(-0.2, 0.2))
                im = imd
            if self.bb_trainer is not None:
                return self._train_step_conv5(words, imd, target_fine, seq_len)
            pf, self._prefetched = self._prefetched, None
            if pf is not None and pf[0] is im:
                feats, ev = pf[1], pf[2]
            else:
                feats, ev = self.features_async(imd, ready)
            between = None
            pre = next_im is not None and torch.is_tensor(next_im) and next_im.is_cuda and self.bb_stream is not None
            if pre and next_gate == "fwd":
                def between(levels_done, nxt=next_im, nready=next_ready):
                    self.bb_stream.wait_event(levels_done)
                    f2, e2 = self.features_async(nxt, nready if nready is not None else levels_done)
                    self._prefetched = (nxt, f2, e2)
            if pre and next_gate == "bwd" and getattr(self, "_bwd_levels", None) is None:
                self._bwd_levels = torch.cuda.Event()
                self._bwd_levels.record()
                _lib.call("cmpc_set_bwd_levels_event", self.eng.h, ctypes.c_void_p(self._bwd_levels.cuda_event))
            self.loss_and_grads(feats, words, target_fine, seq_len, after=ev, im=imd, between=between)
            if pre and next_gate == "bwd":          # behind the levels' backward: beside the grouped dW launch and the text encoder's backward
                self.bb_stream.wait_event(self._bwd_levels)
                f2, e2 = self.features_async(next_im, next_ready if next_ready is not None else self._bwd_levels)
                self._prefetched = (next_im, f2, e2)
            sv = self.eng.tap("scalars").clone()
            # Optimizer, bucket by bucket in the order the backward pass finalises them (exchange modules + ConvLSTM, levels c5 / c4 /
            # c3, text encoder): every bucket's Adam + repack waits on the device for that bucket only, so all but the last run
            # beside the rest of the backward pass; on the optimizer stream, so the next step's backbone does not wait for them.
            # Data-parallel: the bucket's all-reduce (RCCL over xGMI, communication stream) goes in between; 1/world in the Adam kernel.
            cur = torch.cuda.current_stream(self.device)
            ost = self.opt_stream if self.opt_stream is not None else cur
            gscale = 1.0 / self.world
            buckets = self.eng.grad_buckets() if self.dp_on else None
            for b in range(self.eng.n_buckets):
                if self.dp_on:
                    dist.allreduce_bucket_(self.eng, b, buckets[b], self.comm_stream)
                    ost.wait_stream(self.comm_stream)
                with torch.cuda.stream(ost):
                    lr = self.eng.optimizer_bucket(b, gscale)
            done = torch.cuda.Event()
            done.record(self.opt_stream if self.opt_stream is not None else torch.cuda.current_stream(self.device))
            self._inflight.append(done)
        scal = {k: sv[i] for i, k in enumerate(self._SCALARS)}
        scal["mean_IOU"] = scal.pop("mIoU")
        scal["learning_rate"] = lr
        self.last = scal
        return self.eng.step, scal

    def _train_step_conv5(self, words, imd, target_fine, seq_len):
        """conv5=True (CMPC_model.py:427-430, v5:521-525; finetune=True of the video model, vid:554-557): backbone forward with its
        activations kept, head forward / backward (which also returns d cost / d taps), backbone backward, one TF-Adam step over the head and
        the res3-res5 convolution weights.  Everything on the caller's stream (no side streams: the optional mode is not the benchmarked
        one).  imd: the image batch -- for the video model the 5 sampled frames."""
        h, w = self.cfg.vf_h, self.cfg.vf_w
        nb = imd.shape[0]
        feats = self.bb_trainer.forward(imd)
        self.loss_and_grads(feats, words, target_fine, seq_len, im=imd if (self.v5 and self.cfg.hsv) else None)
        sv = self.eng.tap("scalars").clone()
        dt = {5: self.eng.tap("dc5").view(nb, h, w, -1), 4: self.eng.tap("dc4").view(nb, h, w, -1)}
        if not self.v5:
            dt[3] = self.eng.tap("dc3").view(nb, h, w, -1)
        self.bb_trainer.backward(dt)
        step0 = self.eng.step
        cur = torch.cuda.current_stream(self.device)
        gscale = 1.0 / self.world
        buckets = self.eng.grad_buckets() if self.dp_on else None
        for b in range(self.eng.n_buckets):
            if self.dp_on:                      # data-parallel: the head's buckets as in train_step, then the backbone's gradient buffer
                dist.allreduce_bucket_(self.eng, b, buckets[b], self.comm_stream)
                cur.wait_stream(self.comm_stream)
            lr = self.eng.optimizer_bucket(b, gscale)
        if self.dp_on:
            for o in range(0, self.bb_trainer.total, 16 << 20):
                torch.distributed.all_reduce(self.bb_trainer.grads[o: o + (16 << 20)])
        t = float(step0 + 1)
        lr_t = lr * (1.0 - 0.999 ** t) ** 0.5 / (1.0 - 0.9 ** t)
        self.bb_trainer.adam(lr_t, gscale / self.eng.loss_scale)
        scal = {k: sv[i] for i, k in enumerate(self._SCALARS)}
        scal["mean_IOU"] = scal.pop("mIoU")
        scal["learning_rate"] = lr
        self.last = scal
        return self.eng.step, scal

    def grad_nonfinite(self) -> int:
        """Gradient elements the LAST optimizer step skipped because they were inf / nan (f16 storage overflow; synchronises).  Non-zero =
        lower the loss scale; parameter and Adam moments of those elements were left untouched (cmpc_adam_step)."""
        torch.cuda.synchronize(self.device)
        return int(self.eng.tap("grad_nonfinite").sum())

    # ------------------------------------------------------------------------------------------
    def state_dict(self):
        return self.eng.state_dict()

    def load_weights(self, named: Dict[str, torch.Tensor]):
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            self.eng.load_state(named)

    def load_backbone(self, named: Dict[str, torch.Tensor]):
        """Restore the frozen backbone from TensorFlow-named variables (`conv1/weights`, `bn2a_branch2a/gamma`, ...:
        trainval_model.py:50-54 loads exactly this subset from deeplab_resnet_init.ckpt).  Frozen batch-norms are re-folded; the
        captured backbone graphs stay valid (weights are updated in place)."""
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            self._backbone_vars = dict(named)
            self.backbone.load_tf(self._backbone_vars)
            if getattr(self, "bb_trainer", None) is not None:
                self.bb_trainer.load(self._backbone_vars)
            torch.cuda.synchronize(self.device)

    def enable_data_parallel(self):
        """One process per GPU; rank 0's weights are broadcast; gradients are summed bucket by bucket while the backward pass
        runs (dist.allreduce_bucket_) and divided by the world size in the Adam kernel.  Active whenever a process group exists -- also a
        group of ONE rank, so that the RCCL path (collectives on engine-owned hipMalloc memory, issued on the communication stream
        behind the bucket events) can be exercised on a single GPU."""
        self.world = dist.world_size()
        self.dp_on = dist.is_initialized()
        if self.dp_on:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                dist.broadcast_params_(self.eng.params, 0)
                self.eng.pack()
                if getattr(self, "bb_trainer", None) is not None:        # conv5: the trained backbone weights start from rank 0's as well
                    dist.broadcast_params_(self.bb_trainer.params, 0)
                    self.bb_trainer.refold()
        return self.world


def get_segmentation_model(name, **kwargs):
    """get_model.get_segmentation_model (get_model.py:15-17): name -> <module>.LSTM_model(**kwargs).  Built: CMPC_model,
    CMPCv5_BiLSTM_model, CMPCv5_BiLSTM_HSV_model (get_model.py:1,10,11)."""
    if name not in MODELS:
        raise ValueError("model %r is not built (have: %s)" % (name, ", ".join(sorted(MODELS))))
    return LSTM_model(model=name, **kwargs)
