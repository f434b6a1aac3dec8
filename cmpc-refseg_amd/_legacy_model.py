"""LSTM_model -- drop-in counterpart of the reference's CMPC_model.LSTM_model (CMPC_model.py:13-492).

Same constructor keywords (CMPC_model.py:15-40), same feeds (words int32[B,T], im f32[B,H,W,3] BGR
minus mean, target_fine f32[B,H,W,1], seq_len int32[B]; :67-71) and fetches (pred, up, sigm :140-142;
up_c3/4/5 :129-133; words_parse :354; gw_w/gw_v :395,399; loss scalars and mIoU :481-491), driven as
    sess.run([train, train_step, merged], feed)   ->  model.train_step(words, im, target_fine, seq_len)
    sess.run([pred, up, sigm], feed)              ->  model.forward(words, im, seq_len)
(trainval_model.py:98-107, test.py:286-296).  The head runs on the HIP kernels of libcmpc_hip.so; the
frozen DeepLab-ResNet-101 backbone runs on PyTorch-ROCm.  There is no CPU fallback.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, backbone as bb, dist, ops
from ._lib import DT_BF16, DT_F32
from .params import EXG, LEVELS, HeadCfg, ParamStore, init_head_params

MU = (104.00698793, 116.66876762, 122.67891434)       # trainval_model.py:371


class _OnMain(torch.autograd.Function):
    """Identity placed on the main stream between two forked phases.  autograd runs a node's backward on the
    stream of its forward, so with this node every cross-lane gradient goes lane -> main -> lane; direct
    lane -> lane event edges inside a stream capture crash hipStreamEndCapture on ROCm 7.2."""
    @staticmethod
    def forward(ctx, x, model=None, tag=None):
        ctx.model, ctx.tag = model, tag
        if model is not None:
            model._mark(tag + ":fwd")
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.model is not None:
            ctx.model._mark(ctx.tag + ":bwd")
        return g, None, None


class LSTM_model(object):
    def __init__(self, batch_size=1, num_steps=20, vf_h=40, vf_w=40, H=320, W=320, vf_dim=2048,
                 vocab_size=12112, w_emb_dim=1000, v_emb_dim=1000, mlp_dim=500, start_lr=0.00025,
                 lr_decay_step=800000, lr_decay_rate=1.0, rnn_size=1000, keep_prob_rnn=1.0, keep_prob_emb=1.0,
                 keep_prob_mlp=1.0, num_rnn_layers=1, optimizer='adam', weight_decay=0.0005, mode='eval',
                 conv5=False, glove_dim=300, emb_name='Gref', emb_dir='data',
                 # --- extensions (not in the reference signature) ---
                 device="cuda:0", dtype="bf16", c4_dim=1024, c3_dim=512, parse_dim=500,
                 backbone_width=64, backbone_blocks=(3, 4, 23, 3), head_params: Optional[Dict] = None,
                 backbone_params: Optional[Dict] = None, seed=1234, **ignored):
        # `ignored` swallows kwargs the reference driver passes but CMPC_model does not accept
        # (freeze_bn, is_aug: trainval_model.py:40).
        if optimizer != 'adam':
            raise ValueError("Unknown optimizer type %s!" % optimizer)          # CMPC_model.py:458
        if conv5:
            raise NotImplementedError("conv5=True (backbone fine-tuning, CMPC_model.py:427-430) is out of scope")
        if keep_prob_rnn != 1.0 or keep_prob_emb != 1.0 or keep_prob_mlp != 1.0 or num_rnn_layers != 1:
            raise NotImplementedError("dropout / stacked LSTM are unused by the reference graph")
        if dtype not in ("bf16", "f32"):
            raise ValueError("dtype must be 'bf16' or 'f32'")
        if not torch.cuda.is_available():
            raise RuntimeError("LSTM_model needs an MI355X (gfx950): the CMPC head has no CPU path")
        _lib.load()
        self.mode, self.device = mode, torch.device(device)
        self.dt = DT_BF16 if dtype == "bf16" else DT_F32
        self.batch_size, self.num_steps, self.H, self.W = batch_size, num_steps, H, W
        self.cfg = HeadCfg(batch_size=batch_size, num_steps=num_steps, vf_h=vf_h, vf_w=vf_w, H=H, W=W, vf_dim=vf_dim,
                           c4_dim=c4_dim, c3_dim=c3_dim, vocab_size=vocab_size, v_emb_dim=v_emb_dim, mlp_dim=mlp_dim,
                           rnn_size=rnn_size, glove_dim=glove_dim, parse_dim=parse_dim, start_lr=start_lr,
                           lr_decay_step=lr_decay_step, weight_decay=weight_decay)
        for name, v in (("vf_dim", vf_dim), ("c4_dim", c4_dim), ("c3_dim", c3_dim)):
            if v % 64:
                raise ValueError(f"{name}={v} must be a multiple of 64 (MFMA K tile)")
        self.store = ParamStore(self.cfg, self.device, self.dt)
        if head_params is None:
            head_params = init_head_params(self.cfg, seed=seed)
            path = '{}/{}_emb.npy'.format(emb_dir, emb_name)                     # CMPC_model.py:79
            if os.path.exists(path):
                glove = np.load(path)                                             # allow_pickle=False
                head_params["text_objseg/Variable"] = torch.from_numpy(np.asarray(glove, dtype=np.float32))
        self.store.load_state(head_params)
        self.cx = ops.Ctx(self.cfg, self.store, self.dt)
        # the three pyramid levels (and the three exchange modules of a round) are independent: each gets its own
        # HIP stream; weight-gradient GEMMs go to a fourth one.  autograd replays every backward on its forward's stream.
        self.set_streams(int(os.environ.get("CMPC_STREAMS", "3")))
        self.backbone = bb.DeepLabResNet(backbone_width, backbone_blocks)
        self.backbone.load_tf(backbone_params if backbone_params is not None else bb.init_params(backbone_width, backbone_blocks))
        self.backbone = self.backbone.to(self.device).to(ops.tdt(self.dt)).to(memory_format=torch.channels_last).eval()
        self.world = 1
        self.last = {}
        # CMPC_GRAPH=1: train_step replays forward + backward from ONE captured HIP graph (about 1200 launches
        # per step, no Python in the loop).  Off by default: on ROCm 7.2 the replay of the 4-stream graph is
        # slower (13.6-14.1 ms) than eager launches on the same 4 streams (11.9 ms); the one-stream graph takes 14.8 ms.
        self.use_graph = os.environ.get("CMPC_GRAPH", "0") != "0"
        self._graph, self._gin, self._gout, self._eager_steps = None, None, None, 0
        self._opt_pending = False
        self._opt_stage0 = None
        self._inflight = []
        self._bb_graph_on = os.environ.get("CMPC_BACKBONE_GRAPH", "1") != "0"
        self._bb = {"calls": 0, "next": 0, "graph": [None, None], "inp": [None, None], "out": [None, None]}
        self.marks = [] if os.environ.get("CMPC_MARKS") else None
        # the three pyramid levels (and the three exchange modules of a round) are independent: each gets
        # its own HIP stream so HBM-bound stage kernels of one level overlap MFMA-bound GEMMs of another.
        # autograd replays every backward on the stream of its forward, so the backward overlaps too.

    _SIDE_STREAMS = {}          # device -> 3 side streams shared by every model on that device

    def set_streams(self, n: int):
        """n > 1: independent levels / exchange modules run on 3 side streams; n = 1: everything on the caller's stream."""
        if getattr(self, "_opt_pending", False):
            self._params_ready()
        self.n_streams = n
        if n > 1:
            key = str(self.device)
            if key not in LSTM_model._SIDE_STREAMS:
                LSTM_model._SIDE_STREAMS[key] = [torch.cuda.Stream(device=self.device) for _ in range(3)]
            self.side = LSTM_model._SIDE_STREAMS[key]
            if key + "/opt" not in LSTM_model._SIDE_STREAMS:
                LSTM_model._SIDE_STREAMS[key + "/opt"] = torch.cuda.Stream(device=self.device)
            self.opt_stream = LSTM_model._SIDE_STREAMS[key + "/opt"]
            # CMPC_WGRAD_OVERLAP=1 starts the weight-gradient flush beside the text encoder's backward.  Off by default:
            # it gains nothing measurable and in about one run out of three the serial chain of small kernels then
            # crawls behind the long-running grouped kernel (13.5 -> 51 ms per step; hardware-queue scheduling).
            early = os.environ.get("CMPC_WGRAD_OVERLAP", "0") != "0"
            self.cx.flush_stream, self.cx.lanes = (self.side[1] if early else None), tuple(self.side)
            if key + "/wg" not in LSTM_model._SIDE_STREAMS:
                LSTM_model._SIDE_STREAMS[key + "/wg"] = torch.cuda.Stream(device=self.device)
            self.cx.wg = LSTM_model._SIDE_STREAMS[key + "/wg"] if os.environ.get("CMPC_WGRAD_STREAM", "0") != "0" else None
        else:
            self.side = None
            self.opt_stream = None
            self.cx.flush_stream, self.cx.lanes = None, ()
            self.cx.wg = None

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
        return self.backbone(self._dev(im, torch.float32))

    def _mark(self, name):
        """CMPC_MARKS=1: timing events at the phase boundaries of a step (scripts/step_timeline.py)."""
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.marks.append((name, ev))

    def _params_ready(self, stage=1):
        """The optimizer of the previous train_step runs on its own stream (it overlaps the next step's frozen
        backbone); everything that touches parameters, packed operands or the gradient buffer waits for it here.
        stage 0: Adam done and the text encoder's / parser's operands repacked (all the LSTM needs, so it runs
        while the level weights are still being packed); stage 1: everything."""
        if self._opt_pending:
            cur = torch.cuda.current_stream(self.device)
            if stage == 0 and self._opt_stage0 is not None:
                cur.wait_event(self._opt_stage0)
                return
            cur.wait_stream(self.opt_stream)
            self._opt_pending = False

    def features_async(self, im, ready=None):
        """Backbone on side stream 0 so that it overlaps the (sequential, latency-bound) text LSTM on the
        caller's stream.  Returns (feats, stream-to-wait-on or None).
        ready: optional torch.cuda.Event recorded after `im` (a device tensor) was produced; the backbone then
        waits for that event only, not for everything queued on the caller's stream (the tail of the previous
        train step), which is how a prefetched batch overlaps the previous step's backward tail and optimizer."""
        if self.side is None:
            return self.features(im), None
        main = torch.cuda.current_stream(self.device)
        if ready is not None and not torch.is_tensor(im):
            ready = None
        im = self._dev(im, torch.float32)
        st = self.side[0]
        if ready is not None:
            st.wait_event(ready)
        else:
            st.wait_stream(main)
        if self._bb_graph_on and not self.use_graph and not torch.cuda.is_current_stream_capturing():
            with torch.cuda.stream(st):
                feats = self._backbone_graphed(im, st)
                self._mark("backbone")
            im.record_stream(st)
            if feats is not None:
                return feats, st
        with torch.cuda.stream(st):
            feats = self.backbone(im)
            self._mark("backbone")
        for f in feats:
            f.record_stream(main)
            for s2 in self.side:
                f.record_stream(s2)
        im.record_stream(st)
        return feats, st

    def _backbone_graphed(self, im, st):
        """The frozen backbone is a static single-stream chain of ~105 launches: after two eager passes it is replayed
        from a captured HIP graph (the host then spends ~0.1 ms on it instead of ~2.7 ms; a single-stream graph replays
        at eager speed on the GPU).  Two graphs with their own static input / output buffers alternate: the taps of step n
        are read until the end of step n (lateral weight gradients) while step n+1's pass may already be running.
        Returns None while still warming up (caller runs eagerly)."""
        bb = self._bb
        if bb["calls"] < 2:
            bb["calls"] += 1
            return None
        k = bb["next"]
        bb["next"] = 1 - k
        if bb["graph"][k] is None:
            inp = torch.empty_like(im)
            inp.copy_(im)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                out = self.backbone(inp)
            bb["graph"][k], bb["inp"][k], bb["out"][k] = g, inp, out
        bb["inp"][k].copy_(im, non_blocking=True)
        bb["graph"][k].replay()
        return bb["out"][k]

    def head(self, feats, words, seq_len, target=None, after=None):
        """build_graph() on given backbone taps (CMPC_model.py:89-142).  Returns the fetch dict.
        `after`: stream that produces `feats` (waited for once the text encoder has been queued)."""
        cfg, cx, O = self.cfg, self.cx, ops
        B, T, N = cfg.batch_size, cfg.num_steps, cfg.N
        self._params_ready(0)
        if after is None:
            c3, c4, c5 = [f.to(ops.tdt(self.dt)).contiguous() for f in feats]
        words = self._dev(words, torch.int32).view(-1)
        seq_len = self._dev(seq_len, torch.int32)
        tgt = self._dev(target, torch.float32) if target is not None else None
        a = cx.anchor
        wf, mask = O.TextEncoder.apply(a, words, seq_len, cx)
        if self.marks is not None:
            wf = _OnMain.apply(wf, self, "text")
        parse = O.LangParser.apply(wf, mask, cx)
        vl = O.LangPool.apply(parse, wf, 2, cx)                 # valid_lang: entity + attribute
        self._mark("text_fwd_done")
        self._params_ready(1)
        out = {"words_feat": wf, "seq_mask": mask, "words_parse": parse}
        fus, losses = {}, {}
        main = torch.cuda.current_stream(self.device)
        if after is not None:
            main.wait_stream(after)
            c3, c4, c5 = [f.to(ops.tdt(self.dt)).contiguous() for f in feats]

        fmask = int(os.environ.get("CMPC_FORK", "7"))          # debug: bit 0 levels, 1 / 2 exchange rounds
        phase = [0]

        def fork(i):
            """run a block on side stream i after everything queued so far on the main stream"""
            if self.side is None or not (fmask >> phase[0]) & 1:
                return main
            st = self.side[i]
            st.wait_stream(main)
            return st

        def join():
            if self.side is not None:
                for st in self.side:
                    main.wait_stream(st)

        for i, (lv, f) in enumerate((("c5", c5), ("c4", c4), ("c3", c3))):
            with torch.cuda.stream(fork(i)):
                X0 = O.Lateral.apply(a, f, lv, cx)
                X1 = O.Mutan.apply(X0, vl, lv, cx)
                X2, gw_w, gw_v = O.SpaGraph.apply(X1, wf, parse, mask, lv, cx)
                fus[lv] = O.Fusion.apply(X1, X2, vl, lv, cx)
                out[f"lat_{lv}"], out[f"vis_la_sp_{lv}"], out[f"spa_graph_{lv}"], out[f"fusion_{lv}"] = X0, X1, X2, fus[lv]
                out[f"gw_w_{lv}"], out[f"gw_v_{lv}"] = gw_w, gw_v
                l, sc, up, _s, _iu = O.ScoreHead.apply(fus[lv], f"score_{lv}", tgt, 0.1, cx)
                out[f"score_{lv}"], out[f"up_{lv}"], losses[lv] = sc, up, l
        join()
        fus = {k: _OnMain.apply(v, self if self.marks is not None and k == "c5" else None, "levels") for k, v in fus.items()}
        phase[0] = 1
        nec = O.LangPool.apply(parse, wf, 3, cx)                # nec_lang: entity + attribute + relation
        out["nec_lang"] = nec
        f3, f4, f5 = fus["c3"], fus["c4"], fus["c5"]
        ex = {}
        for i, (nm, fa, fb, fc) in enumerate((("c3", f3, f4, f5), ("c4", f4, f3, f5), ("c5", f5, f3, f4))):
            with torch.cuda.stream(fork(i)):
                ex[nm] = O.Exchange.apply(fa, fb, fc, nec, nm, cx)
        join()
        phase[0] = 2
        e3, e4, e5 = (_OnMain.apply(ex[k], self if self.marks is not None and k == "c5" else None, "exch1") for k in ("c3", "c4", "c5"))
        for i, (nm, fa, fb, fc) in enumerate((("c3_2", e3, e4, e5), ("c4_2", e4, e3, e5), ("c5_2", e5, e3, e4))):
            with torch.cuda.stream(fork(i)):
                ex[nm] = O.Exchange.apply(fa, fb, fc, nec, nm, cx)
        join()
        e32, e42, e52 = ex["c3_2"], ex["c4_2"], ex["c5_2"]
        out.update(exg_c3=e3, exg_c4=e4, exg_c5=e5, exg_c3_2=e32, exg_c4_2=e42, exg_c5_2=e52)
        self._mark("exch2_done")
        fused = O.ConvLSTM.apply(e32, e42, e52, cx)
        out["fused"] = fused
        l, pred, up, sigm, iu = O.ScoreHead.apply(fused, "score", tgt, 0.7, cx)
        out.update(pred=pred, up=up, sigm=sigm, iu=iu)
        self._mark("fwd_done")
        if tgt is not None:
            # cls_loss_all = 0.7 L + 0.1 (L_c5 + L_c4 + L_c3), CMPC_model.py:444-445 (scalar bookkeeping only)
            out["loss_last"], out["loss_c5"], out["loss_c4"], out["loss_c3"] = l.mean(), losses["c5"].mean(), losses["c4"].mean(), losses["c3"].mean()
            out["loss_all"] = 0.7 * out["loss_last"] + 0.1 * out["loss_c5"] + 0.1 * out["loss_c4"] + 0.1 * out["loss_c3"]
            out["mIoU"] = (iu[0].double() / iu[1].double()).mean()               # CMPC_model.py:486-490
        return out

    @torch.no_grad()
    def forward(self, words, im, seq_len):
        """sess.run([pred, up, sigm, ...], {words, im, seq_len}) (test.py:286-296)."""
        self._check_feeds(words, im, seq_len)
        feats, st = self.features_async(im)
        o = self.head(feats, words, seq_len, after=st)
        B, h, w, H, W, T, N = self.batch_size, self.cfg.vf_h, self.cfg.vf_w, self.H, self.W, self.num_steps, self.cfg.N
        res = {"pred": o["pred"], "up": o["up"], "sigm": o["sigm"],
               "up_c3": o["up_c3"], "up_c4": o["up_c4"], "up_c5": o["up_c5"],
               "words_parse": o["words_parse"].view(B, 1, T, 4),
               # the reference keeps the attributes of the LAST level built, c3 (CMPC_model.py:395,399)
               "gw_w": o["gw_w_c3"][:, :, :T], "gw_v": o["gw_v_c3"][:, :, :T]}
        return res

    def predict(self, images, sentences, sequence_lenghts):
        """TF-serving signature of export_model_serving.py:57-71: images, sentences, sequence_lenghts -> masks."""
        return self.forward(sentences, images, sequence_lenghts)["sigm"]

    def loss_and_grads(self, feats, words, target_fine, seq_len, after=None):
        """forward + backward of `cost` (CMPC_model.py:447) into the flat gradient buffer (L2 and the
        x2 bias multiplier are applied inside the Adam kernel)."""
        self._params_ready(0)          # Adam has consumed the gradient buffer
        self.store.zero_grads()
        o = self.head(feats, words, seq_len, target_fine, after=after)
        o["loss_all"].backward()
        if self.marks is not None:
            self._mark("main_bwd_end")
            for i, st in enumerate(self.side or ()):
                with torch.cuda.stream(st):
                    self._mark("lane%d_bwd_end" % i)
        if self.side is not None:
            # parameter gradients are written by the kernels themselves (not autograd leaves): the
            # optimizer on the main stream must wait for every side stream's backward
            main = torch.cuda.current_stream(self.device)
            for st in self.side:
                main.wait_stream(st)
            if self.cx.wg is not None:
                main.wait_stream(self.cx.wg)
        self._mark("bwd_done")
        self.cx.flush_wgrad()
        self._mark("dW_done")
        return o

    _SCALARS = ("loss_all", "loss_c3", "loss_c4", "loss_c5", "loss_last", "mIoU")
    MAX_STEPS_IN_FLIGHT = int(os.environ.get("CMPC_STEPS_IN_FLIGHT", "2"))
    GRAPH_WARMUP = 2            # eager steps before capture (sizes the library workspaces, MIOpen find, allocator)

    def _fwd_bwd(self, words, im, target_fine, seq_len, ready=None):
        feats, st = self.features_async(im, ready)
        o = self.loss_and_grads(feats, words, target_fine, seq_len, after=st)
        return torch.stack([o[k].detach().float() for k in self._SCALARS])

    def _fwd_bwd_graphed(self, words, im, target_fine, seq_len):
        """forward + backward through a captured HIP graph: feeds are copied into static device buffers,
        the graph (backbone, head forward, head backward into the flat gradient buffer, on 4 streams)
        is replayed, the six summary scalars are copied out."""
        if self._gin is None:
            B, T, H, W = self.batch_size, self.num_steps, self.H, self.W
            d = self.device
            self._gstream = torch.cuda.Stream(device=d)
            self._gin = (torch.zeros(B, T, dtype=torch.int32, device=d), torch.zeros(B, H, W, 3, device=d),
                         torch.zeros(B, H, W, 1, device=d), torch.zeros(B, dtype=torch.int32, device=d))
        for dst, src in zip(self._gin, (words, im, target_fine, seq_len)):
            dst.copy_(src if torch.is_tensor(src) else torch.as_tensor(np.asarray(src)), non_blocking=True)
        cur = torch.cuda.current_stream(self.device)
        if self._eager_steps < self.GRAPH_WARMUP:
            # eager passes run on the stream the capture will use: the library's partial-sum workspaces are
            # per stream and must have their final size before capture
            self._eager_steps += 1
            self._gstream.wait_stream(cur)
            with torch.cuda.stream(self._gstream):
                sv = self._fwd_bwd(*self._gin)
            cur.wait_stream(self._gstream)
            return sv
        if self._graph is None:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._gstream):
                self._gout = self._fwd_bwd(*self._gin)
            self._graph = g
        self._graph.replay()
        return self._gout.clone()

    def capture(self, words, im, target_fine, seq_len):
        """Optional set-up call: run the eager warm-up passes and capture the train graph now (no optimizer
        step is taken), so that the first train_step already replays it."""
        self._check_feeds(words, im, seq_len, target_fine)
        while self._graph is None and self.use_graph:
            self._fwd_bwd_graphed(words, im, target_fine, seq_len)

    def train_step(self, words, im, target_fine, seq_len, ready=None):
        """sess.run([train, train_step, merged], feed) (trainval_model.py:98-107).
        ready: optional torch.cuda.Event recorded once the (device-resident, prefetched) feeds were complete."""
        if self.mode != 'train':
            raise RuntimeError("model was built with mode='eval' (CMPC_model.py:85-86)")
        self._check_feeds(words, im, seq_len, target_fine)
        # Bound the host's lead to MAX_STEPS_IN_FLIGHT steps: the enqueue of a step costs less host time than the
        # step takes on the GPU, and an unbounded lead makes the caching allocator grow by one step's tensors per step
        # of lead (their blocks are pending on stream events) until every step pays hipMalloc calls: 12.4 -> 40-54 ms
        # per step after ~25 unsynchronised steps.
        if len(self._inflight) >= self.MAX_STEPS_IN_FLIGHT:
            self._inflight.pop(0).synchronize()
        self._mark("step_start")
        if self.use_graph:
            sv = self._fwd_bwd_graphed(words, im, target_fine, seq_len)
        else:
            sv = self._fwd_bwd(words, im, target_fine, seq_len, ready)
        gscale = dist.allreduce_grads_(self.store.grads)          # RCCL over xGMI: one flat buffer
        if self.opt_stream is not None:
            # Adam + repack on the optimizer stream: the next step's backbone does not depend on them
            self.opt_stream.wait_stream(torch.cuda.current_stream(self.device))
            ev0 = torch.cuda.Event()
            with torch.cuda.stream(self.opt_stream):
                lr = self.store.adam_step(gscale, on_stage0=lambda: ev0.record(self.opt_stream))
                self._mark("adam_done")
            self._opt_stage0 = ev0
            self._opt_pending = True
        else:
            lr = self.store.adam_step(gscale)
        ev = torch.cuda.Event()
        ev.record(self.opt_stream if self.opt_stream is not None else torch.cuda.current_stream(self.device))
        self._inflight.append(ev)
        scal = {k: sv[i] for i, k in enumerate(self._SCALARS)}
        scal["mean_IOU"] = scal.pop("mIoU")
        scal["learning_rate"] = lr
        self.last = scal
        return self.store.step, scal

    # ------------------------------------------------------------------------------------------
    def state_dict(self):
        self._params_ready()
        return self.store.state_dict()

    def load_weights(self, named: Dict[str, torch.Tensor]):
        self._params_ready()
        self.store.load_state(named)

    def enable_data_parallel(self):
        """One process per GPU; identical weights are assumed (same seed); gradients are summed with
        one all-reduce of the flat buffer and divided by the world size in the Adam kernel."""
        self.world = dist.world_size()
        if self.world > 1:
            dist.broadcast_params_(self.store.params, 0)
            self.store.pack()
        return self.world


def get_segmentation_model(name, **kwargs):
    """get_model.get_segmentation_model (get_model.py:15-17): name -> <module>.LSTM_model(**kwargs)."""
    if name not in ("CMPC_model",):
        raise ValueError("only CMPC_model is built in this round (got %r)" % (name,))
    return LSTM_model(**kwargs)
