"""DeepLab-ResNet-101 (output stride 8), frozen -- the reference's visual backbone
(external/tensorflow-deeplab-resnet/deeplab_resnet/model.py:19-401 on kaffe/tensorflow/network.py:105-270).

BASELINE.json's north_star allows the backbone on PyTorch-ROCm; here only the 7x7 stem convolution and the
max-pool are torch (MIOpen) calls: every 1x1 / 3x3 convolution runs on the implicit-GEMM HIP kernel
(cmpc_conv_nhwc, csrc/gemm.hip) with the folded-BN shift, residual add and ReLU in its epilogue.  The backbone is
inference-only (is_training=False, CMPC_model.py:73), so every slim
batch_norm (epsilon 1e-3, network.py:260-270) is folded into its convolution at load time.
TF 'SAME' padding is reproduced explicitly: the 7x7/2 stem pads (2,3), the 3x3/2 max-pool pads
(0,1) with -inf, dilated 3x3 convolutions pad by their rate.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}        # dtype codes of include/cmpc.h


def _same_pad(size, k, stride, dil):
    out = -(-size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - size, 0)
    return total // 2, total - total // 2


def block_layout(width: int = 64, blocks=(3, 4, 23, 3)):
    """[(stage, suffix, has_branch1, cin, mid, cout, stride, dilation)] in graph order."""
    w = width
    stage_cfg = [(2, blocks[0], w, 4 * w, 1, 1), (3, blocks[1], 2 * w, 8 * w, 2, 1),
                 (4, blocks[2], 4 * w, 16 * w, 1, 2), (5, blocks[3], 8 * w, 32 * w, 1, 4)]
    out, cin = [], w
    for stage, n, mid, cout, stride, dil in stage_cfg:
        for b in range(n):
            suf = "abc"[b] if stage in (2, 5) else ("a" if b == 0 else f"b{b}")   # res2a.., res3b1.., res4b22, res5c
            out.append((stage, suf, b == 0, cin, mid, cout, stride if b == 0 else 1, dil))
            cin = cout
    return out


def param_specs(width: int = 64, blocks=(3, 4, 23, 3)) -> List[Tuple[str, Tuple[int, ...]]]:
    specs = []

    def add(conv, bn, k, cin, cout):
        specs.append((f"{conv}/weights", (k, k, cin, cout)))
        for s in ("gamma", "beta", "moving_mean", "moving_variance"):
            specs.append((f"{bn}/{s}", (cout,)))

    add("conv1", "bn_conv1", 7, 3, width)
    for stage, suf, b1, cin, mid, cout, stride, dil in block_layout(width, blocks):
        p = f"{stage}{suf}"
        if b1:
            add(f"res{p}_branch1", f"bn{p}_branch1", 1, cin, cout)
        add(f"res{p}_branch2a", f"bn{p}_branch2a", 1, cin, mid)
        add(f"res{p}_branch2b", f"bn{p}_branch2b", 3, mid, mid)
        add(f"res{p}_branch2c", f"bn{p}_branch2c", 1, mid, cout)
    return specs


def init_params(width: int = 64, blocks=(3, 4, 23, 3), seed: int = 4321, stem_gamma: float = 1.0) -> Dict[str, torch.Tensor]:
    """Synthetic weights (deeplab_resnet_init.ckpt, trainval_model.py:50, is not in the reference tree):
    He-normal convolutions; BN gamma 1 (0.2 on each block's last BN), beta 0, mean 0, variance 1.  stem_gamma scales bn_conv1/gamma
    and with it every tap (the frozen inference network is positively homogeneous): the CMPCv5 models use 1/256 so that random weights
    give taps of rms ~1 like a trained network's instead of ~1e3 (the image's 0..255 scale), which would saturate their tanh laterals."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_specs(width, blocks):
        if name.endswith("/weights"):
            fan_in = shape[0] * shape[1] * shape[2]
            t = torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / fan_in)
        elif name.endswith("/gamma"):
            t = torch.full(shape, 0.2 if "branch2c" in name else (stem_gamma if name == "bn_conv1/gamma" else 1.0), dtype=torch.float64)
        elif name.endswith("/moving_variance"):
            t = torch.ones(shape, dtype=torch.float64)
        else:
            t = torch.zeros(shape, dtype=torch.float64)
        out[name] = t.float()
    return out


class _ConvBN(nn.Module):
    """conv (no bias) + frozen BN folded into weight / bias (+ optional ReLU)."""

    def __init__(self, k, cin, cout, stride=1, dilation=1, relu=True):
        super().__init__()
        self.k, self.stride, self.dilation, self.relu = k, stride, dilation, relu
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self.register_buffer("bias32", torch.zeros(cout, dtype=torch.float32), persistent=False)
        # [Cout][kh][kw][Cin]: the K-contiguous operand of the implicit-GEMM kernel (cmpc_conv_nhwc)
        self.register_buffer("w_ohwi", torch.zeros(cout, k * k * cin), persistent=False)   # 2-D: immune to memory_format casts
        self.register_buffer("zeros", torch.zeros(256, dtype=torch.uint8), persistent=False)
        self.use_hip = k in (1, 3) and cin % 64 == 0 and cout % 8 == 0

    def load(self, p, conv, bn):
        sc = p[f"{bn}/gamma"].double() / torch.sqrt(p[f"{bn}/moving_variance"].double() + 1e-3)
        sh = p[f"{bn}/beta"].double() - p[f"{bn}/moving_mean"].double() * sc
        w = p[f"{conv}/weights"].double().permute(3, 2, 0, 1) * sc.view(-1, 1, 1, 1)      # HWIO -> OIHW
        self.weight.data.copy_(w.to(self.weight.dtype))
        self.bias.data.copy_(sh.to(self.bias.dtype))
        self._shift32 = sh.float().cpu()              # exact fp32 master of the folded-BN shift: a plain attribute, so no
        self.bias32.copy_(self._shift32)              # module-wide dtype cast can round it (see _apply)
        self.w_ohwi.copy_(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(self.w_ohwi.dtype))

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        # the epilogue kernel always reads an fp32 shift.  super()._apply has just pushed the buffer through `fn`
        # (.to(bfloat16) rounds it to 8 bits): restore it from the exact master instead of widening the rounded copy.
        shift = getattr(self, "_shift32", None)
        self.bias32 = shift.to(self.bias32.device) if shift is not None else self.bias32.float()
        self.zeros = self.zeros.to(torch.uint8)
        return out

    def forward(self, x, res=None):
        """Implicit-GEMM HIP convolution with the folded-BN shift, residual add and ReLU in its epilogue
        (1x1 / 3x3 with Cin % 64 == 0); otherwise MIOpen conv + one fused HIP epilogue pass."""
        k, s, d = self.k, self.stride, self.dilation
        if self.use_hip and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last) and \
                (res is None or res.is_contiguous(memory_format=torch.channels_last)):
            B, Cin, H, W = x.shape
            Ho, Wo, Cout = -(-H // s), -(-W // s), self.weight.shape[0]
            y = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
            a = _lib.ConvArgs()
            a.dtype = _DT[x.dtype]
            a.X, a.ldx = x.data_ptr(), Cin
            a.Wt, a.ldw = self.w_ohwi.data_ptr(), k * k * Cin
            a.bias, a.res = self.bias32.data_ptr(), (res.data_ptr() if res is not None else None)
            a.Y, a.ldy = y.data_ptr(), Cout
            a.B, a.H, a.W, a.Cin, a.Cout, a.ksize, a.stride, a.dil, a.relu = B, H, W, Cin, Cout, k, s, d, int(self.relu)
            a.zeros = self.zeros.data_ptr()
            _lib.call("cmpc_conv_nhwc", ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            return y.permute(0, 3, 1, 2)                  # NCHW view of the NHWC buffer (= channels_last)
        pt, pb = _same_pad(x.shape[2], k, s, d)
        pl, pr = _same_pad(x.shape[3], k, s, d)
        if pt == pb and pl == pr:
            y = F.conv2d(x, self.weight, None, stride=s, padding=(pt, pl), dilation=d)
        else:
            y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), self.weight, None, stride=s, dilation=d)
        B, C, H, W = y.shape
        if C % 8 or not y.is_contiguous(memory_format=torch.channels_last) or (res is not None and not res.is_contiguous(memory_format=torch.channels_last)):
            y = y + self.bias32.to(y.dtype).view(1, -1, 1, 1)
            if res is not None:
                y = y + res
            return F.relu(y, inplace=True) if self.relu else y
        dt = _DT[y.dtype]
        _lib.call("cmpc_bias_act_res", dt, y.data_ptr(), self.bias32.data_ptr(), res.data_ptr() if res is not None else None,
                  int(self.relu), B * H * W, C, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return y


class _Bottleneck(nn.Module):
    def __init__(self, has_b1, cin, mid, cout, stride, dil):
        super().__init__()
        self.b1 = _ConvBN(1, cin, cout, stride, relu=False) if has_b1 else None
        self.a = _ConvBN(1, cin, mid, stride)
        self.b = _ConvBN(3, mid, mid, 1, dil)
        self.c = _ConvBN(1, mid, cout, relu=True)       # ReLU after the residual add, fused with it

    def forward(self, x):
        sc = self.b1(x) if self.b1 is not None else x
        return self.c(self.b(self.a(x)), res=sc)


class DeepLabResNet(nn.Module):
    """im [B,H,W,3] (BGR, mean-subtracted, NHWC like the reference placeholder CMPC_model.py:68)
    -> (c3, c4, c5) = res3b3_relu, res4b22_relu, res5c_relu as NHWC maps (CMPC_model.py:74-76)."""

    def __init__(self, width: int = 64, blocks=(3, 4, 23, 3)):
        super().__init__()
        self.width, self.blocks_cfg = width, tuple(blocks)
        self.stem = _ConvBN(7, 3, width, 2)
        self.layout = block_layout(width, blocks)
        self.blocks = nn.ModuleList([_Bottleneck(b1, cin, mid, cout, st, dil) for (_, _, b1, cin, mid, cout, st, dil) in self.layout])

    def load_tf(self, p: Dict[str, torch.Tensor]):
        self.stem.load(p, "conv1", "bn_conv1")
        for blk, (stage, suf, b1, *_r) in zip(self.blocks, self.layout):
            n = f"{stage}{suf}"
            if b1:
                blk.b1.load(p, f"res{n}_branch1", f"bn{n}_branch1")
            blk.a.load(p, f"res{n}_branch2a", f"bn{n}_branch2a")
            blk.b.load(p, f"res{n}_branch2b", f"bn{n}_branch2b")
            blk.c.load(p, f"res{n}_branch2c", f"bn{n}_branch2c")

    @torch.no_grad()
    def forward_segment(self, x, seg):
        """One of three segments of forward(): 0 = stem + pool + res2 + res3 (takes the image, NHWC), 1 = res4, 2 = res5 (take the previous
        segment's running map).  Returns (running map, {tap: NHWC tensor} for the wanted taps this segment produces): c3 / res2b are complete
        a fifth of the way into the backbone, c4 after res4 -- a caller that captures the segments separately can hand each tap to its
        consumer when it is ready."""
        if seg == 0:
            x = x.permute(0, 3, 1, 2).to(self.stem.weight.dtype).contiguous(memory_format=torch.channels_last)
            x = self.stem(x)
            pt, pb = _same_pad(x.shape[2], 3, 2, 1)
            pl, pr = _same_pad(x.shape[3], 3, 2, 1)
            x = F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
        stages = ((2, 3), (4,), (5,))[seg]
        taps = {}
        for blk, (stage, suf, *_r) in zip(self.blocks, self.layout):
            if stage in stages:
                x = blk(x)
                taps[stage] = x
                taps[f"{stage}{suf}"] = x
        return x, {s_: taps[s_].permute(0, 2, 3, 1).contiguous() for s_ in self.taps_wanted if s_ in taps}

    taps_wanted = (3, 4, 5)     # stages (int) or block names ("2b" = res2b_relu, CMPCv5_BiLSTM_model.py:88) returned by forward, in this order

    @torch.no_grad()
    def forward(self, im_nhwc):
        x = im_nhwc.permute(0, 3, 1, 2).to(self.stem.weight.dtype).contiguous(memory_format=torch.channels_last)
        x = self.stem(x)
        pt, pb = _same_pad(x.shape[2], 3, 2, 1)
        pl, pr = _same_pad(x.shape[3], 3, 2, 1)
        x = F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
        taps = {}
        for blk, (stage, suf, *_r) in zip(self.blocks, self.layout):
            x = blk(x)
            taps[stage] = x
            taps[f"{stage}{suf}"] = x
        # NCHW(channels_last) -> NHWC views are free: permute gives a contiguous NHWC tensor
        return tuple(taps[s].permute(0, 2, 3, 1).contiguous() for s in self.taps_wanted)
