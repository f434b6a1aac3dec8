"""conv5=True: the res3 / res4 / res5 convolution weights of the DeepLab-ResNet are trained with the head (reference
CMPC_model.py:427-430: `tvars` += variables whose name starts with 'res5' / 'res4' / 'res3'; the batch-norm variables are named
'bn...' and stay frozen, `deeplab_resnet/model.py:19-401`, `kaffe/tensorflow/network.py:261-270`).

The forward pass is the frozen backbone's (folded batch-norm: y = conv(x, w) * s + t with constant s, t per output channel); this
module keeps what its backward needs and runs it through the library's own kernels (op-level C ABI, include/cmpc.h):

    dX of a 1x1      cmpc_gemm_nt      dY [R, cout] x (w * s) as [cin][cout]  (the HWIO matrix itself)
    dX of a 3x3      cmpc_conv_nhwc    the same atrous convolution of dY with the flipped, transposed taps
    dW               cmpc_gemm_tn      X^T dY, one product per tap with conv addressing for the 3x3; scaled by s afterwards
    relu'            cmpc_act_bwd      through the saved OUTPUT
    residual fan-in  cmpc_axpy
    update           cmpc_adam_step    TF-Adam with L2 on '.../weights' (CMPC_model.py:433: name[-9:-2] == 'weights'), then the folded
                                       operands (forward OHWI, dX) are rebuilt from the fp32 masters

PyTorch supplies device memory, views and the strided-row gather of the two stride-2 convolutions of res3a."""
import ctypes
from typing import Dict

import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
ACT_RELU = 1


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class BackboneTrainer:
    def __init__(self, net, vars_tf: Dict[str, torch.Tensor], device, weight_decay: float):
        self.net, self.device, self.wd = net, device, float(weight_decay)
        self.tdt = net.stem.weight.dtype
        self.dt = _DT[self.tdt]
        self.layers = []                      # trainable convolutions in graph order
        self.blocks = []                      # (module, layout row, [layer indices b1?, a, b, c])
        off = 0
        for blk, row in zip(net.blocks, net.layout):
            stage, suf, has_b1, cin, mid, cout, stride, dil = row
            if stage < 3:
                continue
            n = f"{stage}{suf}"
            idx = {}
            for br, mod, k, ci, co, st, dl in ((("branch1", blk.b1, 1, cin, cout, stride, 1),) if has_b1 else ()) + (
                    ("branch2a", blk.a, 1, cin, mid, stride, 1), ("branch2b", blk.b, 3, mid, mid, 1, dil), ("branch2c", blk.c, 1, mid, cout, 1, 1)):
                if not mod.use_hip:
                    raise NotImplementedError(f"conv5: res{n}_{br} ({ci}->{co}) is not on the library's convolution path (needs cin % 64 == 0)")
                L = dict(name=f"res{n}_{br}/weights", bn=f"bn{n}_{br}", mod=mod, k=k, cin=ci, cout=co, stride=st, dil=dl, off=off, count=k * k * ci * co)
                idx[br] = len(self.layers)
                self.layers.append(L)
                off += L["count"]
            self.blocks.append((blk, row, idx))
        self.total = off
        f32 = dict(dtype=torch.float32, device=device)
        self.params, self.grads = torch.empty(off, **f32), torch.zeros(off, **f32)
        self.m, self.v = torch.zeros(off, **f32), torch.zeros(off, **f32)
        self.scale = torch.empty(off, **f32)                 # the frozen batch-norm scale of every weight's output channel
        self.zero_page = torch.zeros(4096, dtype=torch.uint8, device=device)
        self.zero_bias = torch.zeros(4096, **f32)
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=device)
        segs = (_lib.AdamSeg * len(self.layers))()
        for i, L in enumerate(self.layers):
            segs[i].off, segs[i].count, segs[i].wd, segs[i].gmult = L["off"], L["count"], self.wd, 1.0
        self.segs_dev = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(device)
        self.load(vars_tf)

    # ------------------------------------------------------------------------------------------
    def view(self, buf, L):
        return buf[L["off"]: L["off"] + L["count"]].view(L["k"], L["k"], L["cin"], L["cout"])

    def load(self, vars_tf):
        for L in self.layers:
            w = vars_tf[L["name"]].to(self.device, torch.float32)
            bn = L["bn"]
            sc = (vars_tf[f"{bn}/gamma"].double() / torch.sqrt(vars_tf[f"{bn}/moving_variance"].double() + 1e-3)).float().to(self.device)
            self.view(self.params, L).copy_(w)
            self.view(self.scale, L).copy_(sc.view(1, 1, 1, -1).expand(L["k"], L["k"], L["cin"], L["cout"]))
        self.refold()

    def named_weights(self) -> Dict[str, torch.Tensor]:
        return {L["name"]: self.view(self.params, L).detach().cpu().clone() for L in self.layers}

    def named_slots(self):
        out = {}
        for L in self.layers:
            out[L["name"] + "/Adam"] = self.view(self.m, L).detach().cpu().clone()
            out[L["name"] + "/Adam_1"] = self.view(self.v, L).detach().cpu().clone()
        return out

    def load_slots(self, named):
        for L in self.layers:
            for buf, suf in ((self.m, "/Adam"), (self.v, "/Adam_1")):
                if L["name"] + suf in named:
                    self.view(buf, L).copy_(torch.as_tensor(named[L["name"] + suf]).to(self.device, torch.float32))

    def refold(self):
        """fp32 masters -> the operands the kernels read: forward [cout][kh][kw][cin] (w * s), dX of a 1x1 [cin][cout], dX of a 3x3
        [cin][kh][kw][cout] with the taps reversed."""
        wf = self.params * self.scale
        for L in self.layers:
            w = self.view(wf, L)                                            # HWIO, folded
            L["mod"].w_ohwi.copy_(w.permute(3, 0, 1, 2).reshape(L["cout"], -1))
            if L["k"] == 1:
                L["wn"] = w[0, 0].to(self.tdt).contiguous()               # [cin][cout]: Bt of dX = dY . (w s)^T
            else:
                L["wn"] = w.flip(0, 1).permute(2, 0, 1, 3).reshape(L["cin"], -1).to(self.tdt).contiguous()

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, im_nhwc):
        """The backbone's forward pass with the activations of res3-res5 kept (NHWC tensors).  Returns the taps in net.taps_wanted order."""
        net = self.net
        x = im_nhwc.permute(0, 3, 1, 2).to(self.tdt).contiguous(memory_format=torch.channels_last)
        x = net.stem(x)
        from .backbone import _same_pad
        import torch.nn.functional as F
        pt, pb = _same_pad(x.shape[2], 3, 2, 1)
        pl, pr = _same_pad(x.shape[3], 3, 2, 1)
        x = F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
        taps, self.saved = {}, []
        for blk, (stage, suf, *_r) in zip(net.blocks, net.layout):
            if stage < 3:
                x = blk(x)
            else:
                sc = blk.b1(x) if blk.b1 is not None else x
                a = blk.a(x)
                b = blk.b(a)
                y = blk.c(b, res=sc)
                nh = lambda t: t.permute(0, 2, 3, 1)                     # NCHW view of an NHWC buffer -> the NHWC tensor itself
                self.saved.append(dict(x=nh(x), a=nh(a), b=nh(b), y=nh(y)))
                x = y
            taps[stage] = x
            taps[f"{stage}{suf}"] = x
        return tuple(taps[s].permute(0, 2, 3, 1).contiguous() for s in net.taps_wanted)

    # ---- kernels -------------------------------------------------------------------------------
    def _relu_bwd(self, dy, y):
        R, C = dy.numel() // dy.shape[-1], dy.shape[-1]
        out = torch.empty_like(dy)
        e = dy.element_size()
        for c0 in range(0, C, 2048):
            w = min(2048, C - c0)
            _lib.call("cmpc_act_bwd", self.dt, ctypes.c_void_p(dy.data_ptr() + c0 * e), ctypes.c_void_p(y.data_ptr() + c0 * e), ctypes.c_void_p(out.data_ptr() + c0 * e),
                      ACT_RELU, R, C, w, w, None, None, 0, 0, _st())
        return out

    def _gemm_nt(self, A, Bt, N):
        R, K = A.numel() // A.shape[-1], A.shape[-1]
        out = torch.empty(A.shape[:-1] + (N,), dtype=A.dtype, device=A.device)
        a = _lib.GemmNtArgs()
        a.dtype, a.nseg = self.dt, 1
        a.A[0], a.lda[0], a.Bt[0], a.ldb[0], a.K[0] = A.data_ptr(), K, Bt.data_ptr(), K, K
        a.C, a.ldc = out.data_ptr(), N
        a.M, a.N, a.n_valid, a.batch = R, N, N, 1
        a.alpha = 1.0
        _lib.call("cmpc_gemm_nt", ctypes.byref(a), _st())
        return out

    def _wgrad(self, L, X, D, H, W):
        """grads[L] += X^T D per tap (X [B,H,W,cin] = the convolution's input at the OUTPUT resolution, D [B,H,W,cout])."""
        R = X.numel() // X.shape[-1]
        k, cin, cout, dil = L["k"], L["cin"], L["cout"], L["dil"]
        for t in range(k * k):
            a = _lib.GemmTnArgs()
            a.dtype = self.dt
            a.A, a.lda, a.Ka = X.data_ptr(), cin, cin
            a.D, a.ldd, a.Nd = D.data_ptr(), cout, cout
            a.out, a.ldo = self.grads.data_ptr() + 4 * (L["off"] + t * cin * cout), cout
            a.R, a.Kv, a.Nv = R, cin, cout
            a.nb, a.nb2 = 1, 1
            tiles = ((cin + 127) // 128) * ((cout + 127) // 128)
            br = 64 if self.dt != 0 else 32
            a.rsplit = max(1, min((R + 4 * br - 1) // (4 * br), (512 + tiles - 1) // tiles))
            a.alpha = 1.0
            a.zeros = self.zero_page.data_ptr()
            if k == 3:
                a.conv_H, a.conv_W, a.conv_dy, a.conv_dx = H, W, (t // 3 - 1) * dil, (t % 3 - 1) * dil
            _lib.call("cmpc_gemm_tn", ctypes.byref(a), _st())

    def _conv_dx3(self, L, D):
        B, H, W, cout = D.shape
        out = torch.empty((B, H, W, L["cin"]), dtype=D.dtype, device=D.device)
        a = _lib.ConvArgs()
        a.dtype = self.dt
        a.X, a.ldx = D.data_ptr(), cout
        a.Wt, a.ldw = L["wn"].data_ptr(), 9 * cout
        a.bias, a.res = self.zero_bias.data_ptr(), None
        a.Y, a.ldy = out.data_ptr(), L["cin"]
        a.B, a.H, a.W, a.Cin, a.Cout, a.ksize, a.stride, a.dil, a.relu = B, H, W, cout, L["cin"], 3, 1, L["dil"], 0
        a.zeros = self.zero_page.data_ptr()
        _lib.call("cmpc_conv_nhwc", ctypes.byref(a), _st())
        return out

    def _axpy(self, x, y):
        _lib.call("cmpc_axpy", self.dt, _p(x), _p(y), 1.0, x.numel(), _st())

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def backward(self, dtaps: Dict[int, torch.Tensor]):
        """dtaps: stage (3, 4, 5) -> d cost / d tap as an NHWC tensor [B,h,w,C] in the storage dtype (the handle's taps "dc3", "dc4",
        "dc5").  Fills self.grads (d cost / d weights, still carrying the handle's loss scale)."""
        self.grads.zero_()
        g = None
        last_of_stage = {}
        for i, (blk, row, idx) in enumerate(self.blocks):
            last_of_stage[row[0]] = i
        for i in range(len(self.blocks) - 1, -1, -1):
            blk, (stage, suf, has_b1, cin, mid, cout, stride, dil), idx = self.blocks[i]
            S = self.saved[i]
            if last_of_stage[stage] == i and stage in dtaps:
                d = dtaps[stage].view(S["y"].shape).contiguous()
                if g is None:
                    g = d.clone()
                else:
                    self._axpy(d, g)
            if g is None:
                continue
            B, H, W, _ = S["y"].shape
            gy = self._relu_bwd(g.contiguous(), S["y"].contiguous())                  # through the block's final ReLU
            Lc, Lb, La = self.layers[idx["branch2c"]], self.layers[idx["branch2b"]], self.layers[idx["branch2a"]]
            b_act, a_act = S["b"].contiguous(), S["a"].contiguous()
            self._wgrad(Lc, b_act, gy, H, W)
            gb = self._relu_bwd(self._gemm_nt(gy, Lc["wn"], mid), b_act)
            self._wgrad(Lb, a_act, gb, H, W)
            ga = self._relu_bwd(self._conv_dx3(Lb, gb), a_act)
            xin = S["x"] if stride == 1 else S["x"][:, ::stride, ::stride, :]
            xin = xin.contiguous()
            self._wgrad(La, xin, ga, H, W)
            if has_b1:
                self._wgrad(self.layers[idx["branch1"]], xin, gy, H, W)
            if i == 0:
                break                                                                 # nothing trainable below res3a
            if stride != 1:
                raise NotImplementedError("conv5: a strided block below the first trainable one")
            gx = self._gemm_nt(ga, La["wn"], cin)
            if has_b1:
                self._axpy(self._gemm_nt(gy, self.layers[idx["branch1"]]["wn"], cin), gx)
            else:
                self._axpy(gy, gx)
            g = gx
        self.grads.mul_(self.scale)                                                   # d / d w = s[cout] * d / d (w s)
        self.saved = []

    def adam(self, lr_t: float, gscale: float):
        """One TF-Adam update of the trained backbone weights (L2 on every '.../weights'), then the folded operands."""
        _lib.call("cmpc_adam_step", _p(self.params), _p(self.grads), _p(self.m), _p(self.v), _p(self.segs_dev), len(self.layers),
                  float(lr_t), 0.9, 0.999, 1e-8, float(gscale), _p(self.nonfinite), _st())
        self.refold()
