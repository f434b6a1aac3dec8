"""Checkpoints in the reference's variable name space -- the counterpart of the tf.train.Saver calls of the reference's driver
(trainval_model.py:46-63 restore, :56 `Saver(max_to_keep=4)`, :136-142 save, :82 resume via -lastiter).

Two on-disk forms, same variable name space:
  * TensorFlow's own: `<prefix>-<step>.index` + `.data-00000-of-00001` (tf_bundle.py reads and writes the tensor-bundle format without
    TensorFlow), so `deeplab_resnet_init.ckpt` and the reference's snapshots load directly and the snapshots written here load in the
    reference: `Saver(fmt="tf")`;
  * ONE `.npz` file (plain NumPy arrays, loaded with allow_pickle=False): `Saver(fmt="npz")`, the default.
The keys / variable names are the ones the reference graph creates:
    text_objseg/<scope>/DW, .../biases, .../beta, .../gamma, text_objseg/Variable (GloVe table), text_objseg/rnn/...   head
    <var>/Adam, <var>/Adam_1                    AdamOptimizer slots m and v of every trainable head variable
    beta1_power, beta2_power, global_step       AdamOptimizer's non-slot variables and the step counter (CMPC_model.py:450)
    conv1/weights, bn_conv1/gamma, res2a_branch2a/weights, bn2a_branch2a/moving_mean, ...   frozen backbone (deeplab_resnet/model.py)
The tensor-bundle reader is pinned by round trips and format known-answers only: TensorFlow is not available in this image and the
reference ships no checkpoint to test it against (tests/test_tf_bundle.py).
"""
from __future__ import annotations

import glob
import os
import re
from typing import Callable, Dict, Iterable, Optional

import numpy as np
import torch

from . import tf_bundle

BACKBONE_PREFIXES = ("res", "bn", "conv1")          # the subset `trainval_model.py:50-54` restores from deeplab_resnet_init.ckpt


def is_backbone_var(name: str) -> bool:
    return name.startswith(BACKBONE_PREFIXES)


def model_variables(model) -> Dict[str, np.ndarray]:
    """Every variable tf.train.Saver() would write for `model` (tf.global_variables()), by TensorFlow name."""
    eng = model.eng
    torch.cuda.synchronize(model.device)
    out: Dict[str, np.ndarray] = {}
    p, m, v = eng.params.cpu().numpy(), eng.m.cpu().numpy(), eng.v.cpu().numpy()
    for name, (off, shape) in eng.index.items():
        n = int(np.prod(shape))
        out[name] = p[off: off + n].reshape(shape).copy()
        out[name + "/Adam"] = m[off: off + n].reshape(shape).copy()
        out[name + "/Adam_1"] = v[off: off + n].reshape(shape).copy()
    t = eng.step
    out["global_step"] = np.asarray(t, dtype=np.int64)
    # TF1 AdamOptimizer: beta*_power start at beta and are multiplied by beta after every apply_gradients
    out["beta1_power"] = np.asarray(0.9 ** (t + 1), dtype=np.float32)
    out["beta2_power"] = np.asarray(0.999 ** (t + 1), dtype=np.float32)
    for k, val in getattr(model, "backbone_vars", {}).items():
        out[k] = val.cpu().numpy() if torch.is_tensor(val) else np.asarray(val)
    return out


def restore_variables(model, variables: Dict[str, np.ndarray], var_filter: Optional[Callable[[str], bool]] = None, strict: bool = True):
    """tf.train.Saver(var_list).restore: set the variables selected by `var_filter` (default: all the file holds).
    strict: every selected head variable of the model must be present (Saver raises NotFoundError otherwise)."""
    eng = model.eng
    keep = (lambda n: True) if var_filter is None else var_filter
    names = [n for n in variables if keep(n)]
    head = {n for n in names if n in eng.index}
    if strict and var_filter is None:
        missing = [n for n in eng.index if n not in variables]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} head variables, e.g. {missing[:3]}")
    torch.cuda.synchronize(model.device)
    with torch.cuda.device(model.device):
        for n in head:
            off, shape = eng.index[n]
            a = np.asarray(variables[n], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"{n}: checkpoint shape {tuple(a.shape)} != {shape}")
            cnt = a.size
            eng.params[off: off + cnt].copy_(torch.from_numpy(a.reshape(-1)))
            for slot, buf in (("/Adam", eng.m), ("/Adam_1", eng.v)):
                if n + slot in variables and keep(n + slot):
                    buf[off: off + cnt].copy_(torch.from_numpy(np.asarray(variables[n + slot], dtype=np.float32).reshape(-1)))
        if head:
            eng.pack()
        if "global_step" in variables and keep("global_step"):
            eng.step = int(variables["global_step"])
        bbv = {n: torch.from_numpy(np.asarray(variables[n], dtype=np.float32)) for n in names if is_backbone_var(n) and "/Adam" not in n}
        if bbv:
            have = dict(getattr(model, "backbone_vars", {}))
            have.update(bbv)
            model.load_backbone(have)
        torch.cuda.synchronize(model.device)


class Saver:
    """tf.train.Saver(var_list=None, max_to_keep=4) for LSTM_model.  save() writes `<prefix>-<global_step>` (fmt "npz": one `.npz`
    file; fmt "tf": TensorFlow's `.index` + `.data-00000-of-00001` pair and the `checkpoint` state file) and deletes all but the
    newest `max_to_keep` snapshots of that prefix; restore() reads either form (optionally only the variables `var_filter` selects:
    `Saver(var_filter=is_backbone_var)` is the backbone-only restore of trainval_model.py:50-54)."""

    def __init__(self, var_filter: Optional[Callable[[str], bool]] = None, max_to_keep: int = 4, fmt: str = "npz"):
        if fmt not in ("npz", "tf"):
            raise ValueError("fmt must be 'npz' or 'tf'")
        self.var_filter, self.max_to_keep, self.fmt = var_filter, max_to_keep, fmt

    def save(self, model, prefix: str, global_step: Optional[int] = None) -> str:
        vs = model_variables(model)
        if self.var_filter is not None:
            vs = {k: v for k, v in vs.items() if self.var_filter(k)}
        step = int(vs.get("global_step", 0)) if global_step is None else int(global_step)
        os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
        if self.fmt == "tf":
            path = f"{prefix}-{step}"
            tf_bundle.write_bundle(path, vs)
        else:
            path = f"{prefix}-{step}.npz"
            np.savez(path, **{k.replace("/", "|"): v for k, v in vs.items()})      # '/' is not portable inside zip member names
        kept = _snapshots(prefix)
        if self.max_to_keep and self.max_to_keep > 0:
            for _, p in kept[:-self.max_to_keep]:
                for f in ([p] if p.endswith(".npz") else [p + ".index"] + glob.glob(glob.escape(p) + ".data-*")):
                    os.remove(f)
            kept = kept[-self.max_to_keep:]
        if self.fmt == "tf":
            d = os.path.dirname(os.path.abspath(prefix))
            tf_bundle.write_checkpoint_state(d, os.path.basename(path), [os.path.basename(p) for _, p in kept if not p.endswith(".npz")])
        return path

    def restore(self, model, path: str, strict: bool = True):
        if os.path.exists(path + ".index"):
            vs = tf_bundle.read_bundle(path)
        else:
            with np.load(path, allow_pickle=False) as z:
                vs = {k.replace("|", "/"): z[k] for k in z.files}
        restore_variables(model, vs, self.var_filter, strict=strict)
        return vs


def _snapshots(prefix: str):
    """[(step, path)] of the snapshots of `prefix`, oldest first; path = the .npz file or the TensorFlow checkpoint prefix."""
    pat = re.compile(re.escape(os.path.basename(prefix)) + r"-(\d+)(\.npz|\.index)$")
    out = []
    for f in glob.glob(glob.escape(prefix) + "-*"):
        m = pat.search(os.path.basename(f))
        if m:
            out.append((int(m.group(1)), f if m.group(2) == ".npz" else f[: -len(".index")]))
    return sorted(out)


def latest_checkpoint(prefix: str) -> Optional[str]:
    """tf.train.latest_checkpoint for snapshots written by Saver.save (either form): the one with the largest step."""
    snaps = _snapshots(prefix)
    return snaps[-1][1] if snaps else None
