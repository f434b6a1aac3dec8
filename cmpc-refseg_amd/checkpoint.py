"""Checkpoints in the reference's variable name space -- the counterpart of the tf.train.Saver calls of the reference's driver
(trainval_model.py:46-63 restore, :56 `Saver(max_to_keep=4)`, :136-142 save, :82 resume via -lastiter).

Two on-disk forms, same variable name space:
  * TensorFlow's own: `<prefix>-<step>.index` + `.data-00000-of-00001` (tf_bundle.py reads and writes the tensor-bundle format without
    TensorFlow): `Saver(fmt="tf")`;
  * ONE `.npz` file (plain NumPy arrays, loaded with allow_pickle=False): `Saver(fmt="npz")`, the default.
The keys / variable names are the ones the reference graph creates (tf.global_variables() of CMPC_model.LSTM_model(mode='train')):
    text_objseg/<scope>/DW, .../biases, .../beta, .../gamma, text_objseg/Variable (GloVe table), text_objseg/rnn/...   head
    text_objseg/Variable_1                      int32 step counter: the unnamed `tf.Variable(0, trainable=False)` of CMPC_model.py:450, created
                                                inside variable_scope('text_objseg') after the GloVe table `text_objseg/Variable` (:146)
    text_objseg/beta1_power, /beta2_power       AdamOptimizer's non-slot variables (created by apply_gradients inside that scope, :478)
    text_objseg/<var.op.name>/Adam, /Adam_1     AdamOptimizer slots m and v: slot_creator opens variable_scope(None, primary.op.name + '/Adam')
                                                UNDER the current scope, hence the doubled prefix `text_objseg/text_objseg/c5_lateral/DW/Adam`
    conv1/weights, bn_conv1/gamma, res2a_branch2a/weights, bn2a_branch2a/moving_mean, ...   frozen backbone (deeplab_resnet/model.py)
Rounds 1-2 wrote `<var>/Adam`, `<var>/Adam_1`, `global_step` (int64), `beta1_power`, `beta2_power`; restore still accepts those as aliases.

PARITY UNPINNED: the optimizer-state names above follow TF1's naming rules as read from its source conventions; TensorFlow is not
available in this image and the reference ships no checkpoint, so neither they nor the tensor-bundle reader have met a TensorFlow-written
file (tests/test_tf_bundle.py pins the names and the format by round trips and known-answers only).
"""
from __future__ import annotations

import glob
import os
import re
import warnings
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import tf_bundle

BACKBONE_PREFIXES = ("res", "bn", "conv1")          # the subset `trainval_model.py:50-54` restores from deeplab_resnet_init.ckpt
STEP_VAR = "text_objseg/Variable_1"                 # CMPC_model.py:450
STEP_ALIASES = (STEP_VAR, "global_step")
BETA_POWER = ("text_objseg/beta1_power", "text_objseg/beta2_power")


def is_backbone_var(name: str) -> bool:
    return name.startswith(BACKBONE_PREFIXES)


def slot_names(var: str) -> Tuple[str, str]:
    """(m, v) slot variable names of a trainable head variable `text_objseg/...` as AdamOptimizer creates them inside the scope."""
    return f"text_objseg/{var}/Adam", f"text_objseg/{var}/Adam_1"


def _slot_aliases(var: str):
    m, v = slot_names(var)
    return (m, var + "/Adam"), (v, var + "/Adam_1")


def model_variables(model) -> Dict[str, np.ndarray]:
    """Every variable tf.train.Saver() would write for `model` (tf.global_variables()), by TensorFlow name."""
    eng = model.eng
    torch.cuda.synchronize(model.device)
    out: Dict[str, np.ndarray] = {}
    p, m, v = eng.params.cpu().numpy(), eng.m.cpu().numpy(), eng.v.cpu().numpy()
    for name, (off, shape) in eng.index.items():
        n = int(np.prod(shape))
        sm, sv = slot_names(name)
        out[name] = p[off: off + n].reshape(shape).copy()
        out[sm] = m[off: off + n].reshape(shape).copy()
        out[sv] = v[off: off + n].reshape(shape).copy()
    for name, val in getattr(model, "extra_vars", lambda: {})().items():      # non-trainable head state (batch-norm moving statistics)
        out[name] = np.asarray(val)
    t = eng.step
    out[STEP_VAR] = np.asarray(t, dtype=np.int32)
    # TF1 AdamOptimizer: beta*_power start at beta and are multiplied by beta after every apply_gradients
    out[BETA_POWER[0]] = np.asarray(0.9 ** (t + 1), dtype=np.float32)
    out[BETA_POWER[1]] = np.asarray(0.999 ** (t + 1), dtype=np.float32)
    for k, val in getattr(model, "backbone_vars", {}).items():
        out[k] = val.cpu().numpy() if torch.is_tensor(val) else np.asarray(val)
    return out


def restore_variables(model, variables: Dict[str, np.ndarray], var_filter: Optional[Callable[[str], bool]] = None, strict: bool = True):
    """tf.train.Saver(var_list).restore: set the variables selected by `var_filter` (default: all the file holds).
    strict: every variable of the MODEL that the filter selects (head variables by their manifest names, backbone variables by
    model.backbone_vars) must be present -- tf.train.Saver raises NotFoundError otherwise.  A full (unfiltered) restore that finds neither
    a step counter nor Adam slots warns: training would resume with step 0 and zero moments (a weights-only file)."""
    eng = model.eng
    keep = (lambda n: True) if var_filter is None else var_filter
    names = [n for n in variables if keep(n)]
    head = {n for n in names if n in eng.index}
    if strict:
        want = [n for n in eng.index if keep(n)] + [n for n in getattr(model, "backbone_vars", {}) if keep(n)]
        missing = [n for n in want if n not in variables]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} of the {len(want)} selected model variables, e.g. {missing[:3]}")
    torch.cuda.synchronize(model.device)
    n_slots = 0
    with torch.cuda.device(model.device):
        for n in head:
            off, shape = eng.index[n]
            a = np.asarray(variables[n], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"{n}: checkpoint shape {tuple(a.shape)} != {shape}")
            cnt = a.size
            eng.params[off: off + cnt].copy_(torch.from_numpy(a.reshape(-1)))
            for aliases, buf in zip(_slot_aliases(n), (eng.m, eng.v)):
                for key in aliases:
                    if key in variables and keep(key):
                        buf[off: off + cnt].copy_(torch.from_numpy(np.asarray(variables[key], dtype=np.float32).reshape(-1)))
                        n_slots += 1
                        break
        if head:
            eng.pack()
        have_step = False
        for key in STEP_ALIASES:
            if key in variables and keep(key):
                eng.step = int(variables[key])
                have_step = True
                break
        extra = {n: np.asarray(variables[n]) for n in names if n in getattr(model, "extra_var_names", lambda: ())()}
        if extra:
            model.load_extra_vars(extra)
        if var_filter is None and head and not (have_step and n_slots):
            warnings.warn("checkpoint restore: no step counter (text_objseg/Variable_1) and / or no Adam slots found -- weights only; "
                          "a resumed training run restarts the learning-rate schedule and Adam's bias correction", stacklevel=2)
        bbv = {n: torch.from_numpy(np.asarray(variables[n], dtype=np.float32)) for n in names if is_backbone_var(n) and "/Adam" not in n}
        if bbv:
            have = dict(getattr(model, "backbone_vars", {}))
            have.update(bbv)
            model.load_backbone(have)
        torch.cuda.synchronize(model.device)


class Saver:
    """tf.train.Saver(var_list=None, max_to_keep=4) for LSTM_model.  save() writes `<prefix>-<global_step>` (fmt "npz": one `.npz`
    file; fmt "tf": TensorFlow's `.index` + `.data-00000-of-00001` pair) plus the `checkpoint` state file, which lists the snapshots in
    SAVE order (what max_to_keep prunes by and latest_checkpoint returns -- not the step number: a resumed run may reset it), and deletes
    all but the newest `max_to_keep`; restore() reads either form (optionally only the variables `var_filter` selects:
    `Saver(var_filter=is_backbone_var)` is the backbone-only restore of trainval_model.py:50-54)."""

    def __init__(self, var_filter: Optional[Callable[[str], bool]] = None, max_to_keep: int = 4, fmt: str = "npz"):
        if fmt not in ("npz", "tf"):
            raise ValueError("fmt must be 'npz' or 'tf'")
        self.var_filter, self.max_to_keep, self.fmt = var_filter, max_to_keep, fmt

    def save(self, model, prefix: str, global_step: Optional[int] = None) -> str:
        vs = model_variables(model)
        if self.var_filter is not None:
            vs = {k: v for k, v in vs.items() if self.var_filter(k)}
        step = int(vs.get(STEP_VAR, 0)) if global_step is None else int(global_step)
        d = os.path.dirname(os.path.abspath(prefix))
        os.makedirs(d, exist_ok=True)
        if self.fmt == "tf":
            path = f"{prefix}-{step}"
            tf_bundle.write_bundle(path, vs)
        else:
            path = f"{prefix}-{step}.npz"
            np.savez(path, **{k.replace("/", "|"): v for k, v in vs.items()})      # '/' is not portable inside zip member names
        kept = [p for p in _snapshots(prefix) if os.path.abspath(p) != os.path.abspath(path)] + [path]      # save order, this one newest
        if self.max_to_keep and self.max_to_keep > 0:
            for p in kept[:-self.max_to_keep]:
                for f in ([p] if p.endswith(".npz") else [p + ".index"] + glob.glob(glob.escape(p) + ".data-*")):
                    if os.path.exists(f):
                        os.remove(f)
            kept = kept[-self.max_to_keep:]
        tf_bundle.write_checkpoint_state(d, os.path.basename(path), [os.path.basename(p) for p in kept])
        return path

    def restore(self, model, path: str, strict: bool = True):
        if os.path.exists(path + ".index"):
            vs = tf_bundle.read_bundle(path)
        elif tf_bundle.is_v1_checkpoint(path):
            # a V1 (single-file) TensorFlow checkpoint: what `deeplab_resnet_init.ckpt` is (trainval_model.py:50)
            vs = tf_bundle.read_v1_checkpoint(path)
        else:
            with np.load(path, allow_pickle=False) as z:
                vs = {k.replace("|", "/"): z[k] for k in z.files}
        restore_variables(model, vs, self.var_filter, strict=strict)
        return vs


def _on_disk(prefix: str) -> List[str]:
    pat = re.compile(re.escape(os.path.basename(prefix)) + r"-(\d+)(\.npz|\.index)$")
    out = []
    for f in glob.glob(glob.escape(prefix) + "-*"):
        m = pat.search(os.path.basename(f))
        if m:
            out.append(f if m.group(2) == ".npz" else f[: -len(".index")])
    return out


def _snapshots(prefix: str) -> List[str]:
    """Snapshots of `prefix` that exist on disk, OLDEST SAVE first: the order of the `checkpoint` state file where it lists them
    (tf.train.Saver's own bookkeeping), modification time for files it does not know."""
    d = os.path.dirname(os.path.abspath(prefix))
    disk = {os.path.abspath(p): p for p in _on_disk(prefix)}
    ordered: List[str] = []
    for name in tf_bundle.read_checkpoint_state_all(d):
        p = os.path.abspath(name if os.path.isabs(name) else os.path.join(d, name))
        if p in disk and disk[p] not in ordered:
            ordered.append(disk[p])

    def mtime(p):
        return os.path.getmtime(p if p.endswith(".npz") else p + ".index")
    rest = sorted((p for p in disk.values() if p not in ordered), key=mtime)
    return rest + ordered if ordered else rest


def latest_checkpoint(prefix: str) -> Optional[str]:
    """tf.train.latest_checkpoint for snapshots written by Saver.save (either form): the most recently SAVED one."""
    snaps = _snapshots(prefix)
    return snaps[-1] if snaps else None
