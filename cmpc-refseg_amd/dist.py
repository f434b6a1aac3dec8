"""Batch data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all;
images are independent except the per-replica l2_normalize(gv_lang) (CMPC_model.py:241), so the only
exchange is ONE all-reduce of the flat fp32 gradient buffer per step; the 1/world factor is folded
into the Adam kernel (gscale)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl", device: torch.device | None = None):
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kw)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def allreduce_grads_(flat: torch.Tensor) -> float:
    """Sum the flat gradient buffer over all ranks in place; returns the scale (1/world) the
    optimizer must apply."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def broadcast_params_(flat: torch.Tensor, src: int = 0):
    if world_size() > 1:
        dist.broadcast(flat, src)
