"""Batch data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the tests).  The reference has no distributed code at all;
images are independent except the per-replica l2_normalize(gv_lang) (CMPC_model.py:241), so the only
exchange is the sum of the flat fp32 gradient buffer, issued as ~60 MB buckets as soon as each is final
(allreduce_bucket_: exchange modules + ConvLSTM first, then the three pyramid levels, the text encoder last), so
that all but the last bucket travel while the backward pass is still running; the 1/world factor is folded into
the Adam kernel (gscale)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl", device: torch.device | None = None):
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or os.environ.get("CMPC_DP_SINGLE")) and not dist.is_initialized():      # CMPC_DP_SINGLE: a group of one rank (tests)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kw)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def is_initialized() -> bool:
    return dist.is_initialized()


def broadcast_params_(flat: torch.Tensor, src: int = 0):
    if dist.is_initialized():
        dist.broadcast(flat, src)


def allreduce_bucket_(eng, b: int, ranges, comm_stream: torch.cuda.Stream, max_elems: int = 16 << 20):
    """Sum gradient bucket b of `eng` (engine.Engine) over all ranks: the communication stream first waits (on the device) for the
    event cmpc_backward recorded when the bucket became final, then its ranges are all-reduced in chunks of <= max_elems (64 MB) on
    that stream -- while the rest of the backward pass is still running on the compute streams.  The caller orders the bucket's
    optimizer update after `comm_stream`."""
    if not dist.is_initialized():
        return
    eng.bucket_wait(b, comm_stream)
    with torch.cuda.stream(comm_stream):
        for off, cnt in ranges:
            for o in range(off, off + cnt, max_elems):
                dist.all_reduce(eng.grads[o: min(o + max_elems, off + cnt)], op=dist.ReduceOp.SUM)
