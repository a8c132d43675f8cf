"""Multi-GPU plumbing: one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI).

The sampler shards by SAMPLE: each trajectory is independent (the reference only runs B = 1), so
there is no per-step collective.  Two collectives exist, both outside the per-step path:
  * broadcast_state_dict - rank 0's UNet / LGP weights to every rank, as a few large fp16 buckets
    (xGMI is point-to-point, ~153 GB/s per link: few large transfers, not one per tensor);
  * gather_images - the decoded uint8 images of every rank (VAE decode runs on the rank that sampled) to rank 0;
    gather_latents - the same for the fp32 latents, for callers that decode elsewhere.
The reference has no multi-GPU inference path (app.py:45-46 is a single pipe.to("cuda")); its only
collectives are the implicit DDP all-reduces of the training scripts (trainer.py:91), out of scope.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.distributed as dist

BUCKET_ELEMS = 256 * 1024 * 1024        # 512 MB of fp16 per broadcast


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], shapes: Optional[Dict[str, tuple]], device,
                         src: int = 0) -> Dict[str, torch.Tensor]:
    """Floating tensors travel as fp16 (every synthetic / checkpoint weight is fp16-representable by
    construction); integer buffers (num_batches_tracked) as int64.  ``shapes`` may be None, then the key /
    shape manifest itself is broadcast first (small object broadcast)."""
    rank = dist.get_rank()
    if shapes is None:
        meta = [None]
        if rank == src:
            meta = [[(k, tuple(v.shape), v.dtype.is_floating_point) for k, v in sd.items()]]
        dist.broadcast_object_list(meta, src=src)
        manifest = meta[0]
    else:
        manifest = [(k, tuple(s), True) for k, s in shapes.items()]
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    floats = [(k, s) for k, s, f in manifest if f]
    i = 0
    while i < len(floats):
        j, n = i, 0
        while j < len(floats) and (n == 0 or n + _numel(floats[j][1]) <= BUCKET_ELEMS):
            n += _numel(floats[j][1]); j += 1
        buf = torch.empty(n, device=device, dtype=torch.float16)
        if rank == src:
            off = 0
            for k, s in floats[i:j]:
                m = _numel(s)
                buf[off:off + m] = sd[k].reshape(-1).to(device, torch.float16)
                off += m
        dist.broadcast(buf, src=src)
        off = 0
        for k, s in floats[i:j]:
            m = _numel(s)
            out[k] = buf[off:off + m].view(s)
            off += m
        i = j
    for k, s, f in manifest:
        if not f:
            t = sd[k].to(device, torch.int64).reshape(-1) if rank == src else torch.zeros(max(1, _numel(s)), device=device, dtype=torch.int64)
            dist.broadcast(t, src=src)
            out[k] = t.view(s)
    return OrderedDict((k, out[k]) for k, _, _ in manifest)       # the sender's key order


def _numel(s) -> int:
    n = 1
    for d in s:
        n *= d
    return n


def gather_latents(x: torch.Tensor, world: int, dst: int = 0):
    """x [S,4,h,h] fp32 on every rank -> list of ``world`` tensors on rank dst (None elsewhere)."""
    return _gather(x, world, dst)


def _gather(x: torch.Tensor, world: int, dst: int):
    x = x.contiguous()
    dev = x.device
    if dist.get_backend() == "gloo" and x.is_cuda:       # CPU tests / several ranks on one GPU: stage through the host
        x = x.cpu()
    bufs = [torch.empty_like(x) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(x, bufs, dst=dst)
    return None if bufs is None else [b.to(dev) for b in bufs]


def gather_images(img: torch.Tensor, world: int, dst: int = 0):
    """img uint8 [S, H, W, 3] (decoded on the rank that sampled it: 786 432 B per 512x512 image) on every rank ->
    list of ``world`` tensors on rank dst (None elsewhere).  The final gather north_star / SURVEY 8(e) describe."""
    assert img.dtype == torch.uint8
    return _gather(img, world, dst)


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of sample indices owned by ``rank`` (remainder spread over the first ranks)."""
    q, r = divmod(total, world)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def allreduce_mean_(flat: torch.Tensor, bucket_bytes: int = 15 << 20) -> torch.Tensor:
    """In-place mean of a flat gradient vector over all ranks, in buckets of <= bucket_bytes (the reference's DDP:
    trainer.py:91 bucket_cap_mb=15).  The one per-step collective of the LGP training path (RCCL all-reduce over
    xGMI on GPUs; gloo in the CPU tests).  No-op when torch.distributed is not initialised or on one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat
    per = max(1, bucket_bytes // flat.element_size())
    for o in range(0, flat.numel(), per):
        dist.all_reduce(flat[o:o + per], op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    return flat
