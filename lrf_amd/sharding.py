"""Per-image sharding of a dataset over the GPUs of one node (one process per GPU).

Images are independent, so there is no data-path collective: every rank encodes its contiguous block and only the
per-image metrics are gathered at the end (RCCL on GPUs, gloo in the CPU tests)."""
from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: ceil(n/world) items per rank, the last ranks may get fewer or none."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def collective_device() -> torch.device:
    """Where the payload of this process group's collectives must live: the current GPU under RCCL ("nccl"), the CPU
    under gloo.  Every rank must hand tensors of the same kind to a collective, also the ranks whose block is empty."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_metrics(local: torch.Tensor, n_items: int) -> torch.Tensor:
    """local: [n_local, k] float32 metrics of this rank's block (in block order) -> [n_items, k] on every rank
    (on collective_device())."""
    if not (dist.is_available() and dist.is_initialized()):
        return local  # (a process group of ONE rank still runs the collective: the N > 1 code path, executed)
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_items + world - 1) // world
    k = local.shape[1]
    local = local.to(collective_device())
    padded = torch.zeros((per, k), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)  # a few bytes per image: latency only
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        rows.append(out[r][: hi - lo])
    return torch.cat(rows, dim=0)


def encode_sharded(n_items: int, load_block: Callable[[int, int], torch.Tensor], encode_block: Callable[[torch.Tensor], List[bytes]],
                   metrics_of: Callable[[torch.Tensor, Sequence[bytes]], torch.Tensor], n_metrics: int):
    """Runs `encode_block` on this rank's block and gathers per-image metrics.

    load_block(lo, hi) -> images of the block; encode_block(images) -> list of byte streams;
    metrics_of(images, streams) -> [n_local, n_metrics] float32 (a rank with an empty block contributes no rows).
    Returns (this rank's streams, [n_items, n_metrics] metrics)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(n_items, rank, world)
    if hi > lo:
        images = load_block(lo, hi)
        streams = encode_block(images)
        local = metrics_of(images, streams).float().reshape(hi - lo, n_metrics)
    else:
        streams, local = [], torch.zeros((0, n_metrics), dtype=torch.float32, device=collective_device())
    return streams, gather_metrics(local, n_items)
