"""Child process of tests/test_a_two_rank_gpu.py: one rank of a two-rank run of the REAL encoder (lrf_amd.qmf_encode_batch on
the GPU) through lrf_amd.sharding.encode_sharded.  The ranks share the box's one GPU, so the (tiny) metrics gather runs
over gloo; on a multi-GPU node the same code runs with backend "nccl" and one GPU per rank (bench.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def dataset(n, H, W):
    g = torch.Generator().manual_seed(77)
    return torch.randint(0, 256, (n, 3, H, W), dtype=torch.uint8, generator=g)


def metrics_of(images, streams):
    import lrf_amd
    dec = lrf_amd.qmf_decode_batch(streams).cpu()
    rows = []
    for im, s, d in zip(images, streams, dec):
        rows.append([float(len(s)), lrf_amd.psnr(im, d).item(), lrf_amd.bits_per_pixel(im.shape[-2:], s)])
    return torch.tensor(rows, dtype=torch.float32)


def main():
    n, H, W, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; LRF_WORKER_BACKEND=nccl: RCCL with the payload on
    # the GPU (one rank per GPU: on this one-GPU box a world of one rank), as the multi-GPU runs do
    backend = os.environ.get("LRF_WORKER_BACKEND", "gloo")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    import lrf_amd
    from lrf_amd.sharding import encode_sharded, shard_range
    data = dataset(n, H, W)
    streams, table = encode_sharded(n, lambda lo, hi: data[lo:hi].pin_memory(), lambda imgs: lrf_amd.qmf_encode_batch(imgs, rank=7),
                                    metrics_of, 3)
    lo, hi = shard_range(n, dist.get_rank(), dist.get_world_size())
    assert len(streams) == hi - lo and tuple(table.shape) == (n, 3)
    from lrf_amd.sharding import collective_device
    assert collective_device().type == ("cuda" if backend == "nccl" else "cpu") and table.device.type == collective_device().type
    table = table.cpu()
    with open(f"{out}.rank{dist.get_rank()}", "w") as f:
        json.dump({"table": table.tolist(), "span": [lo, hi], "first_stream_len": len(streams[0]) if streams else None}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
