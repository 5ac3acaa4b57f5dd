"""CPU suite: the N > 1 path (per-image sharding + metrics gather) with two gloo ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lrf_amd.sharding import encode_sharded, shard_range
    g = torch.Generator().manual_seed(3)
    data = torch.randint(0, 256, (n_items, 3, 8, 8), dtype=torch.uint8, generator=g)

    def load(lo, hi):
        return data[lo:hi]

    def encode(images):  # stand-in encoder (the GPU one is exercised by the -m gpu suite)
        return [bytes([int(im.sum()) % 251]) * (1 + int(im[0, 0, 0]) % 5) for im in images]

    def metrics(images, streams):
        return torch.tensor([[float(len(s)), float(im.float().mean())] for im, s in zip(images, streams)])

    streams, table = encode_sharded(n_items, load, encode, metrics, 2)
    lo, hi = shard_range(n_items, rank, world)
    assert len(streams) == hi - lo
    q.put((rank, table.tolist()))  # plain lists: a tensor would travel as a shared-memory handle the exiting worker can drop
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_rank_sharding(n_items):
    from lrf_amd.sharding import shard_range
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    tables = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds the full table, identical to a single-process run
    g = torch.Generator().manual_seed(3)
    data = torch.randint(0, 256, (n_items, 3, 8, 8), dtype=torch.uint8, generator=g)
    want = torch.tensor([[float(1 + int(im[0, 0, 0]) % 5), float(im.float().mean())] for im in data])
    for r in range(world):
        assert torch.allclose(torch.tensor(tables[r]).reshape(want.shape), want)
    spans = [shard_range(n_items, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n_items and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_shard_range_properties():
    from lrf_amd.sharding import shard_range
    for n in (0, 1, 5, 256, 4096, 4097):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert sum(hi - lo for lo, hi in spans) == n
            assert all(0 <= lo <= hi <= n for lo, hi in spans)


def test_bench_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment starts two child ranks before it imports
    torch.  Without a GPU each rank stops at its "needs a GPU" assertion: two of them must show,
    and the parent must return their failure."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box this test runs the no-GPU branch
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0
    assert "[bench] rank 0 of 2 started" in err and "[bench] rank 1 of 2 started" in err, err[-2000:]
    assert "AssertionError: bench.py needs a GPU" in err  # (the first rank to fail ends the other)
    assert p.stdout.decode().strip() == ""


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in p.stderr.decode(errors="replace")
