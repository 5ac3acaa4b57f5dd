"""GPU suite (-m gpu), first file of the suite on purpose: the N > 1 path with the REAL encoder.  Two fresh child processes
(gloo, sharing the box's one GPU) run lrf_amd.sharding.encode_sharded with lrf_amd.qmf_encode_batch and per-image
(bytes, PSNR, bpp) metrics; the table every rank gathers must equal the single-process table.  The children are started
before this process has touched the GPU (a GPU-initialised process must not exec), which is why the file sorts first."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two_ranks(n, H, W, out):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), str(n), str(H), str(W), out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), logs
    return [json.load(open(f"{out}.rank{r}")) for r in range(2)]


def _bench(args, timeout=600):
    """bench.py with NO launcher (no RANK / WORLD_SIZE in the environment): it has to start its own ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["LRF_BENCH_REHEARSAL"] = "1"  # the ranks share this box's one GPU; the tiny collectives run over gloo
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's) is expected"
    return json.loads(lines[0])


def _bench_forced_dist(args, timeout=600):
    """bench.py as ONE rank with LRF_BENCH_FORCE_DIST=1: the RCCL ("nccl") process group of the N > 1 runs, at world size 1"""
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT", "LRF_BENCH_REHEARSAL")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", LRF_BENCH_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_rccl_path_executes_at_world_size_1():
    """`init_process_group("nccl", device_id=...)`, the float64 all_reduce(MAX) on a device tensor, the
    barriers, the final all_gather and destroy_process_group of bench.py had never run anywhere (every N > 1 test is a gloo
    rehearsal).  LRF_BENCH_FORCE_DIST=1 runs exactly that code with one rank on this box's GPU; the line also carries the
    per-rank figures (own ms_per_step, NUMA binding, packer threads) and SURVEY 8(d)'s bytes-out leg."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run the file first (it is the first of the suite)")
    out = _bench_forced_dist(["--gpus", "1", "--steps", "2", "--warmup", "1", "--regions", "2", "--batch", "16", "--no-cpu-baseline"])
    assert out["n_gpus"] == 1 and out["collective_backend"] == "nccl" and out["forced_dist"] is True and out["rccl_ranks"] == 1
    r0 = out["ranks"][0]
    assert r0["device"] == 0 and r0["pixels_per_region"] == 2 * 16 * 512 * 768
    assert 0 < r0["ms_per_step"] <= out["ms_per_step_max"] * 1.001
    assert set(r0["numa"]) >= {"node", "cpus", "bound", "reason"} and r0["numa"]["cpus"] >= 1 and r0["packer_threads"] >= 1
    assert out["host_to_host_mpix_s"] > 0
    assert out["end_to_end_bytes_mpix_s"] > 0 and out["zlib_ms_per_step"] > 0 and out["packer_threads"] == r0["packer_threads"]
    assert out["end_to_end_bytes_ms_per_step"] >= out["zlib_ms_per_step"] * 0.5
    assert 0.1 < out["bits_per_pixel"] < 3.0  # zlib-9 over the int8 factors of uniform noise at ranks (7,3,3): ~0.39 bits per pixel


def test_gather_metrics_under_rccl(tmp_path):
    """lrf_amd.sharding.gather_metrics / collective_device with backend "nccl" (payload on the GPU) in a fresh child: one
    rank (RCCL refuses two ranks on one device), so the all_gather is of one block — the code path of the 8-GPU run."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run the file first (it is the first of the suite)")
    out = str(tmp_path / "nccl_table.json")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", LRF_WORKER_BACKEND="nccl")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), "5", "96", "160", out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    got = json.load(open(out + ".rank0"))
    assert got["span"] == [0, 5] and len(got["table"]) == 5 and all(row[0] > 0 and row[1] > 5 for row in got["table"])


def test_bench_bare_gpus_2_runs_two_ranks():
    """`python bench.py --gpus 2` with no launcher must run TWO ranks (it used to run one and print
    n_gpus 1).  Weak scaling: both ranks encode --batch images; the line carries the host->host leg of both."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run the file first (it is the first of the suite)")
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--regions", "2", "--batch", "8", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["self_launched"] is True
    assert out["steps"] == 2 and out["timed_regions"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and len(out["ranks"]) == 2
    assert all(0 < r["ms_per_step"] <= out["ms_per_step_max"] * 1.001 and r["packer_threads"] >= 1 for r in out["ranks"])
    assert out["end_to_end_bytes_mpix_s"] > 0 and out["zlib_ms_per_step"] > 0
    assert out["ms_per_step_min"] <= out["ms_per_step_median"] <= out["ms_per_step_max"]
    px = 8 * 512 * 768
    assert all(r["pixels_per_region"] == 2 * px for r in out["ranks"])
    assert abs(out["value"] - 2 * px / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * out["value"]
    assert out["host_to_host_mpix_s"] > 0 and out["value_definition"].startswith("hbm_resident")


def test_bench_strong_scaling_generates_only_the_ranks_block():
    """`--scaling strong` shards ONE global batch; a rank draws its own block image by image (per-image seeds), so its
    input never takes more than its share of the memory (it used to draw the global batch and slice it)."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run the file first (it is the first of the suite)")
    out = _bench(["--gpus", "2", "--config", "clic", "--scaling", "strong", "--batch", "6", "--steps", "1", "--warmup", "0",
                  "--regions", "1", "--no-extras"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["global_batch"] == 6
    share = 3 * 3 * 1365 * 2048
    assert out["config"]["images_this_rank"] == [0, 3] and out["config"]["rank_input_bytes"] == share
    assert out["config"]["input_generation_peak_bytes"] < 2 * share
    assert sum(r["pixels_per_region"] for r in out["ranks"]) == 6 * 1365 * 2048


def test_two_ranks_real_encoder_equal_single_process(tmp_path):
    """n = 9: blocks of 5 and 4 images; n = 1: rank 1 has nothing to encode and still takes part in the gather."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run the file first (it is the first of the suite)")
    H, W = 96, 160
    results = {n: _run_two_ranks(n, H, W, str(tmp_path / f"table{n}.json")) for n in (9, 1)}  # all children before any GPU use here
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _two_rank_worker as w
    import lrf_amd
    for n, tables in results.items():
        assert tables[0]["table"] == tables[1]["table"], "the ranks gathered different tables"
        assert tables[0]["span"][0] == 0 and tables[0]["span"][1] == tables[1]["span"][0] and tables[1]["span"][1] == n
        # the single-process table, on this process's GPU context
        data = w.dataset(n, H, W)
        streams = lrf_amd.qmf_encode_batch(data.pin_memory(), rank=7)
        want = w.metrics_of(data, streams)
        got = torch.tensor(tables[0]["table"], dtype=torch.float32)
        assert torch.equal(got[:, 0], want[:, 0]), "stream sizes differ between the sharded and the single-process run"
        assert torch.allclose(got, want, rtol=0, atol=1e-6)
