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


def test_bench_bare_gpus_2_runs_two_ranks():
    """`python bench.py --gpus 2` with no launcher must run TWO ranks (it used to run one and print
    n_gpus 1).  Weak scaling: both ranks encode --batch images; the line carries the host->host leg of both."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run the file first (it is the first of the suite)")
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--regions", "2", "--batch", "8", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["self_launched"] is True
    assert out["steps"] == 2 and out["timed_regions"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and len(out["ranks"]) == 2
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
