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
