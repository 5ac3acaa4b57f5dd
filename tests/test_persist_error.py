"""The error half of the asynchronous boundary (include/lrf_hip.h, lrf_ctx_check): a persistent launch (k_bcd_p) whose bounded
poll expires must raise in the call it belongs to on every product path, and the context must go on working.  The failure is
injected by a build of the 64-column unit (lrf_amd/liblrf_hip_skipflag.so, made by __graft_entry__.build()); a child process
loads it through the package (tests/_persist_error_worker.py) and a second child runs the same calls on the shipped library."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = os.path.join(ROOT, "tests", "_persist_error_worker.py")


def _run(env_extra):
    env = dict(os.environ, LRF_PERSIST="1", **env_extra)
    r = subprocess.run([sys.executable, WORKER, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return r.stdout.strip().splitlines()[-1].split(), r.stderr


@pytest.mark.gpu
def test_expired_poll_raises_in_the_call_that_failed():
    lib = os.path.join(ROOT, "lrf_amd", "liblrf_hip_skipflag.so")
    assert os.path.exists(lib), "liblrf_hip_skipflag.so missing: run __graft_entry__.build()"
    good, _ = _run({})
    injected, err = _run({"LRF_TEST_LIB": "liblrf_hip_skipflag.so"})
    for what in ("qmf_encode_batch(device tensor)", "qmf_factorize_batch + ctx.synchronize()", "Pipe.encode_rgb_host",
                 "Pipe.encode_rgb_host_iter", "the next persistent call's entry"):
        assert f"raised in {what}" in err, err[-3000:]
    assert injected == good  # after every failure the next call gave the shipped library's bytes


def test_check_is_exported_and_bound():
    from lrf_amd import _lib
    assert "lrf_ctx_check" in _lib.EXPORTS and hasattr(_lib.Context, "check") and hasattr(_lib.Context, "to_host")
