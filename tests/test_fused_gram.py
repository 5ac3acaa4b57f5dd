"""GPU suite (-m gpu): k_planes16_gram (lrf_amd/csrc/lrf_planes_gram_kernel.hip) — the patch matrices and the luma planes' exact
Gram partials in one kernel, the default for calls of 256 x 512x768 images or more — forced on for small calls
(LRF_FUSED_GRAM_MIN_CHUNKS=1, a child process) against the two-kernel form k_planes16 + k_gram64: identical int8 factors for
every geometry class, through lrf_qmf_encode_rgb_u8, the sweep entry point and the host -> host pipe.  (At the bench size the
kernel runs by default: tests/test_hip_parity.py::test_full_size_batch_properties compares that path with the oracle.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(env_extra):
    env = dict(os.environ)
    env.pop("LRF_FUSED_GRAM_MIN_CHUNKS", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fused_gram_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.splitlines(), [l for l in r.stderr.splitlines() if l.startswith("launches:")]


def test_fused_planes_gram_kernel_gives_the_factors_of_the_two_kernel_form():
    fused, which_f = _run({"LRF_FUSED_GRAM_MIN_CHUNKS": "1"})
    plain, which_p = _run({"LRF_FUSED_GRAM_MIN_CHUNKS": str(1 << 40)})
    assert which_f == ["launches: k_planes16_gram 1 k_planes16 0"] and which_p == ["launches: k_planes16_gram 0 k_planes16 1"], (which_f, which_p)
    assert len(fused) == 10 and fused == plain, "\n".join(a + "   |   " + b for a, b in zip(fused, plain))
