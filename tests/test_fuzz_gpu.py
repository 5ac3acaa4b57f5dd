"""GPU suite (-m gpu): short runs of the randomised HIP-vs-oracle sweeps of tools/ (each prints its mismatches and exits
non-zero on one).  The long runs are a developer's (`python tools/dev_fuzz_*.py <cases> <seed>`); a fixed seed here keeps the
suite deterministic.  Child processes: the tools are scripts, and their GPU use is sequential."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args", [
    ("dev_fuzz_anyshape.py", ["45", "11"]),   # any-shape path: own initialisation + iterations, three matrices per call
    ("dev_fuzz_svd.py", ["24", "3"]),         # svd_encode bytes, ranks 1..13 (byte-matrix and fp32-matrix paths)
    ("dev_fuzz_parity.py", ["5", "30"]),      # 64-column path: shapes, ranks up to 24, bounds, batches
    ("dev_fuzz_bcd.py", ["4", "16"]),         # BCD from given factors: 64-column and RGB colour-space paths
])
def test_fuzz_tool_is_clean(tool, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])


def test_fuzz_wave_kernels_for_ranks_17_to_32():
    """tools/dev_fuzz_w32.py with the thresholds at one block, so that runs of any size iterate on k_bcd_w32 / k_bcd_w32f:
    random shapes and batches, ranks 17..32 per plane, K, bounds inside and outside the int16-table range — bit for bit."""
    env = dict(os.environ, LRF_FAMILY_SPLIT_BLOCKS="1", LRF_BCDW32_MIN_BLOCKS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev_fuzz_w32.py"), "7", "24"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert r.stdout.count(": ok") == 24
