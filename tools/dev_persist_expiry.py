"""The bounded poll of the persistent iteration kernel k_bcd_p: a library built with -DLRF_BCDP_TEST_SKIP_FLAG
-DLRF_BCDP_MAX_POLLS=64 (LRF_LIB names it) never publishes the first V update of matrix 0, so the polls of that matrix's later
blocks expire and the error path runs: the launch must drain (no hang), the
call or the synchronisation after it must return an error, and the context must work again afterwards (the queue state is
zeroed after a failed launch).  python tools/dev_persist_expiry.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib as _l0
if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
errors = 0
t0 = time.perf_counter()
for i in range(6):
    try:
        U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
        torch.cuda.synchronize()
        ctx.synchronize()
        print(f"run {i}: no poll expired")
    except Exception as e:  # the library's error, raised by the Python layer
        errors += 1
        print(f"run {i}: error reported: {str(e)[:110]}")
print(f"{errors} of 6 runs reported an expired poll; no hang ({time.perf_counter() - t0:.2f} s for all)")
# (-DLRF_BCDP_MAX_POLLS=0 alone — give up at the first poll that finds its flag unset — reports nothing on this workload:
#  in steady state no wave ever waits)
