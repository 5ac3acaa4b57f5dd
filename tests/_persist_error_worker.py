"""Child process of tests/test_persist_error.py: loads the failure-injection build of the library (liblrf_hip_skipflag.so:
-DLRF_BCDP_TEST_SKIP_FLAG -DLRF_BCDP_MAX_POLLS=4096; in a call of FOUR iterations the persistent kernel k_bcd_p never publishes
the first V update of matrix 0, so the polls of that matrix's later blocks expire) with LRF_PERSIST=1 (k_bcd_p from 1024
blocks) and drives it through the PRODUCT entry points — lrf_amd.qmf_encode_batch on a device tensor (one context, results to
the host through Context.to_host), on a host tensor (the pipe: lrf_pipe_wait_next), qmf_factorize_batch + ctx.synchronize().
Each failing call must raise in THAT call; the call after it must work and give the default library's bytes (sha256 on
stdout, compared by the parent)."""
import hashlib
import os
import sys

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lrf_amd import _lib  # noqa: E402

if os.environ.get("LRF_TEST_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LRF_TEST_LIB"])
import lrf_amd  # noqa: E402

assert os.environ.get("LRF_PERSIST") == "1"
inject = bool(os.environ.get("LRF_TEST_LIB"))
g = torch.Generator().manual_seed(11)
imgs = torch.randint(0, 256, (48, 3, 512, 768), dtype=torch.uint8, generator=g)  # 1152 blocks: k_bcd_p under LRF_PERSIST=1
dev = imgs.cuda()


def sha(streams):
    h = hashlib.sha256()
    for s in streams:
        h.update(s)
    return h.hexdigest()


def expect_failure(fn, what):
    if not inject:
        fn()
        return
    try:
        fn()
    except _lib.LrfError as e:
        assert "k_bcd_p" in str(e) and "expired" in str(e), str(e)
        print(f"raised in {what}: {str(e)[:120]}", file=sys.stderr)
        return
    raise AssertionError(f"{what}: the call with the expired poll returned normally")


out = []
# 1. device tensor in, byte streams out (Context.to_host after the factorisation)
expect_failure(lambda: lrf_amd.qmf_encode_batch(dev, rank=7, num_iters=4), "qmf_encode_batch(device tensor)")
out.append(sha(lrf_amd.qmf_encode_batch(dev, rank=7, num_iters=10)))  # the next call works
# 2. device tensors out: the failure surfaces at ctx.synchronize(), and the context goes on working
ctx = _lib.context(0)


def factorize_then_sync():
    lrf_amd.qmf_factorize_batch(dev, [7, 3, 3], num_iters=4)
    ctx.synchronize()


expect_failure(factorize_then_sync, "qmf_factorize_batch + ctx.synchronize()")
U, V = lrf_amd.qmf_factorize_batch(dev, [7, 3, 3], num_iters=10)
Uh, Vh = ctx.to_host(U, V)
out.append(hashlib.sha256(Uh.numpy().tobytes() + Vh.numpy().tobytes()).hexdigest())
# 3. host tensor in: the pipelined encoder (pieces of 48 / 2 images would be below 1024 blocks: one piece per slot)
pinned = imgs.pin_memory()
pipe = _lib.pipe(None, 2, 48)
expect_failure(lambda: pipe.encode_rgb_host(pinned, [7, 3, 3], 4, -16, 15), "Pipe.encode_rgb_host")
Uh2, Vh2 = pipe.encode_rgb_host(pinned, [7, 3, 3], 10, -16, 15)
out.append(hashlib.sha256(Uh2.numpy().tobytes() + Vh2.numpy().tobytes()).hexdigest())
assert out[1] == out[2], "pipe and one-shot encoder disagree"


def iter_all():
    for _ in pipe.encode_rgb_host_iter(pinned, [7, 3, 3], 4, -16, 15):
        pass


expect_failure(iter_all, "Pipe.encode_rgb_host_iter")
# 4. a failure nobody looked at is refused at the next persistent call's entry (never silently overwritten)
if inject:
    lrf_amd.qmf_factorize_batch(dev, [7, 3, 3], num_iters=4)
    torch.cuda.synchronize()
    expect_failure(lambda: lrf_amd.qmf_factorize_batch(dev, [7, 3, 3], num_iters=10), "the next persistent call's entry")
    U, V = lrf_amd.qmf_factorize_batch(dev, [7, 3, 3], num_iters=10)
    Uh, Vh = ctx.to_host(U, V)
    assert hashlib.sha256(Uh.numpy().tobytes() + Vh.numpy().tobytes()).hexdigest() == out[1]
print(" ".join(out))
