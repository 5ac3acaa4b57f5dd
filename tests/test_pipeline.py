"""GPU suite (-m gpu): the host -> host pipelined encoder (lrf_pipe, include/lrf_hip.h) against the one-shot HBM-resident
encoder.  Images are independent, so cutting a batch into sub-batches on several streams must not change a single byte."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)


@pytest.mark.parametrize("B,sub,slots,pinned", [(13, 5, 3, True), (8, 8, 1, True), (7, 2, 2, False), (3, 0, 3, True), (40, 16, 3, True)])
def test_pipelined_factors_equal_one_shot_factors(B, sub, slots, pinned):
    import lrf_amd
    from lrf_amd import _lib
    H, W, ranks = 96, 160, [7, 3, 3]
    imgs = _images(B, H, W, 11 + B)
    if pinned:
        imgs = imgs.pin_memory()
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks)
    pipe = _lib.Pipe(0, slots=slots, sub_batch=sub)
    try:
        for _ in range(2):  # second call: every slot's tables and workspace are warm, the two-geometry table cache is in use
            U, V = pipe.encode_rgb_host(imgs, ranks, 10, -16, 15)
            assert not U.is_cuda and U.dtype == torch.int8
            assert torch.equal(U, U0.cpu()) and torch.equal(V, V0.cpu())
        assert pipe.workspace_bytes() > 0
    finally:
        pipe.close()


@pytest.mark.parametrize("slots,pinned", [(2, True), (3, True), (2, False), (1, True)])
def test_tapered_schedule_of_unequal_pieces(slots, pinned, monkeypatch):
    """Left to the library (sub_batch 0) a submission is cut into regular pieces and a tapered tail (pipe_schedule, lrf_api.hip);
    here with a regular size of 8 images and the tail 5, 2, 1: 37 images -> 8, 8, 7, 6, 5, 2, 1.  A slot then sees up to four
    piece sizes (its descriptor tables are all resident from the second call on); downloads run on a stream of their own.
    Byte for byte the one-shot result, also from pageable memory (the two-thread submission)."""
    import lrf_amd
    from lrf_amd import _lib
    monkeypatch.setenv("LRF_PIPE_BULK", "8")
    monkeypatch.setenv("LRF_PIPE_TAIL", "5,2,1")
    B, H, W, ranks = 37, 96, 160, [7, 3, 3]
    imgs = _images(B, H, W, 5)
    if pinned:
        imgs = imgs.pin_memory()
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks)
    pipe = _lib.Pipe(0, slots=slots, sub_batch=0)
    try:
        for _ in range(3):
            seen = []
            for first, n, U, V in pipe.encode_rgb_host_iter(imgs, ranks, 10, -16, 15):
                seen.append((first, n))
                assert torch.equal(U[first:first + n], U0[first:first + n].cpu())
            assert [n for _, n in seen] == [8, 8, 7, 6, 5, 2, 1] and [f for f, _ in seen] == [0, 8, 16, 23, 29, 34, 36]
            assert torch.equal(U, U0.cpu()) and torch.equal(V, V0.cpu())
    finally:
        pipe.close()


def test_default_schedule_on_a_batch_large_enough_to_taper():
    """2500 images of 32x32: the regular piece is capped at 1024 images (80 MB of input would be 27306), the tail is three
    quarters and one quarter of it: 738, 738, 768, 256."""
    import lrf_amd
    from lrf_amd import _lib
    B, H, W, ranks = 2500, 32, 32, [4, 2, 2]
    imgs = _images(B, H, W, 8).pin_memory()
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks, num_iters=2)
    pipe = _lib.Pipe(0, slots=2, sub_batch=0)
    try:
        seen = [(first, n) for first, n, U, V in pipe.encode_rgb_host_iter(imgs, ranks, 2, -16, 15)]
        assert [n for _, n in seen] == [738, 738, 768, 256]
        U, V = pipe.encode_rgb_host(imgs, ranks, 2, -16, 15)
        assert torch.equal(U, U0.cpu()) and torch.equal(V, V0.cpu())
    finally:
        pipe.close()


def test_pipelined_with_signs_and_odd_geometry(oracle):
    """reflect-padded geometry, per-image sign vectors, and the generator form that hands back finished sub-batches"""
    import lrf_amd
    from lrf_amd import _lib
    from lrf_amd.codec import split_factors
    B, H, W = 9, 61, 117
    ranks = lrf_amd.qmf_ranks((H, W), quality=9)
    imgs = _images(B, H, W, 3).pin_memory()
    g = torch.Generator().manual_seed(1)
    sign = (torch.randint(0, 2, (B, sum(ranks)), generator=g) * 2 - 1).to(torch.int8)
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks, init_sign=sign)
    pipe = _lib.Pipe(0, slots=2, sub_batch=4)
    try:
        seen = []
        for first, n, U, V in pipe.encode_rgb_host_iter(imgs, ranks, 10, -16, 15, sign=sign):
            seen.append((first, n))
            assert torch.equal(U[first:first + n], U0[first:first + n].cpu())  # final as soon as it is reported
        assert seen == [(0, 4), (4, 4), (8, 1)]
        assert torch.equal(U, U0.cpu()) and torch.equal(V, V0.cpu())
    finally:
        pipe.close()
    b = 8  # and against the oracle, with that image's signs
    got = split_factors(U[b].numpy(), V[b].numpy(), (H, W), ranks)
    X = oracle.rgb_to_planes(imgs[b].numpy())
    off = 0
    for c in range(3):
        u, v = oracle.qmf_decompose(X[c], ranks[c], 10, (-16, 15), sign=sign[b, off:off + ranks[c]].numpy())
        off += ranks[c]
        assert np.array_equal(got[2 * c], u.astype(np.int8)) and np.array_equal(got[2 * c + 1], v.astype(np.int8))


def test_host_batch_streams_equal_single_image_streams():
    import lrf_amd
    imgs = _images(11, 72, 104, 21)
    streams = lrf_amd.qmf_encode_batch(imgs.pin_memory(), rank=7)
    assert len(streams) == 11
    for b in (0, 5, 10):
        assert streams[b] == lrf_amd.qmf_encode(imgs[b], rank=7)
    U, V = lrf_amd.qmf_factorize_host(imgs, [7, 3, 3])
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs.cuda(), [7, 3, 3])
    assert torch.equal(U, U0.cpu()) and torch.equal(V, V0.cpu())


def test_pipe_argument_errors():
    from lrf_amd import _lib
    with pytest.raises(ValueError):
        _lib.Pipe(0, slots=0)
    pipe = _lib.Pipe(0, slots=2, sub_batch=2)
    try:
        imgs = _images(3, 32, 32, 0)
        with pytest.raises(ValueError):
            pipe.encode_rgb_host(imgs, [7, 3, 3], 10, -16, 15, out=(torch.empty((3, 5), dtype=torch.int8), torch.empty((3, 5), dtype=torch.int8)))
        with pytest.raises(ValueError):
            pipe.encode_rgb_host(imgs, [0, 3, 3], 10, -16, 15)
        with pytest.raises(NotImplementedError):
            pipe.encode_rgb_host(imgs, [7, 3, 3], 0, -16, 15)
        U, V = pipe.encode_rgb_host(imgs, [2, 1, 1], 3, -16, 15)  # still usable after the errors
        assert tuple(U.shape) == (3, 16 * 2 + 4 + 4)
    finally:
        pipe.close()


def test_encode_out_buffers_are_validated():
    import lrf_amd
    imgs = _images(2, 32, 48, 2).cuda()
    bad = (torch.empty((2, 10), dtype=torch.int8, device="cuda"), torch.empty((2, 64 * 13), dtype=torch.int8, device="cuda"))
    with pytest.raises(ValueError):
        lrf_amd.qmf_factorize_batch(imgs, [7, 3, 3], out=bad)
    ctx = lrf_amd._lib.context(0)
    U, V = lrf_amd.qmf_factorize_batch(imgs, [7, 3, 3])
    with pytest.raises(ValueError):
        ctx.decode_rgb(U[:, :-1].contiguous(), V, 32, 48, [7, 3, 3])
    with pytest.raises(ValueError):
        ctx.decode_rgb(U, V, 32, 48, [8, 3, 3])


@pytest.mark.parametrize("ranks", [[16, 8, 8], [20, 10, 10]])
def test_pipelined_mixed_rank_families(ranks):
    """Pieces large enough (130 images of 512x768 = 3120 blocks) for the per-family launch plan and the family streams inside
    a slot: the slot's kernel stream forks into the families' streams and joins before the download.  Bytes as the one-shot
    encoder's, which itself equals the single-family run (tests/test_configs_at_size.py)."""
    import lrf_amd
    from lrf_amd import _lib
    imgs = _images(260, 512, 768, 5).pin_memory()
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks, num_iters=3)
    pipe = _lib.Pipe(0, slots=2, sub_batch=130)
    try:
        for _ in range(2):
            U, V = pipe.encode_rgb_host(imgs, ranks, 3, -16, 15)
            assert torch.equal(U, U0.cpu()) and torch.equal(V, V0.cpu())
    finally:
        pipe.close()


def test_small_anyshape_call_does_not_slow_the_pipe_down():
    """a one-image any-shape qmf_encode runs its three planes on three extra contexts / streams ("plane
    lanes"); left alive they cost every later pipelined batch encode of the process 15-45 % (one idle context and two streams:
    6.1 -> 7-9 ms per 256 x 512x768).  The host path of qmf_encode_batch / qmf_factorize_host releases them first: the
    host -> host time after such a call stays that of before.  The functional part is asserted exactly (the lanes are gone after
    the first host-path call, come back on demand, same bytes); the wall-clock part only against the failure it guards against
    (a third slower: medians of nine runs on a shared box moved 12 % between two identical runs in round 5)."""
    import time
    import lrf_amd
    from lrf_amd import codec
    g = torch.Generator().manual_seed(3)
    host = torch.randint(0, 256, (128, 3, 512, 768), dtype=torch.uint8, generator=g).pin_memory()
    small = torch.randint(0, 256, (3, 64, 96), dtype=torch.uint8, generator=g)

    def median_ms():
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lrf_amd.qmf_factorize_host(host, (7, 3, 3))
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[4] * 1e3

    for _ in range(3):
        lrf_amd.qmf_factorize_host(host, (7, 3, 3))
    before = median_ms()
    lrf_amd.qmf_encode(small, quality=20, patch_size=(16, 16))  # three plane lanes are created ...
    assert len(codec._PLANE_LANES) >= 1
    after = median_ms()  # ... and released by the first host-path call
    assert len(codec._PLANE_LANES) == 0
    assert after <= before * 1.30, (before, after)
    again = lrf_amd.qmf_encode(small, quality=20, patch_size=(16, 16))  # re-created on demand, same bytes
    assert again == lrf_amd.qmf_encode(small, quality=20, patch_size=(16, 16))
