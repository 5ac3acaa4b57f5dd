"""CPU suite: byte container against the reference's streams; the C-ABI library loads and exports its header."""
import os
import re

import numpy as np
import pytest

from conftest import QMF_CASES, ROOT, Case


@pytest.mark.parametrize("name", QMF_CASES + ["tiny_it0"])
def test_repack_is_byte_identical(name):
    from lrf_amd.codec import pack_image, parse_stream
    case = Case(name)
    meta, fac = parse_stream(case.encoded)
    H, W = case.image.shape[-2:]
    again = pack_image(fac, (H, W), meta["rank"], meta["bounds"], meta["patch size"], meta["dtype"])
    assert again == case.encoded


def test_combine_separate_errors():
    from lrf_amd import combine_bytes, separate_bytes
    parts = [b"a", b"", b"xyz" * 5, b"\x00\x01"]
    assert list(separate_bytes(combine_bytes(parts), 4)) == parts
    with pytest.raises(TypeError):
        combine_bytes([b"a", "b"])
    with pytest.raises(ValueError):
        separate_bytes(b"\x00", 2)


def test_rank_rule():
    import lrf_amd
    assert lrf_amd.qmf_ranks((512, 768), quality=7) == [4, 2, 2]
    assert lrf_amd.qmf_ranks((512, 768), rank=7) == [7, 3, 3]
    assert lrf_amd.qmf_ranks((64, 96), quality=7) == [4, 1, 1]
    assert lrf_amd.qmf_ranks((64, 96), rank=1) == [1, 1, 1]
    with pytest.raises(AssertionError):
        lrf_amd.qmf_ranks((64, 96), quality=101)


def test_library_exports_every_declared_symbol():
    import ctypes

    from lrf_amd import _lib
    header = open(os.path.join(ROOT, "include", "lrf_hip.h")).read()
    declared = set(re.findall(r"\b(lrf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)  # loads without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lrf_version() == 1


def test_kernel_ids_of_the_binding_are_the_header_ones():
    """LRF_K_* (include/lrf_hip.h) index the per-kernel timers of lrf_ctx_kernel_time: the Python constants and the names bench.py
    prints must follow the header when a kernel class is added (round 5: LRF_K_PLANES_GRAM)."""
    from lrf_amd import _lib
    header = open(os.path.join(ROOT, "include", "lrf_hip.h")).read()
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(LRF_K_[A-Z_]+)\s+(\d+)", header)}
    count = ids.pop("LRF_K_COUNT")
    assert sorted(ids.values()) == list(range(count))
    for name, value in ids.items():
        assert getattr(_lib, name) == value, name
    assert set(_lib.KERNEL_NAMES) == set(ids.values())


def test_plane_dims_match_reference_metadata():
    from lrf_amd import _lib
    from lrf_amd.codec import parse_stream
    for name in ("odd_q7", "nat_q7", "s2odd_q7", "s1_q7"):
        case = Case(name)
        meta, _ = parse_stream(case.encoded)
        H, W = case.image.shape[-2:]
        dims = _lib.plane_dims(H, W)
        assert [list(d[:2]) for d in dims] == meta["original size"]
        assert [list(d[2:4]) for d in dims] == meta["padded size"]


def test_no_gpu_means_loud_failure():
    import torch

    import lrf_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        lrf_amd.qmf_encode(torch.zeros(3, 16, 16, dtype=torch.uint8), quality=7)


def test_unsupported_branches_raise():
    import torch

    import lrf_amd
    img = torch.zeros(3, 16, 16, dtype=torch.uint8)
    with pytest.raises(AssertionError):
        lrf_amd.qmf_encode(img)
    with pytest.raises(AssertionError):
        lrf_amd.qmf_encode(img, quality=7, color_space="HSV")
    with pytest.raises(ValueError):
        lrf_amd.qmf_encode(img, quality=7, scale_factor=(0.0, 0.5))
    with pytest.raises(NotImplementedError):
        lrf_amd.svd_encode(img, quality=7, color_space="YCbCr")
    if not torch.cuda.is_available():  # the implemented branches need the GPU: no CPU fallback
        for kw in (dict(color_space="RGB"), dict(patch=False), dict(patch_size=(4, 4)), dict(color_space="RGB", patch_size=(16, 16)),
                   dict(color_space="RGB", patch=False), dict(scale_factor=(0.25, 0.25))):
            with pytest.raises(RuntimeError):
                lrf_amd.qmf_encode(img, quality=7, **kw)


def test_native_packer_matches_reference_streams():
    """liblrf_pack.so (include/lrf_pack.h) rebuilds the reference's byte streams from their own factors, and exports
    what its header declares."""
    import ctypes

    from lrf_amd.codec import pack_streams_native, parse_stream
    header = open(os.path.join(ROOT, "include", "lrf_pack.h")).read()
    declared = set(re.findall(r"\b(lrf_pack_[a-z0-9_]+)\s*\(", header))
    lib = ctypes.CDLL(os.path.join(ROOT, "lrf_amd", "liblrf_pack.so"))
    assert declared == {"lrf_pack_qmf_streams", "lrf_pack_qmf_streams_planes", "lrf_pack_free", "lrf_pack_zlib_version", "lrf_pack_unpack_qmf_factors"}
    for name in declared:
        assert hasattr(lib, name)
    for name in ("s1_r7", "odd_q7", "tiny_rank1", "tiny_q20", "zero_q7", "nat_q7"):
        case = Case(name)
        meta, fac = parse_stream(case.encoded)
        U = np.concatenate([f.ravel() for f in fac[0::2]])[None]
        V = np.concatenate([f.ravel() for f in fac[1::2]])[None]
        U3, V3 = np.repeat(U, 3, 0), np.repeat(V, 3, 0)
        out = pack_streams_native(U3, V3, case.image.shape[-2:], meta["rank"], meta["bounds"], meta["patch size"], threads=2)
        assert out == [case.encoded] * 3


def test_native_packer_survives_fork():
    """the packer's worker threads outlive a call; a forked child must not reuse the parent's pool (its
    condition variables still count the parent's sleeping workers as waiters: the child's SECOND call used to hang)."""
    import signal

    from lrf_amd.codec import pack_streams_native
    rng = np.random.default_rng(5)
    H, W, ranks = 64, 96, (4, 2, 2)
    from lrf_amd import _lib
    dims = _lib.plane_dims(H, W)
    U = rng.integers(-16, 16, (6, sum(d[4] * r for d, r in zip(dims, ranks))), dtype=np.int8)
    V = rng.integers(-16, 16, (6, 64 * sum(ranks)), dtype=np.int8)
    want = pack_streams_native(U, V, (H, W), ranks, (-16, 15), threads=4)
    assert pack_streams_native(U, V, (H, W), ranks, (-16, 15), threads=4) == want  # the pool's threads exist and sleep
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child: three calls (the second one used to dead-lock), then report through the pipe
        code = 1
        try:
            signal.alarm(60)
            ok = all(pack_streams_native(U, V, (H, W), ranks, (-16, 15), threads=4) == want for _ in range(3))
            os.write(w, b"ok" if ok else b"bad")
            code = 0
        finally:
            os._exit(code)
    os.close(w)
    status = os.waitpid(pid, 0)[1]
    got = os.read(r, 16)
    os.close(r)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"child ended with status {status:#x} (SIGALRM = hang)"
    assert got == b"ok"
    assert pack_streams_native(U, V, (H, W), ranks, (-16, 15), threads=4) == want  # the parent's pool still works


def test_native_unpacker_matches_python_parser_and_refuses_crafted_streams():
    """lrf_pack_unpack_qmf_factors (the decode side of liblrf_pack.so): the factors it inflates from the reference's own
    streams equal this module's Python container code; a stream that is not exactly the int8 / column layout its metadata
    describes — truncated, corrupt, another rank — is refused (the Python path then raises its error)."""
    from lrf_amd import codec
    from lrf_amd.container import combine_bytes, dict_to_bytes, encode_matrix, separate_bytes
    for name in ("s1_r7", "odd_q7", "tiny_rank1", "tiny_q20", "zero_q7", "nat_q7"):
        case = Case(name)
        streams = [case.encoded] * 3
        got = codec._factors_native(streams)
        assert got is not None, name
        metas, U, V = got
        pm, PU, PV = codec._factors_python(streams)
        assert metas == pm and np.array_equal(U, PU) and np.array_equal(V, PV)
    case = Case("s1_r7")
    meta_b, fac_b = separate_bytes(case.encoded, 2)
    assert codec._factors_native([combine_bytes([meta_b, fac_b[:-7]])]) is None          # truncated
    broken = bytearray(fac_b)
    broken[len(broken) // 2] ^= 0x55
    assert codec._factors_native([combine_bytes([meta_b, bytes(broken)])]) is None       # corrupt deflate data
    meta, fac = codec.parse_stream(case.encoded)
    wrong = [f if i else f[:, :-1] for i, f in enumerate(fac)]                           # u_Y with one column less
    crafted = combine_bytes([dict_to_bytes(meta), combine_bytes([encode_matrix(np.ascontiguousarray(f)) for f in wrong])])
    assert codec._factors_native([crafted]) is None
    with pytest.raises(ValueError):
        codec._factors_python([crafted])
    wide = [f.astype(np.int16) for f in fac]                                             # another dtype
    crafted = combine_bytes([dict_to_bytes(meta), combine_bytes([encode_matrix(np.ascontiguousarray(f)) for f in wide])])
    assert codec._factors_native([crafted]) is None


def test_ssim_matches_direct_window_statistics():
    """metrics.ssim against a direct evaluation of the SSIM definition on every full 7x7 window (the cropped region
    never touches the border, so the filter's border mode drops out) — and its defining properties."""
    import numpy as np
    import torch
    from numpy.lib.stride_tricks import sliding_window_view

    from lrf_amd.metrics import ssim

    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (3, 24, 31), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    rng_a = float(a.max() - a.min())
    c1, c2 = (0.01 * rng_a) ** 2, (0.03 * rng_a) ** 2
    vals = []
    for x, y in zip(a.astype(np.float64), b.astype(np.float64)):
        wx = sliding_window_view(x, (7, 7)).reshape(-1, 49)
        wy = sliding_window_view(y, (7, 7)).reshape(-1, 49)
        ux, uy = wx.mean(1), wy.mean(1)
        vx, vy = wx.var(1, ddof=1), wy.var(1, ddof=1)
        vxy = ((wx - ux[:, None]) * (wy - uy[:, None])).sum(1) / 48.0
        vals.append((((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))).mean())
    got = ssim(torch.from_numpy(a), torch.from_numpy(b)).item()
    assert abs(got - float(np.mean(vals))) < 1e-9
    assert abs(ssim(torch.from_numpy(a), torch.from_numpy(a)).item() - 1.0) < 1e-12
    assert got < 1.0
    with pytest.raises(ValueError):
        ssim(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4))


def test_crafted_streams_are_rejected_before_any_kernel_runs():
    """A stream whose factor blobs disagree with its metadata (truncated, or crafted: rank 64 in the metadata with
    2-column factors) would make the decode kernel read past its buffers: qmf_decode_batch must refuse it on the host."""
    import lrf_amd
    from lrf_amd.codec import pack_image, parse_stream, qmf_decode_batch
    case = Case("tiny_q7")
    meta, fac = parse_stream(case.encoded)
    H, W = case.image.shape[-2:]
    lying = pack_image(fac, (H, W), [64, 64, 64], meta["bounds"], meta["patch size"], meta["dtype"])  # metadata says rank 64
    with pytest.raises(ValueError, match="metadata describes"):
        qmf_decode_batch([lying])
    short = [f.copy() for f in fac]
    short[2] = short[2][:-1]  # one patch row missing in U_Cb
    with pytest.raises(ValueError, match="metadata describes"):
        qmf_decode_batch([pack_image(short, (H, W), meta["rank"], meta["bounds"], meta["patch size"], meta["dtype"])])
    wide = [f.astype(np.int16) for f in fac]
    with pytest.raises(ValueError, match="metadata describes"):
        qmf_decode_batch([pack_image(wide, (H, W), meta["rank"], meta["bounds"], meta["patch size"], meta["dtype"])])
    other = Case("tiny_r7")
    with pytest.raises(ValueError, match="differ"):
        qmf_decode_batch([case.encoded, other.encoded])
    assert lrf_amd is not None
