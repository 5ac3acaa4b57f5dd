"""RGB colour-space branch of qmf_encode / qmf_decode (lrf/compression/qmf.py:164-187, 309-323):
one [M,192] matrix per image.  Fixtures: tools/gen_golden.py rgbspace (reference run at one thread)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_image

RGBSP_CASES = ["rgbsp_tiny_q4", "rgbsp_tiny_r1", "rgbsp_tiny_r3_it2", "rgbsp_odd_q6", "rgbsp_smooth_q2", "rgbsp_smooth_q10",
               "rgbsp_nat_q5"]


class RgbCase:
    def __init__(self, name):
        import hashlib
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z, self.name = z, name
        self.spec = json.loads(str(z["spec"]))
        self.kwargs = json.loads(str(z["kwargs"]))
        self.encoded = z["encoded"].tobytes()
        self.R = int(z["rank"])
        self.K = self.kwargs.get("num_iters", 10)
        if "image" in z:
            self.image = torch.from_numpy(z["image"])
        elif self.spec["kind"] == "natural":
            self.image = torch.from_numpy(np.load(os.path.join(GOLDEN, "nat_q7.npz"))["image"])
        else:
            self.image = make_image(self.spec)
        assert hashlib.sha256(self.image.numpy().tobytes()).hexdigest() == str(z["image_sha256"])
        self.psnr = float(z["psnr"])
        self.decoded_sha256 = str(z["decoded_sha256"])

    def ref_factors(self):
        from lrf_amd.container import decode_tensor, separate_bytes
        _, fac = separate_bytes(self.encoded, 2)
        return [decode_tensor(f) for f in separate_bytes(fac, 2)]


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _psnr(a, b):
    mse = np.mean((np.asarray(a, np.float32) - np.asarray(b, np.float32)) ** 2)
    return 20 * np.log10(255 / np.sqrt(mse))


# ---------------------------------------------------------------- CPU: the oracle against the reference
@pytest.mark.parametrize("name", RGBSP_CASES)
def test_oracle_decode_and_bcd_match_reference(name, oracle):
    c = RgbCase(name)
    u_ref, v_ref = c.ref_factors()
    H, W = c.image.shape[-2:]
    assert _sha(oracle.qmf_rgbspace_decode(u_ref, v_ref, H, W)) == c.decoded_sha256
    # from the reference's own initial factors the K iterations give the reference's int8 factors, bit for bit
    u, v = oracle.qmf_rgbspace_decompose(c.image.numpy(), c.R, c.K, init=(c.z["u0"], c.z["v0"]))
    assert np.array_equal(u.astype(np.int8), u_ref) and np.array_equal(v.astype(np.int8), v_ref)


@pytest.mark.parametrize("name", ["rgbsp_tiny_q4", "rgbsp_smooth_q2", "rgbsp_smooth_q10"])
def test_oracle_own_init_reaches_reference_quality(name, oracle):
    c = RgbCase(name)
    H, W = c.image.shape[-2:]
    u, v = oracle.qmf_rgbspace_decompose(c.image.numpy(), c.R, c.K, sign=c.z["sign"])
    dec = oracle.qmf_rgbspace_decode(u, v, H, W)
    assert abs(_psnr(c.image.numpy(), dec) - c.psnr) < 0.05


def test_rgbspace_metadata_and_rank_rule_without_gpu():
    from lrf_amd.codec import rgbspace_dims
    assert rgbspace_dims(173, 264) == (176, 264, 726) and rgbspace_dims(512, 768) == (512, 768, 6144)
    # max(round(min(M, 192) * q / 100), 1), Python's round (half to even) — the ranks the fixtures record
    for name in RGBSP_CASES:
        c = RgbCase(name)
        if "quality" in c.kwargs:
            M = rgbspace_dims(*c.image.shape[-2:])[2]
            assert max(round(min(M, 192) * c.kwargs["quality"] / 100), 1) == c.R


# ---------------------------------------------------------------- GPU: the HIP path through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("name", RGBSP_CASES)
def test_hip_decode_is_bit_exact(name):
    import lrf_amd
    c = RgbCase(name)
    dec = lrf_amd.qmf_decode(c.encoded)
    assert dec.dtype == torch.uint8 and tuple(dec.shape) == tuple(c.image.shape)
    assert _sha(dec.numpy()) == c.decoded_sha256


@pytest.mark.gpu
@pytest.mark.parametrize("name", RGBSP_CASES)
def test_hip_encode_from_reference_init_reproduces_reference_bytes(name):
    """With the reference's initial factors the HIP encoder emits the reference's byte stream."""
    import lrf_amd
    c = RgbCase(name)
    init = (torch.from_numpy(c.z["u0"]).unsqueeze(0), torch.from_numpy(c.z["v0"]).unsqueeze(0))
    enc = lrf_amd.qmf_encode(c.image, color_space="RGB", init=init, **c.kwargs)
    assert enc == c.encoded


@pytest.mark.gpu
@pytest.mark.parametrize("name", RGBSP_CASES)
def test_hip_encode_own_init(name, oracle):
    """Own SVD initialisation (general-N eigen-solver of the SVD baseline; parity by tolerance for this branch):
    same container fields, the reference's quality and size."""
    import lrf_amd
    from lrf_amd.container import bytes_to_dict, separate_bytes
    c = RgbCase(name)
    sign = torch.from_numpy(c.z["sign"].astype(np.int8))
    enc = lrf_amd.qmf_encode(c.image, color_space="RGB", init_sign=sign, **c.kwargs)
    meta, ref_meta = (bytes_to_dict(separate_bytes(e, 2)[0]) for e in (enc, c.encoded))
    assert meta == ref_meta
    dec = lrf_amd.qmf_decode(enc)
    assert abs(_psnr(c.image.numpy(), dec.numpy()) - c.psnr) < 0.1
    assert abs(len(enc) - len(c.encoded)) <= 0.03 * len(c.encoded) + 16
    # against the oracle run the same way (exact Gram matrix + the restated eigen-solver, lrf_oracle_any.c): bit for bit
    u, v = oracle.qmf_rgbspace_decompose(c.image.numpy(), c.R, c.K, sign=c.z["sign"])
    from lrf_amd.container import decode_tensor
    uh, vh = (decode_tensor(f) for f in separate_bytes(separate_bytes(enc, 2)[1], 2))
    assert np.array_equal(uh, u.astype(np.int8)) and np.array_equal(vh, v.astype(np.int8))


@pytest.mark.gpu
def test_hip_rgbspace_rejects_what_is_not_implemented():
    import lrf_amd
    img = torch.randint(0, 256, (3, 64, 96), dtype=torch.uint8)
    with pytest.raises(NotImplementedError):
        lrf_amd.qmf_encode(img, color_space="RGB", rank=193)  # more columns than the matrix has
    with pytest.raises(ValueError):
        lrf_amd.qmf_encode(img, color_space="RGB", rank=(4, 2, 2))  # the RGB branch takes a scalar rank
    # num_iters=0, patch=False and the other patch sizes run since round 2: RGBANY_CASES below


@pytest.mark.gpu
def test_hip_bcd_rank_30_equals_oracle(oracle):
    """ranks 25..32 of the RGB colour-space kernels (beyond the reference's sweep): BCD from given factors, bit for bit"""
    from lrf_amd import _lib
    rng = np.random.default_rng(30)
    img = rng.integers(0, 256, (3, 72, 104), dtype=np.uint8)
    X = oracle.pad_patchify(img.astype(np.float32))
    R = 30
    u0 = (rng.normal(size=(X.shape[0], R)) * 3).astype(np.float32)
    v0 = (rng.normal(size=(192, R)) * 3).astype(np.float32)
    ctx = _lib.context()
    U, V = ctx.qmf_rgbspace_encode(torch.from_numpy(img).cuda().unsqueeze(0), R, 3, (-16, 15), None,
                                   (torch.from_numpy(u0[None]), torch.from_numpy(v0[None])))
    u, v = oracle.bcd(X, u0, v0, 3, (-16, 15))
    assert np.array_equal(U[0].cpu().numpy(), u.astype(np.int8)) and np.array_equal(V[0].cpu().numpy(), v.astype(np.int8))


# ---------------------------------------------------------------- the same branch for other patch sizes, patch=False, num_iters=0
RGBANY_CASES = ["rgbany_p4_q6", "rgbany_p16_r5", "rgbany_p8x4_q3", "rgbany_nopatch_q8", "rgbany_nopatch_r2", "rgbany_p8_it0"]


class RgbAnyCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.kwargs = json.loads(str(z["kwargs"]))
        self.encoded = z["encoded"].tobytes()
        self.image = torch.from_numpy(z["image"])
        self.decoded = z["decoded"]
        self.R, self.psnr = int(z["rank"]), float(z["psnr"])
        self.K = self.kwargs.get("num_iters", 10)
        self.patch_size = tuple(self.kwargs.get("patch_size", (8, 8))) if self.kwargs.get("patch", True) else None

    def ref_factors(self):
        from lrf_amd.container import decode_tensor, separate_bytes
        _, fac = separate_bytes(self.encoded, 2)
        return [decode_tensor(f) for f in separate_bytes(fac, 2)]


@pytest.mark.parametrize("name", RGBANY_CASES)
def test_oracle_rgb_any_matches_reference(name, oracle):
    """CPU: numpy matrices + the oracle's BCD from the reference's initial factors = the reference's int8 factors; the
    numpy decode of the reference's factors = the reference's pixels."""
    c = RgbAnyCase(name)
    u_ref, v_ref = c.ref_factors()
    H, W = c.image.shape[-2:]
    assert np.array_equal(oracle.rgb_decode_any(u_ref, v_ref, H, W, c.patch_size), c.decoded)
    X = oracle.rgb_matrix_any(c.image.numpy(), c.patch_size)
    u0, v0 = c.z["u0"], c.z["v0"]
    if c.patch_size is not None:
        mats, u0s, v0s = [X], [u0[0]], [v0[0]]
    else:
        mats, u0s, v0s = list(X), list(u0), list(v0)
    us, vs = [], []
    for Xm, a, b in zip(mats, u0s, v0s):
        if c.K == 0:
            u, v = torch.from_numpy(a).to(torch.int8).numpy(), torch.from_numpy(b).to(torch.int8).numpy()  # the truncating cast
        else:
            u, v = oracle.bcd(Xm, a, b, c.K, (-16, 15))
        us.append(u.astype(np.int8)); vs.append(v.astype(np.int8))
    u = us[0] if c.patch_size is not None else np.stack(us)
    v = vs[0] if c.patch_size is not None else np.stack(vs)
    assert np.array_equal(u, u_ref) and np.array_equal(v, v_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RGBANY_CASES)
def test_hip_rgb_any_reproduces_reference(name, oracle):
    import lrf_amd
    from lrf_amd import _lib
    c = RgbAnyCase(name)
    H, W = c.image.shape[-2:]
    ctx = _lib.context(0)
    X = ctx.rgbspace_matrix_any(c.image.cuda().unsqueeze(0), c.patch_size)[0].cpu().numpy()
    assert np.array_equal(X, oracle.rgb_matrix_any(c.image.numpy(), c.patch_size))
    dec = lrf_amd.qmf_decode(c.encoded)
    assert np.array_equal(dec.numpy(), c.decoded)
    enc = lrf_amd.qmf_encode(c.image, color_space="RGB", init=(c.z["u0"], c.z["v0"]), **c.kwargs)
    assert enc == c.encoded, "from the reference's initial factors the encoder must emit the reference's bytes"
    own = lrf_amd.qmf_encode(c.image, color_space="RGB", **c.kwargs)  # own initialisation: by tolerance against LAPACK ...
    d2 = lrf_amd.qmf_decode(own)
    assert abs(_psnr(c.image.numpy(), d2.numpy()) - c.psnr) < (1.5 if c.K == 0 else 0.3)
    assert abs(len(own) / len(c.encoded) - 1) < 0.08
    # ... and bit for bit against the oracle run the same way (8x8 patches: exact Gram matrix of the [M,192] matrix; other
    # shapes: the short-side eigen-problem of the any-shape path)
    from lrf_amd.container import decode_tensor, separate_bytes
    Xo = oracle.rgb_matrix_any(c.image.numpy(), c.patch_size)
    us, vs = [], []
    for Xm in ([Xo] if c.patch_size is not None else list(Xo)):
        u0, v0 = oracle.svd_topr_u8(Xm, c.R) if c.patch_size == (8, 8) else oracle.svd_topr_any(Xm, c.R)
        if c.K == 0:
            u, v = torch.from_numpy(u0).to(torch.int8).numpy(), torch.from_numpy(v0).to(torch.int8).numpy()  # the truncating cast
        else:
            u, v = oracle.bcd(Xm, u0, v0, c.K, (-16, 15))
        us.append(u.astype(np.int8)); vs.append(v.astype(np.int8))
    uo = us[0] if c.patch_size is not None else np.stack(us)
    vo = vs[0] if c.patch_size is not None else np.stack(vs)
    uh, vh = (decode_tensor(f) for f in separate_bytes(separate_bytes(own, 2)[1], 2))
    assert np.array_equal(uh, uo) and np.array_equal(vh, vo), "own-initialisation factors differ from the oracle's"
