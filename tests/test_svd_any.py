"""svd_encode / svd_decode, RGB branch beyond the default (lrf/compression/svd.py:157-193, 310-326): other patch sizes,
patch=False, float factors.  Fixtures: the reference's streams (tools/gen_golden.py svd_any).  Parity as for the default
branch: decoding the reference's bytes is bit-exact; the encoder — own SVD instead of LAPACK — by
tolerance: same metadata, PSNR within 0.05 dB, stream size within 5 % (the quantisation scale is only sanity-checked: a
column whose sign differs from LAPACK's moves the tensor's min / max)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CASES = ["svdany_p4_q5", "svdany_p16_r6", "svdany_nopatch_q6", "svdany_p8_float", "svdany_nopatch_float"]


@pytest.mark.parametrize("name", CASES)
def test_svd_any_against_reference(name):
    import lrf_amd
    from lrf_amd.container import bytes_to_dict, separate_bytes
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = json.loads(str(z["kwargs"]))
    if "dtype" in kw:
        kw["dtype"] = getattr(torch, kw["dtype"])
    img = torch.from_numpy(z["image"])
    ref_enc = z["encoded"].tobytes()
    dec = lrf_amd.svd_decode(ref_enc)
    assert np.array_equal(dec.numpy(), z["decoded"]), "decode of the reference's stream must be bit-exact"
    enc = lrf_amd.svd_encode(img, **kw)
    meta, ref_meta = bytes_to_dict(separate_bytes(enc, 2)[0]), bytes_to_dict(separate_bytes(ref_enc, 2)[0])
    assert {k: v for k, v in meta.items() if k != "quantization"} == {k: v for k, v in ref_meta.items() if k != "quantization"}
    for key in ("u", "v"):
        got, ref = meta["quantization"][key], ref_meta["quantization"][key]
        assert (got is None) == (ref is None)
        if ref is not None:  # a column whose sign differs from LAPACK's moves the tensor's min / max, hence the scale
            assert got[0] > 0 and abs(got[0] / ref[0] - 1) < 0.25, (key, got, ref)
    mine = lrf_amd.svd_decode(enc)
    assert abs(lrf_amd.psnr(img, mine).item() - float(z["psnr"])) < 0.05
    assert abs(len(enc) / len(ref_enc) - 1) < 0.05


@pytest.mark.parametrize("name", CASES + ["svd_tiny_q2p5", "svd_smooth_q2p5", "svd_s1_q2p5"])
def test_svd_encode_equals_oracle_bytes(name, oracle):
    """The same encodes against the oracle run the way the library runs (matrix, top-R factors by the restated eigen-solver of
    lrf_oracle_any.c — the exact Gram matrix for the default 8x8 / uint8 branch —, per-tensor uint8 quantisation,
    container): byte for byte.  What separates both from the reference is LAPACK's SVD, bounded by the test above."""
    import lrf_amd
    from conftest import make_image
    from lrf_amd.container import combine_bytes, dict_to_bytes, encode_tensor
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = json.loads(str(z["kwargs"]))
    if "dtype" in kw:
        kw["dtype"] = getattr(torch, kw["dtype"])
    img = torch.from_numpy(z["image"]) if "image" in z else make_image(json.loads(str(z["spec"])))
    enc = lrf_amd.svd_encode(img, **kw)
    patch = kw.get("patch", True)
    ps = tuple(kw.get("patch_size", (8, 8))) if patch else None
    dtype = kw.get("dtype", torch.uint8)
    X = oracle.rgb_matrix_any(img.numpy(), ps)
    mats = [X] if ps is not None else list(X)
    M, N = mats[0].shape
    R = kw["rank"] if kw.get("rank") is not None else max(round(min(M, N) * kw["quality"] / 100), 1)
    fused = ps == (8, 8) and dtype is torch.uint8
    uv = [oracle.svd_topr_u8(Xm, R) if fused else oracle.svd_topr_any(Xm, R) for Xm in mats]
    u = uv[0][0] if ps is not None else np.stack([a for a, _ in uv])
    v = uv[0][1] if ps is not None else np.stack([b for _, b in uv])
    H, W = img.shape[-2:]
    metadata = {"dtype": "uint8", "color space": "RGB", "patch": patch}
    if ps is not None:
        Hp, Wp = H + (ps[0] - H % ps[0]) % ps[0], W + (ps[1] - W % ps[1]) % ps[1]
        metadata.update({"patch size": kw.get("patch_size", (8, 8)), "original size": [H, W], "padded size": [Hp, Wp]})
    if dtype is torch.uint8:
        qu, su, mu = oracle.quantize_u8(u)
        qv, sv, mv = oracle.quantize_u8(v)
        metadata["quantization"] = {"u": [su, mu], "v": [sv, mv]}
        factors = [qu, qv]
    else:
        metadata["quantization"] = {"u": None, "v": None}
        factors = [u, v]
    want = combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(np.ascontiguousarray(f)) for f in factors])])
    assert enc == want


def test_svd_ycbcr_branch_is_refused_by_name():
    import lrf_amd
    img = torch.zeros((3, 32, 32), dtype=torch.uint8)
    with pytest.raises(NotImplementedError, match="defective in the reference"):
        lrf_amd.svd_encode(img, quality=5, color_space="YCbCr")
