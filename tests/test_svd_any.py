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


def test_svd_ycbcr_branch_is_refused_by_name():
    import lrf_amd
    img = torch.zeros((3, 32, 32), dtype=torch.uint8)
    with pytest.raises(NotImplementedError, match="defective in the reference"):
        lrf_amd.svd_encode(img, quality=5, color_space="YCbCr")
