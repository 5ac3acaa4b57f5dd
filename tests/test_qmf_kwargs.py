"""qmf_encode(**kwargs) beyond the tuned configuration: the reference forwards `l2`, `l1_ratio`, `eps`,
`num_levels` to QMF(rank, bounds, factor=(0, 1), **kwargs) per matrix (lrf/compression/qmf.py:127, 189, 208, 256, 280).
Five reference fixtures (tools/gen_golden.py qmfkw): 8x8 patches with an elastic-net term, SVDInit(num_levels=...), no
patches, 4x4 patches with another eps, the RGB colour space.  CPU: the oracle's general loop from the reference's initial
factors + the container = the reference's bytes.  GPU: lrf_amd.qmf_encode from the same initial factors emits the
reference's bytes; with its own initialisation it lands within tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CASES = ["kw_l2_q20", "kw_levels_r3", "kw_nopatch_l2", "kw_p4_eps", "kw_rgb_l2"]


def _case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = json.loads(str(z["kwargs"]))
    for k in ("patch_size", "l2"):
        if isinstance(kw.get(k), list):
            kw[k] = tuple(kw[k])
    return z, kw


def _split(kw):
    enc = {k: v for k, v in kw.items() if k in ("rank", "quality", "color_space", "patch", "patch_size")}
    opt = {k: v for k, v in kw.items() if k in ("l2", "l1_ratio", "eps", "num_levels")}
    return enc, opt


@pytest.mark.parametrize("name", CASES)
def test_oracle_from_reference_init_reproduces_reference_bytes(name, oracle):
    from lrf_amd.codec import pack_anyshape
    from lrf_amd.container import combine_bytes, dict_to_bytes, encode_tensor
    z, kw = _case(name)
    enc_kw, opt = _split(kw)
    img = z["image"]
    H, W = img.shape[-2:]
    ps = tuple(enc_kw.get("patch_size", (8, 8))) if enc_kw.get("patch", True) else None
    rgb = enc_kw.get("color_space", "YCbCr") == "RGB"
    mats = [oracle.rgb_matrix_any(img, ps)] if rgb else oracle.anyshape_matrices(img, ps)
    ranks = [int(r) for r in z["ranks"]]
    fac = []
    for c, (X, R) in enumerate(zip(mats, ranks)):
        U, V, _ = oracle.bcd_ex(X, z[f"u0_{c}"], z[f"v0_{c}"], 10, bounds=(-16, 15), l2=opt.get("l2", 0.0), l1_ratio=opt.get("l1_ratio", 0.0),
                                factor=(0, 1), w=z[f"w0_{c}"], eps=opt.get("eps", 1e-16))
        fac += [torch.from_numpy(U).to(torch.int8).numpy(), torch.from_numpy(V).to(torch.int8).numpy()]
    if rgb:
        Hp, Wp = H + (8 - H % 8) % 8, W + (8 - W % 8) % 8
        meta = {"dtype": "uint8", "color space": "RGB", "patch": True, "bounds": (-16, 15), "patch size": (8, 8), "original size": [H, W],
                "padded size": [Hp, Wp], "rank": ranks[0]}
        stream = combine_bytes([dict_to_bytes(meta), combine_bytes([encode_tensor(f) for f in fac])])
    else:
        if ps is None:
            fac = [f[None] for f in fac]  # patch=False keeps the plane's channel axis (qmf.py:281-282)
        stream = pack_anyshape(fac, (H, W), ranks, (-16, 15), ps, "uint8")
    assert stream == z["encoded"].tobytes()


def test_keywords_the_reference_rejects_are_rejected():
    """`factor` and `project` are passed by the reference itself (qmf.py:188, 256): a caller's copy is a duplicate keyword
    (TypeError) there, and anything CoordinateDescent does not know is a TypeError too — before any GPU work here."""
    import lrf_amd
    img = torch.zeros((3, 16, 16), dtype=torch.uint8)
    for bad in (dict(factor=(0, 1)), dict(project=None), dict(momentum=0.9)):
        with pytest.raises(TypeError):
            lrf_amd.qmf_encode(img, quality=10, **bad)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_encode_with_qmf_options(name):
    import lrf_amd
    z, kw = _case(name)
    img = torch.from_numpy(z["image"])
    n = len(z["ranks"])
    init = [(z[f"u0_{c}"], z[f"v0_{c}"], z[f"w0_{c}"]) for c in range(n)]
    enc = lrf_amd.qmf_encode(img, init=init, **kw)
    assert enc == z["encoded"].tobytes(), "from the reference's initial factors the stream is the reference's"
    own = lrf_amd.qmf_encode(img, **kw)  # the library's own initialisation
    dec = lrf_amd.qmf_decode(own)
    assert tuple(dec.shape) == tuple(img.shape)
    p = lrf_amd.psnr(img, dec).item()
    assert abs(len(own) - len(z["encoded"])) <= 0.12 * len(z["encoded"]), (len(own), len(z["encoded"]))
    if "num_levels" not in kw:  # with num_levels the stream drops the affine pair w (qmf.py:257): its "image" is not one (8.5 dB)
        assert abs(p - float(z["psnr"])) < 0.6, (p, float(z["psnr"]))
