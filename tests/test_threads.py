"""The reference at 8 torch threads against itself at one thread, and where the oracle / HIP results stand between the two
(VERDICT r02 item 8).  MKL splits the long `x.mT @ u` reduction (lrf/factorization/qmf.py:107 via :139) differently with
more threads, so the reference's own byte stream depends on torch.get_num_threads(): every other fixture of this suite is a
ONE-thread run.  tests/golden/threads8.json (tools/gen_golden.py threads) records, for ten images up to 1365x2048, what a
default multi-threaded reference user gets instead.  Measured there:
  * on 6 of the 10 images the reference's stream does not depend on the thread count — and on exactly those the oracle (own
    initialisation, the reference's LAPACK column signs) and the HIP encoder emit that stream byte for byte;
  * on the other 4 (two random 512x768 / 1365x2048 images, one smooth 512x768, one random 1365x2048 at quality 7) the 8-thread
    stream differs from the 1-thread one in up to 7.2 % of the int8 factor entries (by at most 6), in PSNR by at most 0.0034 dB
    and in size by at most 0.52 %.  These are the images on which the iteration is sensitive to ANY perturbation of the order
    of 1e-7: there the oracle / HIP result — whose initialisation is the exact top-R pairs rather than LAPACK's fp32 ones — is
    a third variant, at the same distance from either reference stream as they are from each other (6.5-6.9 % of the entries,
    PSNR within 0.003 dB, size within 0.7 %).  From the reference's own initial factors the oracle reproduces the 1-thread
    iteration bit for bit on them as well (tools/pin_oracle.py does that check on 30 planes).
So "byte-identical with the reference's signs" holds where the reference is identical to itself; the bound below is the
tolerance everywhere else."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_image

RECORDS = json.load(open(os.path.join(GOLDEN, "threads8.json")))
STABLE = [r for r in RECORDS if r["t1"]["sha256"] == r["tN"]["sha256"]]
SENSITIVE = [r for r in RECORDS if r["t1"]["sha256"] != r["tN"]["sha256"]]
# the stated bound (head-room over the measured maxima in the docstring): reference vs itself, and oracle / HIP vs either
MAX_ENTRY_FRACTION, MAX_PSNR_DB, MAX_SIZE_FRACTION = 0.10, 0.01, 0.01


def _image(rec):
    import torch
    if rec["spec"]["kind"] == "natural":
        img = torch.from_numpy(np.load(os.path.join(GOLDEN, "nat_q7.npz"))["image"])
    else:
        img = make_image(rec["spec"])
    assert hashlib.sha256(img.numpy().tobytes()).hexdigest() == rec["image_sha256"]
    return img


def _oracle_encode(oracle, rec, img):
    from lrf_amd.codec import pack_image
    X = oracle.rgb_to_planes(img.numpy())
    fac = []
    for c in range(3):
        u, v = oracle.qmf_decompose(X[c], rec["ranks"][c], 10, (-16, 15), sign=np.array(rec["signs"][c], np.int8))
        fac += [u.astype(np.int8), v.astype(np.int8)]
    return fac, pack_image(fac, tuple(img.shape[-2:]), rec["ranks"], (-16, 15), (8, 8), "uint8")


def _psnr(a, b):
    return 20 * np.log10(255 / np.sqrt(np.mean((np.asarray(a, np.float32) - np.asarray(b, np.float32)) ** 2)))


def test_reference_against_itself_stays_inside_the_stated_bound():
    assert len(SENSITIVE) >= 3, "the fixture is meant to hold cases where the thread count changes the reference's bytes"
    for rec in RECORDS:
        a, b = rec["t1"], rec["tN"]
        assert rec["differing_entries"] <= MAX_ENTRY_FRACTION * rec["entries"], rec["name"]
        assert abs(a["psnr"] - b["psnr"]) <= MAX_PSNR_DB and abs(a["len"] - b["len"]) <= MAX_SIZE_FRACTION * a["len"], rec["name"]


@pytest.mark.parametrize("rec", STABLE, ids=[r["name"] for r in STABLE])
def test_oracle_emits_the_reference_bytes_where_the_reference_is_thread_independent(rec, oracle):
    _, stream = _oracle_encode(oracle, rec, _image(rec))
    assert hashlib.sha256(stream).hexdigest() == rec["t1"]["sha256"]


@pytest.mark.parametrize("rec", SENSITIVE, ids=[r["name"] for r in SENSITIVE])
def test_oracle_is_no_farther_from_either_reference_than_the_bound(rec, oracle):
    """thread-sensitive images: the oracle's stream against the 1-thread and the 8-thread reference (entry distances where
    both streams are kept — the 512x768 cases —, size and PSNR everywhere)"""
    from lrf_amd.container import decode_tensor, separate_bytes
    img = _image(rec)
    fac, stream = _oracle_encode(oracle, rec, img)
    H, W = img.shape[-2:]
    psnr = _psnr(img.numpy(), oracle.planes_to_rgb(fac[0::2], fac[1::2], H, W))
    for ref in (rec["t1"], rec["tN"]):
        assert abs(psnr - ref["psnr"]) <= MAX_PSNR_DB and abs(len(stream) - ref["len"]) <= MAX_SIZE_FRACTION * ref["len"]
    kept = np.load(os.path.join(GOLDEN, "threads8_streams.npz"))
    if rec["name"] + "_t1" in kept:
        for key in ("_t1", "_tN"):
            ref = [decode_tensor(f) for f in separate_bytes(separate_bytes(kept[rec["name"] + key].tobytes(), 2)[1], 6)]
            ndiff = sum(int((a != b).sum()) for a, b in zip(fac, ref))
            assert ndiff <= MAX_ENTRY_FRACTION * rec["entries"], (key, ndiff)
            assert ndiff <= 1.5 * rec["differing_entries"], "farther from a reference stream than the references are from each other"


@pytest.mark.gpu
@pytest.mark.parametrize("rec", RECORDS, ids=[r["name"] for r in RECORDS])
def test_hip_equals_oracle_and_the_reference_where_it_is_thread_independent(rec, oracle):
    """HIP == oracle byte for byte on all ten (1365x2048 included); == the reference's stream on the thread-independent six;
    inside the bound against both reference streams on the others."""
    import lrf_amd
    img = _image(rec)
    sign = np.concatenate([np.array(s, np.int8) for s in rec["signs"]])
    enc = lrf_amd.qmf_encode(img, init_sign=sign, **rec["kwargs"])
    _, stream = _oracle_encode(oracle, rec, img)
    assert enc == stream
    if rec["t1"]["sha256"] == rec["tN"]["sha256"]:
        assert hashlib.sha256(enc).hexdigest() == rec["t1"]["sha256"]
    psnr = lrf_amd.psnr(img, lrf_amd.qmf_decode(enc)).item()
    for ref in (rec["t1"], rec["tN"]):
        assert abs(psnr - ref["psnr"]) <= MAX_PSNR_DB and abs(len(enc) - ref["len"]) <= MAX_SIZE_FRACTION * ref["len"]
