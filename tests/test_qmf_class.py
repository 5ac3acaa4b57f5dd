"""The QMF class beyond what qmf_encode uses (lrf/factorization/qmf.py:74-231): unbounded factors, elastic-net terms, factor
subsets, the affine pair w.  Fixtures: the reference's own results (tools/gen_golden.py qmfx; the first two are the shape of
the reference's smoke test, test/test_factorization.py:5-10).  CPU part: the oracle from the reference's initial factors —
bit for bit where w stays [0; 1], by tolerance where update_w (LAPACK lstsq in the reference, normal equations here) runs.
GPU part: lrf_amd.QMF / the C ABI against the oracle and the reference."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CASES = ["qmfx_unbounded_f01", "qmfx_unbounded_f012", "qmfx_bounded_l2", "qmfx_unbounded_l2_f012", "qmfx_bounded_f0",
         "qmfx_levels_f01", "qmfx_levels_f012", "qmfx_eps"]  # the last three (round 3): SVDInit(num_levels=...), CoordinateDescent(eps=...)


def _case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    spec, kw = json.loads(str(z["spec"])), json.loads(str(z["kwargs"]))
    g = torch.Generator().manual_seed(spec["seed"])
    x = torch.randint(0, 256, (1, spec["M"], spec["N"]), generator=g).float()
    return z, kw, x


def _loss(x, u, v, w):
    y = w[0] + w[1] * (u.astype(np.float64) @ v.astype(np.float64).T)
    return float(np.linalg.norm(x - y) / (np.linalg.norm(x) + 1e-16))


def _oracle_args(kw):
    return dict(bounds=tuple(kw.get("bounds", (None, None))), l2=kw.get("l2", 0.0), l1_ratio=kw.get("l1_ratio", 0.0),
                factor=tuple(kw.get("factor", (0, 1, 2))), eps=kw.get("eps", 1e-16))


def _w0(z):
    """the initial affine pair of the fixture: [0; 1] unless SVDInit ran with num_levels (stored since round 3)"""
    return z["w0"] if "w0" in z else np.array([0.0, 1.0], np.float32)


@pytest.mark.parametrize("name", CASES)
def test_oracle_general_bcd_against_reference(name, oracle):
    z, kw, x = _case(name)
    X = x[0].numpy()
    U, V, W = oracle.bcd_ex(X, z["u0"], z["v0"], kw["num_iters"], w=_w0(z), **_oracle_args(kw))
    if 2 not in _oracle_args(kw)["factor"]:
        assert np.array_equal(U, z["u"]) and np.array_equal(V, z["v"]), "without update_w the reference is reproduced bit for bit"
        assert np.array_equal(W, z["w"])
    else:  # update_w: lstsq in the reference, normal equations here
        assert np.allclose(W, z["w"], rtol=2e-4, atol=2e-3), (W, z["w"])
        assert abs(_loss(X, U, V, W) - float(z["loss"])) < 2e-4
        assert np.mean(U == z["u"]) > 0.97 and np.mean(V == z["v"]) > 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_general_bcd(name, oracle):
    """The C ABI from the reference's initial factors: equal to the oracle bit for bit where w stays [0; 1] (and therefore
    to the reference), within tolerance of both where update_w runs."""
    from lrf_amd import _lib
    z, kw, x = _case(name)
    X = x[0].numpy()
    a = _oracle_args(kw)
    ctx = _lib.context(0)
    w_init = torch.from_numpy(_w0(z))[None] if "w0" in z and not np.array_equal(_w0(z), [0.0, 1.0]) else None
    U, V, W = ctx.decompose_ex(x.cuda(), int(kw["rank"]), kw["num_iters"], init=(torch.from_numpy(z["u0"])[None], torch.from_numpy(z["v0"])[None]),
                               w_init=w_init, **a)
    U, V, W = U[0].cpu().numpy(), V[0].cpu().numpy(), W[0].cpu().numpy()
    Uo, Vo, Wo = oracle.bcd_ex(X, z["u0"], z["v0"], kw["num_iters"], w=_w0(z), **a)
    if 2 not in a["factor"]:
        assert np.array_equal(U, Uo) and np.array_equal(V, Vo) and np.array_equal(W, Wo)
        assert np.array_equal(U, z["u"]) and np.array_equal(V, z["v"])
    else:
        assert np.allclose(W, Wo, rtol=1e-5, atol=1e-4) and np.allclose(W, z["w"], rtol=2e-4, atol=2e-3)
        assert abs(_loss(X, U, V, W) - float(z["loss"])) < 2e-4
        assert np.mean(U == Uo) > 0.99 and np.mean(V == Vo) > 0.99


@pytest.mark.gpu
def test_reference_smoke_test_runs():
    """test/test_factorization.py:5-10 of the reference, with the three-value return its decompose has: QMF(rank=5,
    num_iters=10) on randint(0, 256, (1, 784, 192)) — unbounded, w updated.  Own SVD initialisation (signs differ from
    LAPACK's, which is immaterial without bounds): the loss lands on the reference's."""
    import lrf_amd
    z, kw, x = _case("qmfx_unbounded_f012")
    qmf = lrf_amd.QMF(rank=5, num_iters=10)
    u, v, w = qmf.decompose(x)
    assert tuple(u.shape) == (1, 784, 5) and tuple(v.shape) == (1, 192, 5) and tuple(w.shape) == (1, 2, 1)
    assert torch.equal(u, torch.round(u)) and torch.equal(v, torch.round(v))
    loss = lrf_amd.QMF.loss(x, u, v, w).item()
    assert abs(loss - float(z["loss"])) < 2e-3, (loss, float(z["loss"]))
    y = qmf.forward(x)
    assert tuple(y.shape) == tuple(x.shape)
    # bounded + elastic net through the class, against the fixture's loss
    z2, kw2, x2 = _case("qmfx_bounded_l2")
    u2, v2, w2 = lrf_amd.QMF(**kw2).decompose(x2)
    assert float(u2.min()) >= -16 and float(u2.max()) <= 15
    assert abs(lrf_amd.QMF.loss(x2, u2, v2, w2).item() - float(z2["loss"])) < 5e-3


@pytest.mark.gpu
def test_num_levels_and_eps_through_the_class():
    """lrf_amd.QMF(num_levels=...) / QMF(eps=...) end to end with the library's own initialisation and the reference's LAPACK
    column signs (the scales are max - min of a factor: they move with the signs): the reference's loss to 1e-5, the same
    affine pair to 2e-3; the scaled initial factors span num_levels steps, w1 is the product of the two scales, and
    num_iters=0 returns exactly that initialisation.  With the default signs the loss stays within 2e-3."""
    import lrf_amd
    for name in ("qmfx_levels_f01", "qmfx_levels_f012", "qmfx_eps"):
        z, kw, x = _case(name)
        u, v, w = lrf_amd.QMF(init_sign=torch.from_numpy(z["sign"]), **kw).decompose(x)
        assert abs(lrf_amd.QMF.loss(x, u, v, w).item() - float(z["loss"])) < 1e-5, name
        assert np.allclose(w.reshape(-1).numpy(), z["w"], rtol=2e-3, atol=2e-2), (name, w.reshape(-1), z["w"])
        assert torch.equal(u, torch.round(u)) and torch.equal(v, torch.round(v))
        u, v, w = lrf_amd.QMF(**kw).decompose(x)
        assert abs(lrf_amd.QMF.loss(x, u, v, w).item() - float(z["loss"])) < 2e-3, name
    z, kw, x = _case("qmfx_levels_it0")
    u, v, w = lrf_amd.QMF(init_sign=torch.from_numpy(z["sign"]), **kw).decompose(x)
    assert abs(float(u.max() - u.min()) - kw["num_levels"]) < 1e-3 and abs(float(v.max() - v.min()) - kw["num_levels"]) < 1e-3
    assert abs(float(w[0, 1, 0]) - float(z["w0"][1])) < 2e-3 * float(z["w0"][1]) and float(w[0, 0, 0]) == 0.0
    assert abs(lrf_amd.QMF.loss(x, u, v, w).item() - float(z["loss"])) < 1e-5
    with pytest.raises(NotImplementedError):
        lrf_amd.QMF(rank=3, project=lambda t: t)


@pytest.mark.gpu
def test_verbose_prints_the_loss_before_every_iteration(capsys, oracle):
    """QMF(verbose=True) (lrf/factorization/qmf.py:206-212): one line per iteration, `iter k: loss = tensor([...])`, the loss of the
    factors BEFORE that iteration — the initial factors at k = 1 — in the reference's format; the values equal the oracle's
    factors' loss (numpy, fp64) to 1e-6, and the returned factors are those of the non-verbose call, bit for bit.  Tuned int8
    path (what qmf_encode configures) on two matrices of a batch, and the general path (unbounded, affine pair refitted)."""
    import re
    import lrf_amd
    from lrf_amd import _lib
    ctx = _lib.context(0)
    g = torch.Generator().manual_seed(5)
    x = (torch.randint(0, 256, (2, 480, 64), generator=g).float() * 0.5 + 40.0)
    K, R = 4, 5
    u, v, w = lrf_amd.QMF(rank=R, num_iters=K, bounds=(-16, 15), factor=(0, 1), verbose=True).decompose(x)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("iter ")]
    assert len(lines) == K and all(re.fullmatch(rf"iter {k + 1}: loss = tensor\(\[[0-9.e+-]+, [0-9.e+-]+\]\)", ln) for k, ln in enumerate(lines)), lines
    got = np.array([[float(t) for t in re.findall(r"[0-9.]+(?:e[+-]?[0-9]+)?", ln.split("tensor")[1])] for ln in lines])
    for b in range(2):
        X = x[b].numpy()
        u0, v0 = oracle.svd_init(X, R)
        for k in range(K):
            uk, vk = (u0, v0) if k == 0 else oracle.bcd(X, u0, v0, k, (-16, 15))
            want = _loss(X, uk, vk, (0.0, 1.0))
            assert abs(got[k, b] - want) < 6e-5, (b, k, got[k, b], want)  # (torch prints four decimals)
            lib = ctx.loss(x[b:b + 1].cuda(), torch.from_numpy(np.ascontiguousarray(uk, np.float32))[None].cuda(),
                           torch.from_numpy(np.ascontiguousarray(vk, np.float32))[None].cuda())
            assert abs(float(lib[0]) - want) < 1e-6 * max(1.0, want), (b, k, float(lib[0]), want)  # lrf_qmf_loss_f32 itself
    u2, v2, w2 = lrf_amd.QMF(rank=R, num_iters=K, bounds=(-16, 15), factor=(0, 1)).decompose(x)
    assert torch.equal(u, u2) and torch.equal(v, v2) and torch.equal(w, w2)
    # the general path: unbounded factors, w refitted every iteration
    u, v, w = lrf_amd.QMF(rank=3, num_iters=3, verbose=True).decompose(x[:1])
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("iter ")]
    assert len(lines) == 3
    u2, v2, w2 = lrf_amd.QMF(rank=3, num_iters=3).decompose(x[:1])
    assert torch.equal(u, u2) and torch.equal(v, v2) and torch.allclose(w, w2, rtol=0, atol=0)
    losses = [float(re.findall(r"[0-9.]+(?:e[+-]?[0-9]+)?", ln.split("tensor")[1])[0]) for ln in lines]
    assert all(0 < v_ < 1 for v_ in losses)
    assert abs(lrf_amd.QMF.loss(x[:1], u, v, w).item() - losses[-1]) < 0.05  # (the last line is the loss before the last iteration)
