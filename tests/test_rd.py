"""LOESS smoothing / R-D interpolation (lrf_amd/rd.py) against the reference's class (lrf/utils/misc.py:276-472;
fixture: tools/gen_golden.py loess)."""
import json
import os

import numpy as np

from conftest import GOLDEN


def test_loess_matches_reference():
    from lrf_amd.rd import LOESS
    fx = json.load(open(os.path.join(GOLDEN, "loess.json")))
    for case in fx.values():
        x, y, grid = np.array(case["x"]), np.array(case["y"]), np.array(case["grid"])
        single = LOESS(frac=0.3, degree=1).fit(x, y).predict(grid)
        assert np.allclose(single, case["single"], rtol=1e-9, atol=1e-9)
        model = LOESS(frac=np.arange(0.15, 0.75, 0.1), degree=[1, 2]).fit(x, y)
        assert abs(model.best_frac - case["best_frac"]) < 1e-12 and int(model.best_degree) == case["best_degree"]
        assert np.allclose(model.predict(grid), case["searched"], rtol=1e-9, atol=1e-9)


def test_interpolate_records_groups_and_flags():
    from lrf_amd.rd import interpolate_records
    rng = np.random.default_rng(1)
    recs = []
    for data in ("img0", "img1"):
        for bpp in np.linspace(0.1, 1.0, 30):
            recs.append({"data": data, "method": "QMF", "bit rate (bpp)": float(bpp),
                         "PSNR (dB)": float(20 + 10 * np.log1p(bpp) + rng.normal(0, 0.05))})
    recs.append(dict(recs[0]))  # duplicate x: dropped like drop_duplicates
    out = interpolate_records(recs, "bit rate (bpp)", "PSNR (dB)", np.linspace(0.05, 1.05, 11), frac=[0.3], degree=[1])
    assert len(out) == 22 and [r["data"] for r in out[:11]] == ["img0"] * 11
    assert out[0]["extrapolated"] and out[10]["extrapolated"] and not out[5]["extrapolated"]
    mid = [r for r in out if not r["extrapolated"]]
    assert all(abs(r["PSNR (dB)"] - (20 + 10 * np.log1p(r["bit rate (bpp)"]))) < 0.15 for r in mid)
