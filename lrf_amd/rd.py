"""Rate-distortion aggregation of sweep records: the LOESS smoother and the per-group interpolation the reference uses to
turn (bpp, PSNR) samples into curves on a common bit-rate grid (lrf/utils/misc.py:276-412 `LOESS`, :435-472
`Plot.interpolate`).  Host-side numpy; nothing here touches the GPU.

Local regression at a point x0: the k = ceil(frac * n) nearest samples, tricube weights of their distances scaled by the
largest of them, a weighted polynomial least-squares fit of the given degree, evaluated at x0.  `frac` / `degree` may be
sequences: the pair with the smallest leave-one-out squared error is used (grid search, first best wins).
"""
import itertools
from typing import Optional, Sequence

import numpy as np


def _local_fit(x, y, x0, k, degree):
    d = np.abs(x - x0)
    idx = np.argsort(d)[:k]                     # same tie order as the reference (numpy's default sort)
    w = np.clip((1.0 - (d[idx] / d[idx][-1]) ** 3) ** 3, 0.0, 1.0)
    A = np.vander(x[idx], degree + 1) * w[:, None]
    beta = np.linalg.lstsq(A, y[idx] * w, rcond=None)[0]
    return np.polyval(beta, x0)


class LOESS:
    def __init__(self, frac=0.3, degree=1) -> None:
        self.frac = np.atleast_1d(frac)
        self.degree = np.atleast_1d(degree)
        self.x: Optional[np.ndarray] = None
        self.y: Optional[np.ndarray] = None
        self.best_frac: Optional[float] = None
        self.best_degree: Optional[int] = None

    def _loocv(self, frac: float, degree: int) -> float:
        n = len(self.x)
        err = np.zeros(n)
        k = int(np.ceil(frac * (n - 1)))
        for i in range(n):
            keep = np.arange(n) != i
            err[i] = (self.y[i] - _local_fit(self.x[keep], self.y[keep], self.x[i], k, int(degree))) ** 2
        return float(np.mean(err))

    def fit(self, x, y) -> "LOESS":
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        if len(self.frac) > 1 or len(self.degree) > 1:
            best = (float("inf"), None, None)
            for frac, degree in itertools.product(self.frac, self.degree):
                score = self._loocv(frac, degree)
                if score < best[0]:
                    best = (score, frac, degree)
            self.best_frac, self.best_degree = best[1], best[2]
        else:
            self.best_frac, self.best_degree = self.frac[0], self.degree[0]
        return self

    def predict(self, x_new) -> np.ndarray:
        x_new = np.asarray(x_new, dtype=np.float64)
        k = int(np.ceil(self.best_frac * len(self.x)))
        return np.array([_local_fit(self.x, self.y, x0, k, int(self.best_degree)) for x0 in x_new])


def interpolate_records(records: Sequence[dict], x: str, y: str, x_values, groupby=("data", "method"),
                        frac=None, degree=(1, 2)) -> list:
    """Plot.interpolate for a list of record dicts: per group (first occurrence of every x kept), a LOESS fit with the
    reference's grid (frac 0.15..0.65 step 0.1, degree 1 or 2) evaluated on `x_values`; `extrapolated` marks grid points
    outside the group's x range.  Groups come out in sorted key order, like pandas' groupby."""
    frac = np.arange(0.15, 0.75, 0.1) if frac is None else frac
    groupby = [groupby] if isinstance(groupby, str) else list(groupby)
    x_values = np.asarray(x_values, dtype=np.float64)
    groups = {}
    for rec in records:
        groups.setdefault(tuple(rec[g] for g in groupby), []).append(rec)
    out = []
    for key in sorted(groups):
        seen, xs, ys = set(), [], []
        for rec in groups[key]:
            if rec[x] not in seen:
                seen.add(rec[x])
                xs.append(rec[x])
                ys.append(rec[y])
        pred = LOESS(frac=frac, degree=degree).fit(xs, ys).predict(x_values)
        lo, hi = min(xs), max(xs)
        for xv, yv in zip(x_values, pred):
            out.append({**dict(zip(groupby, key)), x: float(xv), y: float(yv), "extrapolated": bool(xv < lo or xv > hi)})
    return out
