"""QMF — quantization-aware matrix factorization, the reference's class (lrf/factorization/qmf.py:167-231)
with `decompose` running on the MI355X through liblrf_hip.so."""
import math
from typing import Optional

import torch
from torch import Tensor

from . import _lib


class QMF:
    """X ~ U @ V.T with U, V integer matrices inside `bounds`.

    Same constructor and methods as the reference's `lrf.factorization.QMF`.  The HIP path covers what
    `qmf_encode` uses (lrf/compression/qmf.py:256): factor=(0, 1), l2 = 0, eps = 1e-16, integer bounds
    within int8; anything else raises NotImplementedError.  Extra keyword (not in the reference):
    `init_sign` — int8 [R] or [B,R], the sign to impose on each initial component
    (see include/lrf_hip.h lrf_qmf_decompose_f32).
    """

    def __init__(self, rank: Optional[int], num_iters: int = 10, bounds=(None, None), num_levels=None,
                 verbose: bool = False, **kwargs) -> None:
        self.rank = rank
        self.num_iters = num_iters
        self.bounds = tuple(bounds)
        self.num_levels = num_levels
        self.verbose = verbose
        self.init_sign = kwargs.pop("init_sign", None)
        factor = kwargs.pop("factor", (0, 1, 2))
        l2 = kwargs.pop("l2", 0)
        l1_ratio = kwargs.pop("l1_ratio", 0)
        eps = kwargs.pop("eps", 1e-16)
        kwargs.pop("project", None)
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")  # CoordinateDescent.__init__ would raise
        if tuple(factor) != (0, 1) and factor != (0, 1):
            raise NotImplementedError("only factor=(0, 1) (w fixed at [0; 1]) runs on the HIP path")
        if l2 not in (0, (0, 0)) or eps != 1e-16 or num_levels:
            raise NotImplementedError("l2 / l1_ratio / eps / num_levels other than the defaults are not on the HIP path")
        if self.bounds == (None, None):
            raise NotImplementedError("unbounded factors are not on the HIP path (int8 factors only)")
        self._lo, self._hi = math.ceil(self.bounds[0]), math.floor(self.bounds[1])  # qmf.py:194

    def _ctx(self, x):
        return _lib.context(x.device.index if x.is_cuda else None)

    def decompose(self, x: Tensor, *args, **kwargs):
        """x: [B, M, N].  Returns (u, v, w) like the reference (fp32, integer valued; w = [[0],[1]] per batch)."""
        dev_in = x.device
        ctx = self._ctx(x)
        xd = x.float().contiguous()
        if not xd.is_cuda:
            xd = xd.cuda(ctx.device)
        sign = self.init_sign
        if sign is not None:
            sign = torch.as_tensor(sign, dtype=torch.int8).reshape(-1, self.rank).expand(xd.shape[0], self.rank)
            sign = sign.contiguous().cuda(ctx.device)
        if self.num_iters == 0:
            u, v = ctx.svd_init(xd, self.rank, sign)
        else:
            if self.verbose:
                print("QMF(verbose=True): per-iteration loss is not reported by the fused HIP path")
            u8, v8 = ctx.decompose(xd, self.rank, self.num_iters, self._lo, self._hi, sign)
            u, v = u8.float(), v8.float()
        w = torch.cat([torch.zeros_like(xd[..., 0:1, 0:1]), torch.ones_like(xd[..., 0:1, 0:1])], dim=-2)
        return u.to(dev_in), v.to(dev_in), w.to(dev_in)

    @staticmethod
    def reconstruct(u: Tensor, v: Tensor, w: Optional[Tensor] = None) -> Tensor:
        out = u @ v.mT  # lrf/factorization/qmf.py:216-223
        if w is None:
            return out
        w0, w1 = w.split(split_size=1, dim=-2)
        return w0 + w1 * out

    @staticmethod
    def loss(x: Tensor, u: Tensor, v: Tensor, w: Optional[Tensor] = None) -> Tensor:
        y = QMF.reconstruct(u, v, w)  # relative error, lrf/factorization/utils.py:12-15
        return torch.norm(x - y, p=2, dim=(-2, -1)) / (torch.norm(x, p=2, dim=(-2, -1)) + 1e-16)

    def forward(self, x: Tensor) -> Tensor:
        u, v, w = self.decompose(x)
        return self.reconstruct(u, v, w)

    __call__ = forward
