"""QMF — quantization-aware matrix factorization, the reference's class (lrf/factorization/qmf.py:167-231)
with `decompose` running on the MI355X through liblrf_hip.so."""
import math
from typing import Optional

import torch
from torch import Tensor

from . import _lib


class QMF:
    """X ~ w0 + w1 * (U @ V.T) with U, V integer matrices, inside `bounds` when given.

    Same constructor and methods as the reference's `lrf.factorization.QMF` (lrf/factorization/qmf.py:167-231), every
    option of it: `bounds` or none, `factor` subsets (the default (0, 1, 2) also refits the affine pair w every iteration),
    `l2` / `l1_ratio`.  The configuration `qmf_encode` uses — factor=(0, 1), integer bounds within int8, no penalties — runs on
    the tuned int8 kernels (lrf_qmf_decompose_f32); everything else on the general entry point (lrf_qmf_decompose_ex_f32,
    float factors), also `eps` and `num_levels` (round 3); `verbose=True` prints the loss before every iteration in the
    reference's format (round 5).  Not covered: a user `project`.  Extra keyword (not in the
    reference): `init_sign` — int8 [R] or [B,R], the sign to impose on each initial component (include/lrf_hip.h).
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
        self.factor = (factor,) if isinstance(factor, int) else tuple(factor)
        self.l2 = kwargs.pop("l2", 0)
        self.l1_ratio = kwargs.pop("l1_ratio", 0)
        self.eps = kwargs.pop("eps", 1e-16)
        if kwargs.pop("project", None) is not None:
            raise NotImplementedError("a user `project` is not on the HIP path (QMF builds its own, qmf.py:188)")
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")  # CoordinateDescent.__init__ would raise
        if not (self.eps >= 0):
            raise ValueError("eps must be >= 0")
        if not set(self.factor) <= {0, 1, 2}:
            raise ValueError("factor must be a subset of (0, 1, 2)")
        self._bounded = self.bounds != (None, None)
        l2 = self.l2 if isinstance(self.l2, (tuple, list)) else (self.l2, self.l2)
        no_penalty = l2[0] == 0 and l2[1] == 0
        # what qmf_encode uses (lrf/compression/qmf.py:256): the int8 kernels
        self._int8_path = (self._bounded and set(self.factor) == {0, 1} and no_penalty and self.eps == 1e-16 and not num_levels and
                           math.ceil(self.bounds[0]) >= -128 and math.floor(self.bounds[1]) <= 127)
        if self._bounded:
            self._lo, self._hi = math.ceil(self.bounds[0]), math.floor(self.bounds[1])  # qmf.py:194

    def _ctx(self, x):
        return _lib.context(x.device.index if x.is_cuda else None)

    def decompose(self, x: Tensor, *args, **kwargs):
        """x: [B, M, N].  Returns (u, v, w) like the reference: fp32, integer-valued u and v when num_iters >= 1,
        w [B, 2, 1] = [[w0], [w1]] ([[0], [1]] unless 2 is in `factor`)."""
        dev_in = x.device
        ctx = self._ctx(x)
        xd = x.float().contiguous()
        if not xd.is_cuda:
            xd = xd.cuda(ctx.device)
        sign = self.init_sign
        if sign is not None:
            sign = torch.as_tensor(sign, dtype=torch.int8).reshape(-1, self.rank).expand(xd.shape[0], self.rank)
            sign = sign.contiguous().cuda(ctx.device)
        if self.verbose and self.num_iters > 0:
            return self._decompose_verbose(ctx, xd, sign, dev_in)
        if self.num_levels:
            # SVDInit(num_levels=...) (qmf.py:56-68): both factors scaled to num_levels quantisation steps, w1 = their product.
            # The scaling itself is three elementwise torch operations on the initial factors (amax / amin / divide): the same
            # fp32 arithmetic as the reference's, bit for bit from the same u0, v0
            u0, v0 = ctx.svd_init(xd, self.rank, sign)
            su = (u0.amax(dim=(-2, -1), keepdim=True) - u0.amin(dim=(-2, -1), keepdim=True)) / self.num_levels
            sv = (v0.amax(dim=(-2, -1), keepdim=True) - v0.amin(dim=(-2, -1), keepdim=True)) / self.num_levels
            u, v = u0 / su, v0 / sv
            w = torch.cat([torch.zeros_like(su), (su * sv) * torch.ones_like(su)], dim=-2)
            if self.num_iters > 0:
                u, v, w2 = ctx.decompose_ex(xd, self.rank, self.num_iters, self.bounds, self.l2, self.l1_ratio, self.factor, None,
                                            init=(u, v), eps=self.eps, w_init=w.reshape(-1, 2))
                w = w2.reshape(-1, 2, 1)
        elif self.num_iters == 0:
            u, v = ctx.svd_init(xd, self.rank, sign)
            w = torch.cat([torch.zeros_like(xd[..., 0:1, 0:1]), torch.ones_like(xd[..., 0:1, 0:1])], dim=-2)
        elif self._int8_path:
            u8, v8 = ctx.decompose(xd, self.rank, self.num_iters, self._lo, self._hi, sign)
            u, v = u8.float(), v8.float()
            w = torch.cat([torch.zeros_like(xd[..., 0:1, 0:1]), torch.ones_like(xd[..., 0:1, 0:1])], dim=-2)
        else:
            u, v, w2 = ctx.decompose_ex(xd, self.rank, self.num_iters, self.bounds, self.l2, self.l1_ratio, self.factor, sign, eps=self.eps)
            w = w2.reshape(-1, 2, 1)
        return u.to(dev_in), v.to(dev_in), w.to(dev_in)

    def _decompose_verbose(self, ctx, xd, sign, dev_in):
        """verbose=True (lrf/factorization/qmf.py:206-212): before every iteration the loss of the current factors is printed
        in the reference's format.  The iterations run ONE per library call from the previous call's factors — the same
        arithmetic as the fused loop, bit for bit (from the second iteration on the factors are integers and every kernel
        family reproduces the reference's ordered sums) — and the loss is the library's (lrf_qmf_loss_f32), not a torch product."""
        B = xd.shape[0]
        u, v = ctx.svd_init(xd, self.rank, sign)
        w = torch.tensor([[0.0, 1.0]], device=xd.device).repeat(B, 1)
        if self.num_levels:
            su = (u.amax(dim=(-2, -1), keepdim=True) - u.amin(dim=(-2, -1), keepdim=True)) / self.num_levels
            sv = (v.amax(dim=(-2, -1), keepdim=True) - v.amin(dim=(-2, -1), keepdim=True)) / self.num_levels
            u, v = u / su, v / sv
            w = torch.cat([torch.zeros_like(su), su * sv], dim=-2).reshape(B, 2)
        for it in range(1, self.num_iters + 1):
            loss = ctx.loss(xd, u, v, w).cpu()
            print(f"iter {it}: loss = {loss}")
            if self._int8_path:
                u8, v8 = ctx.bcd(xd, u, v, 1, self._lo, self._hi)
                u, v = u8.float(), v8.float()
            else:
                u, v, w = ctx.decompose_ex(xd, self.rank, 1, self.bounds, self.l2, self.l1_ratio, self.factor, None, init=(u, v),
                                           eps=self.eps, w_init=w)
        return u.to(dev_in), v.to(dev_in), w.reshape(-1, 2, 1).to(dev_in)

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
