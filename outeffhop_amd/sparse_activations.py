"""Sparse normalising activations the reference's registries also offer - `SOFTMAX_MAPPING["entmax"]` (entmax-1.5,
OutEffHop/transformers_language/models/softmax.py:25) and STanHop's Association modes 'entmax' / 'sparsemax'
(STanHop_time_seeries/cross_models/hopfield.py:24-33) - so that every registry key and every constructor default of the
reference constructs and runs here.  They are OUTSIDE the HIP hot path (sort / bisection
based, a different algorithm from the softmax family the kernels fuse): plain torch tensor operations on whatever device
the input lives on, forward only, written from the published algorithms -

    sparsemax   Martins & Astudillo 2016, Alg. 1: tau from the sorted prefix sums, p = [z - tau]_+
    entmax-1.5  Peters, Niculae & Martins 2019, Alg. 2: tau from the sorted prefix mean / variance, p = [z/2 - tau]_+^2
    alpha-entmax by bisection on tau (same paper, Alg. 1): p = [(alpha-1) z - tau]_+^(1/(alpha-1))

- and used through the modules' observable path (`attention.unfused_core`), never through `oeh_attn_fwd`.
"""
from __future__ import annotations

import torch
from torch import nn


def _rows_last(x: torch.Tensor, dim: int):
    return (x, False) if dim in (-1, x.dim() - 1) else (x.transpose(dim, -1), True)


def sparsemax(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    z, swapped = _rows_last(x, dim)
    z = z - z.max(dim=-1, keepdim=True).values
    srt = torch.sort(z, dim=-1, descending=True).values
    k = torch.arange(1, z.shape[-1] + 1, device=z.device, dtype=z.dtype)
    csum = srt.cumsum(dim=-1) - 1.0
    support = (k * srt > csum).sum(dim=-1, keepdim=True)          # the largest k with 1 + k z_(k) > sum_{j<=k} z_(j)
    tau = csum.gather(-1, support - 1) / support.to(z.dtype)
    p = torch.clamp(z - tau, min=0.0)
    return p.transpose(dim, -1) if swapped else p


def entmax15(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    z, swapped = _rows_last(x, dim)
    z = (z - z.max(dim=-1, keepdim=True).values) / 2.0
    srt = torch.sort(z, dim=-1, descending=True).values
    k = torch.arange(1, z.shape[-1] + 1, device=z.device, dtype=z.dtype)
    mean = srt.cumsum(dim=-1) / k
    mean_sq = (srt * srt).cumsum(dim=-1) / k
    delta = (1.0 - k * (mean_sq - mean * mean)) / k
    tau_k = mean - torch.sqrt(torch.clamp(delta, min=0.0))
    support = (tau_k <= srt).sum(dim=-1, keepdim=True)
    tau = tau_k.gather(-1, support - 1)
    p = torch.clamp(z - tau, min=0.0) ** 2
    return p.transpose(dim, -1) if swapped else p


def entmax_bisect(x: torch.Tensor, alpha, dim: int = -1, n_iter: int = 50, ensure_sum_one: bool = True) -> torch.Tensor:
    """alpha-entmax for alpha > 1 (a float, or a tensor broadcastable to x with size 1 along `dim`).  Differentiable in x and,
    when `alpha` is a tensor that requires grad, in alpha (`_AlphaEntmaxFn`: the bisection itself carries no useful gradient)."""
    if torch.is_grad_enabled() and (x.requires_grad or (torch.is_tensor(alpha) and alpha.requires_grad)):
        a = alpha if torch.is_tensor(alpha) else torch.tensor(float(alpha), dtype=x.dtype, device=x.device)
        return _AlphaEntmaxFn.apply(x, a, dim, n_iter, ensure_sum_one)
    return _entmax_bisect_fwd(x, alpha, dim, n_iter, ensure_sum_one)


class _AlphaEntmaxFn(torch.autograd.Function):
    """Gradients of p = alpha-entmax(z) from the closed forms of Peters, Niculae & Martins 2019 (Prop. 2 and its proof), not from
    differentiating the bisection (which only sees zmax and the last clamp / pow - ADVICE r2: input gradients off by 0.4 ... 5 at
    gradient scale ~2, and no gradient at all for alpha).  With s_i = p_i^(2 - alpha) on the support (0 elsewhere) and
    p~ = s / sum(s):
        dp/dz       = diag(s) - s s^T / sum(s)
        dp/dalpha   = (p - p~) / (alpha - 1)^2  -  (p log p - p~ sum_j p_j log p_j) / (alpha - 1)
    The reference trains STanHop's EntmaxAlpha through the same expressions (cross_models/entmax.py:104-135)."""

    @staticmethod
    def forward(ctx, x, alpha, dim, n_iter, ensure_sum_one):
        with torch.no_grad():
            p = _entmax_bisect_fwd(x, alpha, dim, n_iter, ensure_sum_one)
        ctx.dim = dim
        ctx.alpha_shape = alpha.shape
        ctx.save_for_backward(p, alpha.to(p.dtype))
        return p

    @staticmethod
    def backward(ctx, g):
        p, alpha = ctx.saved_tensors
        dim = ctx.dim
        if alpha.dim() != p.dim():  # a scalar (or lower-rank) alpha: one value for every row
            alpha_b = alpha.reshape(alpha.shape + (1,) * (p.dim() - alpha.dim())) if alpha.dim() else alpha
        else:
            alpha_b = alpha
        on = p > 0
        s = torch.where(on, p ** (2.0 - alpha_b), torch.zeros_like(p))
        ssum = s.sum(dim=dim, keepdim=True)
        gx = g * s
        gx = gx - s * (gx.sum(dim=dim, keepdim=True) / ssum)
        ga = None
        if ctx.needs_input_grad[1]:
            plogp = torch.where(on, p * torch.log(torch.where(on, p, torch.ones_like(p))), torch.zeros_like(p))
            skew = s / ssum
            am1 = alpha_b - 1.0
            da = g * (p - skew) / (am1 * am1) - g * (plogp - skew * plogp.sum(dim=dim, keepdim=True)) / am1
            da = da.sum(dim=dim, keepdim=True)
            if alpha.dim() == p.dim():  # reduce over the broadcast axes of alpha
                for ax, n in enumerate(alpha.shape):
                    if n == 1 and da.shape[ax] != 1:
                        da = da.sum(dim=ax, keepdim=True)
                ga = da
            else:
                ga = da.sum().reshape(ctx.alpha_shape) if alpha.numel() == 1 else da.reshape(ctx.alpha_shape)
        return gx, ga, None, None, None


def _entmax_bisect_fwd(x: torch.Tensor, alpha, dim: int = -1, n_iter: int = 50, ensure_sum_one: bool = True) -> torch.Tensor:
    z, swapped = _rows_last(x, dim)
    if not torch.is_tensor(alpha):
        alpha = torch.tensor(float(alpha), dtype=z.dtype, device=z.device)
    alpha = alpha.to(z.dtype)
    if swapped and alpha.dim() == x.dim():
        alpha = alpha.transpose(dim, -1)
    am1 = alpha - 1.0
    z = z * am1
    d = z.shape[-1]
    zmax = z.max(dim=-1, keepdim=True).values
    lo = zmax - 1.0                                  # f(lo) >= 0: the largest entry alone gives p = 1
    hi = zmax - (1.0 / d) ** am1                     # f(hi) <= 0: every entry is at most 1/d there
    p_of = lambda tau: torch.clamp(z - tau, min=0.0) ** (1.0 / am1)  # noqa: E731
    f_lo = p_of(lo).sum(dim=-1, keepdim=True) - 1.0
    width = hi - lo
    for _ in range(n_iter):
        width = width / 2.0
        mid = lo + width
        f_mid = p_of(mid).sum(dim=-1, keepdim=True) - 1.0
        lo = torch.where(f_mid * f_lo >= 0, mid, lo)
    p = p_of(lo + width)
    if ensure_sum_one:
        p = p / p.sum(dim=-1, keepdim=True)
    return p.transpose(dim, -1) if swapped else p


class Sparsemax(nn.Module):
    def __init__(self, dim: int = -1):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return sparsemax(x, self.dim)


class Entmax15(nn.Module):
    def __init__(self, dim: int = -1):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return entmax15(x, self.dim)


class AlphaChooser(nn.Module):
    """alpha = clamp(1 + 2 sigmoid(pre_alpha), 1.0001, 3) (cross_models/entmax.py:10-20)."""

    def __init__(self, head_count):
        super().__init__()
        self.pre_alpha = nn.Parameter(torch.randn(1) * 2.0)
        self.head_count = head_count

    def forward(self):
        return torch.clamp(1 + 2 * torch.sigmoid(self.pre_alpha), min=1.0001, max=3)


class EntmaxAlpha(nn.Module):
    """STanHop's default Association activation: alpha-entmax over the keys with one learnable alpha = 1 + 2 sigmoid(a)
    shared by every head and row (cross_models/entmax.py:23-45; parameter names `alpha`, `alpha_chooser` kept)."""

    def __init__(self, head_count: int = 4, dim: int = -1):
        super().__init__()
        self.dim = dim
        self.alpha_chooser = nn.Parameter(AlphaChooser(1)())
        self.alpha = nn.Parameter(torch.randn(1))

    def forward(self, att_scores):
        alpha = 1 + 2 * torch.sigmoid(self.alpha)   # learnable: the gradient reaches it through _AlphaEntmaxFn
        return entmax_bisect(att_scores, alpha.view(*([1] * att_scores.dim())), dim=self.dim)
