"""The `--attn_softmax` plugin registry, MI355X side.

Mirrors the reference's SOFTMAX_MAPPING (OutEffHop/transformers_language/models/softmax.py:22-64): the same
40 string keys, each mapping to a callable `f(x, dim=-1, **kw) -> Tensor`.  Here every callable is a
`SoftmaxFn` object that (a) runs the HIP row kernel (`oeh_softmax_rows`) when called on a GPU tensor and
(b) carries a `SoftmaxSpec` so the attention modules can fuse it into the attention kernel instead of calling it.

Error behaviour kept from the reference: the softmax_1 family takes no `dtype=` keyword
(vutils/softmax_1.py:24 -> TypeError, the failure OPT hits under fp16); the vanilla family
forwards `dtype=` like torch.nn.functional.softmax (the input is cast first).
"entmax" (entmax-1.5) is a torch-ops callable outside the HIP hot path (sparse_activations.py).
"""
from __future__ import annotations

import re
from typing import Dict, Optional

import torch

from . import ops
from .ops import SoftmaxSpec


def softmax_autograd(data: torch.Tensor, spec: SoftmaxSpec, dim: int = -1) -> torch.Tensor:
    """The registry entry as differentiable torch ops, in the reference's op order (vutils/softmax_1.py:11-21, models/softmax.py:10-19):
    what a `SoftmaxFn` runs while autograd is recording its input - training under the reference's swap-in (run_clm.py:214-233,
    run_mlm.py:200-219) - since the HIP row kernel is forward-only.  Still on the tensor's GPU: rocm ATen kernels, no CPU path."""
    if spec.base == 1:
        m = data.max(dim=dim, keepdim=True).values
        e = torch.exp(torch.subtract(data, m))
        p = torch.divide(e, torch.add(e.sum(dim=dim, keepdim=True), torch.exp(torch.multiply(m, -1))))
    else:
        p = torch.nn.functional.softmax(data, dim=dim)
    if spec.clip:
        p = torch.clip(p * (spec.eta - spec.gamma) + spec.gamma, 0, 1)
    return p


class SoftmaxFn:
    """Callable registry entry; `spec` is what the fused kernel consumes."""

    def __init__(self, name: str, spec: SoftmaxSpec):
        self.name = name
        self.spec = spec
        self.__name__ = ("clipped_softmax1" if spec.base == 1 else "clipped_softmax") if spec.clip else ("softmax_1" if spec.base == 1 else "softmax")

    def __call__(self, data: torch.Tensor, dim: int = -1, **kw) -> torch.Tensor:
        if self.spec.base == 1 and kw:
            raise TypeError(f"softmax_1() got an unexpected keyword argument '{next(iter(kw))}'")
        dtype = kw.pop("dtype", None)
        if kw:
            raise TypeError(f"softmax() got an unexpected keyword argument '{next(iter(kw))}'")
        if dtype is not None:
            data = data.to(dtype)
        if ops.grad_recording(data):  # autograd is recording: the differentiable torch-op form (the HIP kernel is forward-only)
            ops._need_gpu(data, allow_grad=True)
            return softmax_autograd(data, self.spec, dim)
        return ops.softmax_rows(data, self.spec, dim=dim)

    def __repr__(self):
        return f"SoftmaxFn({self.name!r}, base={self.spec.base}, clip={self.spec.clip}, gamma={self.spec.gamma}, eta={self.spec.eta})"


def _num(tok: str) -> float:
    return float(tok.replace("-.", "-0.") if tok.startswith("-.") else tok)


def _entmax15(data, dim=-1, **kw):
    """`entmax15` of the reference's registry (models/softmax.py:25): torch ops on the tensor's device, outside the HIP hot
    path (sparse_activations.py); the attention modules run it through their observable path (`spec_of` gives None)."""
    from .sparse_activations import entmax15

    return entmax15(data, dim=dim)


def _build() -> Dict[str, object]:
    etas = ("1.0003", "1.001", "1.002", "1.003", "1.004", "1.01", "1.02", "1.03", "1.1")
    gammas = ("-.00001", "-.00003", "-.0001", "-.0003", "-.0005", "-.001", "-.002", "-.0025", "-.003", "-.004", "-.005", "-.01",
              "-.015", "-.02", "-.025", "-.03", "-.04")
    sym = ("-.001:1.001", "-.002:1.002", "-.003:1.003", "-.005:1.005", "-.01:1.01", "-.03:1.03", "-.1:1.1")
    keys = ["vanilla", "softmax1", "entmax"]
    keys += [f"clipped(0:{e})" for e in etas] + ["clipped(-.1:1)"] + [f"clipped({g}:1)" for g in gammas]
    keys += [f"clipped({s})" for s in sym]
    keys += ["clippedsoftmax1(-.025:1)", "clippedsoftmax1(-.00001:1)", "clippedsoftmax1(-.0001:1)"]
    table: Dict[str, object] = {}
    for k in keys:
        if k == "vanilla":
            table[k] = SoftmaxFn(k, SoftmaxSpec(0, False, 0.0, 1.0))
        elif k == "softmax1":
            table[k] = SoftmaxFn(k, SoftmaxSpec(1, False, 0.0, 1.0))
        elif k == "entmax":
            table[k] = _entmax15
        else:
            fam, g, e = re.fullmatch(r"(clipped|clippedsoftmax1)\(([^:]+):([^)]+)\)", k).groups()
            gamma, eta = _num(g), _num(e)
            # the reference's two literal quirks (softmax.py:57 and :61)
            if k == "clipped(-.005:1.005)":
                gamma = -0.003
            if k == "clippedsoftmax1(-.025:1)":
                eta = 1.1
            table[k] = SoftmaxFn(k, SoftmaxSpec(1 if fam == "clippedsoftmax1" else 0, True, gamma, eta))
    return table


SOFTMAX_MAPPING: Dict[str, object] = _build()

softmax_1 = SOFTMAX_MAPPING["softmax1"]


class Softmax_1(torch.nn.Module):
    """`nn.Module` form of softmax_1 (vutils/softmax_1.py:30-50; STanHop's cross_models/softmax_1.py): runs the HIP row kernel."""

    __constants__ = ["dim"]

    def __init__(self, dim: int = -1):
        super().__init__()
        self.dim = dim

    def __setstate__(self, state):
        self.__dict__.update(state)
        if not hasattr(self, "dim"):
            self.dim = None

    def forward(self, input):
        return softmax_1(input, self.dim)

    def extra_repr(self):
        return f"dim={self.dim}"


def clipped_softmax(gamma: float, eta: float) -> SoftmaxFn:
    """`partial(clipped_softmax, gamma=, eta=)` of the reference (softmax.py:10-13)."""
    return SoftmaxFn(f"clipped({gamma}:{eta})", SoftmaxSpec(0, True, float(gamma), float(eta)))


def clipped_softmax1(gamma: float, eta: float) -> SoftmaxFn:
    """`partial(clipped_softmax1, gamma=, eta=)` of the reference (softmax.py:16-19)."""
    return SoftmaxFn(f"clippedsoftmax1({gamma}:{eta})", SoftmaxSpec(1, True, float(gamma), float(eta)))


def spec_of(fn) -> Optional[SoftmaxSpec]:
    """SoftmaxSpec of a softmax callable if it is one the kernel can fuse, else None."""
    if isinstance(fn, SoftmaxFn):
        return fn.spec
    if fn is torch.nn.functional.softmax or fn is torch.softmax:
        return SoftmaxSpec(0, False, 0.0, 1.0)
    return None
