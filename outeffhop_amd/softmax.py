"""The `--attn_softmax` plugin registry, MI355X side.

Mirrors the reference's SOFTMAX_MAPPING (OutEffHop/transformers_language/models/softmax.py:22-64): the same
40 string keys, each mapping to a callable `f(x, dim=-1, **kw) -> Tensor`.  "vanilla" / "softmax1" are `SoftmaxFn` objects, the
clipped keys are `functools.partial(clipped_softmax[1], gamma=, eta=)` as in the reference (a subclass that also has `.spec`); every
entry (a) runs the HIP row kernel (`oeh_softmax_rows`) when called on a GPU tensor and (b) yields a `SoftmaxSpec` through `spec_of`,
so the attention modules fuse it into the attention kernel instead of calling it.

Error behaviour kept from the reference: the softmax_1 family takes no `dtype=` keyword
(vutils/softmax_1.py:24 -> TypeError, the failure OPT hits under fp16); the vanilla family
forwards `dtype=` like torch.nn.functional.softmax (the input is cast first).
"entmax" (entmax-1.5) is a torch-ops callable outside the HIP hot path (sparse_activations.py).
"""
from __future__ import annotations

import functools
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
            table[k] = softmax_1
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
            table[k] = (make_clipped_softmax1 if fam == "clippedsoftmax1" else make_clipped_softmax)(gamma, eta, name=k)
    return table


softmax_1 = SoftmaxFn("softmax1", SoftmaxSpec(1, False, 0.0, 1.0))


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


def clipped_softmax(data, dim=1, eta=1.1, gamma=-0.1, **kw):
    """The reference's callable, argument for argument (models/softmax.py:10-13; cross_models/clip_softmax.py:5-8):
    `clip(softmax(data, dim, **kw) * (eta - gamma) + gamma, 0, 1)` - here one launch of the HIP row kernel.  `**kw` goes where the
    reference sends it (`dtype=` casts the input first, as torch.nn.functional.softmax does).  The reference's idiom
    `partial(clipped_softmax, gamma=g, eta=1.0)` (opt_attention.py:75-77, bert_attention.py:92) is recognised by `spec_of`, so a module
    given such a partial as `softmax_fn=` still runs the fused attention kernel."""
    return SoftmaxFn("clipped", SoftmaxSpec(0, True, float(gamma), float(eta)))(data, dim, **kw)


def clipped_softmax1(data, dim=1, eta=1.1, gamma=-0.1, **kw):
    """models/softmax.py:16-19 (STanHop: `clipped_softmax_1`, clip_softmax.py:10-13): the same around softmax_1; `**kw` is forwarded to
    softmax_1, which takes none - `dtype=` raises TypeError exactly as the reference does."""
    return SoftmaxFn("clippedsoftmax1", SoftmaxSpec(1, True, float(gamma), float(eta)))(data, dim, **kw)


clipped_softmax_1 = clipped_softmax1  # STanHop's spelling (cross_models/clip_softmax.py:10)


class FusablePartial(functools.partial):
    """`functools.partial(clipped_softmax[1], gamma=, eta=)` - what the reference's registry holds for its clipped keys (softmax.py:26-63) and
    what its modules build for `alpha` - that also answers `.spec` / `.name` like the other registry entries."""

    name = ""

    @property
    def spec(self) -> SoftmaxSpec:
        return spec_of(self)

    @property
    def __name__(self):
        return self.func.__name__

    def __repr__(self):
        sp = self.spec
        return f"FusablePartial({self.name or self.func.__name__!r}, base={sp.base}, clip={sp.clip}, gamma={sp.gamma}, eta={sp.eta})"


def make_clipped_softmax(gamma: float, eta: float, name: str = "") -> FusablePartial:
    """Factory form (rounds 1-5 exported it as `clipped_softmax`): the reference's `partial(clipped_softmax, gamma=, eta=)`."""
    fp = FusablePartial(clipped_softmax, gamma=float(gamma), eta=float(eta))
    fp.name = name or f"clipped({gamma}:{eta})"
    return fp


def make_clipped_softmax1(gamma: float, eta: float, name: str = "") -> FusablePartial:
    fp = FusablePartial(clipped_softmax1, gamma=float(gamma), eta=float(eta))
    fp.name = name or f"clippedsoftmax1({gamma}:{eta})"
    return fp


class ClipSoftmax(torch.nn.Module):
    """STanHop's module form (cross_models/clip_softmax.py:15-36): `clipped_softmax(input, dim, eta, gamma)` positionally."""

    __constants__ = ["dim"]

    def __init__(self, dim: int = -1, eta: float = 1.1, gamma: float = -0.1):
        super().__init__()
        self.dim, self.eta, self.gamma = dim, eta, gamma

    def __setstate__(self, state):
        self.__dict__.update(state)
        if not hasattr(self, "dim"):
            self.dim = None

    def forward(self, input):
        return clipped_softmax(input, self.dim, self.eta, self.gamma)

    def extra_repr(self):
        return f"dim={self.dim}"


class ClipSoftmax_1(ClipSoftmax):
    """cross_models/clip_softmax.py:38-60.  The reference's constructor calls `super(ClipSoftmax, self).__init__()` (:46) and raises
    TypeError, so the class cannot be instantiated there; this one works and computes what its `forward` (:57-58) says."""

    def forward(self, input):
        return clipped_softmax1(input, self.dim, self.eta, self.gamma)


def spec_of(fn) -> Optional[SoftmaxSpec]:
    """SoftmaxSpec of a softmax callable if it is one the kernel can fuse, else None.  Recognised: the registry entries, torch's softmax,
    `clipped_softmax` / `clipped_softmax1` themselves (their default eta = 1.1, gamma = -0.1) and `functools.partial`s of any of these
    that bind only `eta` / `gamma` / `dim` by keyword (the modules pass `dim=-1` at the call, which overrides a bound `dim`)."""
    if isinstance(fn, SoftmaxFn):
        return fn.spec
    if fn is torch.nn.functional.softmax or fn is torch.softmax:
        return SoftmaxSpec(0, False, 0.0, 1.0)
    if fn is clipped_softmax or fn is clipped_softmax1:
        return SoftmaxSpec(1 if fn is clipped_softmax1 else 0, True, -0.1, 1.1)
    if isinstance(fn, functools.partial) and not fn.args:
        kw = dict(fn.keywords or {})
        kw.pop("dim", None)
        if fn.func is clipped_softmax or fn.func is clipped_softmax1:
            if set(kw) <= {"eta", "gamma"}:
                try:
                    return SoftmaxSpec(1 if fn.func is clipped_softmax1 else 0, True, float(kw.get("gamma", -0.1)), float(kw.get("eta", 1.1)))
                except (TypeError, ValueError):
                    return None
            return None
        if not kw:
            return spec_of(fn.func)
    return None


SOFTMAX_MAPPING: Dict[str, object] = _build()
