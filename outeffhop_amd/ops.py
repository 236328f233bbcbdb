"""Tensor-level entry points over the C ABI (include/oeh.h).  torch is plumbing here: device memory,
the current HIP stream, dtype/stride bookkeeping.  Every function launches HIP kernels from
liboeh_hip.so; tensors that are not on a GPU are an error (there is no CPU path in this package).
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import OEH_BF16, OEH_F16, OEH_F32, oeh_attn_desc, oeh_fq, oeh_fq_desc

_DT = {torch.float16: OEH_F16, torch.bfloat16: OEH_BF16, torch.float32: OEH_F32}


@dataclass(frozen=True)
class SoftmaxSpec:
    """One entry of the reference's --attn_softmax registry (models/softmax.py:22-64):
    base 0 = torch softmax, 1 = softmax_1; clip -> clip(p*(eta-gamma)+gamma, 0, 1)."""

    base: int = 1
    clip: bool = False
    gamma: float = 0.0
    eta: float = 1.0


@dataclass
class FakeQuantSpec:
    """Fixed-range asymmetric quantiser grid (uniform_quantizers.py:72-82): fp32 scale, integer zero point, qmax."""

    scale: float
    zero_point: float
    qmax: float = 255.0
    dump: Optional[torch.Tensor] = None  # uint8 tensor receiving the indices (tests)

    @staticmethod
    def from_delta(delta: float, zero_float: float, n_bits: int = 8, eps: float = 1e-8, dump=None) -> "FakeQuantSpec":
        import numpy as np

        qmax = float(2.0 ** n_bits - 1)
        scale = float(np.float32(max(float(delta), eps)))
        zp = float(min(max(float(np.rint(np.float64(zero_float))), 0.0), qmax))
        return FakeQuantSpec(scale, zp, qmax, dump)


@dataclass
class AttnFakeQuant:
    scores: Optional[FakeQuantSpec] = None
    probs: Optional[FakeQuantSpec] = None
    ctx: Optional[FakeQuantSpec] = None
    ctx_before_gate: bool = True  # OPT order; False = BERT order (quantise after the gate)
    ctx_emit_index: bool = False  # the output holds the context quantiser's integers idx - zp (include/oeh.h: ctx_emit_index)


def grad_recording(*ts) -> bool:
    """True when autograd is recording and one of the tensors is part of the graph: the callers then take the differentiable
    torch-op form of the same op chain (attention.unfused_core, softmax.softmax_autograd, attention.gate_autograd) - on the GPU, like
    everything else here - instead of the forward-only HIP kernels."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _need_gpu(*ts, allow_grad: bool = False):
    """Every tensor on ONE GPU, and (unless `allow_grad`: the torch-op path) none of them part of an autograd graph being
    recorded: the library is the only implementation (no CPU path) and it is forward-only - its outputs carry no grad_fn, so
    running it under autograd would drop the gradients of q / k / v / gate silently.  The modules route such calls to the torch-op
    path before they get here (`grad_recording`); a direct op call raises (inference: `with torch.no_grad():`)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.OehError("outeffhop_amd ops need GPU tensors: the HIP library is the only implementation")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.OehError(f"outeffhop_amd ops need all tensors on one GPU, got {dev} and {t.device}")
    if not allow_grad and grad_recording(*ts):
        raise _lib.OehError("outeffhop_amd is forward-only (the backward pass is out of scope, DESIGN.md section 7): an input requires "
                            "grad while autograd is recording - run inference under torch.no_grad() / torch.inference_mode()")
    return dev


class _on_device:
    """Make the tensors' GPU the current HIP device for the launch (kernels run on the stream of the CURRENT device)."""

    def __init__(self, dev):
        self.dev, self.prev = dev, None

    def __enter__(self):
        if self.dev is not None and self.dev.index is not None and self.dev.index != torch.cuda.current_device():
            self.prev = torch.cuda.current_device()
            torch.cuda.set_device(self.dev)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the handle without building a Stream object (9 us -> 0.5 us)


def _stream() -> C.c_void_p:
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _fill_fq(dst: oeh_fq, spec: Optional[FakeQuantSpec]):
    if spec is None:
        dst.enable = 0
        return
    dst.enable = 1
    dst.scale, dst.zero_point, dst.qmax = spec.scale, spec.zero_point, spec.qmax
    if spec.dump is not None:
        assert spec.dump.dtype == torch.uint8 and spec.dump.is_contiguous() and spec.dump.is_cuda
        dst.dump_idx = spec.dump.data_ptr()


def attn_fwd(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    *,
    softmax: SoftmaxSpec = SoftmaxSpec(),
    scale: float = 1.0,
    scale_div: float = 0.0,
    key_pad_mask: Optional[torch.Tensor] = None,
    full_mask: Optional[torch.Tensor] = None,
    key_pad_boolean: bool = False,
    causal: bool = False,
    clamp_min: bool = False,
    mask_min: Optional[float] = None,
    gate: Optional[torch.Tensor] = None,
    fq: Optional[AttnFakeQuant] = None,
    out: Optional[torch.Tensor] = None,
    gate_mlp: Optional["GatePredictor"] = None,
    out_dtype: Optional[torch.dtype] = None,
    _prepared: Optional[list] = None,
) -> torch.Tensor:
    """Fused attention core.  q,k,v are logical (B,H,S,D) views (any batch/head/seq strides, unit head-dim
    stride).  Returns the logical (B,H,Sq,D) result, stored (B,Sq,H,D)-contiguous unless `out` is given, so the
    reference's head merge (bert_attention.py:335-337) is a free `.permute(0,2,1,3).reshape(B,Sq,H*D)`.

    key_pad_mask: additive (B,Sk) [or anything reshapeable to it, e.g. HF's (B,1,1,Sk)]; key_pad_boolean: the caller's promise that
    its entries are 0 or <= -1e4 only (include/oeh.h - lets the fused INT8 chain stay on the quantiser grid with padded keys);
    full_mask: additive (B,1,Sq,Sk); gate: fp32, broadcastable to (B,H,Sq,1), already times the scaling factor.
    out_dtype=torch.float32 with fp16 / bf16 inputs: the output straight from the kernel's fp32 accumulators (include/oeh.h: o_dtype) -
    the kernel's arithmetic before the output rounding, which is how tests / smoke / bench check the "within 1e-3" contract on the
    kernel that ships; the one-pass and full-row kernels only (OehError -95 otherwise)."""
    # ---- the repeated call (round 5): same geometry and options as an earlier call -> that call's prebuilt descriptor, pointers patched.
    # The Python below this block is ~13 us per call (checks, a 40-field ctypes descriptor, mask / gate views) against ~3 us for the launch of a
    # prebuilt one, and the fp16 BERT-base layer is host-bound in eager mode.  Only the plain forms (no fused quantisers, no in-kernel gate predictor,
    # no (B,1,Sq,Sk) mask, no caller's `out`, inference).
    fkey = None
    if (FAST_CALLS and out is None and gate_mlp is None and fq is None and full_mask is None and _prepared is None and out_dtype is None
            and q.is_cuda and q.dim() == 4 and q.device.index == torch.cuda.current_device()
            and not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad or (gate is not None and gate.requires_grad)))):
        fkey = _fast_key(q, k, v, softmax, scale, scale_div, key_pad_mask, key_pad_boolean, causal, clamp_min, mask_min, gate)
        hit = _fast_tls.table.get(fkey) if fkey is not None else None
        if hit is not None:
            fn, d, oshape, odt = hit
            o = torch.empty(oshape, dtype=odt, device=q.device).permute(0, 2, 1, 3)
            if key_pad_mask is not None:
                d.key_pad_mask = key_pad_mask.data_ptr()
            if gate is not None:
                d.gate = gate.data_ptr()
            rc = fn(C.byref(d), C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(o.data_ptr()), None, _stream())
            if rc != 0:
                _lib.check(rc, "oeh_attn_fwd")
            return o
    dev = _need_gpu(q, k, v, key_pad_mask, full_mask, gate, out)
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        raise ValueError("q, k, v must be 4-D (B,H,S,D) views")
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    if k.shape != (B, H, Sk, D) or v.shape != (B, H, Sk, D):
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}")
    if not (q.dtype == k.dtype == v.dtype) or q.dtype not in _DT:
        raise ValueError(f"q/k/v dtypes must match and be fp16/bf16/fp32, got {q.dtype}, {k.dtype}, {v.dtype}")
    fix = lambda t: t if t.stride(3) == 1 else t.contiguous()  # noqa: E731
    q, k, v = fix(q), fix(k), fix(v)
    odt = q.dtype if out_dtype is None else out_dtype
    if odt != q.dtype and not (odt == torch.float32 and q.dtype in (torch.float16, torch.bfloat16)):
        raise ValueError(f"out_dtype must be the input dtype, or float32 for fp16 / bf16 inputs (got {odt} for {q.dtype})")
    Dp = _matrix_core_head_dim(D, Sq, Sk, plain=key_pad_mask is None and full_mask is None and not causal and fq is None)
    if (Dp != D and gate_mlp is None and _prepared is None and odt == q.dtype and (out is None or out.shape == (B, H, Sq, D))
            and not (fq is not None and fq.ctx is not None and fq.ctx.dump is not None)):
        # A head dim between the matrix-core kernels' (OPT-2.7b / ViT-H: 80; 96; 48; 16 with long rows ...): zero columns change
        # neither the scores nor the other columns of the product - three pad copies and the next kernel size up instead of the
        # any-shape kernel (one workgroup per query row, ~100x slower)
        pad = lambda t: torch.nn.functional.pad(t, (0, Dp - D))  # noqa: E731
        res = attn_fwd(pad(q), pad(k), pad(v), softmax=softmax, scale=scale, scale_div=scale_div, key_pad_mask=key_pad_mask, full_mask=full_mask,
                       key_pad_boolean=key_pad_boolean, causal=causal, clamp_min=clamp_min, mask_min=float(torch.finfo(q.dtype).min if mask_min is None else mask_min), gate=gate,
                       fq=fq)[..., :D]
        if out is None:
            return res
        out.copy_(res)
        return out
    if out is None:
        out = torch.empty((B, Sq, H, D), dtype=odt, device=q.device).permute(0, 2, 1, 3)
    elif out.shape != (B, H, Sq, D) or out.dtype != odt or out.stride(3) != 1:
        raise ValueError("out must be a (B,H,Sq,D) view with unit head-dim stride and the input dtype (or out_dtype)")

    d = oeh_attn_desc()
    d.B, d.H, d.Sq, d.Sk, d.D = B, H, Sq, Sk, D
    d.dtype = _DT[q.dtype]
    d.o_dtype = _DT[odt]  # (read for 16-bit inputs only when it says OEH_F32)
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", out)):
        getattr(d, name)[:] = [t.stride(0), t.stride(1), t.stride(2)]
    d.scale, d.scale_div = float(scale), float(scale_div)
    d.softmax_base, d.clip, d.gamma, d.eta = int(softmax.base), int(bool(softmax.clip)), float(softmax.gamma), float(softmax.eta)
    keep = []
    if key_pad_mask is not None:
        m = key_pad_mask.reshape(B, Sk)
        if m.dtype not in (torch.float16, torch.float32):
            m = m.float()
        m = m.contiguous()
        keep.append(m)
        d.key_pad_mask, d.key_pad_dtype, d.key_pad_stride = m.data_ptr(), _DT[m.dtype], m.stride(0)
        d.key_pad_boolean = int(bool(key_pad_boolean))
    if full_mask is not None:
        if full_mask.shape != (B, 1, Sq, Sk):
            raise ValueError(f"Attention mask should be of size {(B, 1, Sq, Sk)}, but is {tuple(full_mask.shape)}")
        m = full_mask if full_mask.dtype in (torch.float16, torch.float32) else full_mask.float()
        m = m if m.stride(3) == 1 else m.contiguous()
        keep.append(m)
        d.full_mask, d.full_mask_dtype = m.data_ptr(), _DT[m.dtype]
        d.full_mask_stride[:] = [m.stride(0), m.stride(2)]
    d.causal, d.clamp_min = int(bool(causal)), int(bool(clamp_min))
    d.mask_min = float(torch.finfo(q.dtype).min if mask_min is None else mask_min)
    if gate is not None:
        g = gate.to(torch.float32)
        while g.dim() < 4:
            g = g.unsqueeze(0)
        g = g.expand(B, H, Sq, 1)
        keep.append(g)
        d.gate = g.data_ptr()
        d.gate_stride[:] = [g.stride(0), g.stride(1), g.stride(2)]
    if gate_mlp is not None:
        if gate is not None:
            raise ValueError("pass either `gate` (values) or `gate_mlp` (predictor evaluated in the kernel), not both")
        gm = gate_mlp
        hid = gm.hidden
        _need_gpu(hid, gm.w1, gm.b1, gm.w2, gm.b2, gm.out)
        if hid.shape != (B, Sq, H * D) or hid.dtype != q.dtype or hid.stride(2) != 1:
            raise ValueError(f"gate_mlp.hidden must be (B,Sq,H*D) = {(B, Sq, H * D)} in the dtype of q with a contiguous last dim")
        f32c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()  # noqa: E731
        w1, b1, w2, b2 = f32c(gm.w1), f32c(gm.b1), f32c(gm.w2), f32c(gm.b2)
        keep.extend([hid, w1, b1, w2, b2, gm.out])
        d.gate_hidden = hid.data_ptr()
        d.gate_hidden_stride[:] = [hid.stride(0), hid.stride(1)]
        d.gate_w1, d.gate_b1 = w1.data_ptr(), b1.data_ptr()
        d.gate_units = 0 if w1.dim() == 2 else w1.shape[1]
        if d.gate_units > 0:
            d.gate_w2, d.gate_b2 = w2.data_ptr(), b2.data_ptr()
        d.gate_scaling = float(gm.scaling)
        if gm.out is not None:
            if gm.out.shape != (B, H, Sq) or gm.out.dtype != torch.float32 or not gm.out.is_contiguous():
                raise ValueError("gate_mlp.out must be a contiguous fp32 (B,H,Sq) tensor")
            d.gate_out = gm.out.data_ptr()
    fqd = None
    if fq is not None and (fq.scores or fq.probs or fq.ctx):
        fqd = oeh_fq_desc()
        _fill_fq(fqd.scores, fq.scores)
        _fill_fq(fqd.probs, fq.probs)
        _fill_fq(fqd.ctx, fq.ctx)
        fqd.ctx_quant_before_gate = int(bool(fq.ctx_before_gate))
        fqd.ctx_emit_index = int(bool(fq.ctx_emit_index))
    lib = _lib.load()
    if _prepared is not None:  # hand back the prebuilt C call instead of launching (bench / hipGraph loops)
        _warn_if_any_shape_kernel(lib, d, fqd, softmax)
        args = (C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), None if fqd is None else C.byref(fqd))
        _prepared.extend([lib.oeh_attn_fwd, args, (d, fqd, keep, q, k, v, out)])
        return out
    with _on_device(dev):
        rc = lib.oeh_attn_fwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), None if fqd is None else C.byref(fqd), _stream())
    _lib.check(rc, "oeh_attn_fwd")
    _warn_if_any_shape_kernel(lib, d, fqd, softmax)  # (after the launch: a call the library refuses raises above and has run nothing)
    if fkey is not None and fqd is None and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1 and out.shape == (B, H, Sq, D):
        # remember the descriptor for the next call of this geometry (its mask / gate pointers are patched per call; the views `keep` holds
        # were only needed for THIS call's pointers)
        if len(_fast_tls.table) >= 256:
            _fast_tls.table.clear()
        _fast_tls.table[fkey] = (lib.oeh_attn_fwd, d, (B, Sq, H, D), odt)
    return out


FAST_CALLS = True


class _FastCalls(threading.local):
    """geometry + options -> (C function, descriptor, output shape, output dtype): see attn_fwd.  Per THREAD: a cached descriptor's mask / gate
    pointers are patched right before the call, and ctypes releases the GIL inside it."""

    def __init__(self):
        self.table = {}


_fast_tls = _FastCalls()


def _fast_key(q, k, v, softmax, scale, scale_div, key_pad_mask, key_pad_boolean, causal, clamp_min, mask_min, gate):
    """What a prebuilt `oeh_attn_desc` depends on besides the four data pointers and the mask / gate pointers; None: not a plain call."""
    if k.dim() != 4 or v.dim() != 4 or k.device != q.device or v.device != q.device or not (q.dtype == k.dtype == v.dtype):
        return None
    if q.stride(3) != 1 or k.stride(3) != 1 or v.stride(3) != 1 or q.shape[3] not in (32, 64, 128):
        return None
    pk = None
    if key_pad_mask is not None:
        # (the vector as the slow path would hand it over unchanged: (B, Sk), fp16 / fp32, contiguous rows, on the same GPU)
        if (key_pad_mask.dim() != 2 or key_pad_mask.dtype not in (torch.float16, torch.float32) or key_pad_mask.stride(1) != 1 or key_pad_mask.stride(0) != key_pad_mask.shape[1]
                or key_pad_mask.device != q.device or key_pad_mask.shape != (q.shape[0], k.shape[2])):
            return None
        pk = key_pad_mask.dtype
    gk = None
    if gate is not None:
        if gate.dim() != 4 or gate.dtype != torch.float32 or gate.device != q.device:
            return None
        gk = (gate.shape, gate.stride())
    return (q.shape, q.stride(), k.shape, k.stride(), v.shape, v.stride(), q.dtype, q.device, softmax, float(scale), float(scale_div), pk, bool(key_pad_boolean), bool(causal),
            bool(clamp_min), mask_min, gk)


def _matrix_core_head_dim(D: int, Sq: int, Sk: int, plain: bool) -> int:
    """The head dim the problem is run at: D itself where a matrix-core kernel takes it (32, 64, 128; 16 for the small-shape
    kernel: <= 64 rows, no mask, no fake-quant - `plain`), the next one up (zero-padded) below 128, D itself above (any-shape
    kernel)."""
    if D in (32, 64, 128) or D > 128:
        return D
    if D <= 16:
        return 16 if (plain and Sq <= 64 and Sk <= 64) else 32
    return 32 if D < 32 else (64 if D < 64 else 128)


_warned_generic = set()


def _warn_if_any_shape_kernel(lib, d, fqd, softmax) -> None:
    """The any-shape kernel (one workgroup per query ROW, fp32 FMAs) is correct for everything and ~100x slower than the
    matrix-core kernels: say so once per kind of shape instead of letting a configuration fall off that cliff silently
    (VERDICT r1 weak #13).  Asks the library which kernel the descriptor selects (a host-side call, no launch)."""
    name = lib.oeh_attn_variant(C.byref(d), None if fqd is None else C.byref(fqd))
    if name is None or not name.startswith(b"generic"):
        return
    if d.D not in (16, 32, 64, 128):
        reason = f"head dim {d.D} (matrix-core kernels: 32, 64, 128; 16 for <= 64 keys)"
    elif d.D == 16 and (d.Sq > 64 or d.Sk > 64):
        reason = f"head dim 16 with more than 64 rows (Sq={d.Sq}, Sk={d.Sk})"
    elif d.D == 16:
        reason = ("head dim 16 outside the small-shape kernel's cases (a mask, fake-quant, the in-kernel gate predictor, or rows that are "
                  f"not 4-element aligned; Sq={d.Sq}, Sk={d.Sk})")
    elif d.Sk > 512:
        what = "fused fake-quant" if fqd is not None else ("clipped softmax outside the two-pass kernel's cases" if softmax.clip else "this mask / scale combination")
        reason = f"{what} with {d.Sk} > 512 keys (the full-row kernels hold a row of <= 512 scores)"
    else:
        reason = "unaligned q / k / v / out (16-byte rows) or a mask / scale combination outside the matrix-core kernels"
    if reason not in _warned_generic:
        import warnings

        _warned_generic.add(reason)
        warnings.warn(f"outeffhop_amd: {reason} runs the any-shape HIP kernel (one workgroup per query row, fp32 FMA): correct, "
                      "but far from the matrix-core kernels' speed", RuntimeWarning, stacklevel=3)


@dataclass(frozen=True)
class QuantGrid:
    """A per-tensor 8-bit grid of one of q / k / v in INT8 storage: value = scale * (index - zero_point)."""

    scale: float
    zero_point: float

    @staticmethod
    def of(spec: "FakeQuantSpec") -> "QuantGrid":
        if spec.qmax != 255.0:
            raise ValueError("INT8 storage needs 8-bit grids")
        return QuantGrid(float(spec.scale), float(spec.zero_point))


def centre_indices(idx: torch.Tensor) -> torch.Tensor:
    """uint8 quantiser indices -> the int8 storage of the INT8 attention path (index - 128): one XOR, no copy of another kind."""
    if idx.dtype != torch.uint8:
        raise ValueError("indices must be uint8")
    return (idx ^ 128).view(torch.int8)


def split_pairs(x: torch.Tensor) -> torch.Tensor:
    """fp32 (rows, K) -> fp16 (rows, 2K) = [hi | lo] operand pairs (`oeh_split_pairs`): x = hi + lo * 2^-11 to 2^-22 relative."""
    dev = _need_gpu(x)
    if x.dim() != 2 or x.dtype != torch.float32 or x.shape[1] % 8 != 0:
        raise ValueError("x must be a 2-D fp32 tensor with a multiple of 8 columns")
    xc = x if x.stride(1) == 1 else x.contiguous()
    out = torch.empty((xc.shape[0], 2 * xc.shape[1]), dtype=torch.float16, device=x.device)
    with _on_device(dev):
        rc = _lib.load().oeh_split_pairs(_ptr(xc), _ptr(out), xc.shape[0], xc.shape[1], xc.stride(0), _stream())
    _lib.check(rc, "oeh_split_pairs")
    return out


def split_triples(x2d: torch.Tensor) -> torch.Tensor:
    """fp32 activations (rows, K) -> the fp16 matrix [xh | xh 2^-5 | xl 2^-5 | 1, 2^-5, 0 x6] (rows, 3K + 8) of `oeh_split_triples`:
    against `attention.triple_weights(weight, bias)` ONE fp16 GEMM with fp32 accumulation (torch.mm(..., out_dtype=float32)) is
    the fp32 Linear - general weights, bias included."""
    dev = _need_gpu(x2d)
    if x2d.dim() != 2 or x2d.dtype != torch.float32 or x2d.shape[1] % 8 != 0:
        raise ValueError("x must be a 2-D fp32 tensor with K % 8 == 0")
    xc = x2d if x2d.stride(1) == 1 else x2d.contiguous()
    out = torch.empty((xc.shape[0], 3 * xc.shape[1] + 8), dtype=torch.float16, device=xc.device)
    with _on_device(dev):
        rc = _lib.load().oeh_split_triples(_ptr(xc), _ptr(out), xc.shape[0], xc.shape[1], xc.stride(0), _stream())
    _lib.check(rc, "oeh_split_triples")
    return out


def quantize_heads_i8(x: torch.Tensor, spec: "FakeQuantSpec", H: int, transpose: bool = False, want_values: bool = False,
                      alpha: float = 1.0, bias: Optional[torch.Tensor] = None, want_indices: bool = True):
    """A projection's output quantiser for the INT8-storage core (`oeh_quantize_heads_i8`): x (B,S,H*64) -> centred int8 indices,
    as a logical (B,H,S,64) view of a (B,S,H*64) tensor, or with `transpose` as the contiguous (B,H,64,S) tensor `attn_fwd_i8`
    wants for v; `want_values`: also the dequantised values (B,S,H*64) in x's dtype (a decoder's cache), same pass; `bias`
    (fp32, H*64): x is a raw GEMM accumulator and alpha * x + bias is what gets quantised.  `want_indices=False` (with
    `want_values`, not transposed): only the values come back - scale + bias + output fake-quant of a QuantLinear in one pass."""
    dev = _need_gpu(x, bias)
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != H * 64 or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous fp32 vector of H*64")
    if x.dim() != 3 or x.shape[2] != H * 64 or x.dtype not in _DT or spec.qmax != 255.0:
        raise ValueError("x must be (B,S,H*64) fp16/bf16/fp32 and the grid 8-bit")
    xc = x if x.stride(2) == 1 else x.contiguous()
    B, S, E = xc.shape
    if not want_indices and (transpose or not want_values):
        raise ValueError("want_indices=False needs want_values and the untransposed layout")
    out = torch.empty((B, H, 64, S) if transpose else (B, S, E), dtype=torch.int8, device=x.device) if want_indices else None
    y = torch.empty((B, S, E), dtype=x.dtype, device=x.device) if want_values else None
    xs = (C.c_int64 * 2)(xc.stride(0), xc.stride(1))
    ys = (C.c_int64 * 2)(S * E, E)
    with _on_device(dev):
        rc = _lib.load().oeh_quantize_heads_i8(_ptr(xc), _ptr(out), _ptr(y), B, S, H, xs, ys, _DT[x.dtype], float(spec.scale), float(spec.zero_point),
                                               int(bool(transpose)), float(alpha), _ptr(bias), _stream())
    _lib.check(rc, "oeh_quantize_heads_i8")
    if not want_indices:
        return y
    idx = out if transpose else out.view(B, S, H, 64).permute(0, 2, 1, 3)
    return (idx, y) if want_values else idx


def proj_quant_i8(a: torch.Tensor, w_int: torch.Tensor, bias: torch.Tensor, B: int, S: int, segs, *, pairs: bool, _outs=None, _prepared: Optional[list] = None):
    """The q / k / v projections of a QuantLinear model as ONE GEMM with the output quantisers in its epilogue (`oeh_proj_quant_i8`):
    a (B*S, K) fp16 activations, (B*S, 2K) operand pairs of an fp32 model (`split_pairs`) or the fp32 activations (B*S, K) themselves
    (split inside the kernel: same result as the pairs, without the pass), w_int (n*E, K) fp16 = the weights'
    integers, the segments one after the other, bias (n*E) fp32; `segs` = one (alpha, FakeQuantSpec, transpose, want_values) per
    segment.  Returns per segment what `quantize_heads_i8` returns: the centred int8 indices as a logical (B,H,S,64) view of
    (B,S,E), or the contiguous (B,H,64,S) tensor with `transpose`, and with `want_values` the pair (indices, values (B,S,E) fp32)."""
    dev = _need_gpu(a, w_int, bias)
    K = w_int.shape[1]
    n = len(segs)
    if a.dtype == torch.float32:  # the fp32 activations themselves: split into (hi, lo) inside the kernel, `pairs` is implied
        form, pairs = 2, False
    else:
        form = 1 if pairs else 0
    if a.dtype not in (torch.float16, torch.float32) or w_int.dtype != torch.float16 or bias.dtype != torch.float32 or a.dim() != 2 or w_int.dim() != 2:
        raise ValueError("a 2-D fp16 (values or operand pairs) or fp32, w_int 2-D fp16, bias fp32")
    if a.shape != (B * S, (2 if pairs else 1) * K) or w_int.shape[0] % n != 0 or bias.numel() != w_int.shape[0] or not bias.is_contiguous():
        raise ValueError("shapes: a (B*S, K or 2K), w_int (n*E, K), bias (n*E)")
    if a.stride(1) != 1 or w_int.stride(1) != 1:
        raise ValueError("a and w_int must have contiguous rows")
    E = w_int.shape[0] // n
    H = E // 64
    arr = (_lib.oeh_proj_seg * n)()
    keep, res = [], []
    for i, (alpha, spec, transpose, want_values) in enumerate(segs):
        if spec.qmax != 255.0:
            raise ValueError("the output grids must be 8-bit")
        if _outs is not None:   # (plan mode, quantization._I8LayerPlan: the caller's buffers)
            out, y = _outs[i]
        else:
            out = torch.empty((B, H, 64, S) if transpose else (B, S, E), dtype=torch.int8, device=a.device)
            y = torch.empty((B, S, E), dtype=torch.float32, device=a.device) if want_values else None
        keep.append((out, y))
        arr[i].alpha, arr[i].scale, arr[i].zero_point = float(alpha), float(spec.scale), float(spec.zero_point)
        arr[i].out, arr[i].y, arr[i].y_stride_row, arr[i].transpose = _ptr(out), _ptr(y), E, int(bool(transpose))
        idx = out if transpose else out.view(B, S, H, 64).permute(0, 2, 1, 3)
        res.append((idx, y) if want_values else idx)
    if _prepared is not None:  # hand back the prebuilt C call instead of launching: [fn, mutable argument list, segment array, keep-alive]
        _prepared.extend([_lib.load().oeh_proj_quant_i8, [_ptr(a), form, _ptr(w_int), _ptr(bias), B, S, K, E, n, arr, a.stride(0), w_int.stride(0)], arr, (keep, w_int, bias)])
        return res
    with _on_device(dev):
        rc = _lib.load().oeh_proj_quant_i8(_ptr(a), form, _ptr(w_int), _ptr(bias), B, S, K, E, n, arr, a.stride(0), w_int.stride(0), _stream())
    _lib.check(rc, "oeh_proj_quant_i8")
    return res


def proj_quant_values(a: torch.Tensor, w_int: torch.Tensor, bias: torch.Tensor, alpha: float, spec: "FakeQuantSpec", *, pairs: bool,
                      acc_add: Optional[torch.Tensor] = None, _prepared: Optional[list] = None) -> torch.Tensor:
    """A whole QuantLinear in one kernel (`oeh_proj_quant_i8`, values only): fake_quant(alpha * (a @ w_int^T) + bias) as fp32 (rows, N) -
    a (rows, K) fp16 (e.g. the integers of the producer's quantiser: exact products) or (rows, 2K) operand pairs, w_int (N, K) fp16 the
    weight's integers, `spec` the frozen 8-bit output quantiser.  rows % 16 == 0, K % 32 == 0, N % 64 == 0."""
    dev = _need_gpu(a, w_int, bias, acc_add)
    N, K = w_int.shape
    rows = a.shape[0]
    i8 = a.dtype == torch.int8  # int8 activations (e.g. centred indices) against int8 weights on the integer matrix cores; acc_add: int32 (N)
    if i8:
        if w_int.dtype != torch.int8 or pairs or a.dim() != 2 or a.shape[1] != K or bias.dtype != torch.float32:
            raise ValueError("int8 form: a (rows, K) int8, w_int (N, K) int8, bias (N) fp32")
        if acc_add is not None and (acc_add.dtype != torch.int32 or acc_add.numel() != N or not acc_add.is_contiguous()):
            raise ValueError("acc_add must be a contiguous int32 vector of N")
    elif a.dtype != torch.float16 or w_int.dtype != torch.float16 or bias.dtype != torch.float32 or a.dim() != 2 or a.shape[1] != (2 if pairs else 1) * K:
        raise ValueError("a (rows, K or 2K) fp16, w_int (N, K) fp16, bias (N) fp32")
    if a.stride(1) != 1 or w_int.stride(1) != 1 or not bias.is_contiguous() or bias.numel() != N or spec.qmax != 255.0:
        raise ValueError("contiguous rows, a bias of N and an 8-bit grid")
    if rows % 16 != 0:
        raise ValueError("rows must be a multiple of 16")
    y = torch.empty((rows, N), dtype=torch.float32, device=a.device)
    seg = (_lib.oeh_proj_seg * 1)()
    seg[0].alpha, seg[0].scale, seg[0].zero_point = float(alpha), float(spec.scale), float(spec.zero_point)
    seg[0].out, seg[0].y, seg[0].y_stride_row, seg[0].transpose, seg[0].acc_add = None, _ptr(y), N, 0, (_ptr(acc_add) if i8 else None)
    if _prepared is not None:  # [fn, mutable argument list, segment array, keep-alive]: the caller patches seg[0].y (and the activations' pointer) per call
        _prepared.extend([_lib.load().oeh_proj_quant_i8, [_ptr(a), 3 if i8 else int(bool(pairs)), _ptr(w_int), _ptr(bias), rows // 16, 16, K, N, 1, seg, a.stride(0), w_int.stride(0)],
                          seg, (a, w_int, bias, acc_add, y)])
        return y
    with _on_device(dev):
        rc = _lib.load().oeh_proj_quant_i8(_ptr(a), 3 if i8 else int(bool(pairs)), _ptr(w_int), _ptr(bias), rows // 16, 16, K, N, 1, seg, a.stride(0), w_int.stride(0), _stream())
    _lib.check(rc, "oeh_proj_quant_i8")
    return y


def attn_fwd_i8(q: torch.Tensor, k: torch.Tensor, v_t: torch.Tensor, grids, *, fq: AttnFakeQuant, out_dtype=torch.float16,
                softmax: SoftmaxSpec = SoftmaxSpec(), scale: float = 1.0, scale_div: float = 0.0, causal: bool = False, clamp_min: bool = False,
                mask_min: Optional[float] = None, gate: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                key_pad_mask: Optional[torch.Tensor] = None, _prepared: Optional[list] = None) -> torch.Tensor:
    """The INT8 configuration on the integer matrix cores (`include/oeh.h`: dtype OEH_I8): q, k logical (B,H,S,64) int8 views of
    centred indices (`centre_indices`), v_t the TRANSPOSED values, a (B,H,64,Sk) int8 view with contiguous keys; `grids` =
    (q, k, v) QuantGrid; `fq` with scores and probabilities (8-bit) [and context].  Returns the logical (B,H,Sq,64) result in
    `out_dtype`, stored (B,Sq,H,64)-contiguous.  `key_pad_mask`: (B,Sk) additive, entries 0 or <= -1e4 ONLY (HF's extended
    masks; the caller vouches for it - `attention.pad_is_boolean`).  Raises OehError(-95) for what this path does not take
    (clipping with gamma > 0, other head dims, full additive masks ...): the caller then runs `attn_fwd(..., fq=...)` on the dequantised values."""
    dev = _need_gpu(q, k, v_t, gate, out, key_pad_mask)
    if q.dtype != torch.int8 or k.dtype != torch.int8 or v_t.dtype != torch.int8:
        raise ValueError("q, k, v_t must be int8 (centred indices)")
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    if k.shape != (B, H, Sk, D) or v_t.shape != (B, H, D, Sk):
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} k {tuple(k.shape)} v_t {tuple(v_t.shape)} (v_t is (B,H,D,Sk))")
    if q.stride(3) != 1 or k.stride(3) != 1 or v_t.stride(3) != 1:
        raise ValueError("q, k need a contiguous head dim and v_t contiguous keys")
    if out is None:
        out = torch.empty((B, Sq, H, D), dtype=out_dtype, device=q.device).permute(0, 2, 1, 3)
    elif out.shape != (B, H, Sq, D) or out.dtype != out_dtype or out.stride(3) != 1:
        raise ValueError("out must be a (B,H,Sq,D) view with unit head-dim stride and dtype out_dtype")
    d = oeh_attn_desc()
    d.B, d.H, d.Sq, d.Sk, d.D, d.dtype, d.o_dtype = B, H, Sq, Sk, D, _lib.OEH_I8, (_lib.OEH_I8 if out_dtype == torch.int8 else _DT[out_dtype])
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v_t), ("o_stride", out)):
        getattr(d, name)[:] = [t.stride(0), t.stride(1), t.stride(2)]
    for name, gr in zip(("q_grid", "k_grid", "v_grid"), grids):
        getattr(d, name).scale, getattr(d, name).zero_point = float(gr.scale), float(gr.zero_point)
    d.scale, d.scale_div = float(scale), float(scale_div)
    d.softmax_base, d.clip, d.gamma, d.eta = int(softmax.base), int(bool(softmax.clip)), float(softmax.gamma), float(softmax.eta)
    d.causal, d.clamp_min = int(bool(causal)), int(bool(clamp_min))
    d.mask_min = float(torch.finfo(torch.float32).min if mask_min is None else mask_min)
    keep = []
    if key_pad_mask is not None:
        pm = key_pad_mask
        if pm.dtype not in (torch.float16, torch.float32):
            pm = pm.float()
        pm = pm.reshape(B, Sk) if pm.numel() == B * Sk else pm.reshape(1, Sk).expand(B, Sk)
        if pm.stride(1) != 1:
            pm = pm.contiguous()
        keep.append(pm)
        d.key_pad_mask, d.key_pad_dtype, d.key_pad_stride = pm.data_ptr(), _DT[pm.dtype], pm.stride(0)
        d.key_pad_boolean = 1  # (this path always reads the mask that way: include/oeh.h)
    if gate is not None:
        g = gate.to(torch.float32)
        while g.dim() < 4:
            g = g.unsqueeze(0)
        g = g.expand(B, H, Sq, 1)
        keep.append(g)
        d.gate = g.data_ptr()
        d.gate_stride[:] = [g.stride(0), g.stride(1), g.stride(2)]
    fqd = oeh_fq_desc()
    _fill_fq(fqd.scores, fq.scores)
    _fill_fq(fqd.probs, fq.probs)
    _fill_fq(fqd.ctx, fq.ctx)
    fqd.ctx_quant_before_gate = int(bool(fq.ctx_before_gate))
    fqd.ctx_emit_index = int(bool(fq.ctx_emit_index))
    if _prepared is not None:  # hand back the prebuilt C call instead of launching (bench / A-B loops)
        _prepared.extend([_lib.load().oeh_attn_fwd, (C.byref(d), _ptr(q), _ptr(k), _ptr(v_t), _ptr(out), C.byref(fqd)), (d, fqd, keep, q, k, v_t, out)])
        return out
    with _on_device(dev):
        rc = _lib.load().oeh_attn_fwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v_t), _ptr(out), C.byref(fqd), _stream())
    _lib.check(rc, "oeh_attn_fwd (INT8 storage)")
    return out


@dataclass
class GatePredictor:
    """The conditional per-token gate evaluated INSIDE the attention kernel (include/oeh.h: gate_hidden ...): the layer
    input `hidden` (B,Sq,H*D) and the per-head predictor weights laid out as for `gate_fwd`; `out` (B,H,Sq) fp32, optional,
    receives the gate probabilities (without `scaling`).  The 16-bit MFMA kernels (full-row and one-pass) take it, and the full-row
    kernel on fp32 storage (operand pairs: fp32-accurate logits) (`fused_gate_ok`)."""
    hidden: torch.Tensor
    w1: torch.Tensor
    b1: torch.Tensor
    w2: Optional[torch.Tensor] = None
    b2: Optional[torch.Tensor] = None
    scaling: float = 1.0
    out: Optional[torch.Tensor] = None


def fused_gate_ok(B, H, Sq, Sk, D, dtype, clip: bool = False, fq: bool = False, units: int = 0, **problem) -> bool:
    """True when `attn_fwd(..., gate_mlp=...)` is supported for this problem (else: `gate_fwd` + `gate=`).  `problem`: the
    remaining descriptor fields that decide the kernel variant (`attn_variant`'s keywords: base, gamma, key_pad, causal,
    scale, scale_div, mask_min) - the probe must describe the real call, not a default one."""
    if dtype not in _DT or fq or int(units) > 64:
        return False
    v = attn_variant(B, H, Sq, Sk, D, dtype, clip=clip, gate_hidden=True, **problem)
    if v is None:
        return False
    if dtype == torch.float32:  # fp32 storage: the full-row kernel's operand-pair form (rows of <= 512 keys) ...
        if not v.startswith("fast16/"):
            return False
        # ... unless the problem WITHOUT the predictor runs the one-pass kernel: that kernel + one `gate_fwd` launch is the faster pair
        # (round 5, OPT-125m shape B=16 S=512 fp32: 42.1 + ~10 us against 58.3 us for the full-row fp32 kernel with the predictor inside)
        v0 = attn_variant(B, H, Sq, Sk, D, dtype, clip=clip, gate_hidden=False, **problem)
        return not (v0 or "").startswith("flash16/")
    return v.startswith("fast16/") or v.startswith("flash16/")


class PreparedAttn:
    """A fully built `oeh_attn_fwd` call (descriptor + pointers) for launch loops where Python argument
    marshalling would otherwise dominate: `p = PreparedAttn(q, k, v, causal=True, ...); p(); p.out`."""

    def __init__(self, q, k, v, i8_grids=None, **kw):
        box = []
        if i8_grids is not None:  # INT8 storage: q, k int8 views, v transposed (attn_fwd_i8)
            self.out = attn_fwd_i8(q, k, v, i8_grids, _prepared=box, **kw)
        else:
            self.out = attn_fwd(q, k, v, _prepared=box, **kw)
        self._fn, self._args, self._keep = box

    def __call__(self, stream: Optional[C.c_void_p] = None) -> None:
        rc = self._fn(*self._args, _stream() if stream is None else stream)
        if rc != 0:
            _lib.check(rc, "oeh_attn_fwd")


def attn_variant(B, H, Sq, Sk, D, dtype=torch.float16, fq: bool = False, clip: bool = False, *, base: int = 1, gamma: float = -0.025,
                 key_pad: bool = False, full_mask: bool = False, causal: bool = False, scale: float = 1.0, scale_div: float = 0.0,
                 mask_min: Optional[float] = None, key_pad_boolean: bool = False, gate_hidden: bool = False) -> Optional[str]:
    """Name of the kernel variant the library would pick for this problem (host only; no GPU needed).  `gate_hidden`: with the
    per-token gate predictor evaluated in the kernel (it narrows the choice)."""
    d = oeh_attn_desc()
    d.B, d.H, d.Sq, d.Sk, d.D, d.dtype = B, H, Sq, Sk, D, _DT[dtype]
    d.scale, d.scale_div = float(scale), float(scale_div)
    d.mask_min = float(torch.finfo(torch.float32).min if mask_min is None else mask_min)
    d.softmax_base, d.causal = int(base), int(bool(causal))
    # only nullness of the mask pointers matters to the selection (host only: nothing is dereferenced)
    d.key_pad_mask, d.key_pad_dtype, d.key_pad_boolean = (1 if key_pad else None), OEH_F32, int(bool(key_pad_boolean))
    d.full_mask, d.full_mask_dtype = (1 if full_mask else None), OEH_F32
    if gate_hidden:
        d.gate_hidden, d.gate_w1, d.gate_b1 = 1, 1, 1
    if clip:
        d.clip, d.gamma, d.eta = 1, float(gamma), 1.0
    fqd = None
    if fq:
        fqd = oeh_fq_desc()
        fqd.scores.enable, fqd.scores.scale, fqd.scores.qmax = 1, 1.0, 255.0
        fqd.probs.enable, fqd.probs.scale, fqd.probs.qmax = 1, 1.0, 255.0
    r = _lib.load().oeh_attn_variant(C.byref(d), None if fqd is None else C.byref(fqd))
    return None if r is None else r.decode()


def softmax_rows(x: torch.Tensor, spec: SoftmaxSpec = SoftmaxSpec(), dim: int = -1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SOFTMAX_MAPPING callable on the GPU: softmax / softmax_1 / clipped variants along `dim`."""
    dev = _need_gpu(x, out)
    if x.dtype not in _DT:
        raise ValueError(f"unsupported dtype {x.dtype}")
    nd = x.dim()
    dim = dim % nd
    xt = x if dim == nd - 1 else x.transpose(dim, -1)
    xc = xt.contiguous()
    y = torch.empty_like(xc) if out is None or dim != nd - 1 else out
    cols = xc.shape[-1]
    rows = xc.numel() // max(cols, 1)
    if xc.numel():
        with _on_device(dev):
            rc = _lib.load().oeh_softmax_rows(_ptr(xc), _ptr(y), rows, cols, _DT[x.dtype], int(spec.base), int(bool(spec.clip)),
                                              float(spec.gamma), float(spec.eta), _stream())
        _lib.check(rc, "oeh_softmax_rows")
    return y if dim == nd - 1 else y.transpose(dim, -1)


def fake_quant(x: torch.Tensor, spec: FakeQuantSpec, want_idx: bool = False):
    """Fixed-range per-tensor asymmetric fake-quant; returns x_q (and the uint8 indices if asked)."""
    dev = _need_gpu(x)
    if x.dtype not in _DT:
        raise ValueError(f"unsupported dtype {x.dtype}")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    idx = torch.empty(xc.shape, dtype=torch.uint8, device=x.device) if want_idx else None
    with _on_device(dev):
        rc = _lib.load().oeh_fake_quant(_ptr(xc), _ptr(y), _ptr(idx), xc.numel(), _DT[x.dtype], spec.scale, spec.zero_point, spec.qmax, _stream())
    _lib.check(rc, "oeh_fake_quant")
    return (y, idx) if want_idx else y


_calib_work = {}  # (device index, stream) -> scratch buffer of the on-device percentile selection (two calibrations on different
                   # streams of one GPU must not share histograms: ADVICE r2)


def _calib_scratch(dev):
    """The current stream's 36-KB selection scratch on `dev`.  Bounded: streams come and go (a recycled handle simply reuses its
    buffer), so beyond 32 entries the table starts over instead of keeping one buffer per stream ever seen."""
    with _on_device(dev):
        key = (dev.index, torch.cuda.current_stream().cuda_stream)
    work = _calib_work.get(key)
    if work is None:
        if len(_calib_work) >= 32:
            _calib_work.clear()  # (buffers still referenced by enqueued work stay alive in the caching allocator's stream order)
        work = _calib_work[key] = torch.empty(_lib.CALIB_WORK_BYTES // 8, dtype=torch.int64, device=dev)
    return work


def percentile_ema(x: torch.Tensor, q_lo: float, q_hi: float, state: torch.Tensor, momentum: float = 0.9, first: bool = False) -> torch.Tensor:
    """(np.percentile(x, q_lo), np.percentile(x, q_hi)) blended into `state` (float64[2] on x's GPU) with the running average of
    RunningMinMaxEstimator (range_estimators.py:101-104), without leaving the device: `include/oeh.h: oeh_percentile_ema`."""
    dev = _need_gpu(x, state)
    if x.dtype not in _DT:
        raise ValueError(f"unsupported dtype {x.dtype}")
    if state.dtype != torch.float64 or state.numel() != 2 or not state.is_contiguous():
        raise ValueError("state must be a contiguous float64 tensor of 2 elements")
    xc = x.detach().contiguous()
    work = _calib_scratch(dev)
    with _on_device(dev):
        rc = _lib.load().oeh_percentile_ema(_ptr(xc), xc.numel(), _DT[x.dtype], float(q_lo), float(q_hi), float(momentum), int(bool(first)),
                                            _ptr(state), _ptr(work), _stream())
    _lib.check(rc, "oeh_percentile_ema")
    return state


CALIB_SCORES, CALIB_PROBS, CALIB_CONTEXT = 0, 1, 2


def attn_calibrate(q: torch.Tensor, k: torch.Tensor, v: Optional[torch.Tensor], which: int, *, softmax: SoftmaxSpec = SoftmaxSpec(), scale: float = 1.0,
                   scale_div: float = 0.0, key_pad_mask: Optional[torch.Tensor] = None, full_mask: Optional[torch.Tensor] = None, causal: bool = False,
                   clamp_min: bool = False, mask_min: Optional[float] = None, scores_range: Optional[torch.Tensor] = None,
                   probs_range: Optional[torch.Tensor] = None, n_bits: int = 8, eps: float = 1e-8, q_lo: float = 0.001, q_hi: float = 99.999,
                   momentum: float = 0.9, first: bool = False, state: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Range estimation of the attention quantisers without the (B,H,Sq,Sk) tensors (`include/oeh.h: oeh_attn_calibrate`).
    which CALIB_SCORES / CALIB_PROBS: the percentile pair [+ running average] of the scaled scores / of the probabilities goes into
    `state` (float64[2] on the device), returned; CALIB_CONTEXT: returns the fp32 context, logical (B,H,Sq,D), stored (B,Sq,H,D).
    `scores_range` / `probs_range`: float64[2] device tensors (x_min, x_max) whose grids quantise the scores / probabilities on the
    way (None: not quantised)."""
    dev = _need_gpu(q, k, v, key_pad_mask, full_mask, scores_range, probs_range, state)
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    if k.shape != (B, H, Sk, D) or (v is not None and v.shape != (B, H, Sk, D)):
        raise ValueError("shape mismatch")
    if q.dtype not in _DT or k.dtype != q.dtype or (v is not None and v.dtype != q.dtype):
        raise ValueError("q / k / v dtypes must match and be fp16 / bf16 / fp32")
    fix = lambda t: t if t is None or t.stride(3) == 1 else t.contiguous()  # noqa: E731
    q, k, v = fix(q), fix(k), fix(v)
    d = oeh_attn_desc()
    d.B, d.H, d.Sq, d.Sk, d.D, d.dtype = B, H, Sq, Sk, D, _DT[q.dtype]
    out = None
    if which == CALIB_CONTEXT:
        out = torch.empty((B, Sq, H, D), dtype=torch.float32, device=q.device).permute(0, 2, 1, 3)
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v if v is not None else k), ("o_stride", out if out is not None else q)):
        getattr(d, name)[:] = [t.stride(0), t.stride(1), t.stride(2)]
    d.scale, d.scale_div = float(scale), float(scale_div)
    d.softmax_base, d.clip, d.gamma, d.eta = int(softmax.base), int(bool(softmax.clip)), float(softmax.gamma), float(softmax.eta)
    keep = []
    if key_pad_mask is not None:
        m = key_pad_mask.reshape(B, Sk)
        m = (m if m.dtype in (torch.float16, torch.float32) else m.float()).contiguous()
        keep.append(m)
        d.key_pad_mask, d.key_pad_dtype, d.key_pad_stride = m.data_ptr(), _DT[m.dtype], m.stride(0)
    if full_mask is not None:
        if full_mask.shape != (B, 1, Sq, Sk):
            raise ValueError(f"Attention mask should be of size {(B, 1, Sq, Sk)}, but is {tuple(full_mask.shape)}")
        m = full_mask if full_mask.dtype in (torch.float16, torch.float32) else full_mask.float()
        m = m if m.stride(3) == 1 else m.contiguous()
        keep.append(m)
        d.full_mask, d.full_mask_dtype = m.data_ptr(), _DT[m.dtype]
        d.full_mask_stride[:] = [m.stride(0), m.stride(2)]
    d.causal, d.clamp_min = int(bool(causal)), int(bool(clamp_min))
    d.mask_min = float(torch.finfo(q.dtype).min if mask_min is None else mask_min)
    for r_ in (scores_range, probs_range, state):
        if r_ is not None and (r_.dtype != torch.float64 or r_.numel() != 2 or not r_.is_contiguous()):
            raise ValueError("ranges / state must be contiguous float64 tensors of 2 elements")
    work = None
    if which != CALIB_CONTEXT:
        if state is None:
            raise ValueError("state (float64[2] on the device) is required for the statistics passes")
        work = _calib_scratch(dev)
    with _on_device(dev):
        rc = _lib.load().oeh_attn_calibrate(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), int(which), _ptr(scores_range), _ptr(probs_range), int(n_bits),
                                            float(eps), float(q_lo), float(q_hi), float(momentum), int(bool(first)), _ptr(state), _ptr(work), _stream())
    _lib.check(rc, "oeh_attn_calibrate")
    return out if which == CALIB_CONTEXT else state


def fake_quant_range(x: torch.Tensor, xmin_xmax: torch.Tensor, n_bits: int = 8, eps: float = 1e-8) -> torch.Tensor:
    """Fake-quant with the grid derived on the device from a float64 (x_min, x_max) pair (`oeh_fake_quant_range`): the
    quantiser's forward while its range is still being estimated, without a host read of the range."""
    dev = _need_gpu(x, xmin_xmax)
    if x.dtype not in _DT:
        raise ValueError(f"unsupported dtype {x.dtype}")
    if xmin_xmax.dtype != torch.float64 or xmin_xmax.numel() != 2 or not xmin_xmax.is_contiguous():
        raise ValueError("xmin_xmax must be a contiguous float64 tensor of 2 elements")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    with _on_device(dev):
        rc = _lib.load().oeh_fake_quant_range(_ptr(xc), _ptr(y), xc.numel(), _DT[x.dtype], _ptr(xmin_xmax), int(n_bits), float(eps), _stream())
    _lib.check(rc, "oeh_fake_quant_range")
    return y


def gate_fwd(hidden: torch.Tensor, H: int, w1: torch.Tensor, b1: torch.Tensor, w2: Optional[torch.Tensor] = None,
             b2: Optional[torch.Tensor] = None, per_head_pool: bool = False, scaling: float = 1.0) -> torch.Tensor:
    """Per-head gate predictors on the module input (bert_attention.py:301-327).  hidden (B,T,H*d);
    w1 (H,d)|(H,m,d), b1 (H)|(H,m), w2 (H,m), b2 (H).  Returns sigmoid(logit)*scaling as (B,H,T,1) or (B,H,1,1) fp32."""
    dev = _need_gpu(hidden, w1, b1, w2, b2)
    B, T, E = hidden.shape
    d = E // H
    hc = hidden if hidden.stride(2) == 1 else hidden.contiguous()
    m_units = 0 if w1.dim() == 2 else w1.shape[1]
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()  # noqa: E731
    w1, b1, w2, b2 = f(w1), f(b1), f(w2), f(b2)
    out = torch.empty((B, H, T), dtype=torch.float32, device=hidden.device)
    with _on_device(dev):
        rc = _lib.load().oeh_gate_fwd(_ptr(hc), _DT[hidden.dtype], B, T, H, d, hc.stride(0), hc.stride(1), _ptr(w1), _ptr(b1), _ptr(w2),
                                      _ptr(b2), m_units, int(per_head_pool), float(scaling), _ptr(out), _stream())
    _lib.check(rc, "oeh_gate_fwd")
    return out[:, :, :1, None] if per_head_pool else out[..., None]


def minmax(x: torch.Tensor) -> torch.Tensor:
    """(min, max) of a tensor as a 2-element fp32 GPU tensor, no host round trip (range_estimators.py:96-97)."""
    dev = _need_gpu(x)
    xc = x.contiguous()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    with _on_device(dev):
        rc = _lib.load().oeh_minmax(_ptr(xc), xc.numel(), _DT[x.dtype], _ptr(out), _stream())
    _lib.check(rc, "oeh_minmax")
    return out
