"""INT8 post-training fake-quant around the attention path, MI355X side.

Host-side mirror of the part of the reference's `quantization/` package that the attention hot path touches
(SURVEY.md section 2 rows 9-12 and the attention classes of rows 6-7):

  AsymmetricUniformQuantizer / SymmetricUniformQuantizer   quantization/quantizers/uniform_quantizers.py:12-310
  CurrentMinMaxEstimator / RunningMinMaxEstimator           quantization/range_estimators.py:54-106
  QuantizationManager (estimate -> fix state machine)       quantization/quantization_manager.py:11-108
  QuantizedModule / QuantizedActivation                     quantization/base_quantized_classes.py:38-199
  QuantLinear (weight + output-activation quant)            quantization/hijacker.py:27-134, autoquant_utils.py:18-20
  Quantized{Bert,OPT}...AttentionWithExtras                 models/quantized_bert.py:221-440, quantized_opt.py:54-274

Same class / attribute / buffer names (`activation_quantizer.quantizer._delta`, `_zero_float`, ...) so calibrated
state dicts are interchangeable.  Activation fake-quant runs on the GPU: fused into the attention kernel once the
three ranges are fixed (the INT8 validate path), as the stand-alone HIP kernel otherwise.  Range estimation keeps
everything on the device (the reference copies each (B,H,S,S) tensor to the host for np.percentile).
"""
from __future__ import annotations

import copy
import dataclasses
from enum import Enum
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import OehError as _OehError
from .attention import (AttentionGateType, BaseEnumOptions, GateBookkeeping, GateState, attention_core, classify_causal, has_hooks, pad_is_boolean,
                        split_mask, unfused_core)
from .ops import AttnFakeQuant, FakeQuantSpec
from .softmax import spec_of


class QuantizerNotInitializedError(Exception):
    def __init__(self):
        super().__init__("Quantizer has  not been initialized yet")


class Qstates(BaseEnumOptions):
    estimate_ranges = 1
    fix_ranges = 2
    learn_ranges = 4
    estimate_ranges_train = 8


# ------------------------------------------------------------------------------------------------------------
# quantisers
# ------------------------------------------------------------------------------------------------------------
class AsymmetricUniformQuantizer(nn.Module):
    """Per-tensor asymmetric uniform fake-quant: idx = clamp(round(x/scale)+zp, 0, 2^n-1); x_q = scale*(idx-zp)."""

    def __init__(self, n_bits, scale_domain="linear", grad_scaling=False, eps=1e-8, per_channel=False, act_quant=False, **kwargs):
        super().__init__()
        if scale_domain != "linear" or per_channel or grad_scaling:
            raise NotImplementedError("only linear-domain per-tensor quantisers are on the MI355X attention path")
        self.n_bits = n_bits
        self.act_quant = act_quant
        self.per_channel = per_channel
        self.state = None
        self.x_min_fp32 = self.x_max_fp32 = None
        self.register_buffer("_delta", None)
        self.register_buffer("_zero_float", None)
        self.scale_domain = scale_domain
        self.eps = eps
        self._spec_cache = None

    @property
    def delta(self):
        if self._delta is None:
            raise QuantizerNotInitializedError()
        return self._delta

    @property
    def zero_float(self):
        if self._zero_float is None:
            raise QuantizerNotInitializedError()
        return self._zero_float

    @property
    def is_initialized(self):
        return self._delta is not None

    @property
    def symmetric(self):
        return False

    @property
    def int_min(self):
        return 0.0

    @property
    def int_max(self):
        return 2.0 ** self.n_bits - 1

    @property
    def scale(self):
        return torch.clamp(self.delta, min=self.eps)

    @property
    def zero_point(self):
        return torch.clamp(torch.round(self.zero_float), self.int_min, self.int_max)

    @property
    def x_max(self):
        return self.scale * (self.int_max - self.zero_point)

    @property
    def x_min(self):
        return self.scale * (self.int_min - self.zero_point)

    def spec(self) -> FakeQuantSpec:
        """Host-side (scale, zero_point, qmax) for the kernels; one device->host read per range change."""
        key = (self._delta.data_ptr(), self._delta._version, self._zero_float.data_ptr(), self._zero_float._version)
        if self._spec_cache is None or self._spec_cache[0] != key:
            self._spec_cache = (key, FakeQuantSpec.from_delta(float(self._delta), float(self._zero_float), self.n_bits, self.eps))
        return self._spec_cache[1]

    def to_integer_forward(self, x_float):
        _, idx = ops.fake_quant(x_float, self.spec(), want_idx=True)
        return idx.to(x_float.dtype)

    def forward(self, x_float):
        return ops.fake_quant(x_float, self.spec())

    def _tensorize_min_max(self, x_min, x_max):
        if not torch.is_tensor(x_min):
            x_min = torch.tensor(x_min).float()
            x_max = torch.tensor(x_max).float()
        if x_min.dim() > 0 and len(x_min) > 1:
            raise ValueError("x_min and x_max must be a float or 1-D Tensor for per-tensor quantization (per_channel=False)")
        x_min = torch.min(x_min, torch.zeros_like(x_min))
        x_max = torch.max(x_max, torch.ones_like(x_max) * self.eps)
        return x_min, x_max

    def set_quant_range(self, x_min, x_max):
        self.x_min_fp32, self.x_max_fp32 = x_min, x_max
        x_min, x_max = self._tensorize_min_max(x_min, x_max)
        self._delta = ((x_max - x_min) / self.int_max).detach()
        self._zero_float = (-x_min / self._delta).detach()

    def fix_ranges(self):
        pass  # ranges are plain buffers here (no learnable-range mode on this path)

    def make_range_trainable(self):
        raise NotImplementedError("learned ranges (QAT) are outside the inference hot path")

    def reset(self):
        self._delta = None
        self._zero_float = None

    def extra_repr(self):
        return f"n_bits={self.n_bits}, per_channel={self.per_channel}, is_initalized={self.is_initialized}"


class SymmetricUniformQuantizer(AsymmetricUniformQuantizer):
    """Signed symmetric grid for WEIGHTS (quant_configs.py:26); evaluated once per weight and cached by QuantLinear,
    so plain device tensor ops are enough (not on the attention core: SURVEY 8a row a12)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("_signed", None)

    @property
    def signed(self):
        if self._signed is None:
            raise QuantizerNotInitializedError()
        return bool(self._signed.item())

    @property
    def symmetric(self):
        return True

    @property
    def int_min(self):
        return -(2.0 ** (self.n_bits - 1)) if self.signed else 0

    @property
    def int_max(self):
        return 2.0 ** (self.n_bits - int(self.signed)) - 1

    @property
    def zero_point(self):
        return 0.0

    def set_quant_range(self, x_min, x_max):
        self.x_min_fp32, self.x_max_fp32 = x_min, x_max
        x_min, x_max = self._tensorize_min_max(x_min, x_max)
        self._signed = x_min.min() < 0
        self._delta = (torch.max(x_min.abs(), x_max) / self.int_max).detach()

    def forward(self, x_float):
        scale = self.scale
        x_int = torch.clamp(torch.round(x_float / scale), self.int_min, self.int_max)
        return scale * x_int

    def to_integer_forward(self, x_float):
        return torch.clamp(torch.round(x_float / self.scale), self.int_min, self.int_max)


# ------------------------------------------------------------------------------------------------------------
# range estimators
# ------------------------------------------------------------------------------------------------------------
def percentile_pair(x: torch.Tensor, lo_q: float, hi_q: float):
    """np.percentile(x, (lo_q, hi_q)) (linear interpolation, float64 result) computed on the device: the tails are
    extracted with top-k instead of copying the tensor to the host and sorting it (range_estimators.py:90-94)."""
    flat = x.detach().reshape(-1).float()
    n = flat.numel()

    def one(q):
        h = (n - 1) * (q / 100.0)
        i = int(np.floor(h))
        frac = h - i
        from_top = n - 1 - i  # 0 = the maximum
        if from_top + 2 <= min(n, 1 << 20):
            vals = torch.topk(flat, min(n, from_top + 1), largest=True, sorted=True).values  # descending
            a = float(vals[from_top])
            b = float(vals[from_top - 1]) if from_top >= 1 else a
        elif i + 3 <= min(n, 1 << 20):
            vals = torch.topk(flat, min(n, i + 2), largest=False, sorted=True).values  # ascending
            a = float(vals[i])
            b = float(vals[i + 1]) if i + 1 < n else a
        else:
            srt = torch.sort(flat).values
            a = float(srt[i])
            b = float(srt[min(i + 1, n - 1)])
        # numpy's _lerp on float32 data: the difference is formed in float32, the interpolation in float64,
        # anchored at b for frac >= 0.5
        diff = np.float64(np.float32(b) - np.float32(a))
        if frac >= 0.5:
            return np.float64(np.float32(b)) - diff * (1.0 - np.float64(frac))
        return np.float64(np.float32(a)) + diff * np.float64(frac)

    return one(lo_q), one(hi_q)


class RangeEstimatorBase(nn.Module):
    def __init__(self, *args, per_channel=False, quantizer=None, **kwargs):
        super().__init__()
        if per_channel:
            raise NotImplementedError("per-channel ranges are a weights-only option outside the attention path")
        self.register_buffer("current_xmin", None)
        self.register_buffer("current_xmax", None)
        self.per_channel = per_channel
        object.__setattr__(self, "quantizer", quantizer)  # not a submodule (keeps state-dict keys like the reference's repr)
        # GPU tensors with a percentile: the (x_min, x_max) pair lives in ONE float64[2] device tensor that the HIP selection
        # kernel updates in place (ops.percentile_ema); current_xmin / current_xmax are snapshots of it.  Not a buffer: the
        # state dict keeps the reference's keys.
        self.device_state = None

    def reset(self):
        self.current_xmin = None
        self.current_xmax = None
        self.device_state = None

    def _percentile_on_device(self, x, q_lo, q_hi, momentum):
        """(np.percentile(x, q_lo), np.percentile(x, q_hi)) [+ running average] without leaving the GPU: no topk, no host sync."""
        first = self.device_state is None or self.device_state.device != x.device
        if first:
            self.device_state = torch.empty(2, dtype=torch.float64, device=x.device)
        # (momentum None - CurrentMinMaxEstimator: no running average - is "every batch is the first one")
        ops.percentile_ema(x, q_lo, q_hi, self.device_state, momentum=0.0 if momentum is None else momentum, first=first or momentum is None)
        self.current_xmin, self.current_xmax = self.device_state[0].clone(), self.device_state[1].clone()
        return self.current_xmin, self.current_xmax


class CurrentMinMaxEstimator(RangeEstimatorBase):
    def __init__(self, *args, percentile=None, **kwargs):
        self.percentile = percentile
        super().__init__(*args, **kwargs)

    def forward(self, x):
        if self.percentile and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16, torch.float32):
            self.device_state = None  # no running average: every call starts over
            return self._percentile_on_device(x, self.percentile, 100 - self.percentile, 0.0)
        if self.percentile:
            lo, hi = percentile_pair(x, self.percentile, 100 - self.percentile)
            self.current_xmin = torch.tensor(lo).to(x.device)
            self.current_xmax = torch.tensor(hi).to(x.device)
        else:
            mm = ops.minmax(x) if x.is_cuda else torch.stack([x.min(), x.max()]).float()
            self.current_xmin, self.current_xmax = mm[0].to(x.dtype).detach(), mm[1].to(x.dtype).detach()
        return self.current_xmin, self.current_xmax


class RunningMinMaxEstimator(RangeEstimatorBase):
    def __init__(self, *args, momentum=0.9, percentile=None, **kwargs):
        self.momentum = momentum
        self.percentile = percentile
        super().__init__(*args, **kwargs)

    def forward(self, x):
        if self.percentile and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16, torch.float32) and \
                (self.current_xmin is None or self.device_state is not None):
            return self._percentile_on_device(x, 100 - self.percentile, self.percentile, self.momentum)
        if self.percentile:
            lo, hi = percentile_pair(x, 100 - self.percentile, self.percentile)
            x_min, x_max = torch.tensor(lo).to(x.device), torch.tensor(hi).to(x.device)  # float64, like the reference
        else:
            mm = ops.minmax(x)
            x_min, x_max = mm[0].to(x.dtype).detach(), mm[1].to(x.dtype).detach()
        if self.current_xmin is None:
            self.current_xmin, self.current_xmax = x_min, x_max
        else:
            self.current_xmin = (1 - self.momentum) * x_min + self.momentum * self.current_xmin
            self.current_xmax = (1 - self.momentum) * x_max + self.momentum * self.current_xmax
        return self.current_xmin, self.current_xmax


class _ClsEnum(Enum):
    @property
    def cls(self):
        return self.value

    def __call__(self, *a, **k):
        return self.value(*a, **k)

    def __str__(self):
        return self.name

    @classmethod
    def list_names(cls):
        return [m.name for m in cls]


class QMethods(_ClsEnum):
    symmetric_uniform = SymmetricUniformQuantizer
    asymmetric_uniform = AsymmetricUniformQuantizer


class RangeEstimators(_ClsEnum):
    current_minmax = CurrentMinMaxEstimator
    running_minmax = RunningMinMaxEstimator


# ------------------------------------------------------------------------------------------------------------
# manager + module wrappers
# ------------------------------------------------------------------------------------------------------------
class QuantizationManager(nn.Module):
    """estimate_ranges: every forward updates the range, then quantises; fix_ranges: quantise with frozen buffers."""

    def __init__(self, qmethod=SymmetricUniformQuantizer, init=CurrentMinMaxEstimator, per_channel=False, x_min=None, x_max=None,
                 qparams=None, init_params=None):
        super().__init__()
        self.state = Qstates.estimate_ranges
        self.qmethod, self.init, self.per_channel = qmethod, init, per_channel
        self.qparams = qparams if qparams else {}
        self.init_params = init_params if init_params else {}
        self.range_estimator = None
        self.quantizer = self.qmethod(per_channel=self.per_channel, **self.qparams)
        self.quantizer.state = self.state
        if x_min is not None and x_max is not None:
            self.set_quant_range(x_min, x_max)
            self.fix_ranges()
        else:
            self.range_estimator = self.init(per_channel=self.per_channel, quantizer=self.quantizer, **self.init_params)

    @property
    def n_bits(self):
        return self.quantizer.n_bits

    def estimate_ranges(self):
        self.state = self.quantizer.state = Qstates.estimate_ranges

    def fix_ranges(self):
        if not self.quantizer.is_initialized:
            raise QuantizerNotInitializedError()
        self.state = self.quantizer.state = Qstates.fix_ranges
        self.quantizer.fix_ranges()

    def estimate_ranges_train(self):
        self.state = self.quantizer.state = Qstates.estimate_ranges_train

    def reset_ranges(self):
        self.range_estimator.reset()
        self.quantizer.reset()
        self.estimate_ranges()

    def set_quant_range(self, x_min, x_max):
        self.quantizer.set_quant_range(x_min, x_max)

    @property
    def is_fixed(self) -> bool:
        return self.state == Qstates.fix_ranges or (self.state == Qstates.estimate_ranges_train and not self.training)

    def forward(self, x):
        if not self.is_fixed:
            self.set_quant_range(*self.range_estimator(x))
            st = getattr(self.range_estimator, "device_state", None)
            if st is not None and x.is_cuda and type(self.quantizer) is AsymmetricUniformQuantizer:
                # the range is still moving: quantise with the grid derived on the device from the estimator's float64 pair
                # (the same arithmetic as set_quant_range above) instead of reading it back for a host-side descriptor
                return ops.fake_quant_range(x, st, self.quantizer.n_bits, self.quantizer.eps)
        return self.quantizer(x)

    def extra_repr(self):
        return f"state={self.state.name}"


def _apply_qm(root: nn.Module, fn):
    for m in root.modules():
        if isinstance(m, QuantizationManager):
            fn(m)


class QuantizedModule(nn.Module):
    """Switches a module between quantised and full-precision mode (`_quant_a` / `_quant_w` buffers)."""

    def __init__(self, *args, method=AsymmetricUniformQuantizer, act_method=None, weight_range_method=CurrentMinMaxEstimator,
                 act_range_method=RunningMinMaxEstimator, n_bits=8, n_bits_act=None, per_channel_weights=False, percentile=None,
                 weight_range_options=None, act_range_options=None, scale_domain="linear", **kwargs):
        for junk in ("act_quant_dict", "quant_dict", "quant_setup", "bayesian_bits_kwargs", "prune_method", "prune_kwargs"):
            kwargs.pop(junk, None)
        super().__init__(*args, **kwargs)
        self.method = method
        self.act_method = act_method or method
        self.n_bits = n_bits
        self.n_bits_act = n_bits_act or n_bits
        self.per_channel_weights = per_channel_weights
        self.percentile = percentile
        self.weight_range_method = weight_range_method
        self.weight_range_options = weight_range_options if weight_range_options else {}
        self.act_range_method = act_range_method
        self.act_range_options = act_range_options if act_range_options else {}
        self.scale_domain = scale_domain
        self.cached_params = None
        self._caching = True
        self.register_buffer("_quant_w", torch.BoolTensor([False]))
        self.register_buffer("_quant_a", torch.BoolTensor([False]))
        self._qa = self._qw = False  # host copies of the flags (no device read per forward)
        self.act_qparams = dict(n_bits=self.n_bits_act, scale_domain=self.scale_domain, act_quant=True)
        self.weight_qparams = dict(n_bits=self.n_bits, scale_domain=self.scale_domain, act_quant=False)

    def quantized_weights(self):
        self.cached_params = None
        self._quant_w = torch.BoolTensor([True]).to(self._quant_w.device)
        self._qw = True

    def full_precision_weights(self):
        self.cached_params = None
        self._quant_w = torch.BoolTensor([False]).to(self._quant_w.device)
        self._qw = False

    def quantized_acts(self):
        self._quant_a = torch.BoolTensor([True]).to(self._quant_a.device)
        self._qa = True

    def full_precision_acts(self):
        self._quant_a = torch.BoolTensor([False]).to(self._quant_a.device)
        self._qa = False

    def quantized(self):
        self.quantized_weights()
        self.quantized_acts()

    def full_precision(self):
        self.full_precision_weights()
        self.full_precision_acts()

    def get_quantizer_status(self):
        return dict(quant_a=self._qa, quant_w=self._qw)

    def fix_ranges(self):
        _apply_qm(self, lambda m: m.fix_ranges() if m.quantizer.is_initialized else None)

    def estimate_ranges(self):
        _apply_qm(self, lambda m: m.estimate_ranges())

    def estimate_ranges_train(self):
        _apply_qm(self, lambda m: m.estimate_ranges_train() if m.quantizer.is_initialized else None)

    def train(self, mode=True):
        super().train(mode)
        if mode:
            self.cached_params = None
        return self

    def _apply(self, *args, **kwargs):
        self.cached_params = None
        return super()._apply(*args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        # a quantised checkpoint carries `_quant_a` / `_quant_w` = True: the host copies must follow the loaded buffers,
        # or the module would silently run in full precision (one device read per load, none per forward)
        super()._load_from_state_dict(*args, **kwargs)
        self.cached_params = None
        self._qa, self._qw = bool(self._quant_a.item()), bool(self._quant_w.item())

    def extra_repr(self):
        return f"weight_quant={self._qw}, act_quant={self._qa}"


class QuantizedActivation(QuantizedModule):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.activation_quantizer = QuantizationManager(qmethod=self.act_method, qparams=self.act_qparams, init=self.act_range_method,
                                                        init_params=self.act_range_options)

    def quantize_activations(self, x):
        return self.activation_quantizer(x) if self._qa else x

    def forward(self, x):
        return self.quantize_activations(x)

    def fixed_spec(self) -> Optional[FakeQuantSpec]:
        """FakeQuantSpec if this quantiser is active with frozen ranges (fusable), None if inactive."""
        return self.activation_quantizer.quantizer.spec() if self._qa else None

    @property
    def fusable(self) -> bool:
        return (not self._qa) or self.activation_quantizer.is_fixed


class QuantLinear(QuantizedModule, nn.Linear):
    """nn.Linear with fake-quantised weights (cached in eval) and fake-quantised output activations."""

    def __init__(self, *args, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.activation_function = copy.deepcopy(activation) if activation else None
        self.activation_quantizer = QuantizationManager(qmethod=self.act_method, init=self.act_range_method, qparams=self.act_qparams,
                                                        init_params=self.act_range_options)
        w_init = dict(percentile=self.percentile) if self.weight_range_method is CurrentMinMaxEstimator else self.weight_range_options
        self.weight_quantizer = QuantizationManager(qmethod=self.method, init=self.weight_range_method,
                                                    per_channel=self.per_channel_weights, qparams=self.weight_qparams, init_params=w_init)

    def get_params(self):
        if not self.training and self.cached_params:
            return self.cached_params
        weight, bias = self.weight, self.bias
        if self._qw:
            weight = self.weight_quantizer(weight)
        if self._caching and not self.training and self.cached_params is None:
            self.cached_params = (weight.detach().clone(), None if bias is None else bias.detach().clone())
        return weight, bias

    def _pair_weights(self):
        """[Iw ; Iw * 2^-11] (2K, N) fp16 and the fp32 weight scale, cached: the quantised weight is scale * Iw with Iw an
        8-bit integer matrix - exact in fp16 - so `ops.split_pairs(x) @ this` is the fp32 linear at fp16 matrix-core speed."""
        qz = self.weight_quantizer.quantizer
        key = (self.weight.data_ptr(), self.weight._version, qz._delta.data_ptr(), qz._delta._version)
        hit = self.__dict__.get("_pair_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                iw = qz.to_integer_forward(self.weight.detach()).t().contiguous()  # (K, N), integers in [-128, 127]
                ww = torch.cat([iw, iw * (1.0 / 2048.0)], dim=0).to(torch.float16).contiguous()
                s32 = float(np.float32(float(qz.scale)))
            hit = (key, ww, s32)
            self.__dict__["_pair_cache"] = hit
        return hit[1], hit[2]

    def _int_weights(self):
        """The quantised weight's integers Iw (N, K) as fp16 (exact) and the fp32 weight scale, cached - the weight operand of
        `ops.proj_quant_i8`, which forms Iw * 2^-11 for the lo half of the operand pairs in registers."""
        qz = self.weight_quantizer.quantizer
        key = (self.weight.data_ptr(), self.weight._version, qz._delta.data_ptr(), qz._delta._version)
        hit = self.__dict__.get("_int_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                iw = qz.to_integer_forward(self.weight.detach()).to(torch.float16).contiguous()
                s32 = float(np.float32(float(qz.scale)))
            hit = (key, iw, s32)
            self.__dict__["_int_cache"] = hit
        return hit[1], hit[2]

    def pair_gemm_ok(self, x) -> bool:
        """fp32 inference on the GPU with per-tensor symmetric <= 8-bit weights: the operand-pair GEMM applies."""
        qz = self.weight_quantizer.quantizer
        return (PAIR_GEMM and not self.training and self._qw and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and type(qz) is SymmetricUniformQuantizer and qz.is_initialized and qz.n_bits <= 8 and self.in_features % 8 == 0)

    def linear_pairs(self, x, pairs=None, raw: bool = False):
        """x @ W_q^T + b through one fp16 GEMM on operand pairs (`pairs` = ops.split_pairs(x 2-D), shared by projections of the
        same input): fp32-accurate (the weight side exactly), ~3x faster than the fp32 library GEMM at these sizes.
        raw: return (accumulator, weight scale, bias) instead - the caller folds scale and bias into its next pass."""
        ww, s32 = self._pair_weights()
        if pairs is None:
            pairs = ops.split_pairs(x.reshape(-1, x.shape[-1]))
        acc = torch.mm(pairs, ww, out_dtype=torch.float32).view(*x.shape[:-1], self.out_features)
        if raw:
            return acc, s32, self.bias
        return torch.add(self.bias.detach(), acc, alpha=s32) if self.bias is not None else acc * s32

    def _pair_gemm_quantised(self, x):
        """The pair GEMM with its epilogue in ONE pass over the accumulator (`oeh_quantize_heads_i8`, values only): weight scale,
        bias and the frozen 8-bit output quantiser.  None when that does not apply (the caller then runs the separate ops)."""
        if (not self._qa or self.activation_function is not None or self.bias is None or self.out_features % 64 != 0 or x.dim() < 2
                or not self.activation_quantizer.is_fixed or type(self.activation_quantizer.quantizer) is not AsymmetricUniformQuantizer
                or self.activation_quantizer.quantizer.n_bits != 8):
            return None
        acc, s32, bias = self.linear_pairs(x, raw=True)
        sp = self.activation_quantizer.quantizer.spec()
        y = ops.quantize_heads_i8(acc.reshape(1, -1, self.out_features), sp, self.out_features // 64, want_values=True, alpha=s32,
                                  bias=bias.detach(), want_indices=False)
        return y.view(*x.shape[:-1], self.out_features)

    def index_gemm_ok(self, like) -> bool:
        """`linear_index` applies: what `pair_gemm_ok` asks of the weights, for an fp32 model on the GPU."""
        return self.pair_gemm_ok(like) and self.bias is not None and self.activation_function is None

    def int8_index_ok(self, rows: int) -> bool:
        """`linear_index` on int8 centred indices applies: frozen 8-bit asymmetric output quantiser, K % 64 == 0, whole 16-row groups."""
        aq = self.activation_quantizer
        return (FUSED_PROJ and self._qa and self.out_features % 64 == 0 and aq.is_fixed and type(aq.quantizer) is AsymmetricUniformQuantizer
                and aq.quantizer.n_bits == 8 and self.in_features % 64 == 0 and rows % 16 == 0 and self.bias is not None
                and self._int8_weights_fit())

    def _int8_weights_fit(self) -> bool:
        """The weight's integers fit a SIGNED byte.  A SymmetricUniformQuantizer whose weights are all >= 0 is unsigned: its integers
        run to 255, exact in fp16 (the fp16-integer form of `linear_index` serves them) but not as int8 - 128 ... 255 would wrap
        (ADVICE r4).  One host read per weight / weight-range version, cached."""
        qz = self.weight_quantizer.quantizer
        sg = getattr(qz, "_signed", None)   # (the symmetric quantiser's signed / unsigned switch and its bit width decide whether the integers run to 255: ADVICE r5)
        key = (self.weight.data_ptr(), self.weight._version, qz._delta.data_ptr(), qz._delta._version, getattr(qz, "n_bits", None),
               None if sg is None else (sg.data_ptr(), sg._version) if torch.is_tensor(sg) else sg)
        hit = self.__dict__.get("_int8_fit_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                iw = qz.to_integer_forward(self.weight.detach())
                hit = (key, bool(((iw >= -128) & (iw <= 127)).all()))
            self.__dict__["_int8_fit_cache"] = hit
        return hit[1]

    def _int8_weights(self, xzero: float):
        """The weight's integers as int8 (N, K), (128 - xzero) * their row sums as int32 (N) and the fp32 weight scale, cached."""
        qz = self.weight_quantizer.quantizer
        key = (self.weight.data_ptr(), self.weight._version, qz._delta.data_ptr(), qz._delta._version, float(xzero))
        hit = self.__dict__.get("_int8_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                iw = qz.to_integer_forward(self.weight.detach())
                if not self._int8_weights_fit():
                    raise ValueError("the weight's integers do not fit int8 (an unsigned grid): use the fp16 integer form (int8_index_ok)")
                iw8 = iw.to(torch.int8).contiguous()
                add = (int(128 - int(xzero)) * iw.to(torch.int64).sum(dim=1)).to(torch.int32).contiguous()
                s32 = float(np.float32(float(qz.scale)))
            hit = (key, iw8, add, s32)
            self.__dict__["_int8_cache"] = hit
        return hit[1], hit[2], hit[3]

    def linear_index(self, rel, xscale: float, out_dtype=torch.float32, xzero: float = 0.0):
        """This projection of x = xscale * rel, `rel` the INTEGERS idx - zp of the producer's 8-bit quantiser as a 16-bit float
        tensor (`ops.attn_fwd_i8(..., fq.ctx_emit_index)`): x W_q^T = (xscale * w_scale) * (rel . Iw), ONE fp16 GEMM of integers
        with exact products and fp32 accumulation - half the operand-pair GEMM and no split pass - then scale, bias and the
        frozen output quantiser in one pass over the accumulator (the reference: a float GEMM of the dequantised values,
        quantized_opt.py:271; the two agree to fp32 rounding of the sum)."""
        K = self.in_features
        shape = (*rel.shape[:-1], self.out_features)
        aq = self.activation_quantizer
        fixed8 = (self._qa and self.out_features % 64 == 0 and aq.is_fixed and type(aq.quantizer) is AsymmetricUniformQuantizer and aq.quantizer.n_bits == 8)
        rel2 = rel.reshape(-1, K)
        if rel2.dtype == torch.int8:
            # the producer's CENTRED indices idx - 128 (`ops.attn_fwd_i8(..., out_dtype=torch.int8)`): both sides int8 on the integer matrix
            # cores, exact int32 sums; the per-column integer (128 - zp) * sum_k Iw turns them into the sums over idx - zp
            iw8, add, s32 = self._int8_weights(xzero)
            self.__dict__["_int8_index_calls"] = self.__dict__.get("_int8_index_calls", 0) + 1  # (tests: which form ran)
            y = ops.proj_quant_values(rel2, iw8, self.bias.detach(), float(np.float32(xscale) * np.float32(s32)), aq.quantizer.spec(), pairs=False, acc_add=add)
            return y.view(shape).to(out_dtype)
        if (FUSED_PROJ and fixed8 and K % 32 == 0 and rel2.shape[0] % 16 == 0 and rel2.dtype == torch.float16 and rel2.stride(1) == 1
                and self.bias is not None):
            # ONE kernel: the GEMM of integers, scale, bias and the output quantiser in its epilogue (`oeh_proj_quant_i8`, values only)
            iw, s32 = self._int_weights()
            y = ops.proj_quant_values(rel2, iw, self.bias.detach(), float(np.float32(xscale) * np.float32(s32)), aq.quantizer.spec(), pairs=False)
            return y.view(shape).to(out_dtype)
        ww, s32 = self._pair_weights()
        acc = torch.mm(rel2, ww[:K], out_dtype=torch.float32)
        alpha = float(np.float32(xscale) * np.float32(s32))
        if fixed8:
            y = ops.quantize_heads_i8(acc.view(1, -1, self.out_features), aq.quantizer.spec(), self.out_features // 64, want_values=True,
                                      alpha=alpha, bias=self.bias.detach(), want_indices=False)
            return y.view(shape).to(out_dtype)
        res = torch.add(self.bias.detach(), acc, alpha=alpha).view(shape).to(out_dtype)
        return aq(res) if self._qa else res

    def forward(self, x, offsets=None):
        weight, bias = self.get_params()
        if self.pair_gemm_ok(x):
            fused = self._pair_gemm_quantised(x)
            if fused is not None:
                return fused
            res = self.linear_pairs(x)
        else:
            res = nn.functional.linear(x.contiguous(), weight.contiguous(), bias=bias)
        if self.activation_function is not None:
            res = self.activation_function(res)
        if self._qa:
            res = self.activation_quantizer(res)
        return res


def quantize_model(model: nn.Module, **quant_params) -> nn.Module:
    """The nn.Linear case of the reference's recursive rewriter (autoquant_utils.py:236-270): the only one the attention
    classes need (query/key/value, q_proj/k_proj/v_proj/out_proj)."""
    if type(model) is nn.Linear:
        q = QuantLinear(model.in_features, model.out_features, bias=model.bias is not None, **quant_params)
        q.weight.data = model.weight.data
        if model.bias is not None:
            q.bias.data = model.bias.data
        return q.to(model.weight.device)
    raise NotImplementedError(f"quantize_model: {type(model).__name__} is outside the attention path")


class QuantizedModel(nn.Module):
    """Convenience switches over every QuantizedModule inside (base_quantized_model.py:18-162)."""

    def _each(self, fn):
        for m in self.modules():
            if isinstance(m, QuantizedModule):
                fn(m)

    def quantized_weights(self):
        self._each(lambda m: m.quantized_weights())

    def full_precision_weights(self):
        self._each(lambda m: m.full_precision_weights())

    def quantized_acts(self):
        self._each(lambda m: m.quantized_acts())

    def full_precision_acts(self):
        self._each(lambda m: m.full_precision_acts())

    def quantized(self):
        self._each(lambda m: m.quantized())

    def full_precision(self):
        self._each(lambda m: m.full_precision())

    def fix_ranges(self):
        _apply_qm(self, lambda m: m.fix_ranges() if m.quantizer.is_initialized else None)

    def estimate_ranges(self):
        _apply_qm(self, lambda m: m.estimate_ranges())

    def estimate_ranges_train(self):
        _apply_qm(self, lambda m: m.estimate_ranges_train() if m.quantizer.is_initialized else None)

    def set_quant_state(self, weight_quant, act_quant):
        (self.quantized_acts if act_quant else self.full_precision_acts)()
        (self.quantized_weights if weight_quant else self.full_precision_weights)()


# ------------------------------------------------------------------------------------------------------------
# the two quantised attention classes
# ------------------------------------------------------------------------------------------------------------

# ------------------------------------------------------------------------------------------------------------
# the frozen-range INT8 layer as a PLAN (round 5; VERDICT r4 weak #9: the eager quantised modules were host-bound - 97 us per BERT-base layer for
# 46 us of GPU work, 132 us per OPT-125m layer for 107).  `_int8_storage_core` is ~80 Python statements of eligibility checks, cache look-ups,
# descriptor filling and allocations around three launches; once the ranges are frozen nothing of that changes from one forward to the next except
# the input's address.  After a successful run the three C calls are kept PREBUILT (ops: `_prepared` forms): the int8 q / k / v^T index buffers and
# (OPT) the int8 context are the plan's own - stream-ordered reuse, keyed by stream -, a forward then allocates only what it returns, patches four or
# five pointers and launches.  A plan is valid for ONE input geometry and ONE state of the module: every tensor whose value was baked in (weights,
# biases, every quantiser's range buffers) is watched by identity + version counter, the quantisation state flags, training mode, forward hooks and
# this file's feature switches are re-checked on every call; anything else rebuilds it through the full path.
# ------------------------------------------------------------------------------------------------------------
I8_PLAN = True


class _I8LayerPlan:
    __slots__ = ("xkey", "flags", "watch", "hookmods", "mods", "stream", "padkey", "proj", "attn", "outp", "bufs", "B", "T", "E", "H", "want_values", "hdtype", "as_index", "keep")

    @staticmethod
    def watch_entries(owner, lins, consumer):
        ents = []
        for m in (*lins, consumer):
            if m is None:
                continue
            ents += [(m._parameters, "weight"), (m._parameters, "bias")]
            for mg in (m.weight_quantizer, m.activation_quantizer):
                ents += [(mg.quantizer._buffers, n_) for n_ in mg.quantizer._buffers]
        for aq in (owner.attn_scores_act_quantizer, owner.attn_probs_act_quantizer, owner.context_act_quantizer):
            qz = aq.activation_quantizer.quantizer
            ents += [(qz._buffers, n_) for n_ in qz._buffers]
        out = []
        for d_, n_ in ents:
            t_ = d_.get(n_)
            out.append((d_, n_, t_, None if t_ is None else t_._version, None if t_ is None else t_.data_ptr()))   # (data_ptr: `p.data = other` keeps identity AND version)
        return out

    @staticmethod
    def state_flags(owner, lins, consumer):
        fl = [owner.training, INT8_STORAGE, INDEX_GEMM, FUSED_PROJ, PAIR_GEMM, I8_PLAN]
        for m in (*lins, consumer):
            if m is not None:
                fl += [m.training, m._qa, m._qw, m.activation_quantizer.state, m.weight_quantizer.state]
        for aq in (owner.attn_scores_act_quantizer, owner.attn_probs_act_quantizer, owner.context_act_quantizer):
            fl += [aq._qa, aq.activation_quantizer.state]
        # the quantisers' Python attributes that are baked into the plan's specs (not registered buffers: the watch list cannot see them)
        for m in (*lins, consumer):
            if m is not None:
                for mg in (m.weight_quantizer, m.activation_quantizer):
                    qz = mg.quantizer
                    fl += [type(qz).__name__, getattr(qz, "n_bits", None), getattr(qz, "eps", None)]
        for aq in (owner.attn_scores_act_quantizer, owner.attn_probs_act_quantizer, owner.context_act_quantizer):
            qz = aq.activation_quantizer.quantizer
            fl += [type(qz).__name__, getattr(qz, "n_bits", None), getattr(qz, "eps", None)]
        return tuple(fl)

    def valid(self, owner, x, lins, consumer, padvec, stream) -> bool:
        if self.xkey != (x.shape, x.stride(), x.dtype, x.device) or self.stream != stream or (x.data_ptr() & 15) or x.device.index != torch.cuda.current_device():
            return False
        mods = (*lins, consumer)
        if len(mods) != len(self.mods) or any(a is not b for a, b in zip(mods, self.mods)):   # (a projection swapped for another module)
            return False
        if self.padkey != (None if padvec is None else (padvec.dtype, padvec.shape, padvec.stride())):
            return False
        if self.flags != _I8LayerPlan.state_flags(owner, lins, consumer):
            return False
        # The plan replays forward-only kernels: when autograd is recording and the input or a watched parameter requires grad, the full
        # path must run - it raises / takes the differentiable route as before (a plan built under no_grad used to be
        # replayed here and returned tensors without grad_fn - gradients dropped silently).  The predicate of ops.grad_recording.
        if torch.is_grad_enabled() and (x.requires_grad or any(t_ is not None and t_.requires_grad for _, _, t_, _, _ in self.watch)):
            return False
        for d_, n_, t_, v_, p_ in self.watch:
            cur = d_.get(n_)
            if cur is not t_ or (t_ is not None and (t_._version != v_ or t_.data_ptr() != p_)):
                return False
        return not has_hooks(*self.hookmods)

    def run(self, x, padvec, stream):
        B, T, E = self.B, self.T, self.E
        fn, args, arr, _ = self.proj
        args[0] = ops._ptr(x)
        yk = yv = None
        if self.want_values:
            yk = torch.empty((B, T, E), dtype=torch.float32, device=x.device)
            yv = torch.empty((B, T, E), dtype=torch.float32, device=x.device)
            arr[1].y, arr[2].y = yk.data_ptr(), yv.data_ptr()
        rc = fn(*args, stream)
        if rc != 0:
            ops._lib.check(rc, "oeh_proj_quant_i8 (plan)")
        afn, aargs, (adesc, _fqd) = self.attn
        if padvec is not None:
            adesc.key_pad_mask = padvec.data_ptr()
        if self.as_index:   # OPT: int8 context (the plan's buffer) -> out_proj on the integer matrix cores -> a fresh fp32 output
            rc = afn(*aargs, stream)
            if rc != 0:
                ops._lib.check(rc, "oeh_attn_fwd (plan)")
            ofn, oargs, seg, _ = self.outp
            y = torch.empty((B * T, E), dtype=torch.float32, device=x.device)
            seg[0].y = y.data_ptr()
            rc = ofn(*oargs, stream)
            if rc != 0:
                ops._lib.check(rc, "oeh_proj_quant_i8 (plan, out_proj)")
            return y.view(B, T, E).to(self.hdtype), (yk, yv), True
        out = torch.empty((B, T, E), dtype=self.hdtype, device=x.device)   # (B, Sq, H, D)-contiguous: the merged context
        aargs[4] = ops._ptr(out)
        rc = afn(*aargs, stream)
        if rc != 0:
            ops._lib.check(rc, "oeh_attn_fwd (plan)")
        return out, (yk, yv), False


class _QuantAttnBase(GateBookkeeping, QuantizedModel):
    def _init_common(self, org_model, quant_params):
        self.attn_scores_act_quantizer = QuantizedActivation(**quant_params)
        self.attn_probs_act_quantizer = QuantizedActivation(**quant_params)
        self.context_act_quantizer = QuantizedActivation(**quant_params)
        self.softmax_fn = org_model.softmax_fn
        self.attn_gate_type = org_model.attn_gate_type
        self.attn_gate_init = org_model.attn_gate_init
        self.attn_gate_mlp = org_model.attn_gate_mlp
        self.attn_gate_mlp2 = org_model.attn_gate_mlp2
        self.attn_gate_linear_all_features = org_model.attn_gate_linear_all_features
        self.alpha = org_model.alpha  # the gate is not quantised (quantized_bert.py:256)
        self.gate_fn = org_model.gate_fn
        self.pooling_fn = org_model.pooling_fn
        self.last_gate_avg_prob = None
        self.last_gate_all_probs = None

    def _fq(self, ctx_before_gate: bool) -> Optional[AttnFakeQuant]:
        trio = (self.attn_scores_act_quantizer, self.attn_probs_act_quantizer, self.context_act_quantizer)
        if not all(t.fusable for t in trio):
            return None  # still estimating ranges: the tensors must be materialised
        return AttnFakeQuant(trio[0].fixed_spec(), trio[1].fixed_spec(), trio[2].fixed_spec(), ctx_before_gate=ctx_before_gate)



    def _calibrate_fused(self, q, k, v, *, scale=1.0, scale_div=0.0, attention_mask=None, clamp_min=False, detect_causal=False):
        """Qstates.estimate_ranges without the (B,H,Sq,Sk) tensors (VERDICT r2 missing #3; range_estimators.py:83-106 as driven by
        transformers_language/utils.py:50-71): the score quantiser's percentile range, then - with the scores quantised on that
        fresh range - the probability quantiser's, from `ops.attn_calibrate`, which recomputes the values tile by tile for every pass
        of the exact selection; then the context with both quantisers applied.  The recomputed values are fp32 from the stored
        (16-bit or fp32) q / k - what the fused eval kernels quantise; the observable path hands the estimators scores and
        probabilities ROUNDED to a 16-bit storage dtype, so for fp16 / bf16 modules the two paths' ranges differ by up to one storage
        ulp of the percentile value (2^-11 / 2^-8 relative; the context quantiser, which sees the rounded probabilities' product, up to 1 %
        in bf16 - tests/test_modules_gpu.py::test_fused_calibration_vs_observable_16bit).
        Returns the context (B,H,Sq,D) in q's dtype, or None when this does not apply (a quantiser that is off or already fixed, min-max estimators, host-side estimator state,
        an unregistered softmax, ...): the caller then runs the observable path."""
        if not FUSED_CALIBRATION or not q.is_cuda or q.dtype not in (torch.float16, torch.bfloat16, torch.float32) or q.shape[-1] not in (32, 64, 128):
            return None
        spec = spec_of(self.softmax_fn)
        mgrs = [m.activation_quantizer for m in (self.attn_scores_act_quantizer, self.attn_probs_act_quantizer)]
        if spec is None or not (self.attn_scores_act_quantizer._qa and self.attn_probs_act_quantizer._qa):
            return None
        for mg in mgrs:
            est = mg.range_estimator
            if (mg.is_fixed or type(mg.quantizer) is not AsymmetricUniformQuantizer or not isinstance(est, RunningMinMaxEstimator) or not est.percentile
                    or est.momentum is None or (est.current_xmin is not None and est.device_state is None)):
                return None
        # forward hooks on the quantiser modules (attach_act_hooks registers one on every named module) see the score / probability
        # tensors only on the observable path
        hooked = [m for aq in (self.attn_scores_act_quantizer, self.attn_probs_act_quantizer) for m in aq.modules()]
        if has_hooks(*hooked):
            return None
        B, H, Sq, D = q.shape
        Sk = k.shape[2]
        if B * H * Sq * Sk >= 2 ** 32:
            return None
        pad, full = split_mask(attention_mask, B, Sq, Sk)
        causal = False
        if full is not None and detect_causal and Sq <= Sk:
            causal, padvec = classify_causal(full)
            if causal:
                full, pad = None, padvec
        mdt = attention_mask.dtype if attention_mask is not None and attention_mask.is_floating_point() else q.dtype
        kw = dict(softmax=spec, scale=scale, scale_div=scale_div, key_pad_mask=pad, full_mask=full, causal=causal, clamp_min=clamp_min,
                  mask_min=float(torch.finfo(mdt).min))
        states = []
        for which, mg in enumerate(mgrs):
            est, qz = mg.range_estimator, mg.quantizer
            first = est.device_state is None or est.device_state.device != q.device
            if first:
                est.device_state = torch.empty(2, dtype=torch.float64, device=q.device)
            ops.attn_calibrate(q, k, None, which, scores_range=states[0] if which == 1 else None, n_bits=qz.n_bits, eps=qz.eps,
                               q_lo=100 - est.percentile, q_hi=est.percentile, momentum=est.momentum, first=first, state=est.device_state, **kw)
            est.current_xmin, est.current_xmax = est.device_state[0].clone(), est.device_state[1].clone()
            mg.set_quant_range(est.current_xmin, est.current_xmax)  # the buffers the state dict / fix_ranges read (device arithmetic, no sync)
            states.append(est.device_state)
        qz = mgrs[0].quantizer
        ctx = ops.attn_calibrate(q, k, v, ops.CALIB_CONTEXT, scores_range=states[0], probs_range=states[1], n_bits=qz.n_bits, eps=qz.eps, **kw)
        self.__dict__["_fused_calib_calls"] = self.__dict__.get("_fused_calib_calls", 0) + 1  # (tests: which path ran)
        return ctx.to(q.dtype)

    def _qkv_pair_weights(self, lins):
        """The three projections' operand-pair weights (`QuantLinear._pair_weights`) side by side, (2K, 3E) fp16, and their fp32
        weight scales; rebuilt when any of the three was (a changed weight or weight range rebuilds that projection's own cache)."""
        parts = [m._pair_weights() for m in lins]
        hit = self.__dict__.get("_qkv_pair_cache")
        if hit is None or any(a is not b for a, (b, _) in zip(hit[0], parts)):
            hit = (tuple(p[0] for p in parts), torch.cat([p[0] for p in parts], dim=1).contiguous())
            self.__dict__["_qkv_pair_cache"] = hit
        return hit[1], [p[1] for p in parts]

    def _qkv_int_weights(self, lins):
        """The three projections' integer weights one after the other, (3E, K) fp16, their biases (3E) fp32 and their fp32 weight
        scales (`ops.proj_quant_i8`); rebuilt when a weight, a weight range or a bias changed."""
        parts = [m._int_weights() for m in lins]
        bkey = tuple((m.bias.data_ptr(), m.bias._version) for m in lins)
        hit = self.__dict__.get("_qkv_int_cache")
        if hit is None or hit[1] != bkey or any(a is not b for a, (b, _) in zip(hit[0], parts)):
            with torch.no_grad():
                hit = (tuple(p[0] for p in parts), bkey, torch.cat([p[0] for p in parts], dim=0).contiguous(),
                       torch.cat([m.bias.detach().float() for m in lins]).contiguous())
            self.__dict__["_qkv_int_cache"] = hit
        return hit[2], hit[3], [p[1] for p in parts]

    def _int8_storage_core(self, hidden_states, lins, H, head_dim, *, scale, scale_div, causal, padvec, mask_min, gate, fq, want_values,
                           consumer=None):
        """the q/k/v projections are QuantLinear - their outputs ARE 8-bit indices on calibrated grids
        (hijacker.py:78-127; quantized_opt.py:67-75, quantized_bert.py:236-238) - so the attention core takes the indices
        themselves and runs both products on the integer matrix cores (`ops.attn_fwd_i8`, include/oeh.h dtype OEH_I8).  Applies
        when the three output quantisers are 8-bit with frozen ranges, head_dim = 64, the softmax is unclipped or clipped with gamma <= 0 and the mask is
        none / causal / a key-padding vector of 0 / finfo.min entries (`padvec`, vouched for by the caller); returns the
        merged context (B, T, E) and the (k, v) float values (`want_values`: a decoder's cache), or None -> the caller runs the
        fake-quant path on floats.  `consumer`: the QuantLinear that takes the context next (OPT's out_proj); when the context
        quantiser is the core's last op the core hands it the quantiser's integers (`ctx_emit_index`) and the returned tensor is
        the consumer's OUTPUT (`QuantLinear.linear_index`), marked by the third element of the result."""
        if not INT8_STORAGE or not hidden_states.is_cuda or head_dim != 64:
            return None
        ckey = (H, float(scale), float(scale_div), bool(causal), float(mask_min), bool(want_values), fq.ctx_before_gate, id(consumer), self.softmax_fn)
        if I8_PLAN and gate is None:
            plan = self.__dict__.get("_oeh_i8_plan")
            if plan is not None and plan[0] == ckey and not torch.cuda.is_current_stream_capturing():
                # (under graph capture the full path runs: its intermediates then live in the graph's own pool, not in buffers a later plan rebuild would free)
                stream = ops._stream()
                if plan[1].valid(self, hidden_states, lins, consumer, padvec, stream.value):
                    d_ = self.__dict__   # (the counters tests / tools read to see which kernels ran)
                    d_["_i8_plan_runs"] = d_.get("_i8_plan_runs", 0) + 1
                    d_["_i8_calls"] = d_.get("_i8_calls", 0) + 1
                    d_["_fused_proj_calls"] = d_.get("_fused_proj_calls", 0) + 1
                    if plan[1].as_index:
                        d_["_index_gemm_calls"] = d_.get("_index_gemm_calls", 0) + 1
                        consumer.__dict__["_int8_index_calls"] = consumer.__dict__.get("_int8_index_calls", 0) + 1
                    return plan[1].run(hidden_states, padvec, stream)
        # this path reads the QuantLinear weights and quantiser grids directly: the forwards of the projections, of the consumer and
        # of the three activation quantisers never run, so forward hooks on any of them (attach_act_hooks registers one on every
        # named module) would be bypassed silently - the module path then (as bert_attention.py / opt_attention.py do)
        # (the flattened module list is remembered per set of top modules: walking `.modules()` of seven modules was a sixth of this
        # path's host time, and the eager forward is host-bound; hooks registered later on any of them are still seen - the hook
        # dictionaries are read on every call - only a submodule swapped in afterwards under the same top module would not be)
        tops = (*lins, consumer, self.attn_scores_act_quantizer, self.attn_probs_act_quantizer, self.context_act_quantizer)
        bkey = tuple(id(t) for t in tops)
        bhit = self.__dict__.get("_bypassed_cache")
        if bhit is None or bhit[0] != bkey:
            bhit = (bkey, [m for top in tops if top is not None for m in top.modules()])
            self.__dict__["_bypassed_cache"] = bhit
        if has_hooks(*bhit[1]):
            return None
        spec = spec_of(self.softmax_fn)
        bsz, tgt_len, _ = hidden_states.shape
        if spec is None or (spec.clip and spec.gamma > 0.0) or tgt_len % 16 != 0 or tgt_len > 512 or fq.probs is None or fq.probs.qmax != 255.0 or fq.scores is None:
            return None
        if not all(isinstance(m, QuantLinear) for m in lins):
            return None
        aqs = [m.activation_quantizer for m in lins]
        qzs = [a.quantizer for a in aqs]
        if not all(m._qa and a.is_fixed and z.n_bits == 8 and m.activation_function is None for m, a, z in zip(lins, aqs, qzs)):
            return None
        E = H * head_dim
        outs, grids = [], []
        pairs, acc3 = None, None
        fused_consts = None
        all_pairs = all(m.pair_gemm_ok(hidden_states) and m.bias is not None for m in lins)
        K_in = hidden_states.shape[-1]
        if (FUSED_PROJ and all_pairs and K_in % 32 == 0 and all(m.in_features == K_in and m.out_features == E for m in lins)
                and all(type(z) is AsymmetricUniformQuantizer for z in qzs)):
            # fp32 model, ONE kernel: the pair GEMM against the three integer weight matrices with weight scale, bias and the three output
            # quantisers in its epilogue (`oeh_proj_quant_i8`) - the (B*T, 3E) accumulator never reaches memory
            # (the fp32 activations go in as they are: the (hi, lo) operand split happens when a wave reads its fragments - the same
            # values as `ops.split_pairs` would write, without the pass)
            x2 = hidden_states.reshape(-1, K_in)
            if x2.stride(1) != 1 or (x2.stride(0) * 4) % 16 != 0:
                x2 = x2.contiguous()
            if x2.data_ptr() % 16 != 0:   # (a view at an odd storage offset: the kernel's 16-byte row loads need an aligned copy)
                x2 = x2.clone(memory_format=torch.contiguous_format)
            w3, b3, scales3 = self._qkv_int_weights(lins)
            specs = [z.spec() for z in qzs]
            try:
                outs = ops.proj_quant_i8(x2, w3, b3, bsz, tgt_len, [(scales3[n_], specs[n_], n_ == 2, n_ > 0 and want_values) for n_ in range(3)], pairs=True)
                grids = [ops.QuantGrid.of(sp) for sp in specs]
                fused_consts = (w3, b3, scales3, specs) if x2.data_ptr() == hidden_states.data_ptr() else None   # (plan: only for an input used as it is)
                self.__dict__["_fused_proj_calls"] = self.__dict__.get("_fused_proj_calls", 0) + 1  # (tests: which path ran)
            except _OehError as e:  # a size / alignment the fused kernel refuses (32-bit lane offsets ...): the library pair GEMM below
                if e.code not in (-95, -14):
                    raise
                outs = []
        if outs:
            pass
        elif all_pairs:
            # fp32 model: the input as fp16 operand pairs, split once, and ONE fp16 GEMM against the three integer weight
            # matrices side by side; each projection's weight scale and bias are folded into its quantiser pass
            pairs = ops.split_pairs(hidden_states.reshape(-1, hidden_states.shape[-1]))
            ww3, scales3 = self._qkv_pair_weights(lins)
            acc3 = torch.mm(pairs, ww3, out_dtype=torch.float32).view(bsz, tgt_len, 3 * E)
        for n_, m in enumerate(() if outs else lins):  # (not fused:) GEMM, then ONE kernel: centred int8 indices in the core's layout (v transposed) [+ the cache's floats]
            alpha, qbias = 1.0, None
            if acc3 is not None:
                res, alpha, qbias = acc3[..., n_ * E:(n_ + 1) * E], scales3[n_], m.bias.detach()
            elif m.pair_gemm_ok(hidden_states) and m.bias is not None:
                if pairs is None:
                    pairs = ops.split_pairs(hidden_states.reshape(-1, hidden_states.shape[-1]))
                res, alpha, qbias = m.linear_pairs(hidden_states, pairs, raw=True)
                qbias = qbias.detach()
            else:
                w, b = m.get_params()
                res = nn.functional.linear(hidden_states.contiguous(), w.contiguous(), bias=b)
            sp = m.activation_quantizer.quantizer.spec()
            outs.append(ops.quantize_heads_i8(res, sp, H, transpose=(n_ == 2), want_values=(n_ > 0 and want_values), alpha=alpha, bias=qbias))
            grids.append(ops.QuantGrid.of(sp))
        qc = outs[0]
        kc, yk = outs[1] if want_values else (outs[1], None)
        vt, yv = outs[2] if want_values else (outs[2], None)
        as_index = (INDEX_GEMM and consumer is not None and fq.ctx is not None and fq.ctx.qmax == 255.0 and (gate is None or not fq.ctx_before_gate)
                    and isinstance(consumer, QuantLinear) and consumer.in_features == E and consumer.index_gemm_ok(hidden_states))
        fq_call = dataclasses.replace(fq, ctx_emit_index=True) if as_index else fq
        # ... as int8 centred indices when the consumer takes them on the integer matrix cores (K % 64 == 0, a whole zero point), else as
        # the integers idx - zp in fp16
        as_int8 = as_index and consumer.int8_index_ok(bsz * tgt_len) and float(fq.ctx.zero_point) == float(int(fq.ctx.zero_point))
        try:
            out = ops.attn_fwd_i8(qc, kc, vt, grids, fq=fq_call, out_dtype=(torch.int8 if as_int8 else torch.float16) if as_index else hidden_states.dtype, softmax=spec, scale=scale,
                                  scale_div=scale_div, causal=causal, clamp_min=causal or padvec is not None, mask_min=mask_min, gate=gate,
                                  key_pad_mask=padvec)
        except _OehError as e:
            if e.code != -95:
                raise
            return None
        self.__dict__["_i8_calls"] = self.__dict__.get("_i8_calls", 0) + 1  # (tests: which core ran)
        merged = out.permute(0, 2, 1, 3).reshape(bsz, tgt_len, E)
        if as_index:
            self.__dict__["_index_gemm_calls"] = self.__dict__.get("_index_gemm_calls", 0) + 1
            result = consumer.linear_index(merged, fq.ctx.scale, out_dtype=hidden_states.dtype, xzero=fq.ctx.zero_point), (yk, yv), True
        else:
            result = merged, (yk, yv), False
        if (I8_PLAN and gate is None and fused_consts is not None and (as_int8 if as_index else consumer is None) and hidden_states.is_contiguous()
                and hidden_states.dtype == torch.float32 and not torch.cuda.is_current_stream_capturing() and hidden_states.device.index == torch.cuda.current_device()
                and (padvec is None or (padvec.dtype in (torch.float16, torch.float32) and padvec.shape == (bsz, tgt_len) and padvec.stride(1) == 1))):
            try:
                self.__dict__["_oeh_i8_plan"] = (ckey, self._build_i8_plan(hidden_states, lins, consumer, H, E, fused_consts, grids, fq_call, spec, scale, scale_div, causal,
                                                                           padvec, mask_min, want_values, as_index, bhit[1]))
            except _OehError as e:   # (a geometry one of the prepared calls refuses: every forward keeps taking the full path)
                self.__dict__["_oeh_i8_plan"] = None
                self.__dict__["_oeh_i8_plan_error"] = str(e)
        return result

    def _build_i8_plan(self, x, lins, consumer, H, E, fused_consts, grids, fq_call, spec, scale, scale_div, causal, padvec, mask_min, want_values, as_index, hookmods):
        """The three launches of the run that has just succeeded, prebuilt (see _I8LayerPlan): same operands, same descriptors - only where the
        outputs go differs (the int8 intermediates into the plan's own buffers)."""
        B, T, K = x.shape
        dev = x.device
        w3, b3, scales3, specs = fused_consts
        pl = _I8LayerPlan()
        pl.B, pl.T, pl.E, pl.H, pl.want_values, pl.hdtype, pl.as_index = B, T, E, H, bool(want_values), x.dtype, bool(as_index)
        qi, ki = torch.empty((B, T, E), dtype=torch.int8, device=dev), torch.empty((B, T, E), dtype=torch.int8, device=dev)
        vt = torch.empty((B, H, 64, T), dtype=torch.int8, device=dev)
        ytmp = torch.empty((B, T, E), dtype=torch.float32, device=dev) if want_values else None   # (pointer patched per call)
        box = []
        ops.proj_quant_i8(x.view(B * T, K), w3, b3, B, T, [(scales3[n_], specs[n_], n_ == 2, n_ > 0 and want_values) for n_ in range(3)], pairs=True,
                          _outs=[(qi, None), (ki, ytmp), (vt, ytmp)], _prepared=box)
        pl.proj = (box[0], box[1], box[2], box[3])
        heads = lambda t: t.view(B, T, H, 64).permute(0, 2, 1, 3)  # noqa: E731
        if as_index:
            ctx8 = torch.empty((B, T, H, 64), dtype=torch.int8, device=dev)
            out_t, odt = ctx8.permute(0, 2, 1, 3), torch.int8
        else:
            ctx8 = None
            out_t, odt = torch.empty((B, T, H, 64), dtype=x.dtype, device=dev).permute(0, 2, 1, 3), x.dtype   # (pointer patched per call)
        box2 = []
        ops.attn_fwd_i8(heads(qi), heads(ki), vt, grids, fq=fq_call, out_dtype=odt, softmax=spec, scale=scale, scale_div=scale_div, causal=causal,
                        clamp_min=causal or padvec is not None, mask_min=mask_min, gate=None, key_pad_mask=padvec, out=out_t, _prepared=box2)
        pl.attn = (box2[0], list(box2[1]), (box2[2][0], box2[2][1]))
        pl.outp = None
        keep = [box2[2]]
        if as_index:
            iw8, add, s32 = consumer._int8_weights(fq_call.ctx.zero_point)
            box3 = []
            ops.proj_quant_values(ctx8.view(B * T, E), iw8, consumer.bias.detach(), float(np.float32(fq_call.ctx.scale) * np.float32(s32)),
                                  consumer.activation_quantizer.quantizer.spec(), pairs=False, acc_add=add, _prepared=box3)
            pl.outp = (box3[0], box3[1], box3[2], box3[3])
        pl.bufs = (qi, ki, vt, ctx8)
        pl.keep = keep
        pl.xkey = (x.shape, x.stride(), x.dtype, x.device)
        pl.padkey = None if padvec is None else (padvec.dtype, padvec.shape, padvec.stride())
        pl.stream = ops._stream().value
        pl.flags = _I8LayerPlan.state_flags(self, lins, consumer)
        pl.watch = _I8LayerPlan.watch_entries(self, lins, consumer)
        pl.hookmods = hookmods
        pl.mods = (*lins, consumer)
        return pl


class QuantizedBertSelfAttentionWithExtras(_QuantAttnBase):
    """quantized_bert.py:221-440: fake-quant on scores (after /sqrt(d), before the mask), on probs, and on the context
    AFTER gating and head merge; the gate is applied without gate_scaling_factor (:422)."""

    def __init__(self, org_model, **quant_params):
        super().__init__()
        self.num_attention_heads = org_model.num_attention_heads
        self.attention_head_size = org_model.attention_head_size
        self.all_head_size = org_model.all_head_size
        self.position_embedding_type = getattr(org_model, "position_embedding_type", None)
        self.is_decoder = org_model.is_decoder
        self.query = quantize_model(org_model.query, **quant_params)
        self.key = quantize_model(org_model.key, **quant_params)
        self.value = quantize_model(org_model.value, **quant_params)
        self.dropout = org_model.dropout
        self._init_common(org_model, quant_params)

    def transpose_for_scores(self, x):
        return x.view(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_value=None, output_attentions=False):
        if self.position_embedding_type in ("relative_key", "relative_key_query"):
            raise NotImplementedError("relative position embeddings are not on the quantised MI355X path")
        if (encoder_hidden_states is None and past_key_value is None and head_mask is None and not output_attentions
                and not (self.training and self.dropout.p > 0.0)):
            # self-attention in inference: query / key / value are QuantLinear (quantized_bert.py:236-238) - their outputs are 8-bit
            # indices, and the whole core (scores / sqrt(d) -> fq -> + mask -> softmax -> fq -> P V -> gate -> fq) runs on them
            fq8 = self._fq(ctx_before_gate=False)
            pad8, ok = None, fq8 is not None
            if ok and attention_mask is not None:
                B_, T_ = hidden_states.shape[0], hidden_states.shape[1]
                pad8, full8 = split_mask(attention_mask, B_, T_, T_)
                ok = full8 is None and pad_is_boolean(attention_mask)
            if ok:
                gate8 = GateState.evaluate(self, hidden_states, self.num_attention_heads)
                mdt = attention_mask.dtype if attention_mask is not None and attention_mask.is_floating_point() else hidden_states.dtype
                done = self._int8_storage_core(hidden_states, (self.query, self.key, self.value), self.num_attention_heads, self.attention_head_size,
                                               scale=1.0, scale_div=float(np.sqrt(self.attention_head_size)), causal=False, padvec=pad8,
                                               mask_min=float(torch.finfo(mdt).min), gate=gate8, fq=fq8, want_values=self.is_decoder)
                if done is not None:
                    context, (yk, yv), _ = done
                    outputs = (context,)
                    if self.is_decoder:
                        outputs = outputs + ((self.transpose_for_scores(yk), self.transpose_for_scores(yv)),)
                    return outputs
        q = self.transpose_for_scores(self.query(hidden_states))
        src = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        if encoder_hidden_states is not None:
            attention_mask = encoder_attention_mask
        if encoder_hidden_states is not None and past_key_value is not None:
            k, v = past_key_value[0], past_key_value[1]
        else:
            k, v = self.transpose_for_scores(self.key(src)), self.transpose_for_scores(self.value(src))
            if encoder_hidden_states is None and past_key_value is not None:
                k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        new_past = (k, v) if self.is_decoder else None
        gate = GateState.evaluate(self, hidden_states, self.num_attention_heads)
        div = float(np.sqrt(self.attention_head_size))
        fq = self._fq(ctx_before_gate=False)
        fusable = (fq is not None and spec_of(self.softmax_fn) is not None and head_mask is None and not output_attentions
                   and not (self.training and self.dropout.p > 0.0))
        probs = None
        if fusable:
            context = attention_core(q, k, v, softmax_fn=self.softmax_fn, scale_div=div, attention_mask=attention_mask, gate=gate, fq=fq)
        else:
            ctx = None
            if fq is None and head_mask is None and not output_attentions and not (self.training and self.dropout.p > 0.0):
                ctx = self._calibrate_fused(q, k, v, scale_div=div, attention_mask=attention_mask)  # ranges without the (B,H,S,S) tensors
            if ctx is None:
                ctx, _, probs = unfused_core(q, k, v, softmax_fn=self.softmax_fn, scale_div=div, attention_mask=attention_mask,
                                             dropout=self.dropout, head_mask=head_mask, fq_scores=self.attn_scores_act_quantizer,
                                             fq_probs=self.attn_probs_act_quantizer)
            if gate is not None:
                ctx = ctx * gate.to(ctx.dtype)
            context = ctx.permute(0, 2, 1, 3).contiguous().view(ctx.shape[0], ctx.shape[2], self.all_head_size)
            context = self.context_act_quantizer(context)
        outputs = (context, probs) if output_attentions else (context,)
        if self.is_decoder:
            outputs = outputs + (new_past,)
        return outputs


class QuantizedOPTAttentionWithExtras(_QuantAttnBase):
    """quantized_opt.py:54-274: fake-quant on the bmm output (before mask), on probs, and on P@V BEFORE gating; gate without
    scaling (:257); q/k/v/out projections are QuantLinear."""

    def __init__(self, org_model, **quant_params):
        super().__init__()
        self.embed_dim = org_model.embed_dim
        self.num_heads = org_model.num_heads
        self.dropout = org_model.dropout
        self.head_dim = org_model.head_dim
        self.scaling = org_model.scaling
        self.is_decoder = org_model.is_decoder
        self.k_proj = quantize_model(org_model.k_proj, **quant_params)
        self.v_proj = quantize_model(org_model.v_proj, **quant_params)
        self.q_proj = quantize_model(org_model.q_proj, **quant_params)
        self.out_proj = quantize_model(org_model.out_proj, **quant_params)
        self._init_common(org_model, quant_params)

    def _heads(self, t, bsz):
        return t.view(bsz, -1, self.num_heads, self.head_dim).permute(0, 2, 1, 3)

    def _shape(self, tensor, seq_len, bsz):
        return tensor.view(bsz, seq_len, self.num_heads, self.head_dim).transpose(1, 2).contiguous()

    def forward(self, hidden_states, key_value_states=None, past_key_value=None, attention_mask=None, layer_head_mask=None,
                output_attentions=False):
        bsz, tgt_len, _ = hidden_states.size()
        if key_value_states is None and past_key_value is None and layer_head_mask is None and not output_attentions and not self.training:
            fq8 = self._fq(ctx_before_gate=True)
            if fq8 is not None:
                if attention_mask is not None and attention_mask.size() != (bsz, 1, tgt_len, tgt_len):
                    raise ValueError(f"Attention mask should be of size {(bsz, 1, tgt_len, tgt_len)}, but is {attention_mask.size()}")
                ok, causal8, pad8 = True, False, None
                if attention_mask is not None:  # HF's decoder mask: causal, plus finfo.min columns for padded keys (checked once per tensor)
                    causal8, pad8 = classify_causal(attention_mask)
                    ok = causal8
                if ok:
                    gate8 = GateState.evaluate(self, hidden_states, self.num_heads)
                    mdt = attention_mask.dtype if attention_mask is not None and attention_mask.is_floating_point() else hidden_states.dtype
                    done = self._int8_storage_core(hidden_states, (self.q_proj, self.k_proj, self.v_proj), self.num_heads, self.head_dim,
                                                   scale=self.scaling, scale_div=0.0, causal=causal8, padvec=pad8, mask_min=float(torch.finfo(mdt).min),
                                                   gate=gate8, fq=fq8, want_values=self.is_decoder, consumer=self.out_proj)
                    if done is not None:
                        merged, (yk, yv), projected = done
                        kv = (self._heads(yk, bsz), self._heads(yv, bsz)) if self.is_decoder else past_key_value
                        return (merged if projected else self.out_proj(merged)), None, kv
        q = self._heads(self.q_proj(hidden_states) * self.scaling, bsz)
        if key_value_states is not None and past_key_value is not None:
            k, v = past_key_value[0], past_key_value[1]
        else:
            src = hidden_states if key_value_states is None else key_value_states
            k, v = self._heads(self.k_proj(src), bsz), self._heads(self.v_proj(src), bsz)
            if key_value_states is None and past_key_value is not None:
                k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        new_past = (k, v) if self.is_decoder else past_key_value
        src_len = k.size(2)
        if attention_mask is not None and attention_mask.size() != (bsz, 1, tgt_len, src_len):
            raise ValueError(f"Attention mask should be of size {(bsz, 1, tgt_len, src_len)}, but is {attention_mask.size()}")
        gate = GateState.evaluate(self, hidden_states, self.num_heads)
        fq = self._fq(ctx_before_gate=True)
        fusable = (fq is not None and spec_of(self.softmax_fn) is not None and layer_head_mask is None and not output_attentions
                   and not (self.training and self.dropout > 0.0))
        weights = None
        if fusable:
            merged = attention_core(q, k, v, softmax_fn=self.softmax_fn, attention_mask=attention_mask, clamp_min=attention_mask is not None,
                                    detect_causal=True, gate=gate, fq=fq)
        else:
            ctx, used = None, None
            if fq is None and layer_head_mask is None and not output_attentions and not (self.training and self.dropout > 0.0):
                ctx = self._calibrate_fused(q, k, v, attention_mask=attention_mask, clamp_min=attention_mask is not None, detect_causal=True)
            if ctx is None:
                hm = None if layer_head_mask is None else layer_head_mask.view(1, -1, 1, 1)
                drop = (lambda p: nn.functional.dropout(p, p=self.dropout, training=self.training))
                ctx, _, used = unfused_core(q, k, v, softmax_fn=self.softmax_fn, attention_mask=attention_mask, clamp_min=attention_mask is not None,
                                            dropout=drop, head_mask=hm, fq_scores=self.attn_scores_act_quantizer,
                                            fq_probs=self.attn_probs_act_quantizer)
            weights = used if output_attentions else None
            ctx = self.context_act_quantizer(ctx)
            if gate is not None:
                ctx = ctx * gate.to(ctx.dtype)
            merged = ctx.transpose(1, 2).reshape(bsz, tgt_len, self.embed_dim)
        return self.out_proj(merged), weights, new_past


# estimate_ranges state: the score and probability quantisers' percentile ranges from `ops.attn_calibrate` (the (B,H,Sq,Sk)
# tensors are recomputed tile by tile inside the kernel, never stored); False: the observable path materialises them.
FUSED_CALIBRATION = True
# OPT's out_proj on the context quantiser's INTEGERS (ops ... ctx_emit_index + QuantLinear.linear_index): one 16-bit GEMM of
# integers instead of the split pass + the operand-pair GEMM.  False: the core writes float values, out_proj runs as any QuantLinear.
INDEX_GEMM = True
# The q / k / v projections of the INT8-storage path as ONE GEMM with the output quantisers in its epilogue (ops.proj_quant_i8;
# include/oeh.h: oeh_proj_quant_i8).  False: the library pair GEMM + three `oeh_quantize_heads_i8` passes (tests compare the two).
FUSED_PROJ = True
# The INT8-storage attention core (integer matrix cores) is used by QuantizedOPTAttentionWithExtras whenever it applies;
# False: always the fake-quant kernels on float values (tests compare the two).
INT8_STORAGE = True
# QuantLinear on fp32 inputs with quantised (8-bit integer x scale) weights: one fp16 GEMM on operand pairs instead of the fp32
# library GEMM (ops.split_pairs; include/oeh.h: oeh_split_pairs).  False: torch's fp32 linear (tests compare the two).
PAIR_GEMM = True


# ------------------------------------------------------------------------------------------------------------
# config helpers (flag names / defaults of transformers_language/quant_configs.py:7-33, utils.py:27-47)
# ------------------------------------------------------------------------------------------------------------
class DotDict(dict):
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __getattr__(self, key):
        if key in self:
            return self[key]
        raise AttributeError(f"DotDict instance has no key '{key}' ({self.keys()})")


def get_quant_config() -> DotDict:
    cfg = DotDict()
    cfg.act_quant = DotDict(cross_entropy_layer=None, num_batches=16, options={}, quant_method=RangeEstimators.running_minmax, std_dev=None)
    cfg.quant = DotDict(act_quant=True, n_bits=8, n_bits_act=8, num_candidates=None, per_channel=False, percentile=None, quant_setup="all",
                        qmethod=QMethods.symmetric_uniform, qmethod_act=QMethods.asymmetric_uniform, weight_quant=True,
                        weight_quant_method=RangeEstimators.current_minmax)
    return cfg


def val_qparams(config) -> dict:
    return {
        "method": config.quant.qmethod.cls,
        "n_bits": config.quant.n_bits,
        "n_bits_act": config.quant.n_bits_act,
        "act_method": config.quant.qmethod_act.cls,
        "per_channel_weights": config.quant.per_channel,
        "percentile": config.quant.percentile,
        "quant_setup": config.quant.quant_setup,
        "weight_range_method": config.quant.weight_quant_method.cls,
        "weight_range_options": {},
        "act_range_method": config.act_quant.quant_method.cls,
        "act_range_options": config.act_quant.options,
    }
