"""outeffhop_amd - MI355X (gfx950) implementation of OutEffHop's modified-softmax attention hot path.

Python host surface mirroring the reference's plugin interface:
    SOFTMAX_MAPPING, AttentionGateType, BertSelfAttentionWithExtras, OPTAttentionWithExtras,
    ViTSelfAttentionWithExtras, Association / Hopfield / HopfieldPooling, QuantizedActivation,
    Quantized{Bert,OPT}...AttentionWithExtras
over the C-ABI library `lib/liboeh_hip.so` (include/oeh.h).  There is no CPU implementation in this package:
every op raises if the HIP library is missing or a tensor is not on a GPU.
"""
from . import _lib, ops  # noqa: F401
from .attention import AttentionGateType, logit  # noqa: F401
from .bert_attention import BertSelfAttentionWithExtras  # noqa: F401
from .hopfield import Association, Hopfield, HopfieldPooling  # noqa: F401
from .opt_attention import OPTAttentionWithExtras  # noqa: F401
from .ops import AttnFakeQuant, FakeQuantSpec, SoftmaxSpec, attn_fwd, fake_quant, softmax_rows  # noqa: F401
from .quantization import (  # noqa: F401
    AsymmetricUniformQuantizer,
    QMethods,
    QuantizationManager,
    QuantizedActivation,
    QuantizedBertSelfAttentionWithExtras,
    QuantizedOPTAttentionWithExtras,
    QuantLinear,
    RangeEstimators,
    RunningMinMaxEstimator,
    get_quant_config,
    val_qparams,
)
from .softmax import (SOFTMAX_MAPPING, ClipSoftmax, ClipSoftmax_1, Softmax_1, clipped_softmax, clipped_softmax1, clipped_softmax_1,  # noqa: F401
                      make_clipped_softmax, make_clipped_softmax1, softmax_1, spec_of)
from .sparse_activations import EntmaxAlpha, Sparsemax, entmax15, entmax_bisect, sparsemax  # noqa: F401
from .vit_attention import ViTSelfAttentionWithExtras  # noqa: F401

__version__ = "0.1.0"
