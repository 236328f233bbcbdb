"""Seeded synthetic inputs shared by tests/golden/make_golden.py (build container, next to the reference) and the tests (GPU box):
the larger fixtures (round 6: core_attn_long / *_h12 / cfg4_calib) store only the REFERENCE'S OUTPUTS, and both sides regenerate
the inputs and weights from these functions.  numpy's legacy RandomState streams are stable across numpy versions and platforms,
so the same bits come out here and there; nothing below imports the reference or the package."""
import numpy as np


def normal(seed: int, shape, scale: float = 1.0) -> np.ndarray:
    return (np.random.RandomState(seed).standard_normal(size=tuple(shape)) * scale).astype(np.float32)


def fp16_rounded(seed: int, shape, scale: float = 1.0) -> np.ndarray:
    """float32 values that are exactly representable in fp16 (the fp16 oracle's inputs: SURVEY 8c)."""
    return normal(seed, shape, scale).astype(np.float16).astype(np.float32)


def state_dict_like(shapes: dict, seed: int, w_std: float = 0.02, gate_std: float = 0.3) -> dict:
    """name -> float32 array for every (name, shape) of a module's state_dict, drawn in sorted-name order from one stream:
    projection weights and biases N(0, w_std), gate-predictor parameters (`alpha...`) N(0, gate_std)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        std = gate_std if name.startswith("alpha") else w_std
        out[name] = (rs.standard_normal(size=tuple(shapes[name])) * std).astype(np.float32)
    return out


def key_padding(B: int, S: int, left: list, right: list) -> np.ndarray:
    """(B, S) additive key-padding row: sample b keeps keys left[b] .. S - right[b] - 1; the others are float32 finfo.min."""
    m = np.zeros((B, S), np.float32)
    fmin = np.finfo(np.float32).min
    for b in range(B):
        m[b, : left[b]] = fmin
        if right[b]:
            m[b, S - right[b]:] = fmin
    return m


def opt_decoder_mask(B: int, T: int, lengths: list) -> np.ndarray:
    """HF 4.31 OPTDecoder._prepare_decoder_attention_mask: causal(finfo.min) + right padding(finfo.min), (B, 1, T, T) float32
    (the sum may reach -inf; the module clamps with torch.max(., finfo.min): opt_attention.py:220-224)."""
    fmin = np.finfo(np.float32).min
    causal = np.triu(np.full((T, T), fmin, np.float32), 1)[None, None].repeat(B, 0)
    pad = np.zeros((B, 1, T, T), np.float32)
    for b, L in enumerate(lengths):
        pad[b, :, :, L:] = fmin
    with np.errstate(over="ignore"):
        return (causal + pad).astype(np.float32)


# ---- the long-row core cases (core_attn_long.npz)
LONG_S, LONG_H, LONG_D = 512, 2, 64
LONG_PAD_S, LONG_PAD_B = 704, 2
LONG_PAD_LEFT, LONG_PAD_RIGHT = [0, 70], [150, 0]   # sample 0 right-padded, sample 1 left-padded


def long_causal_qkv():
    """OPT order at S = 512: q already multiplied by d ** -0.5 and re-rounded to fp16 (opt_attention.py:167 on fp16 storage)."""
    sh = (1, LONG_H, LONG_S, LONG_D)
    q = (fp16_rounded(6101, sh) * np.float32(LONG_D ** -0.5)).astype(np.float16).astype(np.float32)
    return q, fp16_rounded(6102, sh), fp16_rounded(6103, sh)


def long_padded_qkv():
    """BERT order at S = 704 (beyond the full-row kernels' 512 keys): raw q, the scale applied after the product."""
    sh = (LONG_PAD_B, 1, LONG_PAD_S, LONG_D)
    return fp16_rounded(6111, sh), fp16_rounded(6112, sh), fp16_rounded(6113, sh)


# ---- module I/O at 12 heads (bert_attn_h12.npz / opt_attn_h12.npz)
H12_B, H12_S, H12_E, H12_H = 2, 64, 768, 12


def h12_hidden(seed: int) -> np.ndarray:
    return normal(seed, (H12_B, H12_S, H12_E))


# ---- cfg4 calibration (cfg4_calib.npz): OPT-125m attention, B = 16, S = 512, 4 calibration batches + 1 evaluation batch
CFG4_B, CFG4_S, CFG4_E, CFG4_H = 16, 512, 768, 12
CFG4_CALIB_SEEDS, CFG4_EVAL_SEED, CFG4_WEIGHT_SEED = (2000, 2001, 2002, 2003), 2004, 2010


def cfg4_hidden(seed: int) -> np.ndarray:
    """Layer-norm-like hidden states with a few outlier channels (x 12 on 3 of the 768 features - what the reference measures)."""
    x = normal(seed, (CFG4_B, CFG4_S, CFG4_E))
    x[..., [77, 380, 588]] *= 12.0
    return x


# ---- ViT-S/16 attention at its own size (vit_attn_s16.npz): 197 tokens (196 patches + cls: not a multiple of 16), 6 heads of 64
VIT_B, VIT_N, VIT_C, VIT_H = 2, 197, 384, 6


def vit_tokens(seed: int) -> np.ndarray:
    return normal(seed, (VIT_B, VIT_N, VIT_C))


# ---- BERT-base INT8 validate flow at full size (bert_int8_calib.npz): B = 32, S = 128, E = 768, key padding, 4 calibration batches + 1 evaluation batch
BI8_B, BI8_S, BI8_E, BI8_H = 32, 128, 768, 12
BI8_CALIB_SEEDS, BI8_EVAL_SEED, BI8_WEIGHT_SEED = (2100, 2101, 2102, 2103), 2104, 2110


def bi8_hidden(seed: int) -> np.ndarray:
    x = normal(seed, (BI8_B, BI8_S, BI8_E))
    x[..., [12, 300, 701]] *= 10.0
    return x


def bi8_lengths() -> list:
    return [int(v) for v in np.random.RandomState(2120).randint(64, BI8_S + 1, size=BI8_B)]
