/*
 * oeh.h - C ABI of liboeh_hip.so: the MI355X (gfx950) implementation of OutEffHop's
 * modified-softmax attention hot path.
 *
 * The reference (MAGICS-LAB/OutEffHop) is pure PyTorch eager code and has NO native / FFI
 * boundary of its own ("Native-code inventory"); its plugin surface is Python
 * (SOFTMAX_MAPPING, AttentionGateType, the *WithExtras modules).  This header is therefore the
 * boundary a maintainer would bind from that Python surface (ctypes stub in INTEGRATION.md);
 * each entry point names the reference op chain it replaces.
 *
 * Conventions: plain pointers and sizes only (no torch / HIP types: `stream` is a hipStream_t
 * passed as void*); every pointer is a DEVICE pointer unless said otherwise; calls enqueue on
 * `stream` and return immediately; no allocation, no ownership transfer, no host synchronisation
 * (graph-capture safe); return 0 on success or a negative OEH_E* code, never throw.
 */
#ifndef OEH_H_
#define OEH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OEH_ABI_VERSION 6

/* error codes (negative errno style) */
#define OEH_OK 0
#define OEH_EINVAL (-22)     /* bad argument (null pointer, non-positive size, bad enum) */
#define OEH_ENOTSUP (-95)    /* shape / dtype / option combination not implemented */
#define OEH_EALIGN (-14)     /* pointer or stride not aligned as required (16 B rows) */
#define OEH_ELAUNCH (-5)     /* HIP reported a launch error */
#define OEH_ENODEV (-19)     /* no gfx950 device / code object not loadable */

/* storage dtypes of q, k, v, o (arithmetic is always fp32 accumulate) */
#define OEH_F16 0
#define OEH_BF16 1
#define OEH_F32 2
#define OEH_I8 3 /* oeh_attn_fwd only: q, k, v are CENTRED 8-bit quantiser indices (see oeh_attn_desc.q_grid) */

/* softmax base: softmax_n with n = 0 (torch softmax) or n = 1 (softmax_1) */
#define OEH_SOFTMAX_VANILLA 0
#define OEH_SOFTMAX_ONE 1

/* One per-tensor asymmetric uniform fake-quantiser in fixed-range mode
 * (AsymmetricUniformQuantizer.forward, OutEffHop/quantization/quantizers/uniform_quantizers.py:119-148;
 *  QuantizationManager in Qstates.fix_ranges, quantization_manager.py:94-102):
 *    idx = clamp(rint(x / scale) + zero_point, 0, qmax);  x_q = scale * (idx - zero_point)
 * scale = float32(max(delta, 1e-8)), zero_point = clamp(rint(zero_float), 0, qmax), qmax = 2^n_bits - 1,
 * all computed on the host from the calibrated (_delta, _zero_float) buffers.  The division is a
 * true IEEE fp32 division and rint is round-half-to-even (torch.round). */
typedef struct oeh_fq {
  int32_t enable;
  float scale;
  float zero_point;
  float qmax;
  uint8_t* dump_idx; /* optional (tests): receives idx as uint8, dense row-major in the tensor's logical shape */
} oeh_fq;

/* The three activation quantisers inside the attention class
 * (quantized_opt.py:154,182,210; quantized_bert.py:363,374,434). */
typedef struct oeh_fq_desc {
  oeh_fq scores;                 /* on QK^T (after scaling), BEFORE the mask is added; dump shape (B,H,Sq,Sk) */
  oeh_fq probs;                  /* on the softmax output (after clipping);              dump shape (B,H,Sq,Sk) */
  oeh_fq ctx;                    /* on P@V;                                              dump shape (B,H,Sq,D)  */
  int32_t ctx_quant_before_gate; /* 1: OPT order (quantized_opt.py:210 then gate :224-261);
                                    0: BERT order (gate quantized_bert.py:389-426 then quant :434) */
  int32_t ctx_emit_index;        /* 1: o receives the context quantiser's INTEGERS idx - zero_point (exact in every output
                                    dtype, |.| <= 255) instead of scale * (idx - zero_point).  For a consumer that is a
                                    QuantLinear (OPT's out_proj, quantized_opt.py:271 - integer weights): the projection is
                                    then ONE 16-bit GEMM of integers, exact products, with scale_ctx * scale_w folded into its
                                    output pass - no fp32 GEMM, no operand pairs.  Needs ctx.enable and the quantiser as the
                                    last op of the core (no gate after it), else OEH_EINVAL. */
} oeh_fq_desc;

/* Attention problem descriptor.  Tensors are (B,H,S,D) VIEWS given by element strides for
 * (batch, head, sequence); the head dim is contiguous.  This covers the reference's layouts:
 * BERT's permuted view of (B,S,H*D) (bert_attention.py:164-167), OPT's contiguous (B*H,S,D)
 * (opt_attention.py:146-147,198-201), ViT's qkv unbind (vit_attention.py:205-206) and STanHop's
 * (B,L,H,E) (hopfield.py:43-49).  The output is normally given (B,S,H,D) strides so the head merge
 * (bert_attention.py:335-337, opt_attention.py:318-322) costs nothing. */
typedef struct oeh_attn_desc {
  int32_t B, H, Sq, Sk, D;
  int32_t dtype;                 /* OEH_F16 | OEH_BF16 | OEH_F32 : q, k, v and o;  OEH_I8 : q, k, v are centred 8-bit indices on
                                    q_grid / k_grid / v_grid (v transposed) and o is o_dtype - see the ABI-3 fields below */
  int64_t q_stride[3];           /* elements: batch, head, seq */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int64_t o_stride[3];

  /* scores = (q.k) * scale          when scale_div == 0  (OPT: scale = 1, q pre-scaled, opt_attention.py:167;
   *                                  ViT vit_attention.py:71; Association hopfield.py:48)
   * scores = (q.k) / scale_div      when scale_div != 0  (BERT: / sqrt(d), bert_attention.py:265) */
  float scale;
  float scale_div;

  int32_t softmax_base;          /* OEH_SOFTMAX_VANILLA | OEH_SOFTMAX_ONE (vutils/softmax_1.py:4-28) */
  int32_t clip;                  /* 1: p = clip(p*(eta-gamma)+gamma, 0, 1) (models/softmax.py:10-19) */
  float gamma, eta;

  /* additive masks, applied in this order after the scores fake-quant:
   *   key_pad_mask (B,Sk): BERT's (B,1,1,S) extended mask (bert_attention.py:270-272)
   *   full_mask (B,1,Sq,Sk): OPT's causal+padding mask (opt_attention.py:215-220)
   *   causal: analytic causal mask, adds mask_min where k > q + (Sk - Sq)
   *   clamp_min: then max(x, mask_min) (opt_attention.py:221-223) */
  const void* key_pad_mask;
  int32_t key_pad_dtype;         /* OEH_F16 | OEH_F32 */
  int64_t key_pad_stride;        /* elements between batches */
  const void* full_mask;
  int32_t full_mask_dtype;       /* OEH_F16 | OEH_F32 */
  int64_t full_mask_stride[2];   /* elements: batch, query row; keys contiguous */
  int32_t causal;
  int32_t clamp_min;
  float mask_min;                /* finfo.min of the dtype the reference would hold the scores in */

  /* gate: fp32 values already multiplied by gate_scaling_factor, broadcast by zero strides
   * (context *= gate * scaling, bert_attention.py:327; opt_attention.py:309).  NULL = no gate. */
  const float* gate;
  int64_t gate_stride[3];        /* elements: batch, head, query row */

  /* OR the conditional per-token gate computed INSIDE the kernel from the layer input (bert_attention.py:301-327,
   * opt_attention.py:283-309), used when gate == NULL and gate_hidden != NULL: per head h the predictor acts on
   * gate_hidden[b, t, h*D:(h+1)*D] (same dtype as q, D contiguous, rows 16-byte aligned); weights are fp32
   * device arrays laid out as for oeh_gate_fwd (gate_units == 0: Linear(D,1): w1 (H,D), b1 (H); gate_units = m > 0:
   * Linear(D,m), ReLU, Linear(m,1): w1 (H,m,D), b1 (H,m), w2 (H,m), b2 (H)); context *= sigmoid(logit) * gate_scaling.
   * The first layer runs on the matrix cores with the weights rounded to the storage dtype and fp32 accumulation -
   * what the reference's own Linear does in a 16-bit model - so values agree with oeh_gate_fwd (fp32 weights) to
   * ~1e-4, not bit for bit.  gate_out (B,H,Sq) fp32, optional, receives sigmoid(logit) without the scaling (the
   * modules' last_gate_all_probs bookkeeping).  At most 64 hidden units (attn_gate_mlp2: head_dim); the 16-bit MFMA variants
   * ("fast16/...", "flash16/...") and, on fp32 storage, the full-row kernel (rows of <= 512 keys; weights and input as fp16
   * operand pairs: the logits are fp32-accurate, as the reference's fp32 Linear): OEH_ENOTSUP otherwise (use oeh_gate_fwd +
   * `gate`). */
  const void* gate_hidden;
  int64_t gate_hidden_stride[2]; /* elements: batch, token */
  const float* gate_w1;
  const float* gate_b1;
  const float* gate_w2;
  const float* gate_b2;
  int32_t gate_units;
  float gate_scaling;
  float* gate_out;

  /* dtype == OEH_I8 (ABI 3) - the INT8 configuration with q, k, v as the 8-bit indices the reference's QuantLinear
   * projections put them on (hijacker.py:78-127; quantized_opt.py:67-75: q_proj / k_proj / v_proj are QuantLinear, their
   * outputs fake-quantised per tensor before the bmm at :151), both products on the integer matrix cores:
   *   an element is the int8 value c = idx - 128 of the index idx in [0, 255]; its value is grid.scale * (c + 128 - grid.zero_point)
   *   (for q: AFTER OPT's `* scaling`, i.e. fold head_dim^-0.5 into q_grid.scale or pass it as `scale`);
   *   q, k: (B,H,S,D) views as usual (strides in elements = bytes); v TRANSPOSED: a (B,H,D,Sk) view, keys contiguous,
   *   v_stride = (batch, head, d row) - the second product sums over keys;
   *   o has dtype o_dtype (OEH_F16 | OEH_BF16 | OEH_F32; ABI 6: OEH_I8 with oeh_fq_desc.ctx_emit_index on a full 8-bit context grid with a
   *   whole zero point - o then holds the context quantiser's CENTRED indices idx - 128 as int8, what oeh_proj_quant_i8 takes with pairs == 3).
   * Requires D == 64, Sk <= 512 and a multiple of 16, 16-byte aligned rows, masks none | causal | key_pad_mask (with or
   * without causal) whose entries are 0 (visible) or <= -1e4 (padded: HF's extended masks hold 0 / finfo.min) - a padded key is
   * dropped exactly like a causally hidden one, other mask values are NOT added to the scores on this path -, clipping only with gamma <= 0 (the reference's registry), and `fq`
   * with scores and probabilities enabled (probabilities on a full 8-bit grid, qmax == 255; whole-number zero points, as
   * uniform_quantizers.py:79 makes them): OEH_ENOTSUP otherwise (run the fake-quant variants on dequantised values).
   * The test-only index dumps (oeh_fq.dump_idx) are honoured with o_dtype == OEH_F32: scores for every key (the reference
   * quantises before the mask is added), probabilities, context - same layouts as everywhere. */
  struct { float scale; float zero_point; } q_grid, k_grid, v_grid;
  int32_t o_dtype;               /* dtype == OEH_I8: the output's dtype.  dtype OEH_F16 / OEH_BF16 (ABI 5): OEH_F32 here asks for the
                                    output straight from the kernel's fp32 accumulators (o is then fp32, strides in fp32 elements) - the
                                    arithmetic of the kernel that ships before its output rounding, for the "within 1e-3" checks: sibling
                                    instantiations that differ in the epilogue's store only.  Head dim 64: the one-pass kernel's plain
                                    form (masks none / causal / key padding / a (B,1,Sq,Sk) mask; or with the in-kernel gate predictor,
                                    unmasked / causal) and the full-row kernel's plain / clipped forms (+ key padding; the plain form also
                                    with the in-kernel gate predictor).  Head dim 128 (round 5): the plain forms of both (the one-pass
                                    kernel with one block per wave, unmasked / causal).  OEH_ENOTSUP elsewhere.  Any other value: o has `dtype`. */

  /* (appended in ABI 4: fields are only ever added at the end of a descriptor) */
  int32_t key_pad_boolean;       /* key_pad_mask: the caller's promise that every entry is 0 or <= -1e4 (HF's extended masks: 0 / finfo.min):
                                    a padded key is then simply invisible and the INT8 chain can stay on the quantiser grid with
                                    key padding too (full-row kernel's grid form, two-pass form for rows of more than 512 keys)
                                    instead of the reference's op order on dequantised values.  With moderate negative entries
                                    (an additive bias rather than a mask) leave it 0.  Same result as the literal order except
                                    for a row WITHOUT any visible key whose mask entries are not absorbing (> -1e30): such a
                                    row comes out as with finfo.min entries.  dtype OEH_I8 always reads the mask this way. */
} oeh_attn_desc;

/* QK^T -> scale -> [fq] -> mask -> softmax / softmax_1 -> [clip] -> [fq] -> PV -> [fq] -> gate -> [fq]
 * in ONE kernel.  Replaces bert_attention.py:222-337, opt_attention.py:204-322,
 * vit_attention.py:54-75/215-266, hopfield.py:47-49 and, with `fq`, quantized_bert.py:317-434 /
 * quantized_opt.py:151-270.  `fq` may be NULL.  Dropout and head_mask are inference no-ops in the
 * reference and are not part of this path (callers must not be training with p_drop > 0). */
int oeh_attn_fwd(const oeh_attn_desc* desc, const void* q, const void* k, const void* v, void* o,
                 const oeh_fq_desc* fq, void* stream);

/* Row-wise softmax family on a dense (rows, cols) array: the SOFTMAX_MAPPING callables
 * (models/softmax.py:22-64; vutils/softmax_1.py:24-28).  x and y may alias.  dtype of x and y. */
int oeh_softmax_rows(const void* x, void* y, int64_t rows, int32_t cols, int32_t dtype, int32_t softmax_base,
                     int32_t clip, float gamma, float eta, void* stream);

/* Stand-alone per-tensor asymmetric fake-quant (QuantizedActivation in fixed-range mode,
 * base_quantized_classes.py:182-199).  y (same dtype as x) and/or idx (uint8) may be NULL. */
int oeh_fake_quant(const void* x, void* y, uint8_t* idx, int64_t n, int32_t dtype, float scale, float zero_point,
                   float qmax, void* stream);

/* Gate probabilities for the conditional gates (bert_attention.py:301-327):
 *   hidden (B,T,E) [dtype], E = H*d; per-head predictor fc_h on hidden[:, :, h*d:(h+1)*d]:
 *     hidden_units == 0 : Linear(d,1)            w1 (H,d)        b1 (H)          (w2,b2 NULL)
 *     hidden_units  > 0 : Linear(d,m),ReLU,Linear(m,1)  w1 (H,m,d) b1 (H,m) w2 (H,m) b2 (H)
 *   per_head_pool: average the logits over T before the sigmoid (conditional_per_head, :321-322)
 *   gate_out (B,H,T) fp32 (or (B,H,1) when pooling) = sigmoid(logit) * scaling.
 * Weights are fp32 device arrays. */
int oeh_gate_fwd(const void* hidden, int32_t dtype, int32_t B, int32_t T, int32_t H, int32_t d,
                 int64_t hidden_stride_b, int64_t hidden_stride_t, const float* w1, const float* b1, const float* w2,
                 const float* b2, int32_t hidden_units, int32_t per_head_pool, float scaling, float* gate_out,
                 void* stream);

/* min and max of a dense array -> out[0], out[1] (fp32 device scalars): the CurrentMinMax / RunningMinMax
 * (no percentile) range statistics (range_estimators.py:71-72,96-97) without a device->host copy. */
int oeh_minmax(const void* x, int64_t n, int32_t dtype, float* out2, void* stream);

/* Percentile range statistics with the running average, entirely in device memory: RunningMinMaxEstimator with
 * `percentile` (range_estimators.py:83-106; the --est_ranges_pct flow of validate_clm.py:450-454 through
 * pass_data_for_range_estimation, transformers_language/utils.py:50-71), replacing the reference's device->host copy +
 * np.percentile per quantiser and batch:
 *   lo = np.percentile(x, q_lo), hi = np.percentile(x, q_hi)   (percents; "linear" interpolation between the two order
 *        statistics around each rank, float64 - what numpy returns for float32 data; exact, by radix selection)
 *   state[0..1] = first ? (lo, hi) : (1 - momentum) * (lo, hi) + momentum * state[0..1]
 * state: device double[2].  work: device scratch of OEH_CALIB_WORK_BYTES (8-byte aligned; contents irrelevant before and
 * after).  n < 2^32.  No host synchronisation. */
#define OEH_CALIB_WORK_BYTES 36864
int oeh_percentile_ema(const void* x, int64_t n, int32_t dtype, double q_lo, double q_hi, double momentum, int32_t first,
                       double* state, void* work, void* stream);

/* The activation quantiser's forward while ranges are still being estimated (QuantizationManager.forward in
 * Qstates.estimate_ranges, quantization_manager.py:104-112): the grid is derived IN the kernel from the float64
 * (x_min, x_max) pair in device memory exactly as set_quant_range does (uniform_quantizers.py:72-82:
 * x_min <- min(x_min, 0), x_max <- max(x_max, eps), delta = (x_max - x_min) / (2^n_bits - 1), zero = -x_min / delta,
 * scale = float32(max(delta, eps)), zero_point = clamp(rint(zero), 0, 2^n_bits - 1)), then y = scale * (idx - zero_point). */
int oeh_fake_quant_range(const void* x, void* y, int64_t n, int32_t dtype, const double* xmin_xmax, int32_t n_bits, double eps,
                         void* stream);

/* Range estimation of the attention core's activation quantisers WITHOUT materialising the (B,H,Sq,Sk) tensors.  In
 * Qstates.estimate_ranges the reference hands the whole score tensor and then the whole probability tensor to np.percentile
 * (range_estimators.py:83-106 from quantized_opt.py:154,182 / quantized_bert.py:363,374; 201 MB each per OPT-125m layer and
 * batch).  This entry point RECOMPUTES the values tile by tile - fp32 throughout, both products on the fp32 matrix-core
 * instruction, the elementwise chain in the reference's op order - and feeds them to the exact radix selection of
 * oeh_percentile_ema, once per selection pass; nothing of size Sq x Sk is stored.  `desc` as for oeh_attn_fwd (dtype
 * OEH_F16 | OEH_BF16 | OEH_F32, D in {32, 64, 128}, any Sq / Sk, masks key_pad_mask / full_mask / causal, scale or scale_div,
 * softmax_base, clip; gate fields are ignored):
 *   which == OEH_CALIB_SCORES : state <- percentile pair [+ running average] of the scaled scores (before quantiser and masks)
 *   which == OEH_CALIB_PROBS  : scores fake-quantised on the grid of `scores_range` (NULL: not quantised), masks, softmax, clip;
 *                               state <- percentile pair [+ running average] of the probabilities
 *   which == OEH_CALIB_CONTEXT: ... probabilities fake-quantised on the grid of `probs_range` (NULL: not), P V -> ctx_out
 *                               (fp32, o_stride of desc in fp32 elements); no statistics (state / work unused) - the context is
 *                               small, its quantiser runs oeh_percentile_ema / oeh_fake_quant_range on it
 * scores_range / probs_range: device double[2] = (x_min, x_max), e.g. the `state` of the previous call; the grids are derived
 * in the kernel as oeh_fake_quant_range does (n_bits, eps).  q_lo, q_hi, momentum, first, state, work: as oeh_percentile_ema.
 * No host synchronisation; safe under hipGraph capture. */
#define OEH_CALIB_SCORES 0
#define OEH_CALIB_PROBS 1
#define OEH_CALIB_CONTEXT 2
int oeh_attn_calibrate(const oeh_attn_desc* desc, const void* q, const void* k, const void* v, float* ctx_out, int32_t which,
                       const double* scores_range, const double* probs_range, int32_t n_bits, double eps, double q_lo, double q_hi,
                       double momentum, int32_t first, double* state, void* work, void* stream);

/* The producer side of the INT8-storage attention core (oeh_attn_fwd with dtype OEH_I8): a QuantLinear projection's output
 * quantiser (hijacker.py:78-127; AsymmetricUniformQuantizer.forward, uniform_quantizers.py:119-148) that writes what the core
 * consumes - the centred int8 index c = clamp(rint(x / scale) + zero_point, 0, 255) - 128 - instead of a fake-quantised
 * float tensor, split into heads of 64:
 *   x: (B, S, H*64) values in `dtype`, last dim contiguous, x_stride = (batch, row) in elements;
 *   transpose == 0: out is (B, S, H*64) int8 in x's element order (q, k; needs x_stride[0] == S * x_stride[1]);
 *   transpose == 1: out is (B, H, 64, S) int8, keys contiguous (v: the layout the second product wants; S % 16 == 0);
 *   y (optional, may be NULL): the dequantised values scale * (idx - zero_point) in `dtype`, y_stride like x_stride - what a
 *   decoder keeps as its (k, v) cache - in the same pass; with transpose == 0, out may be NULL when y is given: the values
 *   only, i.e. a QuantLinear's scale + bias + output fake-quant in one pass over the GEMM accumulator (out_proj);
 *   bias (optional, fp32 device array of H*64, may be NULL): x is a raw GEMM accumulator and the value that is quantised is
 *   alpha * x + bias[column] - the projection's weight scale and bias folded into this pass (oeh_split_pairs' GEMM);
 *   bias == NULL: x is quantised as it is (alpha ignored).
 * x, out, y and bias must be 16-byte aligned, and so must the row and batch strides of x and y in bytes (16-byte vector
 * accesses): OEH_EALIGN otherwise.
 * One launch per projection instead of fake-quant + index conversion + transpose copy. */
int oeh_quantize_heads_i8(const void* x, int8_t* out, void* y, int64_t B, int32_t S, int32_t H, const int64_t x_stride[2],
                          const int64_t y_stride[2], int32_t dtype, float scale, float zero_point, int32_t transpose, float alpha,
                          const float* bias, void* stream);

/* fp32 activations as fp16 operand pairs for a library GEMM either side of the attention core (the q/k/v and output
 * projections of an fp32 model, opt_attention.py:167-201,318-324 / quantized_opt.py:67-75): out[r][0:K] = hi = RN16(x[r]),
 * out[r][K:2K] = lo = RN16((x[r] - hi) * 2^11); with the weight matrix stacked as [W ; W * 2^-11] one fp16 GEMM with fp32
 * accumulation gives x.W to ~2^-22 relative (the weight side exactly when W holds 8-bit integers, as QuantLinear's weights
 * do up to their scale) at fp16 matrix-core speed.  x: (rows, K) fp32, K % 8 == 0, 16-byte aligned rows; out: (rows, 2K) fp16. */
int oeh_split_pairs(const float* x, void* out_f16, int64_t rows, int32_t K, int64_t x_stride_row, void* stream);

/* The same prologue for Linears with GENERAL fp32 weights (the unquantised fp32 models of validate_clm.py / validate_mlm_config.py:
 * q_proj / k_proj / v_proj / out_proj, query / key / value - opt_attention.py:167-201,318-324, bert_attention.py:183-208): with
 * x = xh + xl 2^-11, W = Wh + Wl 2^-11, b = bh + bl 2^-11 (all halves fp16),
 *     x W^T + b  =  [ xh | xh 2^-5 | xl 2^-5 | 1, 2^-5, 0 x6 ]  @  [ Wh ; Wl 2^-6 ; Wh 2^-6 ; bh ; bl 2^-6 ; 0 x6 ]   (+ O(2^-22) relative)
 * i.e. ONE fp16 GEMM with fp32 accumulation over K' = 3K + 8, bias included - measured closer to the float64 result than the fp32
 * library GEMM and about twice as fast (M = 8192, K = 768: N = 768 46 us against 89 us, N = 2304 122 us against 231 us).
 * This entry writes the activation side: x (rows, K) fp32 with row stride x_stride_row (elements) -> out_f16 (rows, 3K + 8) fp16,
 * contiguous; K % 8 == 0, 16-byte aligned rows.  Values beyond the fp16 range saturate (as in oeh_split_pairs). */
int oeh_split_triples(const float* x, void* out_f16, int64_t rows, int32_t K, int64_t x_stride_row, void* stream);

/* The q / k / v projections of a QuantLinear model as ONE matrix-core GEMM with the output quantisers in its epilogue (SURVEY 8f-1;
 * quantized_opt.py:67-75 q_proj / k_proj / v_proj, quantized_bert.py:236-238 query / key / value: QuantLinear = weight fake-quant +
 * F.linear + output fake-quant, hijacker.py:78-127): what a library GEMM over the stacked weights followed by three
 * oeh_quantize_heads_i8 passes computes, without the (B*S, n_seg*E) accumulator ever reaching memory.
 *   a: the activations (B*S rows, row stride lda elements, 16-byte aligned rows) - pairs == 0: fp16 (rows, K); pairs == 1: fp16
 *      (rows, 2K), the operand pairs [hi | lo] that oeh_split_pairs writes for an fp32 model; pairs == 2: the fp32 activations (rows, K)
 *      themselves - the kernel forms a (hi, lo) pair when a wave reads its operand fragments, without the split pass: 64 x = hi + lo with the
 *      residual unscaled (|x - pair| <= 2^-31: 22 bits down to |x| = 2^-9, as pairs == 1 above that; |x| <= 2 047, saturating beyond), so an index
 *      differs from the pairs == 1 call's only where the value sits on a rounding boundary to within the fp32 accumulation error; pairs == 3: int8 (rows, K) - e.g. the centred indices idx - 128 of the producer's 8-bit
 *      quantiser (oeh_attn_fwd, dtype OEH_I8, o_dtype OEH_I8) - against int8 weights on the integer matrix cores (K % 64 == 0): exact
 *      int32 sums; with acc_add[n] = (128 - zero_point) * sum_k w[n][k] the accumulator is the sum over idx - zero_point;
 *   w: (n_seg*E, K) fp16 (int8 with pairs == 3), row stride ldw: the QuantLinear weights' INTEGERS (w / weight scale: exact in fp16), the segments' rows
 *      one after the other; pairs == 1 multiplies the lo half against w * 2^-11, formed in registers (exact on integers);
 *   bias: (n_seg*E) fp32; segment i covers output columns [i*E, (i+1)*E) and turns the fp32 accumulator into
 *      value = alpha * acc + bias[column],  c = clamp(rint(value / scale) + zero_point, 0, 255) - 128:
 *      out  (may be NULL when y is given): int8, (B, S, E) [transpose == 0: q, k] or (B, E/64, 64, S) [transpose == 1: v, keys
 *           contiguous] - the layouts oeh_attn_fwd takes with dtype OEH_I8; 16-byte aligned;
 *      y    (optional): the dequantised values scale * (c + 128 - zero_point) as fp32, (B*S, E) with row stride y_stride_row
 *           elements - a decoder's (k, v) cache.
 * K % 32 == 0, E % 64 == 0, S % 16 == 0 (and every 32-bit lane offset of a, w and y below 4 GiB): OEH_ENOTSUP otherwise; n_seg outside
 * 1..3: OEH_EINVAL.  Same formulas as oeh_quantize_heads_i8; the
 * accumulation order differs from a library GEMM's, so an index may differ by one step where the value sits on a rounding
 * boundary to within the fp32 accumulation error (both are fp32-grade: |acc - exact| <~ 1e-6 of the row's scale). */
typedef struct {
  float alpha;
  float scale, zero_point;
  int8_t* out;
  float* y;
  int64_t y_stride_row;
  int32_t transpose;
  const int32_t* acc_add;   /* pairs == 3 only: E integers added to the int32 accumulator before alpha (NULL: none) */
} oeh_proj_seg;
int oeh_proj_quant_i8(const void* a, int32_t pairs, const void* w, const float* bias, int64_t B, int32_t S, int32_t K, int32_t E, int32_t n_seg,
                      const oeh_proj_seg* segs, int64_t lda, int64_t ldw, void* stream);

/* library information (host side, no device work) */
int oeh_abi_version(void);
const char* oeh_build_info(void);       /* "gfx950 hipcc <version> ..." */
const char* oeh_strerror(int code);
/* name of the kernel variant oeh_attn_fwd would launch for `desc` ("flash16/MQ2/D64/f16" one-pass kernel,
 * "fast16/NT32/D64/f16/clip" full-row kernel, "mfma16/NT32/D64/f16/fq" general kernel, "generic") or NULL if
 * unsupported; host only.  The returned string lives in thread-local storage until the next call on this thread. */
const char* oeh_attn_variant(const oeh_attn_desc* desc, const oeh_fq_desc* fq);
#ifdef __cplusplus
}
#endif
#endif /* OEH_H_ */
