// extern "C" entry points of liboeh_hip.so (declared in include/oeh.h): argument validation, kernel
// variant selection and launch.  No allocation, no synchronisation: safe under hipGraph capture.
#include "../../include/oeh.h"
#include "../../include/oeh_debug.h"
#include "oeh_gemm.h"
#include "oeh_attn_params.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace oeh {
int launch_attn_mfma_d32(const AttnParams& P, int in, bool fq, hipStream_t st);
int launch_attn_mfma_d64(const AttnParams& P, int in, bool fq, hipStream_t st);
int launch_attn_mfma_d128(const AttnParams& P, int in, bool fq, hipStream_t st);
int launch_attn_fast_d32(const AttnParams& P, int in, hipStream_t st);
int launch_attn_fast_d64(const AttnParams& P, int in, hipStream_t st);
int launch_attn_fast_d128(const AttnParams& P, int in, hipStream_t st);
int launch_attn_flash_d32(const AttnParams& P, int in, int mq, hipStream_t st);
int launch_attn_flash_d64(const AttnParams& P, int in, int mq, hipStream_t st);
int launch_attn_flash_d128(const AttnParams& P, int in, int mq, hipStream_t st);
int launch_attn_generic(const AttnParams& P, int in, hipStream_t st);
int launch_attn_small(const AttnParams& P, int in, hipStream_t st);
int launch_attn_i8(const AttnParams& P, int out, hipStream_t st);
int launch_attn_wide(const AttnParams& P, int in, hipStream_t st);
int launch_softmax_rows(const void* x, void* y, long rows, int cols, int in, int base, int clip, float w, float g, hipStream_t st);
int launch_fake_quant(const void* x, void* y, unsigned char* idx, long n, int in, FqP f, hipStream_t st);
int launch_gate(const void* hidden, int in, int B, int T, int H, int d, long hs_b, long hs_t, const float* w1, const float* b1,
                const float* w2, const float* b2, int m_units, int pool, float scaling, float* out, hipStream_t st);
int launch_minmax(const void* x, long n, int in, float* out2, hipStream_t st);
int launch_percentile_ema(const void* x, long n, int in, double q_lo, double q_hi, double momentum, int first, double* state, void* work,
                          hipStream_t st);
int launch_fake_quant_range(const void* x, void* y, long n, int in, const double* range, float qmax, double eps, hipStream_t st);
int launch_attn_calibrate(const AttnParams& P, int in, int which, const double* s_range, const double* p_range, float qmax, double eps, double q_lo,
                          double q_hi, double momentum, int first, double* state, void* work, hipStream_t st);
int launch_split_pairs(const float* x, void* out, long rows, int K, long x_sr, hipStream_t st);
int launch_split_triples(const float* x, void* out, long rows, int K, long x_sr, hipStream_t st);
int launch_quantize_heads_i8(const void* x, signed char* out, void* y, long B, int S, int H, long x_sb, long x_ss, long y_sb, long y_ss, int in,
                             FqP f, int transpose, float alpha, const float* bias, hipStream_t st);
}  // namespace oeh

using oeh::AttnParams;
using oeh::FqP;

namespace {

bool dtype_ok(int d) { return d == OEH_F16 || d == OEH_BF16 || d == OEH_F32; }
int elem_bytes(int d) { return d == OEH_F32 ? 4 : d == OEH_I8 ? 1 : 2; }

// RN(scale * log2(e)), the product formed in double
float scale_log2e(float scale) { return (float)((double)scale * 1.4426950408889634074); }

FqP make_fq(const oeh_fq* f) {
  FqP r;
  std::memset(&r, 0, sizeof(r));
  if (f != nullptr && f->enable) {
    r.en = 1;
    r.scale = f->scale;
    r.rscale = 1.0f / f->scale;
    r.zp = f->zero_point;
    r.qmax = f->qmax;
    r.lo = -f->zero_point;
    r.hi = f->qmax - f->zero_point;
    r.dump = f->dump_idx;
    r.c2 = scale_log2e(f->scale);
    r.oscale = f->scale;
  }
  return r;
}

enum Variant { V_NONE = 0, V_FLASH, V_FAST, V_MFMA, V_GENERIC, V_SMALL, V_I8 };

// rows must be 16-byte aligned for the MFMA path's 16-B loads / 8..16-B stores
bool aligned16(const void* p, const int64_t* st, int eb) {
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return false;
  for (int i = 0; i < 3; ++i)
    if (((st[i] * eb) & 15) != 0) return false;
  return true;
}

int validate(const oeh_attn_desc* d, const void* q, const void* k, const void* v, void* o, const oeh_fq_desc* fq) {
  if (d == nullptr || q == nullptr || k == nullptr || v == nullptr || o == nullptr) return OEH_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->Sq <= 0 || d->Sk <= 0 || d->D <= 0) return OEH_EINVAL;
  if (!dtype_ok(d->dtype) && d->dtype != OEH_I8) return OEH_EINVAL;
  if (d->dtype == OEH_I8) {
    if (!dtype_ok(d->o_dtype) && d->o_dtype != OEH_I8) return OEH_EINVAL;
    // (an int8 output is the context quantiser's centred indices: only with ctx_emit_index on a full 8-bit grid with a whole zero point)
    if (d->o_dtype == OEH_I8 && (fq == nullptr || !fq->ctx_emit_index || !fq->ctx.enable || fq->ctx.qmax != 255.0f || fq->ctx.zero_point != std::nearbyint(fq->ctx.zero_point)))
      return OEH_EINVAL;
    const float zs[3] = {d->q_grid.zero_point, d->k_grid.zero_point, d->v_grid.zero_point};
    const float ss[3] = {d->q_grid.scale, d->k_grid.scale, d->v_grid.scale};
    for (int i = 0; i < 3; ++i)
      if (!(ss[i] > 0.0f) || !(zs[i] >= 0.0f && zs[i] <= 255.0f) || zs[i] != std::nearbyint(zs[i])) return OEH_EINVAL;
  }
  if (d->softmax_base != OEH_SOFTMAX_VANILLA && d->softmax_base != OEH_SOFTMAX_ONE) return OEH_EINVAL;
  {  // strides are element counts in [0, 2^32): the kernels form batch / head offsets from 32-bit products (views with negative
     // strides are not a layout any caller of the reference produces)
    const int64_t* sts[4] = {d->q_stride, d->k_stride, d->v_stride, d->o_stride};
    for (const int64_t* st : sts)
      for (int i = 0; i < 3; ++i)
        if (st[i] < 0 || st[i] >= ((int64_t)1 << 32)) return OEH_ENOTSUP;
  }
  if (fq != nullptr && fq->ctx_emit_index) {  // the integers idx - zp instead of values: only where the context quantiser is the last op
    const bool gated = d->gate != nullptr || d->gate_hidden != nullptr;
    if (!fq->ctx.enable || (gated && fq->ctx_quant_before_gate)) return OEH_EINVAL;
  }
  if (d->key_pad_mask != nullptr && d->key_pad_dtype != OEH_F16 && d->key_pad_dtype != OEH_F32) return OEH_EINVAL;
  if (d->full_mask != nullptr && d->full_mask_dtype != OEH_F16 && d->full_mask_dtype != OEH_F32) return OEH_EINVAL;
  if (d->gate == nullptr && d->gate_hidden != nullptr) {
    if (d->gate_w1 == nullptr || d->gate_b1 == nullptr || d->gate_units < 0) return OEH_EINVAL;
    if (d->gate_units > 0 && (d->gate_w2 == nullptr || d->gate_b2 == nullptr)) return OEH_EINVAL;
  }
  if (fq != nullptr) {
    const oeh_fq* fs[3] = {&fq->scores, &fq->probs, &fq->ctx};
    for (const oeh_fq* f : fs) {
      if (!f->enable) continue;
      if (!(f->scale > 0.0f) || !(f->qmax >= 1.0f) || f->zero_point < 0.0f || f->zero_point > f->qmax) return OEH_EINVAL;
      if (f->dump_idx != nullptr && f->qmax > 255.0f) return OEH_ENOTSUP;
    }
  }
  return OEH_OK;
}

// 16-bit q / k / v with the output taken from the fp32 accumulators (include/oeh.h: o_dtype)
bool want_out32(const oeh_attn_desc* d) { return (d->dtype == OEH_F16 || d->dtype == OEH_BF16) && d->o_dtype == OEH_F32; }

bool any_fq(const oeh_fq_desc* fq) { return fq != nullptr && (fq->scores.enable || fq->probs.enable || fq->ctx.enable); }

bool is_pow2(float x) {
  int e = 0;
  return x > 0.0f && std::isfinite(x) && std::frexp(x, &e) == 0.5f;
}

// The fast kernel (oeh_attn_fast.inl) covers 16-bit storage, masks in {none, key padding, causal} and a positive
// multiplicative scale (a power-of-two divisor is the same multiply, exactly); fake-quant is its FQ variant.
bool fast_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq) {
  if (d->dtype == OEH_F32 || d->full_mask != nullptr) return false;
  // the LDS-DMA streams address a tile as scalar base + 32-bit lane byte offsets (up to 64 rows of the sequence stride)
  if (d->q_stride[2] >= (1 << 24) || d->k_stride[2] >= (1 << 24) || d->v_stride[2] >= (1 << 24) || d->q_stride[2] < 0 || d->k_stride[2] < 0 || d->v_stride[2] < 0) return false;
  if (any_fq(fq)) {  // the FQ variant: scores and probabilities both quantised (the reference's configuration), no in-kernel predictor
    if (!(fq->scores.enable && fq->probs.enable) || (d->gate == nullptr && d->gate_hidden != nullptr)) return false;
    if (fq->scores.dump_idx != nullptr || fq->probs.dump_idx != nullptr || fq->ctx.dump_idx != nullptr) return false;  // test-only dumps: general kernel
  }
  // a divisor (BERT order, bert_attention.py:265): a power of two is the same multiply exactly; any other positive divisor
  // becomes a multiply by RN(1/div) - one more rounding of 6e-8 relative, far inside the accuracy of the 16-bit paths -
  // except under fake-quant, where the index must come from the reference's own rounding (general kernel: true division)
  if (d->scale_div != 0.0f ? !(is_pow2(d->scale_div) || (!any_fq(fq) && d->scale_div > 0.0f && std::isfinite(d->scale_div)))
                           : !(d->scale > 0.0f && std::isfinite(d->scale))) return false;
  if (d->clip && d->gamma > 0.0f) return false;
  if (d->causal && d->Sq > d->Sk) return false;
  if ((d->causal || d->key_pad_mask != nullptr) && !(d->mask_min < -1.0e4f)) return false;
  return true;
}

int g_force_flash = 0;                   // tools/microbench.py only
int g_no_d128_rule = 0;                 // tests: the full-row kernel also for head dim 128 with clip / INT8 (the comparison against the general kernel)
int g_force_small = 0;                   // tests: the small-shape kernel wherever it can run (no size heuristic)
int g_wide = 0;                          // tools / tests: the 32x32x16 form of the one-pass kernel (oeh_attn_wide.hip) wherever it applies

// The one-pass kernel (oeh_attn_flash.inl) additionally needs the plain softmax_n (no clip).  No Sk limit.
bool flash_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq, bool short_rows_too = false) {
  if (d->full_mask != nullptr) {  // a (B,1,Sq,Sk) mask: only where the general kernel does not reach (rows of more than 512 keys); PAD variant
    oeh_attn_desc t = *d;
    t.full_mask = nullptr;
    return d->Sk > 512 && !d->clip && !any_fq(fq) && fast_eligible(&t, fq) && (d->mask_min < -1.0e4f);
  }
  if (!fast_eligible(d, fq) || d->clip || any_fq(fq)) return false;
  // (key padding under the vanilla softmax: a row without a visible key is uniform over ALL keys in the reference - the PAD variant's
  // epilogue gives such rows the mean of V, oeh_attn_flash.inl)
  if (short_rows_too) return true;
  // short rows (<= 128 keys) fit the full-row kernel's registers in one pass, which measures faster there (BERT-base B=32 S=128:
  // 7.7 vs 9.7 us per launch) - until the batch is large enough for the one-pass kernel's 128-row workgroups to fill the chip by
  // themselves (>= 768 of them): then its halved K / V streaming wins (H=12 S=128, round 3: B=48 12.3 vs 13.0 us, B=64 15.5 vs 14.5,
  // B=96 23.0 vs 20.3, B=128 31.5 vs 24.2)
  // causal rows whose count leaves the last 128-row workgroup at most half full (S = 192, 320, 448): the full-row kernel's 64-row
  // workgroups waste nothing there (16.9 vs 18.3, 15.7 vs 17.1, 18.7 vs 19.4 us); without the causal mask the one-pass kernel keeps its lead
  // (round 4, profiles/r04_dispatch_ab.txt - both sides of every rule in one process: S = 192 4.8 %, S = 320 3.1 % for the full-row kernel,
  // S = 448 a tie (18.3 vs 18.2 us): the rule ends at 384 keys, like its fp32 twin below)
  if (d->causal && d->Sk <= 384 && ((d->Sq - 1) % 128) < 64 && d->Sq > 128 && !g_force_flash && !short_rows_too) return false;
  if (d->Sk <= 128 && !g_force_flash) {
    const long wgs = (long)d->B * d->H * ((d->Sq + 127) / 128);
    // (one threshold for every head dim since round 4: at d = 32 and 768 workgroups the one-pass kernel measured 10.0 vs 11.0 us, at
    // 1536 16.7 vs 21.8 - the separate d <= 32 threshold of 1536 was the wrong way round on the re-measurement)
    if (!(d->Sk > 64 && d->Sq >= 112 && wgs >= 768)) return false;
  }
  return true;
}

// Clipped softmax on rows of MORE than 512 keys: the one-pass kernel's two-pass form (CLIP: statistics, then the product;
// oeh_attn_flash.inl).  Up to 512 keys the full-row kernel computes the scores once and is the faster one.
bool flash_clip_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq) {
  oeh_attn_desc t = *d;
  if (t.dtype == OEH_F32) t.dtype = OEH_F16;  // fp32 storage: the SRC32 form (operand pairs), same conditions on the problem
  // round 4: a (B,1,Sq,Sk) mask on rows of more than 512 keys too (the PAD variants read it per block, as the plain one-pass form does)
  const bool long_full = d->full_mask != nullptr && d->Sk > 512 && (d->mask_min < -1.0e4f);
  if (long_full) t.full_mask = nullptr;
  if (!d->clip || any_fq(fq) || !fast_eligible(&t, fq)) return false;  // (fast_eligible: gamma <= 0, masks, scale)
  if (d->gate == nullptr && d->gate_hidden != nullptr) return false;
  // (key padding / a full mask under the VANILLA softmax since round 4: a row without a visible key - uniform over all Sk keys in the
  // reference - gets clip(w / Sk + gamma) times the sum of V in the epilogue, oeh_attn_flash.inl)
  return d->Sk > 512 || (d->D == 128 && d->Sk >= 384 && d->full_mask == nullptr) || (g_force_flash && d->full_mask == nullptr);  // (d = 128, S = 512: 47.4 vs 58.2 us in the full-row kernel, causal 38.1 vs 41.2)
}

// The fused INT8 chain on rows of MORE than 512 keys: the one-pass kernel's two-pass form of the grid chain (TP = 2).  What
// the full-row kernel's FQ == 1 takes: scores and probabilities both quantised, no clip, no key padding, no test dumps.
bool flash_fq_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq) {
  oeh_attn_desc t = *d;
  if (t.dtype == OEH_F32) t.dtype = OEH_F16;  // fp32 storage: the SRC32 form
  if (!any_fq(fq) || !fast_eligible(&t, fq)) return false;  // (fast_eligible: both quantisers, no dumps, scale, masks, gamma <= 0 when clipped)
  if (d->gate == nullptr && d->gate_hidden != nullptr) return false;
  // key padding: on the grid only as a vector of 0 / <= -1e4 entries (key_pad_boolean), and with softmax_1 (trailing padded tiles are not streamed)
  if (d->key_pad_mask != nullptr && !(d->key_pad_boolean && d->softmax_base == OEH_SOFTMAX_ONE)) return false;
  if (fq->probs.qmax > (d->dtype == OEH_BF16 ? 255.0f : 2047.0f)) return false;  // the integer-valued P operand must be exact
  return d->Sk > 512 || g_force_flash;
}

// fp32 storage on the one-pass kernel (SRC32 variants: tiles staged through registers, fp16 operands, fp32 output): the same
// conditions on the problem; the in-kernel gate predictor reads a 16-bit layer input and is not available
bool flash32_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq) {
  if (d->dtype != OEH_F32 || (d->gate == nullptr && d->gate_hidden != nullptr)) return false;
  oeh_attn_desc t = *d;
  t.dtype = OEH_F16;
  if (!flash_eligible(&t, fq, true)) return false;
  // rows of <= 128 keys: the full-row kernel's fp32 form where it applies, at every batch size measured (round 3, H=12 S=128: B=32 16.5
  // vs 20.0 us, B=64 29.8 vs 30.6, B=128 54.6 vs 57.7); what it does not take stays here (16.7 us in the general kernel)
  // (with >= 768 of its 128-row workgroups the one-pass form is ahead again: B=64 25.8 vs 30.5 us; with a padding vector 30.5 vs 31.5 -
  // round 4: one rule for both, profiles/r04_dispatch_ab.txt)
  const bool many_unpadded = d->Sq >= 112 && d->Sk > 64 && (long)d->B * d->H * ((d->Sq + 127) / 128) >= 768;
  if (d->Sk <= 128 && !g_force_flash && d->full_mask == nullptr && !many_unpadded && fast_eligible(&t, fq)) return false;
  // ... and causal rows that leave the last 128-row workgroup at most half full, up to 320 rows (S = 192: 33.1 vs 40.3 us, S = 320: 38.4 vs
  // 39.0; S = 448: 42.9 vs 42.2 - the one-pass kernel again)
  if (d->causal && d->Sk <= 384 && d->Sq > 128 && ((d->Sq - 1) % 128) < 64 && !g_force_flash && d->full_mask == nullptr && fast_eligible(&t, fq)) return false;
  return true;
}
// ... and on the full-row kernel (clipped softmax, the INT8 chain, vanilla softmax with key padding)
bool fast32_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq) {
  if (d->dtype != OEH_F32) return false;  // (the in-kernel gate predictor: on operand pairs here - fast_eligible refuses it together with fake-quant)
  oeh_attn_desc t = *d;
  t.dtype = OEH_F16;
  return fast_eligible(&t, fq);
}

// The small-shape kernel (oeh_attn_small.hip: one wave per (batch, head), K and V in registers, fp32 matrix-core products):
// many tiny problems - STanHop's Association (L, S ~ 28, H = 4, E in {16, 32, 64}, batch * data_dim problems).  No masks,
// no fake-quant, no in-kernel gate predictor; rows of 4 elements must be 4-element aligned (16 B in fp32, 8 B in 16-bit).
bool small_eligible(const oeh_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const oeh_fq_desc* fq) {
  if (any_fq(fq) || d->key_pad_mask != nullptr || d->full_mask != nullptr || d->causal) return false;
  if (d->gate == nullptr && d->gate_hidden != nullptr) return false;
  if (!(d->D == 16 || d->D == 32 || d->D == 64) || d->Sk > 64 || d->Sq > 64) return false;
  // A wave per problem only pays when there are enough problems to fill the chip, on fp32 data - where it also is the EXACT path (fp32
  // matrix-core products: 3e-6 of the reference; the operand-pair kernels round the probability operand to fp16) - and on rows of at
  // most 32 keys: measured, round 3 (d = 64, against the full-row kernel's fp32 form): B*H = 896 S = 28 9.4 vs 8.9 us (kept here: STanHop's
  // shape, the exact path), B*H = 1536 S = 32 13.3 vs 13.7, but S = 48 29.8 vs 20.2 and S = 64 37.0 vs 22.7; 16-bit data: S = 28 8.3 vs 8.05
  // in the full-row kernel, S = 48 18.6 vs 6.6, S = 64 19.9 vs 7.1 - or when no matrix-core kernel takes the shape (d = 16)
  if (!g_force_small && d->D != 16 && (d->dtype != OEH_F32 || d->Sk > 32 || d->Sq > 32 || (long)d->B * d->H < 256)) return false;
  const int ab = 4 * elem_bytes(d->dtype);
  const int64_t* sts[4] = {d->q_stride, d->k_stride, d->v_stride, d->o_stride};
  const void* ps[4] = {q, k, v, o};
  for (int t = 0; t < 4; ++t) {
    if (q != nullptr && (reinterpret_cast<uintptr_t>(ps[t]) % ab) != 0) return false;
    for (int i = 0; i < 3; ++i)
      if ((sts[t][i] & 3) != 0) return false;
  }
  return true;
}

unsigned long long* g_stamps = nullptr;  // tools/timeline.py only
int g_variant_off = 0;                   // tools/microbench.py only: bit (1 << Variant) disables a variant
int g_flash_mq = 0;                      // tools/microbench.py only: force query blocks per wave
int g_head_group = 0;                    // tools/microbench.py only (OEH_HEAD_GROUP): block order of the fp32-storage kernels in groups of heads
int g_place = 0;                         // tools/microbench.py only: 1 = plain block order in the one-pass kernel (no snake placement)

// query blocks (16 rows) per wave of the one-pass kernel: 2 (128-row workgroups) once that still gives every CU two
// workgroups, else 1
int flash_mq(const oeh_attn_desc* d) {
  if (g_flash_mq != 0 && !(d->D == 128 && d->dtype == OEH_F32)) return g_flash_mq;
  if (d->D == 128 && d->dtype == OEH_F32) return 1;  // two blocks of fp32 operand pairs at d = 128 do not fit the register file (130 spills)
  const long wg2 = (long)((d->Sq + 127) / 128) * d->B * d->H;
  return (d->Sq > 64 && wg2 >= 416) ? 2 : 1;  // (H=12 S=512 causal: B=8 - 384 workgroups - 10.8 vs 12.4 us with one block per wave, B=9 - 432 - 13.3 vs 12.8, B=10 14.5 vs 12.7, B=12 16.2 vs 13.5)
}

// The 32x32x16 form of the one-pass kernel (oeh_attn_wide.hip): plain softmax / softmax_1, 16-bit storage, head dim 64, masks none |
// causal, gate values only.  Behind the diagnostic hook (oeh_debug_set_variant bit 12) until it is measured ahead.
bool wide_eligible(const oeh_attn_desc* d, const oeh_fq_desc* fq) {
#ifndef OEH_WITH_WIDE
  return false;  // the candidate is not part of the production library (csrc/Makefile: `make experiment` builds it in)
#endif
  if (!g_wide || d->D != 64 || (d->dtype != OEH_F16 && d->dtype != OEH_BF16) || want_out32(d)) return false;
  if (d->clip || any_fq(fq) || d->key_pad_mask != nullptr || d->full_mask != nullptr || (d->gate == nullptr && d->gate_hidden != nullptr)) return false;
  return d->Sq > 64;
}

// INT8 storage (oeh_attn_i8.hip): see include/oeh.h, oeh_attn_desc.q_grid
bool i8_eligible(const oeh_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const oeh_fq_desc* fq) {
  if (d->dtype != OEH_I8 || d->D != 64 || d->Sk > 512 || (d->Sk & 15) != 0) return false;
  if (fq == nullptr || !fq->scores.enable || !fq->probs.enable || fq->probs.qmax != 255.0f) return false;
  // the kernel forms the probabilities' indices with the float zero point and centres them with the integer one; the score
  // index is rounded in the integer domain: both zero points must be whole numbers (they are: uniform_quantizers.py:79 rounds)
  if (fq->probs.zero_point != std::nearbyint(fq->probs.zero_point) || fq->scores.zero_point != std::nearbyint(fq->scores.zero_point)) return false;
  const bool dumps = fq->scores.dump_idx != nullptr || fq->probs.dump_idx != nullptr || fq->ctx.dump_idx != nullptr;
  if (dumps && d->o_dtype != OEH_F32) return false;  // the index dumps (tests) exist in the fp32-output form
  if ((d->clip && d->gamma > 0.0f) || d->full_mask != nullptr || (d->gate == nullptr && d->gate_hidden != nullptr)) return false;
  if (d->key_pad_mask != nullptr && !(d->mask_min < -1.0e4f)) return false;  // (a key-padding vector of 0 / <= -1e4 entries: include/oeh.h)
  if (d->scale_div != 0.0f ? !(d->scale_div > 0.0f && std::isfinite(d->scale_div)) : !(d->scale > 0.0f && std::isfinite(d->scale))) return false;
  if (d->causal && (d->Sq > d->Sk || !(d->mask_min < -1.0e4f))) return false;
  if (d->k_stride[2] <= 0 || d->k_stride[2] >= (1 << 24) || d->v_stride[2] <= 0 || d->v_stride[2] >= (1 << 24)) return false;  // 32-bit lane offsets inside a tile
  if (q != nullptr) {
    const int ob = elem_bytes(d->o_dtype);
    if (!aligned16(q, d->q_stride, 1) || !aligned16(k, d->k_stride, 1) || !aligned16(v, d->v_stride, 1) || !aligned16(o, d->o_stride, ob)) return false;
  }
  return true;
}

Variant pick_variant(const oeh_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const oeh_fq_desc* fq) {
  if (d->dtype == OEH_I8) return i8_eligible(d, q, k, v, o, fq) ? V_I8 : V_NONE;
  const int eb = elem_bytes(d->dtype);
  const bool shape_ok = (d->D == 32 || d->D == 64 || d->D == 128) && d->Sk <= 512;
  // integer-valued (idx - zp) must be exact in the 16-bit P operand: |.| <= 2048 (f16) / 256 (bf16)
  bool p_exact = true;
  if (fq != nullptr && fq->probs.enable) p_exact = fq->probs.qmax <= (d->dtype == OEH_BF16 ? 255.0f : 2047.0f);
  const bool al = (q == nullptr) || (aligned16(q, d->q_stride, eb) && aligned16(k, d->k_stride, eb) &&
                                     aligned16(v, d->v_stride, eb) && aligned16(o, d->o_stride, want_out32(d) ? 4 : eb));
  const bool d_ok = d->D == 32 || d->D == 64 || d->D == 128;
  if (small_eligible(d, q, k, v, o, fq) && !(g_variant_off & (1 << V_SMALL))) return V_SMALL;
  if (d_ok && al && flash_eligible(d, fq) && !(g_variant_off & (1 << V_FLASH))) return V_FLASH;
  if (d_ok && al && flash32_eligible(d, fq) && !(g_variant_off & ((1 << V_FLASH) | (1 << 6)))) return V_FLASH;
  if (d_ok && al && flash_fq_eligible(d, fq) && !(g_variant_off & ((1 << V_FLASH) | (d->dtype == OEH_F32 ? (1 << 6) : 0)))) return V_FLASH;
  if (d_ok && al && flash_clip_eligible(d, fq) && !(g_variant_off & ((1 << V_FLASH) | (d->dtype == OEH_F32 ? (1 << 6) : 0)))) return V_FLASH;
  // head dim 128: the full-row kernel fits ONE workgroup per CU (16-KB tiles), and with the clip or the INT8 chain on top the general
  // kernel - smaller workgroups of its own - measures faster (round 3, B=16 H=8 S=512 causal: INT8 44.6 vs 39.8 us, clipped 41.2 vs 39.7;
  // S=256 clipped 30.3 vs 25.0); the plain softmax stays on the one-pass / full-row kernels
  const bool d128_general = !g_no_d128_rule && d->D == 128 && d->dtype != OEH_F32 && (any_fq(fq) || d->clip) && d->key_pad_mask == nullptr && !(d->gate == nullptr && d->gate_hidden != nullptr);
  if (shape_ok && p_exact && al && fast_eligible(d, fq) && !d128_general && !(g_variant_off & (1 << V_FAST))) return V_FAST;
  if (shape_ok && p_exact && al && fast32_eligible(d, fq) && !(g_variant_off & ((1 << V_FAST) | (1 << 7)))) return V_FAST;
  if (shape_ok && p_exact && al) return V_MFMA;
  if ((size_t)(d->D + d->Sk) * 4 <= 64 * 1024) return V_GENERIC;
  return V_NONE;
}

void fill_params(AttnParams& P, const oeh_attn_desc* d, const void* q, const void* k, const void* v, void* o, const oeh_fq_desc* fq) {
  std::memset(&P, 0, sizeof(P));
  P.q = q; P.k = k; P.v = v; P.o = o;
  P.B = d->B; P.H = d->H; P.Sq = d->Sq; P.Sk = d->Sk; P.D = d->D;
  P.qs_b = d->q_stride[0]; P.qs_h = d->q_stride[1]; P.qs_s = d->q_stride[2];
  P.ks_b = d->k_stride[0]; P.ks_h = d->k_stride[1]; P.ks_s = d->k_stride[2];
  P.vs_b = d->v_stride[0]; P.vs_h = d->v_stride[1]; P.vs_s = d->v_stride[2];
  P.os_b = d->o_stride[0]; P.os_h = d->o_stride[1]; P.os_s = d->o_stride[2];
  P.scale = d->scale; P.scale_div = d->scale_div;
  P.base = d->softmax_base;
  P.clip = d->clip ? 1 : 0;
  // (eta - gamma) is formed in double and rounded to fp32 once, as Python does before the tensor multiply
  P.clip_w = (float)((double)d->eta - (double)d->gamma);
  P.clip_g = d->gamma;
  P.pad = d->key_pad_mask; P.pad_f16 = d->key_pad_dtype == OEH_F16; P.pad_sb = d->key_pad_stride;
  P.pad_bool = (d->key_pad_mask != nullptr && d->key_pad_boolean && d->mask_min < -1.0e4f) ? 1 : 0;
  P.full = d->full_mask; P.full_f16 = d->full_mask_dtype == OEH_F16;
  P.full_sb = d->full_mask_stride[0]; P.full_sq = d->full_mask_stride[1];
  P.causal = d->causal ? 1 : 0; P.clamp_min = d->clamp_min ? 1 : 0; P.mask_min = d->mask_min;
  P.gate = d->gate; P.gs_b = d->gate_stride[0]; P.gs_h = d->gate_stride[1]; P.gs_s = d->gate_stride[2];
  if (d->gate == nullptr && d->gate_hidden != nullptr) {
    P.gh = d->gate_hidden; P.ghs_b = d->gate_hidden_stride[0]; P.ghs_t = d->gate_hidden_stride[1];
    P.gw1 = d->gate_w1; P.gb1 = d->gate_b1; P.gw2 = d->gate_w2; P.gb2 = d->gate_b2;
    P.g_units = d->gate_units; P.g_scaling = d->gate_scaling; P.g_out = d->gate_out;
  }
  if (fq != nullptr) {
    P.fq_s = make_fq(&fq->scores); P.fq_p = make_fq(&fq->probs); P.fq_c = make_fq(&fq->ctx);
    P.ctx_before_gate = fq->ctx_quant_before_gate ? 1 : 0;
    if (fq->ctx_emit_index) P.fq_c.oscale = 1.0f;  // (validate(): the context quantiser is the last op)
  }
  P.stamps = g_stamps;
  P.snake = (g_place & 1) ? 0 : 1;
  P.head_major = g_head_group;
  P.nQT = (d->Sq + 63) / 64;
  P.nBH = d->B * d->H;
  P.nBHpad = (P.nBH + 7) & ~7;
  P.magic_nbh = (unsigned)(0x100000000ULL / (unsigned long long)P.nBHpad > 0xffffffffULL ? 0xffffffffULL : 0x100000000ULL / (unsigned long long)P.nBHpad);
  P.magic_h = (unsigned)(0x100000000ULL / (unsigned long long)P.H > 0xffffffffULL ? 0xffffffffULL : 0x100000000ULL / (unsigned long long)P.H);
  // Causal tiles strictly above the diagonal contribute exactly 0 to P@V (and nothing to the row statistics)
  // and may be skipped when: masked probabilities are exactly 0 before the clip and the clip maps 0 to 0
  // (gamma <= 0); the row can never be fully masked under vanilla softmax (which would make it uniform over
  // ALL keys) - guaranteed by the diagonal unless another mask is present; and no index dump wants every element.
  const bool dump = (P.fq_s.dump != nullptr) || (P.fq_p.dump != nullptr);
  const bool other_mask = (P.pad != nullptr) || (P.full != nullptr);
  P.skip_ok = (P.causal && d->Sq <= d->Sk && (!P.clip || d->gamma <= 0.0f) && !dump && (P.base == 1 || !other_mask) &&
               std::isfinite(d->mask_min) && d->mask_min < -1e4f) ? 1 : 0;
}

const char* variant_name(Variant v, const oeh_attn_desc* d, bool fq) {
  static thread_local char buf[64];
  if (v == V_GENERIC) return "generic";
  if (v == V_I8) {
    std::snprintf(buf, sizeof(buf), "i8mfma/NT%d/D64/%s", d->Sk <= 128 ? 8 : (d->Sk <= 256 ? 16 : 32),
                  d->o_dtype == OEH_F16 ? "f16" : (d->o_dtype == OEH_BF16 ? "bf16" : (d->o_dtype == OEH_I8 ? "i8" : "f32")));
    return buf;
  }
  if (v == V_SMALL) {
    std::snprintf(buf, sizeof(buf), "small/ST%d/D%d/%s", d->Sk <= 32 ? 2 : 4, d->D, d->dtype == OEH_F16 ? "f16" : (d->dtype == OEH_BF16 ? "bf16" : "f32"));
    return buf;
  }
  if (v == V_NONE) return nullptr;
  const int nt = d->Sk <= 128 ? 8 : (d->Sk <= 256 ? 16 : 32);
  const char* dt = d->dtype == OEH_F16 ? "f16" : (d->dtype == OEH_BF16 ? "bf16" : "f32");
  if (v == V_FLASH && wide_eligible(d, nullptr) && !fq) std::snprintf(buf, sizeof(buf), "flash16w/D%d/%s", d->D, dt);
  else if (v == V_FLASH) std::snprintf(buf, sizeof(buf), "flash16/MQ%d/D%d/%s%s", flash_mq(d), d->D, dt, fq ? "/fq2p" : (d->clip ? "/clip2p" : ""));
  else if (v == V_FAST) std::snprintf(buf, sizeof(buf), "fast16/NT%d/D%d/%s%s%s", nt, d->D, dt, d->clip ? "/clip" : "", fq ? "/fq" : "");
  else std::snprintf(buf, sizeof(buf), "mfma16/NT%d/D%d/%s%s", nt, d->D, dt, fq ? "/fq" : "");
  return buf;
}

}  // namespace

extern "C" {

int oeh_attn_fwd(const oeh_attn_desc* desc, const void* q, const void* k, const void* v, void* o, const oeh_fq_desc* fq,
                 void* stream) {
  int rc = validate(desc, q, k, v, o, fq);
  if (rc != OEH_OK) return rc;
  const Variant var = pick_variant(desc, q, k, v, o, fq);
  if (var == V_NONE) return OEH_ENOTSUP;
  if (desc->gate == nullptr && desc->gate_hidden != nullptr) {  // fused gate predictor: 16-bit MFMA variants, 16-B aligned rows
    // up to four 16-unit MFMA tiles of hidden units; fp32 storage: the full-row kernel's operand-pair form only
    if ((var != V_FAST && var != V_FLASH) || desc->gate_units > 64 || (desc->dtype == OEH_F32 && var != V_FAST)) return OEH_ENOTSUP;
    const int geb = desc->dtype == OEH_F32 ? 4 : 2;
    if (((reinterpret_cast<uintptr_t>(desc->gate_hidden) | (uintptr_t)(desc->gate_hidden_stride[0] * geb) | (uintptr_t)(desc->gate_hidden_stride[1] * geb)) & 15) != 0) return OEH_EALIGN;
    if (desc->gate_hidden_stride[1] <= 0 || desc->gate_hidden_stride[1] >= (1 << 24)) return OEH_ENOTSUP;  // (32-bit lane offsets of the input rows' LDS-DMA)
  }
  AttnParams P;
  fill_params(P, desc, q, k, v, o, fq);
  P.src32 = (desc->dtype == OEH_F32 && (var == V_FLASH || var == V_FAST)) ? 1 : 0;  // fp32 storage read directly, fp32 output
  if (want_out32(desc)) {
    // the accumulators themselves - sibling instantiations (O32) of what the 16-bit workloads of BASELINE.json run.  Head dim 64: the one-pass
    // kernel's plain form with masks none / causal / key padding / a (B,1,Sq,Sk) mask, or with the in-kernel gate predictor (not both);
    // the full-row kernel's plain and clipped forms (+ key padding), the plain form with the in-kernel gate predictor.  Head dim 128
    // (round 5): the plain forms of both (one block per wave in the one-pass kernel), gate values only.
    const bool gated_in_kernel = desc->gate == nullptr && desc->gate_hidden != nullptr;
    const bool masked = desc->key_pad_mask != nullptr || desc->full_mask != nullptr;
    bool ok = !any_fq(fq);
    if (desc->D == 64) {
      if (var == V_FLASH) ok = ok && !desc->clip && !(masked && gated_in_kernel);
      else if (var == V_FAST) ok = ok && !(desc->clip && gated_in_kernel);
      else ok = false;
    } else if (desc->D == 128) {
      ok = ok && !desc->clip && !gated_in_kernel && ((var == V_FLASH && !masked && flash_mq(desc) == 1) || var == V_FAST);
    } else {
      ok = false;
    }
    if (!ok) return OEH_ENOTSUP;
    P.out32 = 1;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (var == V_SMALL) return oeh::launch_attn_small(P, desc->dtype, st);
  if (var == V_I8) {
    const double mult = desc->scale_div != 0.0f ? 1.0 / (double)desc->scale_div : (double)desc->scale;
    P.i8_cq = 128 - (int)desc->q_grid.zero_point; P.i8_ck = 128 - (int)desc->k_grid.zero_point;
    P.i8_cv = 128 - (int)desc->v_grid.zero_point; P.i8_cp = 128 - (int)fq->probs.zero_point;
    P.i8_k1 = (float)((double)desc->q_grid.scale * (double)desc->k_grid.scale * mult / (double)fq->scores.scale);
    P.i8_so = (float)((double)fq->probs.scale * (double)desc->v_grid.scale);
    P.nBHpad = (P.nBH + 15) & ~15;  // head pairs on one XCD (oeh_attn_i8.hip)
    P.magic_nbh = (unsigned)(0x100000000ULL / (unsigned long long)P.nBHpad);
    return oeh::launch_attn_i8(P, desc->o_dtype, st);
  }
  if (var == V_FLASH) {
    if (desc->scale_div != 0.0f) { P.scale = 1.0f / desc->scale_div; P.scale_div = 0.0f; }  // (fast_eligible: exact for a power of two)
#ifdef OEH_WITH_WIDE
    if (wide_eligible(desc, fq)) {
      P.nQT = (desc->Sq + 127) / 128;
      return oeh::launch_attn_wide(P, desc->dtype, st);
    }
#endif
    const int mq = flash_mq(desc);
    P.nQT = (desc->Sq + 64 * mq - 1) / (64 * mq);
    switch (desc->D) {
      case 32: return oeh::launch_attn_flash_d32(P, desc->dtype, mq, st);
      case 64: return oeh::launch_attn_flash_d64(P, desc->dtype, mq, st);
      default: return oeh::launch_attn_flash_d128(P, desc->dtype, mq, st);
    }
  }
  if (var == V_FAST) {
    if (desc->scale_div != 0.0f) { P.scale = 1.0f / desc->scale_div; P.scale_div = 0.0f; }  // (fast_eligible: exact for a power of two)
    switch (desc->D) {
      case 32: return oeh::launch_attn_fast_d32(P, desc->dtype, st);
      case 64: return oeh::launch_attn_fast_d64(P, desc->dtype, st);
      default: return oeh::launch_attn_fast_d128(P, desc->dtype, st);
    }
  }
  if (var == V_MFMA) {
    switch (desc->D) {
      case 32: return oeh::launch_attn_mfma_d32(P, desc->dtype, any_fq(fq), st);
      case 64: return oeh::launch_attn_mfma_d64(P, desc->dtype, any_fq(fq), st);
      default: return oeh::launch_attn_mfma_d128(P, desc->dtype, any_fq(fq), st);
    }
  }
  return oeh::launch_attn_generic(P, desc->dtype, st);
}

const char* oeh_attn_variant(const oeh_attn_desc* desc, const oeh_fq_desc* fq) {
  if (desc == nullptr || desc->B <= 0 || desc->H <= 0 || desc->Sq <= 0 || desc->Sk <= 0 || desc->D <= 0 || (!dtype_ok(desc->dtype) && desc->dtype != OEH_I8))
    return nullptr;
  return variant_name(pick_variant(desc, nullptr, nullptr, nullptr, nullptr, fq), desc, any_fq(fq));
}

int oeh_softmax_rows(const void* x, void* y, int64_t rows, int32_t cols, int32_t dtype, int32_t softmax_base, int32_t clip,
                     float gamma, float eta, void* stream) {
  if (x == nullptr || y == nullptr || rows < 0 || cols <= 0 || !dtype_ok(dtype)) return OEH_EINVAL;
  if (softmax_base != OEH_SOFTMAX_VANILLA && softmax_base != OEH_SOFTMAX_ONE) return OEH_EINVAL;
  if (rows == 0) return OEH_OK;
  const float w = (float)((double)eta - (double)gamma);
  return oeh::launch_softmax_rows(x, y, rows, cols, dtype, softmax_base, clip ? 1 : 0, w, gamma, reinterpret_cast<hipStream_t>(stream));
}

int oeh_fake_quant(const void* x, void* y, uint8_t* idx, int64_t n, int32_t dtype, float scale, float zero_point, float qmax,
                   void* stream) {
  if (x == nullptr || n < 0 || !dtype_ok(dtype)) return OEH_EINVAL;
  if (!(scale > 0.0f) || !(qmax >= 1.0f) || zero_point < 0.0f || zero_point > qmax) return OEH_EINVAL;
  if (idx != nullptr && qmax > 255.0f) return OEH_ENOTSUP;
  if (n == 0 || (y == nullptr && idx == nullptr)) return OEH_OK;
  FqP f;
  std::memset(&f, 0, sizeof(f));
  f.en = 1; f.scale = scale; f.rscale = 1.0f / scale; f.zp = zero_point; f.qmax = qmax;
  f.lo = -zero_point; f.hi = qmax - zero_point;
  f.c2 = scale_log2e(scale);
  return oeh::launch_fake_quant(x, y, idx, n, dtype, f, reinterpret_cast<hipStream_t>(stream));
}

int oeh_gate_fwd(const void* hidden, int32_t dtype, int32_t B, int32_t T, int32_t H, int32_t d, int64_t hidden_stride_b,
                 int64_t hidden_stride_t, const float* w1, const float* b1, const float* w2, const float* b2, int32_t hidden_units,
                 int32_t per_head_pool, float scaling, float* gate_out, void* stream) {
  if (hidden == nullptr || w1 == nullptr || b1 == nullptr || gate_out == nullptr || !dtype_ok(dtype)) return OEH_EINVAL;
  if (B <= 0 || T <= 0 || H <= 0 || d <= 0 || hidden_units < 0) return OEH_EINVAL;
  if (hidden_units > 0 && (w2 == nullptr || b2 == nullptr)) return OEH_EINVAL;
  return oeh::launch_gate(hidden, dtype, B, T, H, d, hidden_stride_b, hidden_stride_t, w1, b1, w2, b2, hidden_units,
                          per_head_pool ? 1 : 0, scaling, gate_out, reinterpret_cast<hipStream_t>(stream));
}

int oeh_minmax(const void* x, int64_t n, int32_t dtype, float* out2, void* stream) {
  if (x == nullptr || out2 == nullptr || n <= 0 || !dtype_ok(dtype)) return OEH_EINVAL;
  return oeh::launch_minmax(x, n, dtype, out2, reinterpret_cast<hipStream_t>(stream));
}

int oeh_percentile_ema(const void* x, int64_t n, int32_t dtype, double q_lo, double q_hi, double momentum, int32_t first, double* state,
                       void* work, void* stream) {
  if (x == nullptr || state == nullptr || work == nullptr || n <= 0 || n >= (int64_t)1 << 32 || !dtype_ok(dtype)) return OEH_EINVAL;
  if (!(q_lo >= 0.0 && q_lo <= 100.0 && q_hi >= 0.0 && q_hi <= 100.0) || !(momentum >= 0.0 && momentum <= 1.0)) return OEH_EINVAL;
  if ((reinterpret_cast<uintptr_t>(work) & 7) != 0 || (reinterpret_cast<uintptr_t>(state) & 7) != 0) return OEH_EALIGN;
  return oeh::launch_percentile_ema(x, n, dtype, q_lo, q_hi, momentum, first ? 1 : 0, state, work, reinterpret_cast<hipStream_t>(stream));
}

int oeh_fake_quant_range(const void* x, void* y, int64_t n, int32_t dtype, const double* xmin_xmax, int32_t n_bits, double eps,
                         void* stream) {
  if (x == nullptr || y == nullptr || xmin_xmax == nullptr || n < 0 || !dtype_ok(dtype) || n_bits < 1 || n_bits > 16 || !(eps > 0.0)) return OEH_EINVAL;
  if (n == 0) return OEH_OK;
  return oeh::launch_fake_quant_range(x, y, n, dtype, xmin_xmax, (float)((1 << n_bits) - 1), eps, reinterpret_cast<hipStream_t>(stream));
}

int oeh_attn_calibrate(const oeh_attn_desc* desc, const void* q, const void* k, const void* v, float* ctx_out, int32_t which,
                       const double* scores_range, const double* probs_range, int32_t n_bits, double eps, double q_lo, double q_hi,
                       double momentum, int32_t first, double* state, void* work, void* stream) {
  if (desc == nullptr || q == nullptr || k == nullptr) return OEH_EINVAL;
  if (which < OEH_CALIB_SCORES || which > OEH_CALIB_CONTEXT || n_bits < 1 || n_bits > 16 || !(eps > 0.0)) return OEH_EINVAL;
  if (which == OEH_CALIB_CONTEXT ? (v == nullptr || ctx_out == nullptr) : (state == nullptr || work == nullptr)) return OEH_EINVAL;
  if (desc->B <= 0 || desc->H <= 0 || desc->Sq <= 0 || desc->Sk <= 0 || !dtype_ok(desc->dtype)) return OEH_EINVAL;
  if (desc->softmax_base != OEH_SOFTMAX_VANILLA && desc->softmax_base != OEH_SOFTMAX_ONE) return OEH_EINVAL;
  if (desc->key_pad_mask != nullptr && desc->key_pad_dtype != OEH_F16 && desc->key_pad_dtype != OEH_F32) return OEH_EINVAL;
  if (desc->full_mask != nullptr && desc->full_mask_dtype != OEH_F16 && desc->full_mask_dtype != OEH_F32) return OEH_EINVAL;
  if (!(desc->D == 32 || desc->D == 64 || desc->D == 128)) return OEH_ENOTSUP;
  if (which != OEH_CALIB_CONTEXT) {
    if (!(q_lo >= 0.0 && q_lo <= 100.0 && q_hi >= 0.0 && q_hi <= 100.0) || !(momentum >= 0.0 && momentum <= 1.0)) return OEH_EINVAL;
    if ((int64_t)desc->B * desc->H * desc->Sq * desc->Sk >= ((int64_t)1 << 32)) return OEH_ENOTSUP;
    if ((reinterpret_cast<uintptr_t>(work) & 7) != 0 || (reinterpret_cast<uintptr_t>(state) & 7) != 0) return OEH_EALIGN;
  }
  const int eb = elem_bytes(desc->dtype);
  if (!aligned16(q, desc->q_stride, eb) || !aligned16(k, desc->k_stride, eb)) return OEH_EALIGN;
  if (which == OEH_CALIB_CONTEXT && (!aligned16(v, desc->v_stride, eb) || !aligned16(ctx_out, desc->o_stride, 4))) return OEH_EALIGN;
  AttnParams P;
  fill_params(P, desc, q, k, v, ctx_out, nullptr);
  return oeh::launch_attn_calibrate(P, desc->dtype, which, scores_range, probs_range, (float)((1 << n_bits) - 1), eps, q_lo, q_hi, momentum, first ? 1 : 0,
                                    state, work, reinterpret_cast<hipStream_t>(stream));
}

int oeh_quantize_heads_i8(const void* x, int8_t* out, void* y, int64_t B, int32_t S, int32_t H, const int64_t x_stride[2], const int64_t y_stride[2],
                          int32_t dtype, float scale, float zero_point, int32_t transpose, float alpha, const float* bias, void* stream) {
  if (x == nullptr || (out == nullptr && (y == nullptr || transpose)) || x_stride == nullptr || B <= 0 || S <= 0 || H <= 0 || !dtype_ok(dtype)) return OEH_EINVAL;
  if (!(scale > 0.0f) || zero_point < 0.0f || zero_point > 255.0f || zero_point != std::nearbyint(zero_point)) return OEH_EINVAL;
  if (y != nullptr && y_stride == nullptr) return OEH_EINVAL;
  if (!transpose && x_stride[0] != (int64_t)S * x_stride[1]) return OEH_ENOTSUP;
  if (!transpose && y != nullptr && y_stride[0] != (int64_t)S * y_stride[1]) return OEH_ENOTSUP;
  if (transpose && (S & 15) != 0) return OEH_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return OEH_EALIGN;
  {  // 16-byte vector loads of the input rows and of the bias
    const int eb = elem_bytes(dtype);
    if (((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(x_stride[0] * eb) | (uintptr_t)(x_stride[1] * eb) | reinterpret_cast<uintptr_t>(bias)) & 15) != 0) return OEH_EALIGN;
  }
  if (y != nullptr && ((reinterpret_cast<uintptr_t>(y) | (uintptr_t)(y_stride[0] * elem_bytes(dtype)) | (uintptr_t)(y_stride[1] * elem_bytes(dtype))) & 15) != 0) return OEH_EALIGN;  // 16-byte value stores
  FqP f;
  std::memset(&f, 0, sizeof(f));
  f.en = 1; f.scale = scale; f.rscale = 1.0f / scale; f.zp = zero_point; f.qmax = 255.0f; f.lo = -zero_point; f.hi = 255.0f - zero_point;
  return oeh::launch_quantize_heads_i8(x, reinterpret_cast<signed char*>(out), y, B, S, H, x_stride[0], x_stride[1], y != nullptr ? y_stride[0] : 0,
                                       y != nullptr ? y_stride[1] : 0, dtype, f, transpose ? 1 : 0, alpha, bias, reinterpret_cast<hipStream_t>(stream));
}

int oeh_proj_quant_i8(const void* a, int32_t pairs, const void* w, const float* bias, int64_t B, int32_t S, int32_t K, int32_t E, int32_t n_seg,
                      const oeh_proj_seg* segs, int64_t lda, int64_t ldw, void* stream) {
  if (a == nullptr || w == nullptr || bias == nullptr || segs == nullptr || B <= 0 || S <= 0 || K <= 0 || E <= 0 || n_seg < 1 || n_seg > 3) return OEH_EINVAL;
  if ((K % oeh::kGemmBK) != 0 || (E & 63) != 0 || (S & 15) != 0 || B * (int64_t)S > 0x7fffffffLL) return OEH_ENOTSUP;
  if (pairs < 0 || pairs > 3) return OEH_EINVAL;
  const int64_t aeb = pairs == 2 ? 4 : pairs == 3 ? 1 : 2;  // (pairs == 2: a is the fp32 activation matrix itself; 3: int8 a and w)
  const int64_t web = pairs == 3 ? 1 : 2;
  if (pairs == 3 && (K & 63) != 0) return OEH_ENOTSUP;
  if (lda < (pairs == 1 ? 2 : 1) * (int64_t)K || ldw < K) return OEH_EINVAL;
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) | (uintptr_t)(lda * aeb) | (uintptr_t)(ldw * web)) & 15) != 0) return OEH_EALIGN;
  if (B * (int64_t)S * lda * aeb >= 0xffffffffLL || (int64_t)n_seg * E * ldw * web >= 0xffffffffLL) return OEH_ENOTSUP;  // (32-bit lane offsets)
  oeh::GemmParams P;
  std::memset(&P, 0, sizeof(P));
  P.a = a; P.w = w; P.bias = bias; P.lda = lda; P.ldw = ldw; P.M = (int)(B * S); P.N = n_seg * E; P.K = K; P.pairs = pairs;
  P.E = E; P.S = S; P.H = E / 64;
  for (int i = 0; i < n_seg; ++i) {
    const oeh_proj_seg& g = segs[i];
    if (g.out == nullptr && g.y == nullptr) return OEH_EINVAL;
    if (!(g.scale > 0.0f) || g.zero_point < 0.0f || g.zero_point > 255.0f || g.zero_point != std::nearbyint(g.zero_point)) return OEH_EINVAL;
    if ((reinterpret_cast<uintptr_t>(g.out) & 15) != 0 || (reinterpret_cast<uintptr_t>(g.y) & 3) != 0) return OEH_EALIGN;
    if (g.y != nullptr && g.y_stride_row < E) return OEH_EINVAL;
    if (g.y != nullptr && (12 * g.y_stride_row + 16) * 4 >= 0xffffffffLL) return OEH_ENOTSUP;  // (32-bit lane offsets of the value stores: oeh_gemm.hip y_voff)
    oeh::GemmSeg& t = P.seg[i];
    t.alpha = g.alpha; t.out = reinterpret_cast<signed char*>(g.out); t.y = g.y; t.y_ld = g.y_stride_row; t.transpose = g.transpose ? 1 : 0;
    t.acc_add = pairs == 3 ? g.acc_add : nullptr;
    if ((reinterpret_cast<uintptr_t>(g.acc_add) & 3) != 0) return OEH_EALIGN;
    t.f.en = 1; t.f.scale = g.scale; t.f.rscale = 1.0f / g.scale; t.f.zp = g.zero_point; t.f.qmax = 255.0f; t.f.lo = -g.zero_point; t.f.hi = 255.0f - g.zero_point;
  }
  return oeh::launch_gemm(P, reinterpret_cast<hipStream_t>(stream));
}

int oeh_split_triples(const float* x, void* out_f16, int64_t rows, int32_t K, int64_t x_stride_row, void* stream) {
  if (x == nullptr || out_f16 == nullptr || rows <= 0 || K <= 0) return OEH_EINVAL;
  if ((K & 7) != 0) return OEH_ENOTSUP;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out_f16) | (uintptr_t)(x_stride_row * 4)) & 15) != 0) return OEH_EALIGN;
  return oeh::launch_split_triples(x, out_f16, rows, K, x_stride_row, reinterpret_cast<hipStream_t>(stream));
}

int oeh_split_pairs(const float* x, void* out_f16, int64_t rows, int32_t K, int64_t x_stride_row, void* stream) {
  if (x == nullptr || out_f16 == nullptr || rows <= 0 || K <= 0) return OEH_EINVAL;
  if ((K & 7) != 0) return OEH_ENOTSUP;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out_f16) | (uintptr_t)(x_stride_row * 4)) & 15) != 0) return OEH_EALIGN;
  return oeh::launch_split_pairs(x, out_f16, rows, K, x_stride_row, reinterpret_cast<hipStream_t>(stream));
}

// Diagnostic hooks: include/oeh_debug.h (NOT part of the product ABI; inert unless OEH_DEBUG_HOOKS=1 in the environment)
static bool debug_hooks_on() {
  static const bool on = [] { const char* e = std::getenv("OEH_DEBUG_HOOKS"); return e != nullptr && e[0] == '1'; }();
  return on;
}
int oeh_debug_set_variant(int off_mask, int flash_mq_force) {
  if (!debug_hooks_on()) return OEH_ENOTSUP;
  g_wide = (off_mask >> 12) & 1;
  g_variant_off = off_mask & 0xff; g_force_flash = (off_mask >> 8) & 1; g_flash_mq = flash_mq_force; g_place = (off_mask >> 9) & 1; g_force_small = (off_mask >> 10) & 1; g_no_d128_rule = (off_mask >> 11) & 1;
  { const char* e = std::getenv("OEH_HEAD_GROUP"); g_head_group = e != nullptr ? (std::atoi(e) & ~7) : 0; }
  return OEH_OK;
}
int oeh_debug_set_stamps(void* device_buffer) {
  if (!debug_hooks_on()) return OEH_ENOTSUP;
  g_stamps = static_cast<unsigned long long*>(device_buffer);
  return OEH_OK;
}

int oeh_abi_version(void) { return OEH_ABI_VERSION; }

const char* oeh_build_info(void) {
  return "liboeh_hip gfx950 (MI355X, CDNA4) v_mfma_f32_16x16x32_{f16,bf16} + v_mfma_f32_16x16x4_f32 + v_mfma_i32_16x16x64_i8; built " __DATE__ " " __TIME__ " hipcc " __VERSION__;
}

const char* oeh_strerror(int code) {
  switch (code) {
    case OEH_OK: return "ok";
    case OEH_EINVAL: return "invalid argument";
    case OEH_ENOTSUP: return "shape/dtype/option combination not supported";
    case OEH_EALIGN: return "pointer or stride not 16-byte aligned";
    case OEH_ELAUNCH: return "HIP kernel launch failed";
    case OEH_ENODEV: return "no gfx950 device";
    default: return "unknown error";
  }
}

}  // extern "C"
