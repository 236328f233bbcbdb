// Fused modified-softmax attention for gfx950 (MI355X): QK^T -> scale -> [fq] -> mask -> softmax_n ->
// [clip] -> [fq] -> PV -> [fq] -> gate -> [fq] -> (B,S,H,D) store, one kernel, nothing S x S in HBM.
//
// Replaces the eager chains bert_attention.py:222-337, opt_attention.py:204-322,
// vit_attention.py:54-75, hopfield.py:47-49 and (FQ=true) quantized_bert.py:317-434 /
// quantized_opt.py:151-270 of the reference.
//
// Design (DESIGN.md "attention kernel"):
//  * The clip and the probability fake-quant are non-linear in the FINAL normalised probability, so the
//    online-softmax rescaling trick is unusable.  Instead one wave keeps the complete score rows of its
//    16 queries in registers (Sk <= 16*NT keys -> NT accumulator tiles of 4 VGPRs; NT=32 for S=512), which
//    gives the reference's exact two-pass order (max, exp, sum, divide) with ONE pass over K and ONE over V.
//  * "Swapped" products: S^T = K Q^T and O^T = V^T P^T with v_mfma_f32_16x16x32_{f16,bf16}.  In the C/D
//    layout the query is then on the lane (col = lane&15) and the keys run over registers, so the row
//    max / sum are in-lane reductions plus two cross-lane steps, every per-row scalar (max, 1/den, gate)
//    is one VGPR, and the P^T operand of the second product is the lane's own registers (no LDS, no
//    shuffles): element j of k-slot group g is key 16*t0+4g+j (j<4) or 16*t1+4g+j-4 (j>=4); the V^T operand
//    is fetched in the same permuted key order with ds_read_b64_tr_b16.
//  * K and V stream through a 2-deep LDS ring in 64-key tiles (register-staged, 16-B loads, XOR-swizzled so
//    ds_read_b128 / ds_read_b64_tr_b16 / ds_write_b128 are conflict-free - tools/lds_bank_sim.py).
//  * One workgroup = 4 waves = 64 query rows of one (batch, head); q tiles of a head sit on one XCD
//    (block ids a multiple of 8 apart) so K/V re-reads hit that XCD's L2; heaviest causal tiles first.
#pragma once
#include "oeh_attn_params.h"

namespace oeh {

template <int D>
__device__ __forceinline__ int swz_k(int row) {
  if constexpr (D == 64) return row & 7;
  if constexpr (D == 128) return row & 15;
  if constexpr (D == 32) return (0x6C >> (((row >> 2) & 3) * 2)) & 3;  // {0,3,2,1}
  return 0;
}
template <int D>
__device__ __forceinline__ int swz_v(int row) {
  if constexpr (D == 64) return (row >> 1) & 3;
  if constexpr (D == 128) return row & 7;
  if constexpr (D == 32) return (row >> 2) & 1;
  return 0;
}

template <int IN>
__device__ __forceinline__ f4 mfma16(u4 a, u4 b, f4 c) {
  if constexpr (IN == IN_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// test-only index dump: up to 4 consecutive uint8 (byte stores: rows need not be 4-byte aligned)
__device__ __forceinline__ void dump4(unsigned char* p, unsigned int word, int nvalid) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < nvalid) p[r] = (unsigned char)(word >> (8 * r));
}

template <int NT, int D, int IN>
constexpr int occupancy_hint() { return (NT >= 32 || (D >= 128 && IN == IN_F32)) ? 2 : (NT >= 16 ? 3 : 4); }  // (fp32, D = 128: 64 KB of LDS)

// FQ: 0 none | 1 the chain on the quantiser grid (the production INT8 arithmetic of oeh_attn_fast.inl: scores and
// probabilities quantised, multiplicative scale, no padding / full mask / clip) | 2 every other fake-quant configuration
template <int NT, int D, int IN, int FQ>
__global__ __launch_bounds__(256, (occupancy_hint<NT, D, IN>())) void oeh_attn_mfma_kernel(const AttnParams P) {
  constexpr int KT = NT / 4;            // 64-key LDS tiles
  constexpr int ROWB = 2 * D;           // bytes per LDS row (16-bit elements)
  constexpr int TILEB = 64 * ROWB;
  constexpr int CPR = D / 8;            // 16-B chunks per row
  constexpr int CPT = (64 * CPR) / 256; // chunks per thread per tile
  constexpr int KS = D / 32;            // k-steps of the first product
  constexpr int DT = D / 16;            // 16-wide d tiles of the second product
  static_assert(CPT >= 1, "D >= 32");
  constexpr bool OUT16 = (IN != IN_F32);
  // fp32 storage: every operand is the fp16 pair (hi, lo) of oeh_common.h: split8 - a second LDS image `LO` bytes behind the
  // first, three MFMAs per score k-step and two per context k-step (P is an fp16 / integer operand), fp32 accuracy
  constexpr bool SPLIT = (IN == IN_F32);
  constexpr int LO = 2 * TILEB;
  constexpr int MI = SPLIT ? IN_F16 : IN;  // matrix-core operand type

  __shared__ __attribute__((aligned(16))) unsigned char lds_tile[(SPLIT ? 4 : 2) * TILEB];
  __shared__ __attribute__((aligned(16))) float lds_pad[NT * 16];

  const int bid = blockIdx.x;
  const int qt_rev = bid / P.nBHpad;
  const int bh = bid - qt_rev * P.nBHpad;
  if (bh >= P.nBH) return;
  const int qt = P.nQT - 1 - qt_rev;
  const int b = bh / P.H, h = bh - b * P.H;

  if constexpr (SPLIT) fp16_overflow_clamp();  // out-of-range fp32 operands saturate (oeh_common.h)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int q0 = qt * 64 + wave * 16;
  const int qrow = q0 + c;
  const bool qvalid = qrow < P.Sq;
  const int off = P.Sk - P.Sq;

  int kend_wg = P.Sk, kend_wave = P.Sk;
  if (P.skip_ok) {
    kend_wg = min(P.Sk, max(0, qt * 64 + 64 + off));
    kend_wave = min(P.Sk, max(0, q0 + 16 + off));
  }
  const int n_kt = (kend_wg + 63) >> 6;    // workgroup-uniform
  const int nt_wave = (kend_wave + 15) >> 4;  // wave-uniform

  // ---- Q^T operand: this lane's query row, 8 consecutive d per k-step
  u4 qf[KS], ql[SPLIT ? KS : 1];
  {
    const long qoff = (long)b * P.qs_b + (long)h * P.qs_h + (long)qrow * P.qs_s;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = u4{0, 0, 0, 0};
      if constexpr (SPLIT) {
        ql[ks] = u4{0, 0, 0, 0};
        if (qvalid) load8_split(P.q, qoff + ks * 32 + 8 * g, qf[ks], ql[ks]);
      } else {
        if (qvalid) qf[ks] = load8_as16<IN>(P.q, qoff + ks * 32 + 8 * g);
      }
    }
  }
  // ---- key-padding mask row -> LDS (zeros when absent)
  for (int i = tid; i < NT * 16; i += 256) {
    float pv = 0.0f;
    if (P.pad != nullptr && i < P.Sk) pv = load_mask(P.pad, P.pad_f16, (long)b * P.pad_sb + i);
    lds_pad[i] = pv;
  }

  const long kbase = (long)b * P.ks_b + (long)h * P.ks_h;
  const long vbase = (long)b * P.vs_b + (long)h * P.vs_h;
  f4 stage32[SPLIT ? CPT : 1][2];  // fp32 storage: the raw values wait here, split when they are committed
  u4 stage[CPT];
  auto issue_load = [&](const void* base, long boff, long srow, int tile) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int cid = tid + 256 * i;
      const int row = cid / CPR, ch = cid % CPR;
      const int key = tile * 64 + row;
      if constexpr (SPLIT) {
        stage32[i][0] = stage32[i][1] = f4{0.f, 0.f, 0.f, 0.f};
        if (key < P.Sk) {
          const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(base) + boff + (long)key * srow + ch * 8);
          stage32[i][0] = p[0];
          stage32[i][1] = p[1];
        }
      } else {
        stage[i] = u4{0, 0, 0, 0};
        if (key < P.Sk) stage[i] = load8_as16<IN>(base, boff + (long)key * srow + ch * 8);
      }
    }
  };
  auto commit_at = [&](int i, unsigned char* dst) {
    if constexpr (SPLIT) {
      u4 hi, lo;
      split8(stage32[i][0], stage32[i][1], hi, lo);
      *reinterpret_cast<u4*>(dst) = hi;
      *reinterpret_cast<u4*>(dst + LO) = lo;
    } else {
      *reinterpret_cast<u4*>(dst) = stage[i];
    }
  };
  auto commit_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int cid = tid + 256 * i;
      const int row = cid / CPR, ch = cid % CPR;
      commit_at(i, lds_tile + buf * TILEB + row * ROWB + ((ch ^ swz_k<D>(row)) << 4));
    }
  };
  auto commit_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int cid = tid + 256 * i;
      const int row = cid / CPR, ch = cid % CPR;
      commit_at(i, lds_tile + buf * TILEB + row * ROWB + (((ch >> 1) ^ swz_v<D>(row)) << 5) + ((ch & 1) << 4));
    }
  };

  // =========================== phase 1: S^T = K Q^T, all keys, into registers ===========================
  f4 s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s[t] = f4{0.f, 0.f, 0.f, 0.f};

  if (n_kt > 0) issue_load(P.k, kbase, P.ks_s, 0);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      commit_k(kt & 1);
      __syncthreads();
      if (kt + 1 < n_kt) issue_load(P.k, kbase, P.ks_s, kt + 1);
      else issue_load(P.v, vbase, P.vs_s, 0);
      const unsigned char* tb = lds_tile + (kt & 1) * TILEB;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int t = kt * 4 + sub;
        if (t < nt_wave) {
          const int row = sub * 16 + c;
          f4 acc = f4{0.f, 0.f, 0.f, 0.f}, accx = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const unsigned char* ka = tb + row * ROWB + (((ks * 4 + g) ^ swz_k<D>(row)) << 4);
            const u4 kf = *reinterpret_cast<const u4*>(ka);
            acc = mfma16<MI>(kf, qf[ks], acc);
            if constexpr (SPLIT) {
              accx = mfma16<MI>(kf, ql[ks], accx);
              accx = mfma16<MI>(*reinterpret_cast<const u4*>(ka + LO), qf[ks], accx);
            }
          }
          if constexpr (SPLIT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(accx[r], kSplitDown, acc[r]);
          }
          s[t] = acc;
        }
      }
    }
  }

  // =========================== phase 2: elementwise chain + row statistics ===========================
  // lane (c,g) owns query row `qrow`, keys 16t+4g+r.  Every optional step is guarded by a wave-uniform test evaluated
  // once per 16-key tile (the option itself, or "this tile can hold such a key"), so that the per-element work is the
  // arithmetic of the steps that are on - the order of the steps is the reference's.
  const float mask_min = P.mask_min;
  const int klim = qrow + off;  // last key a causal row may see
  const bool use_div = P.scale_div != 0.0f;
  const bool has_pad = P.pad != nullptr, has_full = P.full != nullptr;
  const bool fq_s_on = FQ && P.fq_s.en, fq_p_on = FQ && P.fq_p.en;
  const bool dump_s = fq_s_on && P.fq_s.dump != nullptr, dump_p = fq_p_on && P.fq_p.dump != nullptr;
  const int t_causal = P.causal ? max(0, q0 + off + 1) >> 4 : NT;  // first tile with a key the wave's first row must not see
  const int t_tail = P.Sk >> 4;                                     // first tile with a key >= Sk
  float m = -__builtin_inff();
  u2 ph[NT];
  // The chain on the quantiser grid (oeh_attn_fast.inl, FQ variant - the production INT8 path; this kernel serves its
  // index dumps and must give the same bits): scores and probabilities quantised, multiplicative scale, masks none / causal.
  if constexpr (FQ == 1) {
    constexpr float RELMASK = -1.0e30f;
    const float k1 = P.scale * P.fq_s.rscale, slo = kGridMagic + P.fq_s.lo, shi = kGridMagic + P.fq_s.hi;
    const int klime = P.causal ? min(klim, P.Sk - 1) : P.Sk - 1;
    float mr = RELMASK;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt_wave) {
        const int key0 = 16 * t + 4 * g;
        f4 rel;
#pragma unroll
        for (int r = 0; r < 4; ++r) rel[r] = grid_rel_m(s[t][r], k1, slo, shi) - kGridMagic;  // (oeh_common.h; here as the plain integer)
        if (dump_s && qvalid) dump4(P.fq_s.dump + (((long)b * P.H + h) * P.Sq + qrow) * P.Sk + key0, fq_dump_word(rel, P.fq_s), P.Sk - key0);
        if (t >= t_causal || t >= t_tail) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + r > klime) rel[r] = RELMASK;
        }
        s[t] = rel;
        mr = __builtin_fmaxf(__builtin_fmaxf(mr, __builtin_fmaxf(rel[0], rel[1])), __builtin_fmaxf(rel[2], rel[3]));
      }
    }
    mr = row4_max(mr);
    m = mr * P.fq_s.scale;
    const float c2 = P.fq_s.c2;
    f4 sum4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt_wave) {
        f4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = s[t][r] - mr;
          e[r] = __builtin_amdgcn_exp2f(d * c2);
        }
        s[t] = e;
        sum4 = sum4 + e;
      }
    }
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    sum = row4_sum(sum);
    float den = sum;
    if (P.base != 0) den = sum + exp_acc(m * -1.0f);
    const float inv_g = 1.0f / den;
    const float cinv = inv_g * P.fq_p.rscale, plo = P.fq_p.lo, phi = P.fq_p.hi;
    const float clip_iw = inv_g * P.clip_w, clip_g = P.clip_g;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ph[t] = u2{0u, 0u};
      if (t < nt_wave) {
        const int key0 = 16 * t + 4 * g;
        f4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (P.clip) {  // (the full-row kernel's clipped grid form, same operations)
            const float pc = __builtin_amdgcn_fmed3f(__builtin_fmaf(s[t][r], clip_iw, clip_g), 0.0f, 1.0f);
            pv[r] = __builtin_amdgcn_fmed3f(__builtin_rintf(pc * P.fq_p.rscale), plo, phi);
          } else {
            pv[r] = __builtin_amdgcn_fmed3f(__builtin_rintf(s[t][r] * cinv), plo, phi);
          }
        }
        if (dump_p && qvalid) dump4(P.fq_p.dump + (((long)b * P.H + h) * P.Sq + qrow) * P.Sk + key0, fq_dump_word(pv, P.fq_p), P.Sk - key0);
        if constexpr (MI == IN_BF16) {
          ph[t].x = pack2_bf16(pv[0], pv[1]);
          ph[t].y = pack2_bf16(pv[2], pv[3]);
        } else {
          ph[t].x = pack2_f16(pv[0], pv[1]);
          ph[t].y = pack2_f16(pv[2], pv[3]);
        }
      }
    }
  } else {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt_wave) {
      const int key0 = 16 * t + 4 * g;
      f4 x = s[t];
      if (use_div) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = x[r] / P.scale_div;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = x[r] * P.scale;
      }
      if constexpr (FQ) {
        if (fq_s_on) {
          const f4 rel = fq_rel4(x, P.fq_s);
          if (dump_s && qvalid) dump4(P.fq_s.dump + (((long)b * P.H + h) * P.Sq + qrow) * P.Sk + key0, fq_dump_word(rel, P.fq_s), P.Sk - key0);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = P.fq_s.scale * rel[r];
        }
      }
      if (has_pad) {
        const f4 padv = *reinterpret_cast<const f4*>(&lds_pad[key0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = x[r] + padv[r];
      }
      if (has_full) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (qvalid && key0 + r < P.Sk) x[r] = x[r] + load_mask(P.full, P.full_f16, (long)b * P.full_sb + (long)qrow * P.full_sq + key0 + r);
      }
      if (t >= t_causal) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + r > klim) x[r] = x[r] + mask_min;
      }
      if (P.clamp_min) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = __builtin_fmaxf(x[r], mask_min);
      }
      if (t >= t_tail) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + r >= P.Sk) x[r] = -__builtin_inff();
      }
      s[t] = x;
      m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(x[0], x[1])), __builtin_fmaxf(x[2], x[3]));
    }
  }
  m = row4_max(m);

  float sum = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt_wave) {
      if constexpr (FQ) {
        const f2 m2 = f2{m, m};
        const f2 e01 = exp_acc_nonpos2(f2{s[t][0], s[t][1]} - m2), e23 = exp_acc_nonpos2(f2{s[t][2], s[t][3]} - m2);
        s[t] = f4{e01[0], e01[1], e23[0], e23[1]};
        sum += e01[0];
        sum += e01[1];
        sum += e23[0];
        sum += e23[1];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp_fast(s[t][r] - m);
          s[t][r] = e;
          sum += e;
        }
      }
    }
  }
  sum = row4_sum(sum);
  float den = sum;
  if (P.base != 0) den = sum + exp_acc(m * -1.0f);  // softmax_1: + 1*exp(-max)  (softmax_1.py:18-20)
  const float inv = 1.0f / den;

  // probabilities -> [clip] -> [fq] -> 16-bit P^T operand, two 16-key tiles per 32-key k-step
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ph[t] = u2{0u, 0u};
    if (t < nt_wave) {
      const int key0 = 16 * t + 4 * g;
      f4 pv;
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[r] = s[t][r] * inv;
      if (P.clip) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = pv[r] * P.clip_w;
          p = p + P.clip_g;
          pv[r] = __builtin_fminf(__builtin_fmaxf(p, 0.0f), 1.0f);
        }
      }
      if constexpr (FQ) {
        if (fq_p_on) {
          pv = fq_rel4(pv, P.fq_p);  // integer valued (idx - zp): exact in f16/bf16; scale applied after the product
          if (dump_p && qvalid) dump4(P.fq_p.dump + (((long)b * P.H + h) * P.Sq + qrow) * P.Sk + key0, fq_dump_word(pv, P.fq_p), P.Sk - key0);
        }
      }
      if (t >= t_tail) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + r >= P.Sk) pv[r] = 0.0f;
      }
      if constexpr (MI == IN_BF16) {
        ph[t].x = pack2_bf16(pv[0], pv[1]);
        ph[t].y = pack2_bf16(pv[2], pv[3]);
      } else {
        ph[t].x = pack2_f16(pv[0], pv[1]);
        ph[t].y = pack2_f16(pv[2], pv[3]);
      }
    }
  }

  }

  // =========================== phase 3: O^T = V^T P^T ===========================
  f4 o[DT], ox[SPLIT ? DT : 1];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f4{0.f, 0.f, 0.f, 0.f};
  if constexpr (SPLIT) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ox[dt] = f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      const int buf = (n_kt + kt) & 1;
      commit_v(buf);
      __syncthreads();
      if (kt + 1 < n_kt) issue_load(P.v, vbase, P.vs_s, kt + 1);
      const unsigned char* tb = lds_tile + buf * TILEB;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t0 = kt * 4 + 2 * u;
        if (t0 < nt_wave) {
          const u4 pb = u4{ph[t0].x, ph[t0].y, ph[t0 + 1].x, ph[t0 + 1].y};
          const int row = 32 * u + 4 * g + (c >> 2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const unsigned char* a0 = tb + row * ROWB + ((dt ^ swz_v<D>(row)) << 5) + ((c & 3) << 3);
            const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
            const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 16 * ROWB));
            const u2 l2 = __builtin_bit_cast(u2, lo), h2 = __builtin_bit_cast(u2, hi);
            const u4 va = u4{l2.x, l2.y, h2.x, h2.y};
            o[dt] = mfma16<MI>(va, pb, o[dt]);
            if constexpr (SPLIT) {
              const s4 lol = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + LO));
              const s4 hil = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + LO + 16 * ROWB));
              const u2 l3 = __builtin_bit_cast(u2, lol), h3 = __builtin_bit_cast(u2, hil);
              ox[dt] = mfma16<MI>(u4{l3.x, l3.y, h3.x, h3.y}, pb, ox[dt]);
            }
          }
        }
      }
    }
  }
  if constexpr (SPLIT) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] = __builtin_fmaf(ox[dt][r], kSplitDown, o[dt][r]);
  }

  // =========================== epilogue: [scale_p] [fq] gate [fq] store ===========================
  // lane (c,g) holds O[qrow][16dt + 4g + r]
  if (!qvalid) return;
  float gatev = 1.0f;
  if (P.gate != nullptr) gatev = P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qrow * P.gs_s];
  const long ooff = (long)b * P.os_b + (long)h * P.os_h + (long)qrow * P.os_s;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    float ov[4];
    unsigned int dump_word = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = o[dt][r];
      if constexpr (FQ) {
        if (P.fq_p.en) x = P.fq_p.scale * x;
        if (P.fq_c.en && P.ctx_before_gate) {
          const float idx = fq_index(x, P.fq_c);
          dump_word |= ((unsigned int)idx) << (8 * r);
          x = fq_out(idx, P.fq_c);
        }
      }
      if (P.gate != nullptr) x = x * gatev;
      if constexpr (FQ) {
        if (P.fq_c.en && !P.ctx_before_gate) {
          const float idx = fq_index(x, P.fq_c);
          dump_word |= ((unsigned int)idx) << (8 * r);
          x = fq_out(idx, P.fq_c);
        }
      }
      ov[r] = x;
    }
    const int d0 = 16 * dt + 4 * g;
    if constexpr (OUT16) {
      u2 w;
      if constexpr (IN == IN_BF16) {
        w.x = pack2_bf16(ov[0], ov[1]);
        w.y = pack2_bf16(ov[2], ov[3]);
      } else {
        w.x = pack2_f16(ov[0], ov[1]);
        w.y = pack2_f16(ov[2], ov[3]);
      }
      *reinterpret_cast<u2*>(reinterpret_cast<unsigned short*>(P.o) + ooff + d0) = w;  // 8-B pieces: write-through would cost 2.7x per byte
    } else {
      store_wt16(reinterpret_cast<float*>(P.o) + ooff + d0, u4{f32_bits(ov[0]), f32_bits(ov[1]), f32_bits(ov[2]), f32_bits(ov[3])});
    }
    if constexpr (FQ) {
      if (P.fq_c.en && P.fq_c.dump != nullptr) dump4(P.fq_c.dump + (((long)b * P.H + h) * P.Sq + qrow) * D + d0, dump_word, 4);
    }
  }
}

// ---- explicit instantiations + launcher table ---------------------------------------------------------------
template <int NT, int D, int IN, int FQ>
static int launch_one(const AttnParams& P, hipStream_t st) {
  const unsigned grid = (unsigned)(P.nQT * P.nBHpad);
  hipLaunchKernelGGL((oeh_attn_mfma_kernel<NT, D, IN, FQ>), dim3(grid), dim3(256), 0, st, P);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

template <int NT, int D>
static int launch_nt_d(const AttnParams& P, int in, bool fq, hipStream_t st) {
  if (fq) {
    const bool grid_chain = P.fq_s.en && P.fq_p.en && P.pad == nullptr && P.full == nullptr && P.scale_div == 0.0f;
    if (grid_chain) {
      switch (in) {
        case IN_F16: return launch_one<NT, D, IN_F16, 1>(P, st);
        case IN_BF16: return launch_one<NT, D, IN_BF16, 1>(P, st);
        default: return launch_one<NT, D, IN_F32, 1>(P, st);
      }
    }
    switch (in) {
      case IN_F16: return launch_one<NT, D, IN_F16, 2>(P, st);
      case IN_BF16: return launch_one<NT, D, IN_BF16, 2>(P, st);
      default: return launch_one<NT, D, IN_F32, 2>(P, st);
    }
  }
  switch (in) {
    case IN_F16: return launch_one<NT, D, IN_F16, 0>(P, st);
    case IN_BF16: return launch_one<NT, D, IN_BF16, 0>(P, st);
    default: return launch_one<NT, D, IN_F32, 0>(P, st);
  }
}

template <int D>
static int launch_d(const AttnParams& P, int in, bool fq, hipStream_t st) {
  if (P.Sk <= 128) return launch_nt_d<8, D>(P, in, fq, st);
  if (P.Sk <= 256) return launch_nt_d<16, D>(P, in, fq, st);
  return launch_nt_d<32, D>(P, in, fq, st);
}

}  // namespace oeh
