// The INT8 attention core on the INTEGER matrix cores: q, k, v arrive as 8-bit quantiser indices - what the
// reference's QuantLinear projections really produce (hijacker.py:78-127: the q_proj / k_proj / v_proj outputs are fake-quantised
// onto per-tensor 8-bit grids before quantized_opt.py:151 multiplies them) - both contractions run on v_mfma_i32_16x16x64_i8
// with exact i32 accumulation, and the three activation quantisers of the attention class (scores, probabilities, context)
// are the chain on the quantiser grid of oeh_attn_fast.inl.  Head dim 64: one MFMA k-step is a whole score.
//
// Storage (dtype OEH_I8): an element is the CENTRED index c = idx - 128 (int8), its value x = scale * (c + 128 - zp).
// The integer matrix cores are signed 8 x 8 bit; an asymmetric grid's idx - zp does not fit (zp is data dependent), so the
// products are formed on the centred indices and the offsets cq = 128 - zp_q (...) are put back exactly:
//     sum_d (a + cq)(b + ck) = sum_d a b + ck sum_d a + cq sum_d b + D cq ck
// with sum_d b (per key) from one more MFMA of the same K fragment against a ones operand - it lands in the accumulator
// layout the scores have - and sum_d a (per query) from the lane's own Q registers.  The same identity, with the sums over
// keys, serves P V; the probabilities' indices are formed, rounded (RNE) and saturated to [0, 255] by v_cvt_pk_u8_f32,
// four to a register, and ARE the second product's operand.
//
// Layouts: q, k (B,H,S,64) views as everywhere; v TRANSPOSED, (B,H,64,Sk) with the keys contiguous (v_stride = batch, head,
// d row): the second product sums over keys, so its operand wants 16 consecutive keys of one d per lane, and an LDS-DMA
// cannot transpose.  LDS: K tiles [64 keys][64 B] and V^T tiles [64 d][64 B], 6 + 6 slots of 4 KB, fed by LDS-DMA exactly
// like the 16-bit kernels; the 16-B chunk of a row is XOR-swizzled with {0,3,2,1}[row group] so that ds_read_b128 of either
// operand is conflict-free (the row groups differ: keys are dealt to MFMA rows as key = 16 (row >> 2) + 4 t + (row & 3) so
// that a lane ends up with 16 CONSECUTIVE keys of its query - the k order the V^T operand has).
// Masks: none | analytic causal | a key-padding vector (PAD variants: BERT's (B,1,1,S) mask, OPT's padded batches) whose
// entries are 0 or <= -1e4 (HF's extended masks: 0 / finfo.min - the host checks it once per mask tensor): a padded key
// carries the sentinel like a causally hidden one, its exponential is exactly 0 as in the reference.  Everything else of the
// INT8 configuration (clipping with gamma > 0, other head dims, arbitrary additive masks) runs the fake-quant variants of the 16-bit / fp32
// kernels on dequantised values.
//
// The vector arithmetic per score element is the cost of this kernel (round 2: 21.7 wave-instructions per element, the
// launch VALU-issue bound), so the chain is written for the instruction count:
//   * the offset terms ride the matrix cores: the accumulator starts at the query's integer constant ck sum_d a + D cq ck and
//     a second MFMA of the same K fragment against an operand of cq in every byte adds cq sum_d b - the i32 accumulator IS
//     sum_d (a + cq)(b + ck), exactly; no multiply-add per element;
//   * index of the score by ONE fused multiply-add against 1.5 * 2^23: RN(S k1 + M) = M + rint(S k1) (ties to even on the
//     exact product), clamped in that domain (v_med3 against M + lo, M + hi; a product too large for the trick is beyond the
//     clamp on the same side) - differences of such values are the exact integer differences the exponent needs;
//   * row maximum as v_max3 chains; the causal / tail test of a diagonal tile is one compare + select per element against a
//     per-lane key limit, and only there.
// Measured and dropped (round 3): q tiles in PAIRS per workgroup (nQT-1-j, then j: every workgroup the same nQT + 1 key tiles, half as
// many workgroups, one dispatch round, the second tile's K / V^T L2 hits) - 17.8 against 17.1-17.4 us at S = 256, 14.3-14.6 against
// 13.8-14.0 at S = 128: the dynamic two-round dispatch, heaviest first, already balances, and the pair serialises two prologues.
// DUMP variant (tests): the three index tensors are written out as uint8 (include/oeh.h: oeh_fq.dump_idx), scores for every
// key (the reference quantises before the mask is added).
#include "oeh_attn_fast.inl"

namespace oeh {

typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int perm4(int x) { return (0x6C >> ((x & 3) * 2)) & 3; }  // {0,3,2,1}

template <int NT, int OUT, bool DUMP, bool CQ2, bool PAD>
__global__ __launch_bounds__(256, 3) void oeh_attn_i8_kernel(const AttnParams P) {
  constexpr int D = 64, KT = NT / 4, TILEB = 64 * 64, DT = 4;
  // The K and the V^T phases have eight MFMAs per wave and tile between two barriers: they run at the pace the tiles ARRIVE.
  // R slots per ring, PF tiles requested ahead (PF <= R - 1: a slot is refilled only after the barrier that follows its last
  // readers).  PF = 2 (round 3; 4 before): a launch is one burst of every workgroup's requests against HBM, and tiles requested
  // far ahead only delay the tiles every workgroup needs first - same-process: S = 128 0.94, S = 256 0.97, S = 512 0.99 against
  // PF = 4; PF = 3 in between; PF = 1 exposes the latency (S = 512 +8 %).  (Round 4, with the V^T tiles no longer part of this stream:
  // PF = 3 for the K tiles alone 0.995 ... 1.02 of PF = 2 - no change.)
  constexpr int R = 6, PF = 2;
  constexpr float RELMASK = -1.0e30f;
  constexpr bool OUT32 = (OUT == IN_F32);
  constexpr bool OUT8 = (OUT == 3);  // the context quantiser's CENTRED INDICES idx - 128 as int8 (fq.ctx_emit_index with o_dtype OEH_I8)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * R * TILEB];
  // PAD: per key +big (visible), the sentinel (padded) or -inf (key >= Sk: not even a masked key - a fully masked row of the
  // vanilla softmax is uniform over the Sk keys, as in the reference): rel = min(rel, flag), one instruction per element
  __shared__ __attribute__((aligned(16))) float lds_pad[PAD ? NT * 16 : 4];

  // (kernel arguments of the prologue in one round of scalar loads, block id decoded without integer divisions: oeh_attn_fast.inl)
  asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.nBHpad), "s"(P.nQT), "s"(P.nBH), "s"(P.H), "s"(P.Sq), "s"(P.Sk), "s"(P.skip_ok), "s"(P.causal),
               "s"(P.magic_nbh), "s"(P.magic_h), "s"(P.qs_b), "s"(P.qs_h), "s"(P.qs_s), "s"(P.ks_b), "s"(P.ks_h), "s"(P.ks_s), "s"(P.vs_b), "s"(P.vs_h),
               "s"(P.vs_s), "s"(P.i8_cq), "s"(P.i8_ck), "s"(P.i8_k1));
  const int bid = blockIdx.x;
  int qt_rev, bhr_;
  div_magic((unsigned)bid, (unsigned)P.nBHpad, P.magic_nbh, qt_rev, bhr_);
  // Blocks b and b + 8 run on one XCD.  In the (B,S,H*64) int8 layout two neighbouring heads share every 128-byte line
  // of q and k, so heads 2i and 2i+1 are given block ids 8 apart (nBHpad is a multiple of 16): the second half of a line is
  // an L2 hit instead of a second HBM fetch (57 -> 4x MB per launch on the OPT shape against 44 MB algorithmic).
  const int bhr = bhr_;
  const int bh = (bhr & ~15) | ((bhr & 7) << 1) | ((bhr >> 3) & 1);
  if (bh >= P.nBH) return;
  const int qt = P.nQT - 1 - qt_rev;
  int b, h;
  div_magic((unsigned)bh, (unsigned)P.H, P.magic_h, b, h);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int q0 = qt * 64 + wave * 16, qrow = q0 + c;
  const bool qvalid = qrow < P.Sq;
  const int Sk = P.Sk, off = P.Sk - P.Sq, causal = P.causal;

  int kend_wg = Sk;
  if (P.skip_ok) kend_wg = min(Sk, max(0, qt * 64 + 64 + off));
  const int n_kt = (kend_wg + 63) >> 6;


  // ---- LDS-DMA stream, K tiles then V^T tiles, one 1-KiB piece per wave and tile: lane -> (row of the piece, 16-B chunk)
  const signed char* kbase = reinterpret_cast<const signed char*>(P.k) + bh_offset(b, P.ks_b, h, P.ks_h);
  const signed char* vbase = reinterpret_cast<const signed char*>(P.v) + bh_offset(b, P.vs_b, h, P.vs_h);
  const int prow = wave * 16 + (lane >> 2), pch = lane & 3;
  const unsigned lds_base = lds_offset(lds);
  // scalar base + constant per-lane byte offset (oeh_common.h: glds16_s): the bases advance by one tile per request, nothing
  // is recomputed per lane (the first version rebuilt a clamped 64-bit address per request: ~2 vector instructions per score
  // element of a kernel that is bound by them).  Only a ragged last tile (Sk not a multiple of 64) adjusts the lane offsets:
  // K rows past Sk are redirected to row Sk - 1, V^T chunks past Sk to the last one (finite data, probability 0).
  const unsigned koff = (unsigned)(prow * (int)P.ks_s + (pch ^ perm4(prow >> 4)) * 16);
  const unsigned voff = (unsigned)(prow * (int)P.vs_s + (pch ^ perm4(prow >> 2)) * 16);
  // the lane offsets of the ragged LAST tile (Sk not a multiple of 64), formed once: K rows past Sk -> row Sk - 1, V^T chunks past
  // Sk -> the last chunk; a request then costs one select on a wave-uniform condition, no per-lane arithmetic
  unsigned koff_tail = koff, voff_tail = voff;
  {
    const int t_last = (Sk - 1) >> 6;
    const int over_k = 64 * t_last + prow - (Sk - 1);
    if (over_k > 0) koff_tail -= (unsigned)(over_k * (int)P.ks_s);
    const int over_v = 4 * t_last + (pch ^ perm4(prow >> 2)) - ((Sk >> 4) - 1);  // (Sk is a multiple of 16: host)
    if (over_v > 0) voff_tail -= 16u * (unsigned)over_v;
  }
  const signed char* kcur = kbase;
  const signed char* vcur = vbase;
  const int kstep = 64 * (int)P.ks_s;
  // Round 4: two streams.  K tiles as before (tile t -> K slot t % R, PF ahead).  The V^T tiles are ALL requested at once when the K
  // phase ends - they land during phase 2, a microsecond and more of pure vector work in which the memory system had nothing to do -
  // and they all stay resident (V^T tile j -> V slot j for j < R, the freed K slots j - R beyond: n_kt <= 8 <= 2 R - 4), so the V^T
  // phase has ONE wait + barrier instead of one per tile (it used to run at the pace its tiles arrived: PF = 2 requests in flight).
  int nk = 0;
  auto issue_k = [&]() {
    const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((nk % R) * TILEB + wave * 1024));
    glds16_s(kcur, (64 * nk + 64 > Sk) ? koff_tail : koff, slot);
    kcur += kstep;
    ++nk;
  };
  auto v_slot = [&](const int j) { return j < R ? R + j : j - R; };
  int nv = 0;
  auto issue_v = [&]() {
    const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(v_slot(nv) * TILEB + wave * 1024));
    glds16_s(vcur, (64 * nv + 64 > Sk) ? voff_tail : voff, slot);
    vcur += 64;
    ++nv;
  };
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < n_kt) issue_k();
  auto wait_k = [&](const int i) {  // K tile i has landed: all but the PF - 1 younger requests of this wave (K tiles, then the first V^T tiles)
    const int younger = min(PF, n_kt) - 1;   // (every step so far has issued one request: the stream is min(PF, n_kt) ahead of tile i)
    (void)i;
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(PF <= 4 && PF <= R - 1 && NT / 4 <= 2 * R - 4, "wait_k's immediates; every V^T tile has a slot of its own");

  // ---- Q: global -> registers in the B-operand layout (query q0 + c, head dims 16 g ..), and its row sum
  i4 qf;
  {
    const signed char* qp = reinterpret_cast<const signed char*>(P.q) + bh_offset(b, P.qs_b, h, P.qs_h) + (long)min(qrow, P.Sq - 1) * P.qs_s + 16 * g;
    qf = *reinterpret_cast<const i4*>(qp);
  }
  const int ones = 0x01010101;
  int asum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) asum = __builtin_amdgcn_sdot4(qf[j], ones, asum, false);
  asum = row4_sum(asum);
  if constexpr (PAD) {  // (compiler-visible loads: its wait for them also covers the first transfers, which the first tile wait needs anyway)
    for (int i = threadIdx.x; i < NT * 16; i += 256) {
      float f = -__builtin_inff();
      if (i < Sk) f = load_mask(P.pad, P.pad_f16, (long)b * P.pad_sb + i) <= -1.0e4f ? RELMASK : 3.0e38f;
      lds_pad[i] = f;
    }
  }
  const int cq = P.i8_cq, ck = P.i8_ck, cv = P.i8_cv, cp = P.i8_cp;   // 128 - zero point of q, k, v and of the probabilities
  const float k1 = P.i8_k1;  // (quotient of the score by the score grid's step) = k1 * sum_d (a + cq)(b + ck)
  // the query's part of the offsets, in the accumulator from the start: ck sum_d a + D cq ck (|.| < 2^23)
  const int rowq = ck * asum + D * cq * ck;
  i4 cinit = i4{rowq, rowq, rowq, rowq};
  asm volatile("" : "+v"(cinit));
  // ... and the key's part, cq sum_d b, by a second MFMA against cq in every byte (cq = 128, a q grid with zero point 0,
  // does not fit a signed byte: the CQ2 variant adds 64 twice)
  const int cqb = ((CQ2 ? 64 : cq) & 0xff) * 0x01010101;
  i4 cq4 = i4{cqb, cqb, cqb, cqb};
  asm volatile("" : "+v"(cq4));

  // =========================== phase 1: S^T = K Q^T (i32, all offsets in), kept as integer-valued floats ===========================
  f4 s[NT];
  const int krow_base = 16 * (c >> 2) + (c & 3);                 // MFMA row c of sub-tile t holds key krow_base + 4 t
  const int kswz = (g ^ perm4(c >> 2)) << 4;                     // perm4((key >> 4) & 3), key >> 4 == c >> 2
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      wait_k(kt);
      barrier_mem();
      if (kt + PF < n_kt) issue_k();
      else issue_v();   // the K stream has run out: the first PF V^T tiles (V slots: no K slot is touched before the barrier below)
      const unsigned char* tb = lds + (kt % R) * TILEB;
      i4 kf[4], acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const i4*>(tb + (krow_base + 4 * t) * 64 + kswz);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf[t], qf, cinit, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf[t], cq4, acc[t], 0, 0, 0);
      if constexpr (CQ2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf[t], cq4, acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f4 f;
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = (float)acc[t][r];
        s[kt * 4 + t] = f;
      }
    }
  }

  barrier_mem();   // every wave has read its last K fragments: the K slots are free for V^T tiles R, R + 1
  while (nv < n_kt) issue_v();

  // =========================== phase 2: the chain on the quantiser grid ===========================
  // element (kt, t, r) of lane (c, g) is key 64 kt + 16 g + 4 t + r of query q0 + c.  rel = idx - zp is carried as
  // M + rel, M = 1.5 * 2^23 (oeh_common.h: grid_rel_m); a key the row must not see carries a sentinel far below.
  // (Measured and dropped: the quantiser of a tile right behind its MFMAs inside the K loop, and a tile's probability indices
  // inside the V loop - vector work for the wave while its next tile is on the way: 22.7 and 26.0 us against 22.5 us for the
  // separate passes; the second form spills.)
  constexpr float MAGIC = kGridMagic;
  const int klimc = qrow + off;
  const int klime = causal ? min(klimc, Sk - 1) : Sk - 1;   // last key this lane's row sees
  const int kt_causal = causal ? (max(0, q0 + off + 1) >> 6) : KT;
  const int kt_tail = PAD ? KT : (Sk >> 6);                 // (PAD: the flags carry the tail)
  const float slo = MAGIC + P.fq_s.lo, shi = MAGIC + P.fq_s.hi;
  const int klim_g = (PAD ? (causal ? klimc : Sk - 1) : klime) - 16 * g;   // ... relative to the lane's first key of a tile
  float mr = RELMASK;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      const bool open_tile = kt < kt_causal && kt < kt_tail;
      const int lim = klim_g - 64 * kt;                     // element (t, r) is masked when 4 t + r > lim
      // the score index in place; a tile with hidden keys then masks in place under ONE wave-uniform branch.  (Round 3 wrote the tile's body twice -
      // an open and a masked form - to avoid a test per four elements; the two forms left their results in different registers and the join cost
      // eight v_mov_b64 per tile, 0.5 vector instructions per score element: round 5, from the disassembly.)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt * 4 + t][r] = grid_rel_m(s[kt * 4 + t][r], k1, slo, shi);
        if constexpr (DUMP) {
          if (P.fq_s.dump != nullptr && qvalid) {
            const int zi = (int)P.fq_s.zp - 0x4B400000;
            const unsigned w = (unsigned)((int)f32_bits(s[kt * 4 + t][0]) + zi) | ((unsigned)((int)f32_bits(s[kt * 4 + t][1]) + zi) << 8) |
                               ((unsigned)((int)f32_bits(s[kt * 4 + t][2]) + zi) << 16) | ((unsigned)((int)f32_bits(s[kt * 4 + t][3]) + zi) << 24);
            const int key0 = 64 * kt + 16 * g + 4 * t;
            if (key0 < Sk) *reinterpret_cast<unsigned*>(P.fq_s.dump + (((long)b * P.H + h) * P.Sq + qrow) * Sk + key0) = w;
          }
        }
      }
      if (!open_tile) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt * 4 + t][r] = (4 * t + r > lim) ? RELMASK : s[kt * 4 + t][r];
      }
      if constexpr (PAD) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f4 flag = *reinterpret_cast<const f4*>(&lds_pad[64 * kt + 16 * g + 4 * t]);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt * 4 + t][r] = __builtin_fminf(s[kt * 4 + t][r], flag[r]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {  // (v_max3 from the plain builtins: the operands are v_med3 / select results, known canonical - no
        mr = __builtin_fmaxf(__builtin_fmaxf(mr, s[kt * 4 + t][0]), s[kt * 4 + t][1]);  // canonicalising v_max, and none of the
        mr = __builtin_fmaxf(__builtin_fmaxf(mr, s[kt * 4 + t][2]), s[kt * 4 + t][3]);  // s_nop the compiler puts behind inline asm)
      }
    }
  }
  mr = row4_max(mr);
  const float m = (mr - MAGIC) * P.fq_s.scale;               // the reference's row maximum, fl(scale * rel_max)
  const float c2 = P.fq_s.c2;
  f4 sum4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f((s[kt * 4 + t][r] - mr) * c2);
        s[kt * 4 + t] = e;
        sum4 = sum4 + e;
      }
    }
  }
  float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
  sum = row4_sum(sum);
  float den = sum;
  if (P.base != 0) den = sum + exp_acc(m * -1.0f);
  const float inv_den = 1.0f / den;
  const float cinv = inv_den * P.fq_p.rscale, pzp = P.fq_p.zp;
  const float clip_iw = inv_den * P.clip_w, clip_g = P.clip_g, prs = P.fq_p.rscale;  // clipped softmax (round 3): one clamped fma in front of the index
  const bool clipped = P.clip != 0;
  // probabilities: index = sat_u8(rne(e * cinv + zp)) by v_cvt_pk_u8_f32, four keys to a register, centred by the XOR;
  // a masked key has e == 0, index zp, value 0.  The centred indices' row sum goes with them.
  int psum = 0;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        unsigned w = 0u;
        if (clipped) {  // clip(p (eta - gamma) + gamma, 0, 1) (models/softmax.py:16-19; gamma <= 0: a masked key, e = 0, stays 0), then the index
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pc = __builtin_amdgcn_fmed3f(__builtin_fmaf(s[kt * 4 + t][r], clip_iw, clip_g), 0.0f, 1.0f);
            w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(pc, prs, pzp), r, w);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(s[kt * 4 + t][r], cinv, pzp), r, w);
        }
        if constexpr (DUMP) {
          const int key0 = 64 * kt + 16 * g + 4 * t;
          if (P.fq_p.dump != nullptr && qvalid && key0 < Sk) *reinterpret_cast<unsigned*>(P.fq_p.dump + (((long)b * P.H + h) * P.Sq + qrow) * Sk + key0) = w;
        }
        w ^= 0x80808080u;
        psum = __builtin_amdgcn_sdot4((int)w, ones, psum, false);
        s[kt * 4 + t][0] = bits_f32(w);
      }
    }
  }
  psum = row4_sum(psum);

  // =========================== phase 3: O^T = V^T P^T (i32) and the key sums of V^T ===========================
  // sum_k (v + cv)(p + cp) = o + cp vsum + cv psum + n cv cp: the query's constant cv psum + n cv cp is what the accumulators start at
  const int rowc = cv * psum + 64 * n_kt * cv * cp;
  i4 o[DT], vs[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    o[dt] = i4{rowc, rowc, rowc, rowc};
    vs[dt] = i4{0, 0, 0, 0};
  }
  i4 ones4 = i4{ones, ones, ones, ones};
  asm volatile("" : "+v"(ones4));  // one register quad for the phase (the compiler would rebuild the constant before every use)
  const int vswz = (g ^ perm4(c >> 2)) << 4;                      // perm4((d >> 2) & 3), d = 16 dt + c
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      if (kt == 0) {  // every V^T tile of every wave has landed (they were requested before phase 2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier_mem();
      }
      const unsigned char* tb = lds + v_slot(kt) * TILEB;
      const i4 pb = i4{(int)f32_bits(s[kt * 4 + 0][0]), (int)f32_bits(s[kt * 4 + 1][0]), (int)f32_bits(s[kt * 4 + 2][0]), (int)f32_bits(s[kt * 4 + 3][0])};
      i4 vf[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) vf[dt] = *reinterpret_cast<const i4*>(tb + (16 * dt + c) * 64 + vswz);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        o[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf[dt], pb, o[dt], 0, 0, 0);
        vs[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf[dt], ones4, vs[dt], 0, 0, 0);
      }
    }
  }

  // =========================== epilogue: offsets back, [fq] gate [fq], store ===========================
  // lane (c, g) holds O[q0 + c][16 dt + 4 g + r] (+ the row constant);  cp vsum is still to add
  const float so = P.i8_so;  // scale_p * scale_v
  float gatev = 1.0f;
  if (P.gate != nullptr && qvalid) gatev = P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qrow * P.gs_s];
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int ce = lane_e & 15, ge = lane_e >> 4;
  constexpr int ROWB = OUT8 ? D : 2 * D;
  unsigned char* ebase = lds + wave * (16 * ROWB);  // 16-bit / int8 output: staged through K slots 0 and 1, whole rows stored
  if constexpr (!OUT32) {
    if (n_kt > R) barrier_mem();  // (those slots hold V^T tiles R, R + 1 of a long row: every wave must have left them)
  }
  float xs[DT * 4], crel[DUMP ? DT * 4 : 1];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xs[dt * 4 + r] = so * (float)(o[dt][r] + __mul24(cp, vs[dt][r]));  // |vsum| <= 512 * 128
      if constexpr (DUMP) crel[dt * 4 + r] = 0.0f;
    }
  ctx_chain<DT * 4, DUMP>(xs, P.fq_c, P.ctx_before_gate, P.gate != nullptr, gatev, crel);  // (idx - zp of the context quantiser for the dumps)
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const float* ov = &xs[dt * 4];
    if constexpr (DUMP) {
      unsigned cw = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) cw |= (unsigned)(crel[dt * 4 + r] + P.fq_c.zp) << (8 * r);
      if (P.fq_c.en && P.fq_c.dump != nullptr && qvalid) *reinterpret_cast<unsigned*>(P.fq_c.dump + (((long)b * P.H + h) * P.Sq + qrow) * D + 16 * dt + 4 * g) = cw;
    }
    if constexpr (OUT32) {
      if (q0 + ce < P.Sq)
        store_wt16(reinterpret_cast<float*>(P.o) + bh_offset(b, P.os_b, h, P.os_h) + (long)(q0 + ce) * P.os_s + 16 * dt + 4 * ge,
                   u4{f32_bits(ov[0]), f32_bits(ov[1]), f32_bits(ov[2]), f32_bits(ov[3])});
    } else if constexpr (OUT8) {
      // ov = idx - zp (the quantiser's integers, oscale 1): idx by v_cvt_pk_u8_f32 (exact on whole numbers in [0, 255]), centred by the XOR
      unsigned w8 = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) w8 = __builtin_amdgcn_cvt_pk_u8_f32(ov[r] + P.fq_c.zp, r, w8);
      *reinterpret_cast<unsigned*>(ebase + ce * ROWB + 16 * dt + 4 * ge) = w8 ^ 0x80808080u;
    } else {
      u2 w;
      if constexpr (OUT == IN_BF16) { w.x = pack2_bf16(ov[0], ov[1]); w.y = pack2_bf16(ov[2], ov[3]); }
      else { w.x = pack2_f16(ov[0], ov[1]); w.y = pack2_f16(ov[2], ov[3]); }
      *reinterpret_cast<u2*>(ebase + ce * ROWB + ((((2 * dt + (ge >> 1)) ^ (ce & 7)) << 4) | ((ge & 1) << 3))) = w;
    }
  }
  if constexpr (OUT8) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned char* obase8 = reinterpret_cast<unsigned char*>(P.o) + bh_offset(b, P.os_b, h, P.os_h);
    const int row = lane_e >> 2, lc = lane_e & 3;  // 4 chunks of 16 B per 64-B row, 16 rows in one pass
    const u4 w = *reinterpret_cast<const u4*>(ebase + row * ROWB + (lc << 4));
    if (q0 + row < P.Sq) store_wt16(obase8 + (long)(q0 + row) * P.os_s + lc * 16, w);
  } else if constexpr (!OUT32) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned short* obase = reinterpret_cast<unsigned short*>(P.o) + bh_offset(b, P.os_b, h, P.os_h);
    const int lr = lane_e >> 3, lc = lane_e & 7;  // 8 chunks of 16 B per 128-B row, 8 rows per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 8 + lr;
      const u4 w = *reinterpret_cast<const u4*>(ebase + row * ROWB + ((lc ^ (row & 7)) << 4));
      if (q0 + row < P.Sq) store_wt16(obase + (long)(q0 + row) * P.os_s + lc * 8, w);
    }
  }
}

template <int NT, int OUT, bool DUMP>
static void launch_i8_variant(const AttnParams& P, unsigned grid, hipStream_t st) {
  const bool cq2 = P.i8_cq == 128;  // a q grid with zero point 0 (rare: the projections are two-sided)
  const bool pad = P.pad != nullptr;
  if (cq2) {
    if (pad) hipLaunchKernelGGL((oeh_attn_i8_kernel<NT, OUT, DUMP, true, true>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_i8_kernel<NT, OUT, DUMP, true, false>), dim3(grid), dim3(256), 0, st, P);
  } else {
    if (pad) hipLaunchKernelGGL((oeh_attn_i8_kernel<NT, OUT, DUMP, false, true>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_i8_kernel<NT, OUT, DUMP, false, false>), dim3(grid), dim3(256), 0, st, P);
  }
}

template <int NT>
static int launch_i8_nt(const AttnParams& P, int out, hipStream_t st) {
  const unsigned grid = (unsigned)(P.nQT * P.nBHpad);
  if (P.fq_s.dump != nullptr || P.fq_p.dump != nullptr || P.fq_c.dump != nullptr) {  // tests: the index tensors written out (fp32 output only)
    if (out != IN_F32) return -95;
    launch_i8_variant<NT, IN_F32, true>(P, grid, st);
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  switch (out) {
    case IN_F16: launch_i8_variant<NT, IN_F16, false>(P, grid, st); break;
    case IN_BF16: launch_i8_variant<NT, IN_BF16, false>(P, grid, st); break;
    case 3: launch_i8_variant<NT, 3, false>(P, grid, st); break;  // int8 centred indices (ctx_emit_index)
    default: launch_i8_variant<NT, IN_F32, false>(P, grid, st); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// D == 64, Sk <= 512 and a multiple of 16, scores and probabilities on 8-bit grids (oeh_api.hip: i8_eligible)
int launch_attn_i8(const AttnParams& P, int out, hipStream_t st) {
  if (P.Sk <= 128) return launch_i8_nt<8>(P, out, st);
  if (P.Sk <= 256) return launch_i8_nt<16>(P, out, st);
  return launch_i8_nt<32>(P, out, st);
}

}  // namespace oeh
