// One-pass ("online") fused attention for the PLAIN softmax / softmax_1 case, 16-bit storage - the headline
// configuration (OPT-125m softmax1, gated variants, long BERT/ViT rows).
//
// Why a second structure: the full-row kernel (oeh_attn_fast.inl) must keep a query block's complete score row in
// registers because clipping and fake-quant are non-linear in the final probability; that caps a workgroup at 64
// query rows, so a head's K/V is re-streamed through LDS once per 64 rows (4.5x for causal S=512) and the in-kernel
// timeline (tools/timeline.py) shows its K and V phases paced by that LDS-DMA traffic.  Plain softmax_n has no such
// constraint (hard part 1): with a per-row reference score, exp(x - reference) is accumulated tile by tile
// and the sums rescaled when the reference moves.  softmax_1's "+1" is exp(0 - reference) added to the row sum once
// at the end - the reference formula, including rows that are exactly 0 when every key is masked.  Hence:
//   * one workgroup = 4 waves x MQ query blocks of 16 rows (MQ=2: 128 rows) -> K/V stream 2.3x smaller, and K and V
//     tiles arrive TOGETHER: one barrier per 64 keys instead of two;
//   * registers: Q, O and one 64-key score tile per block only (no Sk limit);
//   * every K fragment read from LDS feeds MQ MFMAs and every V^T fragment feeds MQ MFMAs (LDS bandwidth / MQ);
//   * row max all-reduce by v_permlane16_swap / v_permlane32_swap (VALU), not ds_bpermute (LDS round trip);
//   * row sums by one extra MFMA per 32 keys against a ones operand (sums exactly the rounded P the second product
//     uses, and frees 16 v_add per block and tile - the loop is VALU-issue bound, not MFMA bound);
//   * lazy reference: it moves only when a row's tile maximum exceeds it by 2^8, so the O/l rescale (20 multiplies per
//     block) almost never runs after the first tiles;
//   * Q arrives by LDS-DMA ahead of the first stages and tile 0 starts on Q + K tile 0 alone; O leaves as whole rows,
//     write-through (oeh_common.h: store_wt16 - a plain store parks the output in L2 until the end-of-kernel release);
//   * key padding (PAD variant): the padding row sits in LDS and trailing fully padded key tiles are not streamed.
// Same swapped products (S^T = K Q^T, O^T = V^T P^T on v_mfma_f32_16x16x32), LDS images, swizzles and LDS-DMA ring
// as the full-row kernel.  Masks: none | analytic causal | key padding (softmax_1 only: a fully masked row must be 0).
#pragma once
#include "oeh_attn_fast.inl"

// Knock-out ladder of the one-pass kernel (diagnostic builds only: make -C outeffhop_amd/csrc knockout KO=n; profiles/r05_headline_floor.txt).
// 0 (production): everything.  1: return at entry.  2: + the prologue and the whole LDS-DMA stream with its waits and barriers, nothing else.
// 3: + both products' MFMAs and their LDS fragment reads (the second product on the raw bits of the scores).  4: + the softmax arithmetic (the full
// tile).  2 - 4 skip the epilogue (its stores sit behind a never-true test of an accumulator, so that nothing above is dead code).
#ifndef OEH_KO
#define OEH_KO 0
#endif
#ifndef OEH_NSUB_MASK
#define OEH_NSUB_MASK 1
#endif

#include <type_traits>

namespace oeh {

// all-reduce over the 4 lanes (c, c+16, c+32, c+48) that hold one query row, without LDS
__device__ __forceinline__ float row_allreduce_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(f32_bits(x), f32_bits(x), false, false);
  x = __builtin_fmaxf(bits_f32(a[0]), bits_f32(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(f32_bits(x), f32_bits(x), false, false);
  return __builtin_fmaxf(bits_f32(b[0]), bits_f32(b[1]));
}

template <int D, int MQ, bool SRC32>
constexpr int flash_occupancy() { return D >= 128 ? 1 : (SRC32 ? 2 : 3); }  // fp32 forms: operand pairs + the staged stage, 2 waves per SIMD

// GATE: the conditional per-token gate is computed in the kernel exactly as in the full-row kernel (oeh_attn_fast.inl:
// layer-input rows as K-shaped LDS-DMA tiles, first predictor layer on the matrix cores).  The input rows borrow stage 1
// at start-up, so stage 1 of the K/V stream is issued later (with stage 2, once the gate has been formed).
// SRC32: q, k, v and o are fp32 (the reference's validate_* scripts run fp32 models).  The tiles then come through
// registers - 32 B of fp32 per lane and piece, split into the fp16 operand pair (hi, lo) of oeh_common.h: split8 and written
// as two LDS images in the layout the DMA produces - one 64-key stage ahead: the loads of stage i+1 are issued right after
// the barrier of tile i and committed to LDS at the top of tile i+1, so they have a tile of compute to land; one barrier
// per tile as before, two ring slots (a slot's readers are all behind the barrier that precedes its next commit).  Scores
// from three MFMAs per k-step (q.k = qh.kh + 2^-11 (qh.kl + ql.kh)), the context from two (P is an fp16 operand, V the pair):
// fp32 accuracy on the scores, fp32 accumulation and fp32 output straight from the accumulators (OUT32); Q goes global ->
// registers directly.  No conversion pre-pass (which costs more HBM time than the attention itself).
// CLIP: clipped softmax (softmax.py:10-19: clip(p (eta - gamma) + gamma, 0, 1)) needs the finished denominator before any
// probability can enter the second product, so the key stream runs TWICE: a statistics pass (scores, lazy reference, row
// sums of the exponentials - no V product) and a final pass that recomputes the scores against the now final reference,
// forms p = e / den, clips and multiplies by V.  Rows of any length (the full-row kernel holds at most 512 scores per row in
// registers and is the faster form up to there: it computes the scores once); masked keys have e = 0 and stay 0 (gamma <= 0).
// TP = 2: the fused INT8 chain on the quantiser grid (oeh_attn_fast.inl, FQ == 1: rel = clamp(rint(s k1)), exp2((rel - rel_max) c2),
// index of the probability from e * RN(1 / (den scale_p))) in the same two passes: the statistics pass keeps a running maximum
// and sum per LANE (no cross-lane step inside the loop), the final pass recomputes rel against the row's maximum and feeds the
// integer-valued probability to the second product; context quantiser and gate in the epilogue.  Rows of any length.
// O32: 16-bit storage with the output taken from the fp32 accumulators (include/oeh.h: o_dtype = OEH_F32) - the same loop, only the
// epilogue's store differs (as a runtime switch in the epilogue it cost the production launches +0.7 ... +3 %, round 4).
template <int D, int IN, int MQ, bool PAD, bool GATE, bool SRC32 = false, int TP = 0, bool O32 = false>
__global__ __launch_bounds__(256, (flash_occupancy<D, MQ, SRC32>())) void oeh_attn_flash_kernel(const AttnParams P) {
  static_assert(IN == IN_F16 || IN == IN_BF16, "16-bit matrix-core operands");
  static_assert(!O32 || (!SRC32 && TP == 0 && !(GATE && PAD)), "fp32 output of 16-bit storage: the plain one-pass form [+ key padding | + in-kernel gate]");
  constexpr bool CLIP = (TP == 1), FQ2 = (TP == 2);
  static_assert(TP == 0 || !GATE, "two-pass forms: no in-kernel gate predictor");
  static_assert(!SRC32 || (!GATE && IN == IN_F16), "fp32 storage: fp16 operands, fp32 output, no in-kernel gate predictor");
  constexpr bool OUT32 = SRC32 || O32;
  static_assert(MQ == 1 || MQ == 2, "one or two query blocks per wave");
  constexpr int ROWB = 2 * D;
  constexpr int TILEB = 64 * ROWB;      // one operand tile (64 keys)
  constexpr int STAGEB = 2 * TILEB;     // K tile + V tile
  constexpr int CPR = D / 8;
  constexpr int RPP = 64 / CPR;
  constexpr int G = D / 32;             // LDS-DMA pieces per wave per operand tile
  constexpr int KS = D / 32;
  constexpr int DT = D / 16;
  constexpr int R = 3;                  // stages
  constexpr float NEG = -1.0e30f;       // floor of a padded score (finite: NEG * log2e does not overflow)
  constexpr float NEGT = -1.0e30f;      // exponent argument of a masked key: exp2 -> 0 exactly
  constexpr float kThr = 8.0f;          // lazy reference: P stays <= 2^8 (exact range for f16 / bf16 operands)

  constexpr int SLOT32 = 2 * STAGEB;    // SRC32: one ring slot = hi images (K, V) + lo images (K, V)
  __shared__ __attribute__((aligned(16))) unsigned char lds[SRC32 ? 2 * SLOT32 : R * STAGEB];
  constexpr int PADROW = 1024;          // keys of the padding row kept in LDS (PAD variant)
  __shared__ __attribute__((aligned(16))) float lds_padrow[PAD ? PADROW : 4];
  __shared__ int lds_last[4];

  // The kernel arguments the prologue needs, requested in ONE round of scalar loads at entry (the compiler loads an argument where it
  // is first used: four dependent rounds of ~300 cycles each in front of the first LDS-DMA request, every wave of every workgroup).
  {
    asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.nBHpad), "s"(P.nQT), "s"(P.nBH), "s"(P.H), "s"(P.Sq), "s"(P.Sk), "s"(P.causal), "s"(P.snake),
                 "s"(P.magic_nbh), "s"(P.magic_h), "s"(P.qs_b), "s"(P.qs_h), "s"(P.qs_s), "s"(P.ks_b), "s"(P.ks_h), "s"(P.ks_s), "s"(P.vs_b), "s"(P.vs_h),
                 "s"(P.vs_s));
  }
  const int bid = (P.snake && !(SRC32 && P.head_major)) ? snake_block_id(blockIdx.x, P.nQT * P.nBHpad) : (int)blockIdx.x;  // (= gridDim.x, without the hidden-argument load)
  int qt_rev, bh;
  if (SRC32 && P.head_major) block_to_tile(bid, P.nBHpad, P.nQT, P.head_major, qt_rev, bh);
  else div_magic((unsigned)bid, (unsigned)P.nBHpad, P.magic_nbh, qt_rev, bh);
  if (bh >= P.nBH) return;
#if OEH_KO == 1
  return;
#endif
  int b, h;
  div_magic((unsigned)bh, (unsigned)P.H, P.magic_h, b, h);

  if constexpr (SRC32) fp16_overflow_clamp();  // out-of-range fp32 operands saturate (oeh_common.h)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int Sk = P.Sk, Sq = P.Sq;
  const int off = Sk - Sq;
  const int causal = P.causal;

  // ---- query blocks of this wave: rb[j] = first row of block j, nkb[j] = 64-key tiles it needs (nkb[0] <= nkb[MQ-1])
  // block j of wave w = rows 64*slab[j] + 16*w .. +15: the workgroup's Q is MQ 64-row slabs, each one K-shaped LDS tile
  int slab[MQ], rb[MQ], nkb[MQ];
  const int qt = P.nQT - 1 - qt_rev;    // heaviest causal tiles first
#pragma unroll
  for (int j = 0; j < MQ; ++j) {
    slab[j] = qt * MQ + j;
    rb[j] = 64 * slab[j] + 16 * wave;
  }
  const int last_row_wg = 64 * slab[MQ - 1] + 63;  // last query row of the workgroup (bounds the tiles it streams)
  int n_kt = ((causal ? min(Sk, max(0, last_row_wg + 1 + off)) : Sk) + 63) >> 6;  // tiles the workgroup streams
  int tm0[MQ];                          // first tile that holds a masked key for the block's first row
#pragma unroll
  for (int j = 0; j < MQ; ++j) {
    nkb[j] = ((causal ? min(Sk, max(0, rb[j] + 16 + off)) : Sk) + 63) >> 6;
    tm0[j] = ((causal ? min(rb[j] + off, Sk - 1) : Sk - 1) + 1) >> 6;
  }

  // In-kernel stamps exist only in the diagnostic build of tools/timeline.py (make ... EXTRA=-DOEH_TIMELINE): even a never-taken
  // `if (stamp != nullptr)` per site is a scalar compare + branch on every wave's critical path (three sites per tile).
#ifdef OEH_TIMELINE
  unsigned long long* stamp = nullptr;
  if (P.stamps != nullptr) stamp = P.stamps + ((long)bid * 4 + wave) * 32;
#define OEH_STAMP(slot)                                                                \
  do {                                                                                 \
    if (stamp != nullptr && lane == 0) stamp[(slot)] = __builtin_amdgcn_s_memtime();   \
  } while (0)
  OEH_STAMP(0);
  if (stamp != nullptr && lane == 0) {
    stamp[30] = __builtin_amdgcn_s_memrealtime();
    stamp[29] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
  }
#else
#define OEH_STAMP(slot) do { } while (0)
#endif

  // ---- LDS-DMA stream of (K tile, V tile) stages, strictly in order: the scalar base pointers advance by 64 rows per
  // stage, the per-lane byte offsets (row of the piece, swizzled 16-B chunk) never change
  const unsigned char* const kbase0 = reinterpret_cast<const unsigned char*>(P.k) + 2 * (bh_offset(b, P.ks_b, h, P.ks_h));
  const unsigned char* const vbase0 = reinterpret_cast<const unsigned char*>(P.v) + 2 * (bh_offset(b, P.vs_b, h, P.vs_h));
  const unsigned char* kcur = kbase0;
  const unsigned char* vcur = vbase0;
  const int prow = lane / CPR, pch = lane % CPR;
  const unsigned lds_base = lds_offset(lds);
  auto piece_row = [&](int j) { return (wave * G + j) * RPP + prow; };
  unsigned koff[G], voff[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int row = piece_row(j);
    koff[j] = 2u * (unsigned)(row * P.ks_s + (pch ^ swz_k<D>(row)) * 8);
    voff[j] = 2u * (unsigned)(row * P.vs_s + ((((pch >> 1) ^ swz_v<D>(row)) << 1) | (pch & 1)) * 8);
  }
  const long kstep = 128 * P.ks_s, vstep = 128 * P.vs_s;  // bytes per 64 rows
  int nx_tile = 0, nx_slot = 0;
  auto issue_next = [&]() {
    const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(nx_slot * STAGEB + wave * G * 1024));
    if (nx_tile * 64 + 64 > Sk) {  // ragged last tile: rows past Sk are redirected to row Sk-1 (finite data, masked later)
      unsigned ko[G], vo[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int over = nx_tile * 64 + piece_row(j) - (Sk - 1);
        ko[j] = koff[j] - (over > 0 ? 2u * (unsigned)(over * P.ks_s) : 0u);
        vo[j] = voff[j] - (over > 0 ? 2u * (unsigned)(over * P.vs_s) : 0u);
      }
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(kcur, ko[j], slot + j * 1024);
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(vcur, vo[j], slot + TILEB + j * 1024);
    } else {  // (its own branch: the common path then issues from the loop-invariant offset registers, no copies)
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(kcur, koff[j], slot + j * 1024);        // the K tile first: tile 0 starts on Q + K
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(vcur, voff[j], slot + TILEB + j * 1024);
    }
    kcur += kstep;
    vcur += vstep;
    ++nx_tile;
    nx_slot = (nx_slot == R - 1) ? 0 : nx_slot + 1;
  };
  // (round 6, OEH_PIPE_QK == 1: the same stage request in its 2 G pieces, so that the placed tile can put one behind each sub-tile's score MFMAs;
  // full stages only - the ragged last stage is never requested from inside a tile: the caller tests `nx_tile * 64 + 64 <= Sk`)
  [[maybe_unused]] auto issue_piece = [&](const int p) {
    const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(nx_slot * STAGEB + wave * G * 1024));
    if (p < G) glds16_s(kcur, koff[p], slot + p * 1024);
    else glds16_s(vcur, voff[p - G], slot + TILEB + (p - G) * 1024);
  };
  [[maybe_unused]] auto issue_advance = [&]() {
    kcur += kstep;
    vcur += vstep;
    ++nx_tile;
    nx_slot = (nx_slot == R - 1) ? 0 : nx_slot + 1;
  };
  // ---- Q rides the same LDS-DMA stream, FIRST, into the stage the ring does not use yet (slab j as a K-shaped tile at
  // j*TILEB of stage R-1).  The bytes a workgroup needs before its first MFMA are then Q + K tile 0; with Q as ordinary
  // register loads behind the first two stages (returns are in issue order) they were Q + 2 K tiles + 2 V tiles, and the
  // measured start-up is paced by bytes per CU (~20 B/cycle), not by one memory latency.
  // SRC32: the register-staged stream (see the kernel comment)
  f4 kreg[SRC32 ? G : 1][2], vreg[SRC32 ? G : 1][2];
  auto load_regs = [&](const int t) {
    if constexpr (SRC32) {
      const float* ksrc = reinterpret_cast<const float*>(P.k) + bh_offset(b, P.ks_b, h, P.ks_h);
      const float* vsrc = reinterpret_cast<const float*>(P.v) + bh_offset(b, P.vs_b, h, P.vs_h);
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int row = piece_row(j);
        const int kr = min(t * 64 + row, Sk - 1);  // rows past Sk: finite data, masked later
        const float* kp = ksrc + (long)kr * P.ks_s + (pch ^ swz_k<D>(row)) * 8;
        const float* vp = vsrc + (long)kr * P.vs_s + ((((pch >> 1) ^ swz_v<D>(row)) << 1) | (pch & 1)) * 8;
        kreg[j][0] = *reinterpret_cast<const f4*>(kp);
        kreg[j][1] = *reinterpret_cast<const f4*>(kp + 4);
        vreg[j][0] = *reinterpret_cast<const f4*>(vp);
        vreg[j][1] = *reinterpret_cast<const f4*>(vp + 4);
      }
    }
  };
  auto commit_regs = [&](const int slot) {
    if constexpr (SRC32) {
      unsigned char* base = lds + slot * SLOT32 + (wave * G) * 1024 + lane * 16;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        u4 hi, lo;
        split8(kreg[j][0], kreg[j][1], hi, lo);
        *reinterpret_cast<u4*>(base + j * 1024) = hi;
        *reinterpret_cast<u4*>(base + STAGEB + j * 1024) = lo;
        split8(vreg[j][0], vreg[j][1], hi, lo);
        *reinterpret_cast<u4*>(base + TILEB + j * 1024) = hi;
        *reinterpret_cast<u4*>(base + STAGEB + TILEB + j * 1024) = lo;
      }
    }
  };
  u4 qf[MQ][KS], ql[SRC32 ? MQ : 1][SRC32 ? KS : 1];  // Q^T operands per block (SRC32: the hi / lo pair)
  if constexpr (SRC32) {  // Q: global -> registers in the operand layout (row rb[j] + c, elements 32 ks + 8 g ..)
#pragma unroll
    for (int j = 0; j < MQ; ++j) {
      const int qr = min(rb[j] + c, Sq - 1);  // rows past Sq: finite data, never stored
      const float* qp = reinterpret_cast<const float*>(P.q) + bh_offset(b, P.qs_b, h, P.qs_h) + (long)qr * P.qs_s + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        split8(__builtin_nontemporal_load(reinterpret_cast<const f4*>(qp + 32 * ks)), __builtin_nontemporal_load(reinterpret_cast<const f4*>(qp + 32 * ks + 4)), qf[j][ks], ql[j][ks]);
    }
    load_regs(0);
  } else {
    const unsigned char* qbase = reinterpret_cast<const unsigned char*>(P.q) + 2 * (bh_offset(b, P.qs_b, h, P.qs_h));
    const unsigned qslot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((R - 1) * STAGEB + wave * G * 1024));
#pragma unroll
    for (int t = 0; t < MQ; ++t) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int row = piece_row(j);
        int qrow = 64 * slab[t] + row;
        qrow = qrow < Sq ? qrow : Sq - 1;  // rows past Sq: finite data, never stored
        glds16_s_nt(qbase, 2u * (unsigned)(qrow * P.qs_s + (pch ^ swz_k<D>(row)) * 8), qslot + t * TILEB + j * 1024);
      }
    }
  }
  constexpr int GT = GATE ? 4 : 1;       // 16-unit MFMA tiles of predictor hidden units (<= 64 units)
  u4 gwf[GT][KS];                        // GATE: the lane's share of the first-layer weights, rounded to the storage dtype
  f4 gb1v[GT], gw2v[GT];
  int g_mt = 1;
  if constexpr (GATE) {  // the workgroup's layer-input rows, head h's slice, slab t as a K-shaped tile at t*TILEB of stage 1
    const unsigned char* xbase = reinterpret_cast<const unsigned char*>(P.gh) + 2 * ((long)b * P.ghs_b + (long)h * D);
    const unsigned xslot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(1 * STAGEB + wave * G * 1024));
#pragma unroll
    for (int t = 0; t < MQ; ++t) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int row = piece_row(j);
        int xr = 64 * slab[t] + row;
        xr = xr < Sq ? xr : Sq - 1;
        glds16_s_nt(xbase, 2u * (unsigned)(xr * P.ghs_t + (pch ^ swz_k<D>(row)) * 8), xslot + t * TILEB + j * 1024);
      }
    }
  }
  if constexpr (!SRC32) {
    issue_next();
    if (!GATE && 1 < n_kt) issue_next();
  }

  if constexpr (GATE) {  // weights: hidden unit 16 tau + c, inputs 8g.. of each 32-wide k-step; b1 / w2 of units 16 tau + 4g..4g+3
    const int mm = P.g_units > 0 ? P.g_units : 1;  // <= 64 (host)
    g_mt = (mm + 15) >> 4;
#pragma unroll
    for (int tau = 0; tau < GT; ++tau) {
      gb1v[tau] = gw2v[tau] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) gwf[tau][ks] = u4{0u, 0u, 0u, 0u};
      if (tau < g_mt) {
        const int u = 16 * tau + c;
        const bool uv = u < mm;
        const float* wr = P.gw1 + ((long)h * mm + (uv ? u : 0)) * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          f4 w0 = *reinterpret_cast<const f4*>(wr + 32 * ks), w1 = *reinterpret_cast<const f4*>(wr + 32 * ks + 4);
          if (!uv) w0 = w1 = f4{0.f, 0.f, 0.f, 0.f};
          if constexpr (IN == IN_BF16) gwf[tau][ks] = u4{pack2_bf16(w0[0], w0[1]), pack2_bf16(w0[2], w0[3]), pack2_bf16(w1[0], w1[1]), pack2_bf16(w1[2], w1[3])};
          else gwf[tau][ks] = u4{pack2_f16(w0[0], w0[1]), pack2_f16(w0[2], w0[3]), pack2_f16(w1[0], w1[1]), pack2_f16(w1[2], w1[3])};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ur = 16 * tau + 4 * g + r;
          if (ur < mm) {
            gb1v[tau][r] = P.gb1[(long)h * mm + ur];
            gw2v[tau][r] = P.g_units > 0 ? P.gw2[(long)h * mm + ur] : 1.0f;
          }
        }
      }
    }
  }
  // Q and K tile 0 landed, for every wave: all but the G (V tile 0) + 2G (stage 1) younger transfers
  auto wait_vm = [&](auto nc) {  // s_waitcnt vmcnt(N * G), N compile-time
    if constexpr (SRC32) return;   // no DMA in flight: the compiler waits for its own loads where they are used
    constexpr int N = decltype(nc)::value * G;
    static_assert(N <= 16, "vmcnt immediates used below");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  };
  // Key padding (PAD variant), while the first transfers fly: the row of additive values goes to LDS once (tiles read it
  // from there; in-loop global loads would put compiler waits into the DMA stream), and trailing 64-key tiles in which
  // EVERY key is masked are dropped from the stream - a masked key contributes exp(x - m) = 0 exactly, so this is the
  // reference's result, and right-padded batches are the norm.  The two stages already issued are always consumed.
  const bool pad_in_lds = PAD && Sk <= PADROW;
  if constexpr (PAD) {
    int last = -1;  // last key that is not masked
    for (int key = tid; key < ((Sk + 63) & ~63); key += 256) {  // (no padding vector: the variant serves a (B,1,Sq,Sk) mask alone - zeros)
      const float pv = (key < Sk && P.pad != nullptr) ? load_mask(P.pad, P.pad_f16, (long)b * P.pad_sb + key) : 0.0f;
      if (key < PADROW) lds_padrow[key] = pv;
      if (key < Sk && pv > -1.0e30f) last = key;
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) last = max(last, __shfl_xor(last, sft));
    if (lane == 0) lds_last[wave] = last;
  }
  if (!GATE && 1 < n_kt) wait_vm(std::integral_constant<int, 3>{});
  else wait_vm(std::integral_constant<int, 1>{});  // (GATE: stage 1 is not in flight yet - its slot holds the input rows)
  barrier_mem();
  if constexpr (PAD) {
    const int last = max(max(lds_last[0], lds_last[1]), max(lds_last[2], lds_last[3]));
    n_kt = min(n_kt, max((last >> 6) + 1, min(n_kt, 2)));
  }
  OEH_STAMP(1);

  // lane-constant parts of the LDS fragment addresses
  const unsigned char* kaddr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kaddr[ks] = lds + c * ROWB + (((ks * 4 + g) ^ swz_k<D>(c)) << 4);
  const int vrow = 4 * g + (c >> 2);
  const unsigned char* vaddr[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) vaddr[dt] = lds + TILEB + vrow * ROWB + ((dt ^ swz_v<D>(vrow)) << 5) + ((c & 3) << 3);

  // Q^T operands from the Q stage; read complete before the first loop barrier, after which the stage is refilled
  if constexpr (!SRC32) {
#pragma unroll
    for (int j = 0; j < MQ; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        qf[j][ks] = *reinterpret_cast<const u4*>(kaddr[ks] + (R - 1) * STAGEB + j * TILEB + wave * 16 * ROWB);
  }
  float gate_row[MQ];  // GATE: sigmoid(logit) * scaling of this lane's query row in block j
#pragma unroll
  for (int j = 0; j < MQ; ++j) gate_row[j] = 1.0f;
  if constexpr (GATE) {
#pragma unroll
    for (int j = 0; j < MQ; ++j) {
      u4 xf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + 1 * STAGEB + j * TILEB + wave * 16 * ROWB);
      float a = 0.0f;
#pragma unroll
      for (int tau = 0; tau < GT; ++tau) {
        if (tau < g_mt) {
          f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc = mfma16<IN>(gwf[tau][ks], xf[ks], acc);  // rows = hidden units 16 tau + 4g + r, column = token c
          if (P.g_units > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a = __builtin_fmaf(__builtin_fmaxf(acc[r] + gb1v[tau][r], 0.0f), gw2v[tau][r], a);  // padded units: w2 = 0
          } else {
            a = (g == 0) ? acc[0] + gb1v[0][0] : 0.0f;  // Linear(D,1): unit 0 only
          }
        }
      }
      {  // sum over the 4 lanes (c, c+16, c+32, c+48) of the row
        auto s1 = __builtin_amdgcn_permlane16_swap(f32_bits(a), f32_bits(a), false, false);
        a = bits_f32(s1[0]) + bits_f32(s1[1]);
        auto s2 = __builtin_amdgcn_permlane32_swap(f32_bits(a), f32_bits(a), false, false);
        a = bits_f32(s2[0]) + bits_f32(s2[1]);
      }
      if (P.g_units > 0) a = a + P.gb2[h];
      a = 1.0f / (1.0f + exp_acc(-a));
      gate_row[j] = a * P.g_scaling;
      const int qrow = rb[j] + c;
      if (P.g_out != nullptr && g == 0 && qrow < Sq) P.g_out[((long)b * P.H + h) * Sq + qrow] = a;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  constexpr bool has_pad = PAD;         // a kernel variant, not a branch: merging the two paths inside the loop costs a
                                        // register-to-register copy of the whole score tile on the path without padding
  const float sc = P.scale;
  const float c1 = has_pad ? kLog2e : sc * kLog2e;   // pad mode keeps scaled+masked scores, otherwise raw dot products
  const u4 ones = (IN == IN_BF16) ? u4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                  : u4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
  // Per-row state: mcneg = -(reference score) * c1 in exponent (log2) units, so that t = fma(s, c1, mcneg) is the
  // exponent argument; O and l are sums of exp2(t).  The reference is the first tile's row maximum and afterwards
  // moves only when a tile maximum exceeds it by 2^8.  softmax_1's "+1" is exp2(mcneg) (= exp(-reference)).
  float mcneg[MQ];
  float lsum[MQ], pinv[MQ];             // two-pass forms: this lane's share of the row sum of exp (statistics pass); 1 / denominator (final pass)
  float mrl[MQ];                        // TP = 2: the lane's running maximum of rel, then the row's
  bool dead1[MQ];                       // TP = 1, vanilla softmax with masks: the row has no visible key
#pragma unroll
  for (int j = 0; j < MQ; ++j) dead1[j] = false;
  const float fq_k1 = sc * P.fq_s.rscale, fq_c2 = P.fq_s.c2;
#pragma unroll
  for (int j = 0; j < MQ; ++j) { lsum[j] = 0.0f; pinv[j] = 1.0f; mrl[j] = kGridMagic + (P.fq_s.lo - 1.0f); }  // (one below every index: never a sentinel in exp2 arguments; rel is carried as M + rel, oeh_common.h: grid_rel_m)
  const float fq_slo = kGridMagic + P.fq_s.lo, fq_shi = kGridMagic + P.fq_s.hi;
  f4 lacc[MQ];                          // row sums of the ROUNDED P, accumulated by a ones-row MFMA (every register = l)
  f4 o[MQ][DT], ox[SRC32 ? MQ : 1][SRC32 ? DT : 1];  // ox: the V-lo part of the context (SRC32), scaled by 2^-11 at the end
#pragma unroll
  for (int j = 0; j < MQ; ++j) {
    mcneg[j] = 0.0f;
    lacc[j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[j][dt] = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (SRC32) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) ox[j][dt] = f4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // the ones operand of the row-sum MFMAs, kept in registers across the tile loop (round 6: it was rebuilt - five v_mov - in every tile)
  u4 ones_live = ones;
  asm volatile("" : "+v"(ones_live));
  // Issue priority of the wave's two kinds of phase (matrix-core / vector).  OEH_PRIO_MODE (round 6 experiment, not landed: profiles/r06_headline_tile_asm.txt):
  // 0 = production (1 / 0 for every wave); 1 = the workgroups of the upper half of the causal q tiles outrank the others in both phases (3 / 2 against 1 / 0);
  // 2 = only the heaviest q tile's; 3 = a constant level by q tile, no boost for the matrix-core phases
#ifndef OEH_PRIO_MODE
#define OEH_PRIO_MODE 0
#endif
#if OEH_PRIO_MODE == 0
  auto prio_hi = [&]() { __builtin_amdgcn_s_setprio(1); };
  auto prio_lo = [&]() { __builtin_amdgcn_s_setprio(0); };
#else
#if OEH_PRIO_MODE == 1
  const bool prio_heavy = causal && 2 * qt >= P.nQT;
#elif OEH_PRIO_MODE == 2
  const bool prio_heavy = causal && qt == P.nQT - 1;
#else
  const int prio_lvl = causal ? min(3, (4 * qt) / max(1, P.nQT)) : 0;
#endif
#if OEH_PRIO_MODE == 3
  auto prio_set = [&]() { if (prio_lvl == 3) __builtin_amdgcn_s_setprio(3); else if (prio_lvl == 2) __builtin_amdgcn_s_setprio(2); else if (prio_lvl == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); };
  prio_set();
  auto prio_hi = [&]() {};
  auto prio_lo = [&]() {};
#else
  auto prio_hi = [&]() { if (prio_heavy) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); };
  auto prio_lo = [&]() { if (prio_heavy) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); };
#endif
#endif
  // ---- one 64-key tile for blocks J0..MQ-1 of this wave (J0 = 1: block 0's rows end before this tile)
  // MODE 0: the one-pass tile; CLIP: 1 = statistics pass (no second product), 2 = final pass (final reference, clip)
  auto tile = [&](auto j0c, auto firstc, auto modec, auto nsc, const int i, const int soff, const bool in_tile_issue = false) {
    constexpr int J0 = decltype(j0c)::value;
    constexpr bool FIRST = decltype(firstc)::value;  // tile 0: V tile 0 is awaited between the two products
    constexpr int MODE = decltype(modec)::value;
    // Round 6: NSa[j] = how many of the tile's four 16-key sub-tiles hold a key that ANY row of block j may see (4 = all; the caller passes
    // fewer only for the placed order below).  The sub-tiles behind are masked for the whole block (the causal diagonal, the ragged last
    // tile): their scores would be set to the sentinel, their exponentials are exactly 0 and their products add exactly 0 - so the MFMAs,
    // the scale / max / exp / convert steps and the K fragment reads of those sub-tiles are simply not issued; results bit for bit the same.
    constexpr int NSP = decltype(nsc)::value;
    constexpr int NSa[2] = {NSP & 15, (NSP >> 4) & 15};
    constexpr int NSMAX = (MQ == 2) ? ((J0 == 0 && NSa[0] > NSa[1]) ? NSa[0] : NSa[1]) : NSa[0];
    // Round 5: the plain one-pass tile on 16-bit storage, tiles after the first: the exponentials in two halves - keys 0-31 of every block, then the first
    // half's MFMAs (O^T += V^T P^T over those keys) with the exponentials of keys 32-63 placed BETWEEN them - instead of all 32 v_exp_f32 + 16 conversions
    // in one lump in front of 20 back-to-back MFMAs (the compiler's schedule; profiles/r05_headline_tile_order.txt).
#ifdef OEH_NO_PIPE_PV
    constexpr bool PIPE_PV = false;
#else
    constexpr bool PIPE_PV = (MODE == 0) && !SRC32 && !FIRST && D <= 64 && OEH_KO == 0;   // (D = 128: 1.045 of the plain order - one wave per SIMD there, other limits)
#endif
    static_assert(NSP == 0x44 || PIPE_PV, "partial tiles: the placed order only");
#if OEH_KO == 2
    if constexpr (FIRST) {
      if (!GATE && 1 < n_kt) wait_vm(std::integral_constant<int, 2>{});
      else wait_vm(std::integral_constant<int, 0>{});
      barrier_mem();
      if (GATE && 1 < n_kt) issue_next();
      if (2 < n_kt) issue_next();
    }
    return;
#endif
    // S^T = K Q^T; every K fragment is read once and used by all active blocks
    prio_hi();  // matrix-core phases at a higher issue priority than the other waves' softmax arithmetic (dense S=512: -2.7 %)
    f4 s[MQ][4];
#ifdef OEH_PIPE_QK
    // Round 6, measured and NOT landed (profiles/r06_headline_tile_asm.txt): the FIRST half of the steady-state tile as a placed order too - the scale /
    // reference step of sub-tile s between the score MFMAs of sub-tile s + 1 (pinned by scheduling barriers; the compiler keeps its own hazard distances),
    // and (OEH_PIPE_QK == 1) the next stage's LDS-DMA requests one piece behind each sub-tile's MFMAs instead of all four at the top of the tile.
    constexpr bool PIPE_QK = PIPE_PV && NSP == 0x44 && !has_pad && D == 64;
#else
    constexpr bool PIPE_QK = false;
#endif
    if constexpr (PIPE_QK) {
      auto score_sub = [&](const int sub) {
        u4 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + soff + sub * 16 * ROWB);
#pragma unroll
        for (int j = J0; j < MQ; ++j) {
          f4 acc = f4{0.f, 0.f, 0.f, 0.f};
          for (int ks = 0; ks < KS; ++ks) acc = mfma16<IN>(kf[ks], qf[j][ks], acc);
          s[j][sub] = acc;
        }
      };
      auto scale_sub = [&](const int sub) {
#pragma unroll
        for (int j = J0; j < MQ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[j][sub][r] = __builtin_fmaf(s[j][sub][r], c1, mcneg[j]);
#pragma unroll
        for (int j = J0; j < MQ; ++j) asm volatile("" : "+v"(s[j][sub]));   // (pins the step HERE: the compiler otherwise sinks it to its first use behind the last MFMA)
      };
      auto piece = [&](const int p) {
#if OEH_PIPE_QK == 1
        if (in_tile_issue) issue_piece(p);
#endif
      };
      score_sub(0);
      __builtin_amdgcn_sched_barrier(0);
      piece(0);
      score_sub(1);
      __builtin_amdgcn_sched_barrier(0);
      scale_sub(0);
      piece(1);
      __builtin_amdgcn_sched_barrier(0);
      score_sub(2);
      __builtin_amdgcn_sched_barrier(0);
      scale_sub(1);
      piece(2);
      __builtin_amdgcn_sched_barrier(0);
      score_sub(3);
      __builtin_amdgcn_sched_barrier(0);
      scale_sub(2);
      piece(3);
#if OEH_PIPE_QK == 1
      if (in_tile_issue) issue_advance();
#endif
      __builtin_amdgcn_sched_barrier(0);
      prio_lo();
      scale_sub(3);
    }
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      if (PIPE_QK || sub >= NSMAX) continue;
      u4 kf[KS], kl[SRC32 ? KS : 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + soff + sub * 16 * ROWB);
        if constexpr (SRC32) kl[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + soff + STAGEB + sub * 16 * ROWB);
      }
#pragma unroll
      for (int j = J0; j < MQ; ++j) {
        if (sub >= NSa[j]) continue;
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) acc = mfma16<IN>(kf[ks], qf[j][ks], acc);
        if constexpr (SRC32) {
          f4 accx = f4{0.f, 0.f, 0.f, 0.f};
          for (int ks = 0; ks < KS; ++ks) {
            accx = mfma16<IN>(kf[ks], ql[j][ks], accx);
            accx = mfma16<IN>(kl[ks], qf[j][ks], accx);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(accx[r], kSplitDown, acc[r]);
        }
        s[j][sub] = acc;
      }
    }
    if constexpr (!PIPE_QK) prio_lo();
    u4 pb[MQ][2];
#if OEH_KO == 3
#pragma unroll
    for (int j = J0; j < MQ; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) pb[j][u] = u4{f32_bits(s[j][2 * u][0]), f32_bits(s[j][2 * u][1]), f32_bits(s[j][2 * u + 1][0]), f32_bits(s[j][2 * u + 1][1])};
#else
    // exponent arguments t = (s - reference) * log2e  [key padding: BERT order scale*s + pad first]
    f4 padflag[(has_pad && MODE >= 3) ? 4 : 1];  // the grid chain with key padding (key_pad_boolean): +big for a visible key, the sentinel for a padded one
    if constexpr (has_pad) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub >= NSMAX) continue;
        const int kb = 64 * i + 16 * sub + 4 * g;
        f4 padv;
        if (pad_in_lds) {
          padv = *reinterpret_cast<const f4*>(&lds_padrow[kb]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) padv[r] = (kb + r < Sk && P.pad != nullptr) ? load_mask(P.pad, P.pad_f16, (long)b * P.pad_sb + kb + r) : 0.0f;
        }
        if constexpr (MODE >= 3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) padflag[sub][r] = padv[r] <= -1.0e4f ? NEGT : 3.0e38f;
        } else {
#pragma unroll
          for (int j = J0; j < MQ; ++j) {
            if (sub >= NSa[j]) continue;
            if (P.full != nullptr) {
              // a (B,1,Sq,Sk) additive mask on rows of more than 512 keys (the general kernel takes the shorter ones): read per
              // block from memory - compiler-visible loads inside the LDS-DMA stream, i.e. its waits drain the ring; slow next to
              // the other variants, two orders of magnitude faster than the any-shape kernel this combination used to reach
              const long mrow = (long)b * P.full_sb + (long)min(rb[j] + c, Sq - 1) * P.full_sq;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float x = __builtin_fmaf(s[j][sub][r], sc, padv[r]);
                if (kb + r < Sk) x = x + load_mask(P.full, P.full_f16, mrow + kb + r);
                s[j][sub][r] = __builtin_fmaxf(x, NEG);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) s[j][sub][r] = __builtin_fmaxf(__builtin_fmaf(s[j][sub][r], sc, padv[r]), NEG);
            }
          }
        }
      }
    }
    if constexpr (MODE >= 3) {  // the score quantiser's integer rel = idx - zp (masked keys get the -1e30 sentinel below)
#pragma unroll
      for (int j = J0; j < MQ; ++j)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[j][sub][r] = grid_rel_m(s[j][sub][r], fq_k1, fq_slo, fq_shi);
            if constexpr (has_pad) s[j][sub][r] = __builtin_fminf(s[j][sub][r], padflag[sub][r]);
          }
    } else {
#pragma unroll
    for (int j = J0; j < MQ; ++j)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!PIPE_QK && sub < NSa[j]) s[j][sub][r] = __builtin_fmaf(s[j][sub][r], c1, mcneg[j]);
    }
    // causal / tail mask, classified per 16x16 sub-tile with wave-uniform tests: untouched, all masked, or mixed
#pragma unroll
    for (int j = J0; j < MQ; ++j) {
      if (i < tm0[j]) continue;
      const int lim_lo = causal ? min(rb[j] + off, Sk - 1) : Sk - 1;        // last visible key of the block's first row
      const int lim_hi = causal ? min(rb[j] + 15 + off, Sk - 1) : Sk - 1;   // ... of its last row
      const int klim = causal ? min(rb[j] + c + off, Sk - 1) : Sk - 1;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub >= NSa[j]) continue;   // (masked for the whole block: not computed at all)
        const int k0 = 64 * i + 16 * sub;
        if (k0 > lim_hi) {
          s[j][sub] = f4{NEGT, NEGT, NEGT, NEGT};
        } else if (k0 + 15 > lim_lo) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + 4 * g + r > klim) s[j][sub][r] = NEGT;
        }
      }
    }
    // online softmax per block, P^T packed for the second product
    const float thr = (i == 0) ? -1.0e20f : kThr;
#pragma unroll
    for (int j = J0; j < MQ; ++j) {
      if constexpr (MODE == 3) {  // statistics of the grid chain, per lane
        float mt = max3_raw(s[j][0][0], s[j][0][1], s[j][0][2]);
        mt = max3_raw(mt, s[j][0][3], s[j][1][0]);
        mt = max3_raw(mt, s[j][1][1], s[j][1][2]);
        mt = max3_raw(mt, s[j][1][3], s[j][2][0]);
        mt = max3_raw(mt, s[j][2][1], s[j][2][2]);
        mt = max3_raw(mt, s[j][2][3], s[j][3][0]);
        mt = max3_raw(mt, s[j][3][1], s[j][3][2]);
        mt = max3_raw(mt, s[j][3][3], mrl[j]);
        lsum[j] *= __builtin_amdgcn_exp2f((mrl[j] - mt) * fq_c2);
        mrl[j] = mt;
        f4 t4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) t4[r] += __builtin_amdgcn_exp2f((s[j][sub][r] - mt) * fq_c2);
        lsum[j] += (t4[0] + t4[1]) + (t4[2] + t4[3]);
        continue;
      }
      if constexpr (MODE == 4) {  // final pass of the grid chain: exponential against the row maximum, index of the probability
        const float plo = P.fq_p.lo, phi = P.fq_p.hi;
        if (P.clip) {  // the clipped grid form of the full-row kernel: clip(p (eta - gamma) + gamma, 0, 1) as one clamped fma, then the index
#pragma unroll
          for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __builtin_amdgcn_exp2f((s[j][sub][r] - mrl[j]) * fq_c2);
              const float pc = __builtin_amdgcn_fmed3f(__builtin_fmaf(e, pinv[j], P.clip_g), 0.0f, 1.0f);
              s[j][sub][r] = __builtin_amdgcn_fmed3f(__builtin_rintf(pc * P.fq_p.rscale), plo, phi);
            }
        } else {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f((s[j][sub][r] - mrl[j]) * fq_c2);
            s[j][sub][r] = __builtin_amdgcn_fmed3f(__builtin_rintf(e * pinv[j]), plo, phi);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f4 a = s[j][2 * u], bb = s[j][2 * u + 1];
          if constexpr (IN == IN_BF16) pb[j][u] = u4{pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(bb[0], bb[1]), pack2_bf16(bb[2], bb[3])};
          else pb[j][u] = u4{pack2_f16(a[0], a[1]), pack2_f16(a[2], a[3]), pack2_f16(bb[0], bb[1]), pack2_f16(bb[2], bb[3])};
        }
        continue;
      }
      if constexpr (MODE == 2) {  // final reference: exponentials, p = e / den, clip, pack
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (one fused multiply-add with the instruction's clamp bit, as in the full-row kernel: e * (w / den) + gamma)
            s[j][sub][r] = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_amdgcn_exp2f(s[j][sub][r]), pinv[j] * P.clip_w, P.clip_g), 0.0f, 1.0f);
          }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f4 a = s[j][2 * u], bb = s[j][2 * u + 1];
          if constexpr (IN == IN_BF16) pb[j][u] = u4{pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(bb[0], bb[1]), pack2_bf16(bb[2], bb[3])};
          else pb[j][u] = u4{pack2_f16(a[0], a[1]), pack2_f16(a[2], a[3]), pack2_f16(bb[0], bb[1]), pack2_f16(bb[2], bb[3])};
        }
        continue;
      }
      // row maximum of the exponent arguments (fma / select results: no canonicalising v_max is needed in front)
      // (v_max3 written out: from fmaxf the compiler puts two canonicalising v_max x,x,x in front of every chain)
      float mt;
#ifdef OEH_R5_MAXCHAIN
      mt = max3_raw(s[j][0][0], s[j][0][1], s[j][0][2]);
      mt = max3_raw(mt, s[j][0][3], s[j][1][0]);
      mt = max3_raw(mt, s[j][1][1], s[j][1][2]);
      mt = max3_raw(mt, s[j][1][3], s[j][2][0]);
      mt = max3_raw(mt, s[j][2][1], s[j][2][2]);
      mt = max3_raw(mt, s[j][2][3], s[j][3][0]);
      mt = max3_raw(mt, s[j][3][1], s[j][3][2]);
      mt = max3_raw(mt, s[j][3][3], s[j][3][3]);
#else
      if (NSa[j] == 1) mt = max_first<1>(s[j]);        // (one statement each: oeh_common.h)
      else if (NSa[j] == 2) mt = max_first<2>(s[j]);
      else if (NSa[j] == 3) mt = max_first<3>(s[j]);
      else mt = max16_tree(s[j]);
#endif
      // Move the reference: always on the first tile (to that tile's row maximum, unless every key of it is masked),
      // later only for rows whose maximum exceeds it by 2^8.  The common case is decided on the LANE maxima (no cross-lane
      // step); the row maximum is formed only when some row moves.  Decided per ROW, so that a row's result depends on
      // its own keys only (bitwise causality); the wave-uniform branch merely skips the code when no row moves.
      // Key padding: a row whose keys so far were all absorbed by the mask (l == 0: a left-padded sample) has no reference yet -
      // its first visible tile sets it, as tile 0 does for every other row (else very negative scores behind a masked first tile
      // would all underflow against the initial reference 0, and under the vanilla softmax the row would pass for one without a
      // visible key).  lacc holds the row sum in every register of every lane of the row: the decision stays per row.
      float thr_j = thr;
      if constexpr (has_pad && MODE == 0) thr_j = (lacc[j][0] == 0.0f) ? -1.0e20f : thr;
      if constexpr (has_pad && MODE == 1) {
        // the statistics pass of the two-pass clipped form: the same rule.  Its row sum is dealt over the row's four lanes
        // (c, c + 16, c + 32, c + 48): "nothing accumulated yet" = none of them holds a non-zero share - one ballot, folded to the row's bit
        unsigned long long seen = __builtin_amdgcn_ballot_w64(lsum[j] != 0.0f);
        seen |= seen >> 32;
        seen |= seen >> 16;
        thr_j = ((seen >> c) & 1ull) ? thr : -1.0e20f;
      }
      if (__builtin_amdgcn_ballot_w64(mt > thr_j) != 0) {
        mt = row_allreduce_max(mt);
        const float delta = (mt > thr_j) ? mt : 0.0f;
        mcneg[j] -= delta;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (sub < NSa[j]) s[j][sub][r] -= delta;
        if (i != 0) {
          float alpha = __builtin_amdgcn_exp2f(-delta);
          if constexpr (has_pad && (MODE == 0 || MODE == 1)) alpha = (thr_j < -1.0e19f) ? 1.0f : alpha;  // nothing accumulated yet (and exp2(-delta) may overflow)
          if constexpr (MODE == 1) lsum[j] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) lacc[j][r] *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[j][dt][r] *= alpha;
          if constexpr (SRC32) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
              for (int r = 0; r < 4; ++r) ox[j][dt][r] *= alpha;
          }
        }
      }
      if constexpr (PIPE_PV) continue;   // (the exponentials follow in two halves, the second one between the first half's MFMAs: below)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[j][sub][r] = __builtin_amdgcn_exp2f(s[j][sub][r]);
      if constexpr (MODE == 1) {  // statistics pass: the sum of the exponentials themselves (fp32), nothing else
        f4 t4 = (s[j][0] + s[j][1]) + (s[j][2] + s[j][3]);
        lsum[j] += (t4[0] + t4[1]) + (t4[2] + t4[3]);
        continue;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f4 a = s[j][2 * u], bb = s[j][2 * u + 1];
        if constexpr (IN == IN_BF16) pb[j][u] = u4{pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(bb[0], bb[1]), pack2_bf16(bb[2], bb[3])};
        else pb[j][u] = u4{pack2_f16(a[0], a[1]), pack2_f16(a[2], a[3]), pack2_f16(bb[0], bb[1]), pack2_f16(bb[2], bb[3])};
      }
    }
#endif  // OEH_KO == 3
    if constexpr (FIRST) {
      // V tile 0 landed for every wave (stage 1 may still be in flight); every wave has its Q operands, so the Q stage
      // can now be refilled with stage 2
      if (!GATE && 1 < n_kt) wait_vm(std::integral_constant<int, 2>{});
      else wait_vm(std::integral_constant<int, 0>{});
      barrier_mem();
      if (GATE && 1 < n_kt) issue_next();  // stage 1, held back while its slot carried the gate's input rows
      if (2 < n_kt) issue_next();
    }
    // O^T += V^T P^T and l += 1^T P^T; every V^T fragment is read once and used by all active blocks
    if constexpr (MODE == 1 || MODE == 3) return;
    if constexpr (PIPE_PV) {
      // accumulating MFMA IN PLACE (inline asm, "+v"): from the builtin the register allocator gave the second half's results new registers and copied them
      // back at the loop's end - ten v_mov_b64 behind waits for the matrix core, per tile.  (Outside the compiler's hazard model: the accumulators are next
      // read by vector instructions behind the following tile's score MFMAs, or in the epilogue behind the padding after the loop.)
      auto mfma_acc = [&](f4& acc, const u4 av, const u4 bv) {
        if constexpr (IN == IN_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
      };
      // e = 0..15: element of the block's score tile.  PLACED (volatile asm keeps its position among the MFMA statements; from the builtin the instruction
      // selector sinks all of them to their first use behind the last MFMA of the half)
      auto exp1 = [&](const int j, const int e) { asm volatile("v_exp_f32_e32 %0, %0" : "+v"(s[j][e >> 2][e & 3])); };
      auto pack_half = [&](const int j, const int u) {   // the visible sub-tiles of half u (NSa[j] - 2 u >= 1 of them); a masked one: zeros
        const f4 a = s[j][2 * u];
        f4 bb = f4{0.f, 0.f, 0.f, 0.f};
        if (NSa[j] - 2 * u >= 2) bb = s[j][2 * u + 1];
        if constexpr (IN == IN_BF16) pb[j][u] = u4{pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(bb[0], bb[1]), pack2_bf16(bb[2], bb[3])};
        else pb[j][u] = u4{pack2_f16(a[0], a[1]), pack2_f16(a[2], a[3]), pack2_f16(bb[0], bb[1]), pack2_f16(bb[2], bb[3])};
      };
      auto read_v = [&](const int u, u4 (&va)[DT]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const unsigned char* a0 = vaddr[dt] + soff + u * 32 * ROWB;
          const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
          const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 16 * ROWB));
          const u2 l2 = __builtin_bit_cast(u2, lo), h2 = __builtin_bit_cast(u2, hi);
          va[dt] = u4{l2.x, l2.y, h2.x, h2.y};
        }
      };
      constexpr int NB = MQ - J0;              // active blocks
      // HAZARD RULE of this path (the inline-asm MFMAs are outside the compiler's hazard model): a register an MFMA reads must not have been written by a
      // vector instruction in the two issue slots in front of it (the compiler keeps that distance for its own MFMAs: the s_nop it puts behind a v_mov of
      // the ones operand).  The ones operand is therefore materialised ahead (round 6: once, in front of the tile loop), the packed P
      // of a half is followed by the half's V^T reads, and the second half's conversions by an explicit s_nop.  (Found as NaN row sums at MQ == 1.)
#ifdef OEH_R5_ONES_PER_TILE
      u4 ones_v = ones;
      asm volatile("" : "+v"(ones_v));
#else
      const u4 ones_v = ones_live;
#endif
      // keys 0-31 of every block: exponentials, packed
#pragma unroll
      for (int j = J0; j < MQ; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < 4 * NSa[j]) exp1(j, e);
      }
      // (trans -> VALU: a conversion that reads an exponential needs one wait state; the compiler does not insert it behind inline asm - structural
      // here, and checked on the built library's disassembly by tools/check_disasm.py: ADVICE r5)
      asm volatile("s_nop 0" ::: "memory");
#pragma unroll
      for (int j = J0; j < MQ; ++j) pack_half(j, 0);
      {
      }
      u4 va[DT];
      read_v(0, va);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1" ::: "memory");   // (the hazard rule, whatever the scheduler did with the reads above)
      prio_hi();
      // first half's NB (DT + 1) MFMAs, the exponentials of keys 32-63 (8 per block with all four sub-tiles, 4 with three, none with fewer) between them
      // (about one per gap: a v_exp_f32 is the 8 issue cycles an MFMA of this shape leaves), the conversions behind
      constexpr int E2_0 = (J0 == 0) ? 4 * (NSa[0] > 2 ? NSa[0] - 2 : 0) : 0;             // block 0's exponentials of the second half (not active: none)
      constexpr int E2_1 = (MQ == 2) ? 4 * (NSa[1] > 2 ? NSa[1] - 2 : 0) : 0;
      {
        constexpr int NM = NB * (DT + 1), NE = E2_0 + E2_1;     // MFMAs of the half; exponentials to place (block-major: block 0's E2_0 first)
        constexpr int TWO = NE > NM ? NE - NM : 0;              // the first TWO gaps take two exponentials, the others one (all indices below are closed forms of m:
        auto exp2nd = [&](const int n) {                        // a running counter would make the score tile a dynamically indexed array - in scratch memory)
          if (MQ == 2 && J0 == 0 && n >= E2_0) exp1(1, 8 + n - E2_0);
          else exp1(J0, 8 + n);
        };
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          if (m < NB) mfma_acc(lacc[J0 + m], ones_v, pb[J0 + m][0]);
          else mfma_acc(o[J0 + (m - NB) % NB][(m - NB) / NB], va[(m - NB) / NB], pb[J0 + (m - NB) % NB][0]);
          const int first = m < TWO ? 2 * m : TWO + m, cnt = m < TWO ? 2 : 1;
#pragma unroll
          for (int q = 0; q < cnt; ++q)
            if (first + q < NE) exp2nd(first + q);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = (NM < TWO ? 2 * NM : TWO + NM); n < NE; ++n) exp2nd(n);   // (whatever is left: none when NE <= 2 NM)
      }
      if constexpr (E2_0 + E2_1 > 0) {
        asm volatile("s_nop 0" ::: "memory");   // (the last placed exponential -> its conversion: as above)
#pragma unroll
        for (int j = J0; j < MQ; ++j)
          if (NSa[j] > 2) pack_half(j, 1);
        read_v(1, va);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");   // (the conversions above -> the MFMAs below: the hazard rule)
#pragma unroll
        for (int j = J0; j < MQ; ++j)
          if (NSa[j] > 2) mfma_acc(lacc[j], ones_v, pb[j][1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int j = J0; j < MQ; ++j)
            if (NSa[j] > 2) mfma_acc(o[j][dt], va[dt], pb[j][1]);
      }
      prio_lo();
      return;
    }
    prio_hi();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int j = J0; j < MQ; ++j) lacc[j] = mfma16<IN>(ones, pb[j][u], lacc[j]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const unsigned char* a0 = vaddr[dt] + soff + u * 32 * ROWB;
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 16 * ROWB));
        const u2 l2 = __builtin_bit_cast(u2, lo), h2 = __builtin_bit_cast(u2, hi);
        const u4 va = u4{l2.x, l2.y, h2.x, h2.y};
#pragma unroll
        for (int j = J0; j < MQ; ++j) o[j][dt] = mfma16<IN>(va, pb[j][u], o[j][dt]);
        if constexpr (SRC32) {  // the lo image of V
          const s4 lol = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + STAGEB));
          const s4 hil = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + STAGEB + 16 * ROWB));
          const u2 l3 = __builtin_bit_cast(u2, lol), h3 = __builtin_bit_cast(u2, hil);
          const u4 vl = u4{l3.x, l3.y, h3.x, h3.y};
#pragma unroll
          for (int j = J0; j < MQ; ++j) ox[j][dt] = mfma16<IN>(vl, pb[j][u], ox[j][dt]);
        }
      }
    }
    prio_lo();
  };

  auto finish_stats = [&]() {  // two-pass forms: the row's denominator from the lanes' shares
#pragma unroll
    for (int j = 0; j < MQ; ++j) {
      if constexpr (FQ2) {
        float mr = mrl[j];
        mr = row4_max(mr);
        float l = lsum[j] * __builtin_amdgcn_exp2f((mrl[j] - mr) * fq_c2);
        l = row4_sum(l);
        const float m = (mr - kGridMagic) * P.fq_s.scale;            // the reference's row maximum, fl(scale * rel_max)
        if (P.base != 0) l = l + exp_acc(m * -1.0f);                  // softmax_1: + 1*exp(-max)  (softmax_1.py:18-20)
        mrl[j] = mr;
        pinv[j] = (1.0f / l) * (P.clip ? P.clip_w : P.fq_p.rscale);  // e * this -> the probability's index (before rint); clipped: -> p (eta - gamma)
      } else {
        float l = lsum[j];
        l = row4_sum(l);
        if (P.base != 0) l = l + __builtin_amdgcn_exp2f(mcneg[j]);    // softmax_1: + 1*exp(-reference)
        pinv[j] = 1.0f / l;
        if constexpr (PAD) {  // vanilla softmax, a row without a visible key (l == 0): no 0 * inf in the final pass - the epilogue forms the row
          dead1[j] = (P.base == 0) && (l == 0.0f);
          if (dead1[j]) pinv[j] = 0.0f;
        }
      }
    }
  };
  using J0_0 = std::integral_constant<int, 0>;
  using J0_1 = std::integral_constant<int, 1>;
  using NS_FULL = std::integral_constant<int, 0x44>;
  // bodies specialised for partly masked tiles exist where the placed order does (tile: PIPE_PV)
#if defined(OEH_NO_PIPE_PV) || defined(OEH_NO_NSUB) || defined(OEH_R5_MAXCHAIN) || OEH_KO != 0
  constexpr bool NSV = false;
#else
  constexpr bool NSV = (TP == 0) && !SRC32 && D <= 64;
#endif
  int lh[MQ];   // last key any row of block j may see: a 16-key sub-tile that starts behind it is masked for the whole block
#pragma unroll
  for (int j = 0; j < MQ; ++j) lh[j] = causal ? min(rb[j] + 15 + off, Sk - 1) : Sk - 1;
  if constexpr (SRC32) {
    auto stream32 = [&](auto modec) {  // one pass over the register-staged stream (stage 0 is in the registers on entry)
      int slot_r = 0;
      for (int i = 0; i < n_kt; ++i) {
        commit_regs(slot_r);                  // stage i (its loads were issued a tile ago) -> LDS; the slot's last readers (tile i-2) are behind the previous barrier
        barrier_mem();
        if (i + 1 < n_kt) load_regs(i + 1);   // lands while tile i is computed
        const int soff = slot_r * SLOT32;
        slot_r ^= 1;
        if (i >= nkb[MQ - 1]) continue;
        if (MQ == 2 && i >= nkb[0]) {
          if constexpr (MQ == 2) tile(J0_1{}, std::false_type{}, modec, NS_FULL{}, i, soff);
        } else {
          tile(J0_0{}, std::false_type{}, modec, NS_FULL{}, i, soff);
        }
      }
    };
    if constexpr (TP != 0) {
      stream32(std::integral_constant<int, CLIP ? 1 : 3>{});
      finish_stats();
      barrier_mem();  // the last stages of the first pass have been read by every wave
      load_regs(0);
      stream32(std::integral_constant<int, CLIP ? 2 : 4>{});
    } else {
      stream32(std::integral_constant<int, 0>{});
    }
  } else {
  using MODE_A = std::integral_constant<int, CLIP ? 1 : (FQ2 ? 3 : 0)>;  // the (first) pass over the keys
  tile(J0_0{}, std::true_type{}, MODE_A{}, NS_FULL{}, 0, 0);  // every block sees key 0: tile 0 is computed by every wave, for all its blocks
  OEH_STAMP(6);
  int slot_i = 1;
  for (int i = 1; i < n_kt; ++i) {
    // ---- stage i landed for every wave
    if (i + 1 < n_kt) wait_vm(std::integral_constant<int, 2>{});
    else wait_vm(std::integral_constant<int, 0>{});
    barrier_mem();
    if (i < 8) OEH_STAMP(4 + 3 * i);
    // into the stage every wave finished reading one iteration ago.  (Requesting it later, behind the score MFMAs just
    // queued - a wave spends ~350 cycles per tile issuing its four 1-KiB pieces - measured no better: 21.9-22.4 vs 21.5-21.7 us
    // on dense S=512, equal on the causal shape.)
#if defined(OEH_PIPE_QK) && OEH_PIPE_QK == 1
    // the request rides inside the tile (one piece behind each sub-tile's score MFMAs) where the tile is a full placed one; elsewhere here, as before
    const bool full_tile = (TP == 0) && !SRC32 && !PAD && D == 64 && i < nkb[MQ - 1] && !(MQ == 2 && i >= nkb[0] && NSV && ((lh[1] - 64 * i) >> 4) + 1 < 4);
    const bool ride = (i + 2 < n_kt) && full_tile && (nx_tile * 64 + 64 <= Sk);
    if (i + 2 < n_kt && !ride) issue_next();
#else
    constexpr bool ride = false;
    if (i + 2 < n_kt) issue_next();
#endif
    if (i < 8) OEH_STAMP(5 + 3 * i);
    const int soff = slot_i * STAGEB;
    slot_i = (slot_i == R - 1) ? 0 : slot_i + 1;
    if (i >= nkb[MQ - 1]) continue;  // this wave's rows end before this tile (causal): nothing to compute
    // Round 6: a tile in which only the first n < 4 sixteen-key sub-tiles hold a key the block may see (every block's causal diagonal tile, the ragged last
    // tile) runs a body specialised for n: for the block that ends in this tile (block 0 beside a full block 1; or block 1 alone; or the only block)
    if (MQ == 2 && i >= nkb[0]) {
      if constexpr (MQ == 2) {
        const int n1 = NSV ? ((lh[1] - 64 * i) >> 4) + 1 : 4;
        if constexpr (NSV && (OEH_NSUB_MASK & 1)) {
          if (n1 == 1) tile(J0_1{}, std::false_type{}, MODE_A{}, std::integral_constant<int, 0x14>{}, i, soff);
          else if (n1 == 2) tile(J0_1{}, std::false_type{}, MODE_A{}, std::integral_constant<int, 0x24>{}, i, soff);
          else if (n1 == 3) tile(J0_1{}, std::false_type{}, MODE_A{}, std::integral_constant<int, 0x34>{}, i, soff);
          else tile(J0_1{}, std::false_type{}, MODE_A{}, NS_FULL{}, i, soff, ride);
        } else {
          tile(J0_1{}, std::false_type{}, MODE_A{}, NS_FULL{}, i, soff, ride);
        }
      }
    } else {
      if constexpr (NSV && (OEH_NSUB_MASK & 28) != 0) {
        const int n0 = ((lh[0] - 64 * i) >> 4) + 1;
        const int nl = ((lh[MQ - 1] - 64 * i) >> 4) + 1;   // the last block: full, or (MQ == 1) the block itself
        if (MQ == 2 && nl < 4) tile(J0_0{}, std::false_type{}, MODE_A{}, NS_FULL{}, i, soff);   // (both blocks partial - Sq != Sk layouts: the general body)
        else if (n0 == 1 && (OEH_NSUB_MASK & 4)) tile(J0_0{}, std::false_type{}, MODE_A{}, std::integral_constant<int, 0x41>{}, i, soff);
        else if (n0 <= 2 && (OEH_NSUB_MASK & 8)) tile(J0_0{}, std::false_type{}, MODE_A{}, std::integral_constant<int, 0x42>{}, i, soff);
        else if (n0 == 3 && (OEH_NSUB_MASK & 16)) tile(J0_0{}, std::false_type{}, MODE_A{}, std::integral_constant<int, 0x43>{}, i, soff);
        else tile(J0_0{}, std::false_type{}, MODE_A{}, NS_FULL{}, i, soff, ride);
      } else {
        tile(J0_0{}, std::false_type{}, MODE_A{}, NS_FULL{}, i, soff, ride);
      }
    }
    if (i < 8) OEH_STAMP(6 + 3 * i);
  }
  if constexpr (TP != 0) {
    // ---- denominators (sum over the 4 lanes of a row), then the same stream once more: both slots of the ring are primed
    // again and every tile goes through the loop form (no Q stage this time: the operands are in registers)
    using MODE_B = std::integral_constant<int, CLIP ? 2 : 4>;
    finish_stats();
    barrier_mem();  // every wave has left the last stages of the first pass; nothing is in flight
    kcur = kbase0;
    vcur = vbase0;
    nx_tile = 0;
    nx_slot = 0;
    issue_next();
    if (1 < n_kt) issue_next();
    int slot_b = 0;
    for (int i = 0; i < n_kt; ++i) {
      if (i + 1 < n_kt) wait_vm(std::integral_constant<int, 2>{});
      else wait_vm(std::integral_constant<int, 0>{});
      barrier_mem();
      if (i + 2 < n_kt) issue_next();
      const int soff = slot_b * STAGEB;
      slot_b = (slot_b == R - 1) ? 0 : slot_b + 1;
      if (i >= nkb[MQ - 1]) continue;
      if (MQ == 2 && i >= nkb[0]) {
        if constexpr (MQ == 2) tile(J0_1{}, std::false_type{}, MODE_B{}, NS_FULL{}, i, soff);
      } else {
        tile(J0_0{}, std::false_type{}, MODE_B{}, NS_FULL{}, i, soff);
      }
    }
  }
  }
  OEH_STAMP(2);
#ifndef OEH_NO_PIPE_PV
  if constexpr (!SRC32 && TP == 0) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // (the in-place MFMAs of the last tile, written as inline asm, against the epilogue's vector reads of their results)
#endif

  // ---- epilogue: denominators and gate, O^T staged through a free LDS stage so that global stores are whole rows
  // (per-lane stores of the MFMA layout would touch 16 rows x 32 B per instruction).  Stage n_kt % R is free: its
  // last reader was tile n_kt - 3 and no DMA is in flight.  Each wave owns 16*MQ rows of it: no workgroup barrier.
  // (lane-derived addresses come from an opaque copy of the lane id: formed here, not kept live across the loop where
  // the MQ=2 variant has no register to spare)
#if OEH_KO >= 2 && OEH_KO <= 4
  if (!(o[0][0][0] == 1.2345e-31f && lacc[0][1] == 5.4321e-30f)) return;  // (never true: the accumulators stay live, the epilogue does not run)
#endif
  constexpr int XM = (CPR < 8 ? CPR : 8) - 1;
  unsigned char* ebase = lds + (n_kt % R) * STAGEB + wave * (16 * MQ * ROWB);
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int ce = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
  for (int j = 0; j < MQ; ++j) {
    const int qrow = rb[j] + ce;
    float den = lacc[j][0];
    if (P.base != 0) den = den + __builtin_amdgcn_exp2f(mcneg[j]);  // softmax_1: + 1*exp(-reference)  (vutils/softmax_1.py:18-20)
    float rowscale = 1.0f / den;
    if constexpr (TP != 0) rowscale = 1.0f;  // the clipped probabilities / the probability indices went into the product as they are
    if constexpr (PAD && (TP == 0 || TP == 1)) {
      // Vanilla softmax and a row WITHOUT a visible key (a fully padded sample; a left-padded one under the causal mask): every score
      // of the reference is the same finfo.min, its probabilities are uniform over ALL Sk keys - which a kernel that skips masked
      // tiles has not accumulated (l == 0 here).  Such rows - rare - take the mean of V straight from memory (the clipped two-pass
      // form: clip(w / Sk + gamma, 0, 1) times the sum of V - models/softmax.py:10-13 on a uniform row).
      if (P.base == 0) {
        const bool dead = (TP == 1) ? dead1[j] : (den == 0.0f);
        if (__builtin_amdgcn_ballot_w64(dead) != 0) {
          if (dead) {
            f4 acc[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SRC32) {
              const float* vp = reinterpret_cast<const float*>(P.v) + bh_offset(b, P.vs_b, h, P.vs_h) + 4 * ge;
              for (int kk = 0; kk < Sk; ++kk)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[dt] = acc[dt] + *reinterpret_cast<const f4*>(vp + (long)kk * P.vs_s + 16 * dt);
            } else {
              const unsigned short* vp = reinterpret_cast<const unsigned short*>(P.v) + bh_offset(b, P.vs_b, h, P.vs_h) + 4 * ge;
              for (int kk = 0; kk < Sk; ++kk)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                  const u2 w = *reinterpret_cast<const u2*>(vp + (long)kk * P.vs_s + 16 * dt);
                  acc[dt] = acc[dt] + f4{In<IN>::to_f32((unsigned short)(w.x & 0xffffu)), In<IN>::to_f32((unsigned short)(w.x >> 16)),
                                         In<IN>::to_f32((unsigned short)(w.y & 0xffffu)), In<IN>::to_f32((unsigned short)(w.y >> 16))};
                }
            }
            float rs = 1.0f / (float)Sk;
            if constexpr (TP == 1) rs = __builtin_fminf(__builtin_fmaxf(rs * P.clip_w + P.clip_g, 0.0f), 1.0f);  // the clipped uniform probability
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              o[j][dt] = acc[dt] * rs;
              if constexpr (SRC32) ox[j][dt] = f4{0.f, 0.f, 0.f, 0.f};
            }
            rowscale = 1.0f;
          }
        }
      }
    }
    if (P.gate != nullptr && qrow < Sq) rowscale = rowscale * P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qrow * P.gs_s];
    if constexpr (GATE) rowscale = rowscale * gate_row[j];
    // TP = 2: [scale of the quantised P] [context quantiser] gate [context quantiser] - the full-row kernel's epilogue chain,
    // in whole passes over the block's values (oeh_common.h: ctx_chain), written back into the accumulators
    if constexpr (FQ2) {
      float xs[DT * 4];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = o[j][dt][r];
          if constexpr (SRC32) x = __builtin_fmaf(ox[j][dt][r], kSplitDown, x);
          xs[dt * 4 + r] = P.fq_p.scale * x;
        }
      ctx_chain<DT * 4>(xs, P.fq_c, P.ctx_before_gate, P.gate != nullptr, rowscale);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[j][dt][r] = xs[dt * 4 + r];
    }
    auto finish = [&](float x) { return FQ2 ? x : x * rowscale; };
    if constexpr (OUT32) {  // fp32 output straight from the accumulators: 16 B per lane, 64 B per row and instruction (fp32 storage; O32:
                            // 16-bit storage with o_dtype = OEH_F32 - the kernel's arithmetic before the output rounding, include/oeh.h)
      if (qrow < Sq) {
        float* orow = reinterpret_cast<float*>(P.o) + bh_offset(b, P.os_b, h, P.os_h) + (long)qrow * P.os_s + 4 * ge;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = o[j][dt][r];
            if constexpr (SRC32) { if (!FQ2) x = __builtin_fmaf(ox[j][dt][r], kSplitDown, x); }
            ov[r] = finish(x);
          }
          store_wt16(orow + 16 * dt, u4{f32_bits(ov[0]), f32_bits(ov[1]), f32_bits(ov[2]), f32_bits(ov[3])});
        }
      }
    } else {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u2 w;
      if constexpr (IN == IN_BF16) {
        w.x = pack2_bf16(finish(o[j][dt][0]), finish(o[j][dt][1]));
        w.y = pack2_bf16(finish(o[j][dt][2]), finish(o[j][dt][3]));
      } else {
        w.x = pack2_f16(finish(o[j][dt][0]), finish(o[j][dt][1]));
        w.y = pack2_f16(finish(o[j][dt][2]), finish(o[j][dt][3]));
      }
      *reinterpret_cast<u2*>(ebase + (16 * j + ce) * ROWB + ((((2 * dt + (ge >> 1)) ^ (ce & XM)) << 4) | ((ge & 1) << 3))) = w;
    }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before it reads them back
  if constexpr (!OUT32) {
    unsigned short* obase = reinterpret_cast<unsigned short*>(P.o) + bh_offset(b, P.os_b, h, P.os_h);
    const int lr = lane_e / CPR, lc = lane_e % CPR;
    static_assert(16 % RPP == 0, "a store pass stays inside one query block");
#pragma unroll
    for (int pass = 0; pass < (16 * MQ) / RPP; ++pass) {
      const int row = pass * RPP + lr;                            // row of the wave's staging area
      const int grow = rb[(pass * RPP) / 16] + (pass * RPP) % 16 + lr;  // its query row
      const u4 w = *reinterpret_cast<const u4*>(ebase + row * ROWB + ((lc ^ (row & XM)) << 4));
      if (grow < Sq) store_wt16(obase + (long)grow * P.os_s + lc * 8, w);
    }
  }
  OEH_STAMP(3);
#ifdef OEH_TIMELINE
  if (stamp != nullptr && lane == 0) stamp[31] = __builtin_amdgcn_s_memrealtime();
#endif
}
#undef OEH_STAMP

template <int D, int MQ, int IN>
static void launch_flash_d_mq_in(const AttnParams& P, unsigned grid, hipStream_t st) {
  const bool pad = P.pad != nullptr || P.full != nullptr, gate = P.gh != nullptr;  // (the PAD variants also serve a (B,1,Sq,Sk) mask)
  if (P.src32) {  // fp32 storage read directly, fp32 output; no in-kernel gate predictor on this path
    if constexpr (IN == IN_F16 && !(D == 128 && MQ == 2)) {  // (d = 128 with two blocks per wave: never selected, oeh_api.hip: flash_mq)
      if (P.fq_s.en && pad) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false, true, 2>), dim3(grid), dim3(256), 0, st, P);
      else if (P.fq_s.en) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, true, 2>), dim3(grid), dim3(256), 0, st, P);
      else if (P.clip && pad) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false, true, 1>), dim3(grid), dim3(256), 0, st, P);
      else if (P.clip) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, true, 1>), dim3(grid), dim3(256), 0, st, P);
      else if (pad) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false, true>), dim3(grid), dim3(256), 0, st, P);
      else hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, true>), dim3(grid), dim3(256), 0, st, P);
    }
    return;
  }
  if (P.fq_s.en) {  // (oeh_api.hip: flash_fq_eligible - the grid chain: no clip, no in-kernel predictor; key padding as a 0 / <= -1e4 vector)
    if (pad) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false, false, 2>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, false, 2>), dim3(grid), dim3(256), 0, st, P);
    return;
  }
  if (P.clip) {  // (oeh_api.hip: flash_clip_eligible - no in-kernel predictor)
    if (pad) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false, false, 1>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, false, 1>), dim3(grid), dim3(256), 0, st, P);
    return;
  }
  if constexpr (D == 64) {  // (oeh_api.hip: the out32 rule - head dim 64: plain, + key padding / a (B,1,Sq,Sk) mask, + the in-kernel gate predictor)
    if (P.out32) {
      if (pad) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false, false, 0, true>), dim3(grid), dim3(256), 0, st, P);
      else if (gate) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, true, false, 0, true>), dim3(grid), dim3(256), 0, st, P);
      else hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, false, 0, true>), dim3(grid), dim3(256), 0, st, P);
      return;
    }
  }
  if constexpr (D == 128 && MQ == 1) {  // (head dim 128: the plain form; flash_mq gives one block per wave there)
    if (P.out32) {
      hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false, false, 0, true>), dim3(grid), dim3(256), 0, st, P);
      return;
    }
  }
  if (pad) {
    if (gate) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, true>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, true, false>), dim3(grid), dim3(256), 0, st, P);
  } else {
    if (gate) hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, true>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_flash_kernel<D, IN, MQ, false, false>), dim3(grid), dim3(256), 0, st, P);
  }
}

template <int D, int MQ>
static int launch_flash_d_mq(const AttnParams& P, int in, hipStream_t st) {
  const unsigned grid = (unsigned)(P.nQT * P.nBHpad);
  if (in == IN_BF16) launch_flash_d_mq_in<D, MQ, IN_BF16>(P, grid, st);
  else launch_flash_d_mq_in<D, MQ, IN_F16>(P, grid, st);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// P.nQT = ceil(Sq / (64*MQ)) workgroups per head; MQ is chosen by the host (oeh_api.hip: flash_mq)
template <int D>
static int launch_flash_d(const AttnParams& P, int in, int mq, hipStream_t st) {
  if (mq == 1) return launch_flash_d_mq<D, 1>(P, in, st);
  return launch_flash_d_mq<D, 2>(P, in, st);
}

}  // namespace oeh
