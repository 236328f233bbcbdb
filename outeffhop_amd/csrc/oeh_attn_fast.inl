// Fast fused attention kernel for 16-bit storage (f16 / bf16) - the headline path.
//
// Same structure as the general kernel in oeh_attn_mfma.inl (full score rows in registers, swapped
// products S^T = K Q^T / O^T = V^T P^T on v_mfma_f32_16x16x32) with everything the hot configurations do
// not need taken out of the per-element chain:
//   * masks: none | key-padding vector | analytic causal.  Tiles strictly below the diagonal and inside Sk
//     take a mask-free path (wave-uniform test per 16-key tile); masked / tail elements are set to a large
//     negative constant, their exponential is exactly 0 as in the reference.
//   * scale is folded into the exponent: exp(x*s - m*s) = exp2(fma(x, s*log2e, -m*s*log2e)); the row max is
//     taken on the raw scores (s > 0).  One fma + v_exp_f32 + add per element.
//   * without clipping the probabilities stay un-normalised (e <= 1) through P@V and each output row is
//     multiplied by 1/den once (16 multiplies instead of Sk); with clipping (CLIP) the normalised value is
//     formed first because clip(p*(eta-gamma)+gamma, 0, 1) is not linear.
//   * K/V tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging) into two 3-slot rings (K, V),
//     two tiles in flight; the stream keeps running through the softmax phase so the first V tiles are already
//     resident when P@V starts.  Separate rings make every LDS slot offset a compile-time constant in the unrolled loops.  The XOR swizzle is applied on the per-lane SOURCE address (the LDS image of
//     one wave-instruction is lane-linear), reads use the same swizzle: conflict-free (tools/lds_bank_sim.py).
//     One raw s_barrier per tile, counted s_waitcnt vmcnt(N) (never 0 in the loop).
// Everything else ((B,1,Sq,Sk) masks, fp32 storage, non-power-of-two score division, gamma > 0) runs the general kernel;
// the fused fake-quantisers are the FQ variant below.
#pragma once
#include "oeh_attn_mfma.inl"

namespace oeh {

template <int G>
__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  // wait until all but the youngest G*tiles LDS-DMA instructions of this wave have landed
  if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (tiles == 1) {
    if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (G == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (G == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
}

// waves per SIMD the register allocator must leave room for: 3 at NT=32 (<= 168 VGPRs), LDS allows 3 workgroups (50 KB each)
template <int NT, int D, bool SRC32, bool GATE = false>
constexpr int fast_occupancy() {  // what the LDS rings allow (6 tiles of 64 x 2D bytes); the fp32 forms of the 512-key kernel
  // carry hi+lo operands and the staged tile (and, with the in-kernel gate predictor, its weight pairs): 2 waves per SIMD, no spills
  return D >= 128 ? 1 : ((SRC32 && (NT >= 32 || (GATE && D >= 64))) ? 2 : 3);
}

// GATE: the conditional per-token gate (include/oeh.h: gate_hidden ...) is computed in the kernel.  The layer-input rows
// of the workgroup arrive as one more K-shaped LDS-DMA tile; the first predictor layer is ONE small product on the matrix
// cores per wave, logits^T = W1 X^T, in the same swapped orientation as the scores (token on the lane, hidden units over
// registers), with the weights rounded to the storage dtype - what the reference's Linear does in a 16-bit model; the
// second layer, the sigmoid and the scaling are a few VALU operations per lane and one register carries the result to the
// epilogue.  A separate variant: the others carry none of it.
// FQ: the fused fake-quantisers (scores / probabilities / context); two forms, separate variants because a kernel that holds
// both allocates registers for the union (143 spilled).  FQ == 1: no key padding, no clipping (OPT) - the chain runs on the
// quantiser grid (see phase 2); FQ == 3: the same with a key-padding vector whose entries are 0 or <= -1e4 (include/oeh.h
// key_pad_boolean: BERT's and HF's masks) - a flag per key in LDS, one v_min per element.  FQ == 2: the per-element chain is the reference's op
// order literally - scale, quantise, masks added (not substituted), x - m, 1-ulp exp, normalise, [clip], quantise - as in
// the general kernel (oeh_attn_mfma.inl), on this kernel's data path.  The variant is compiled for the reference's
// configuration, scores AND probabilities quantised (context optional): a run-time test per quantiser and per four elements
// costs a branch and, at the join, register copies (1.5 VALU per element for the clip alone).  Other subsets, and the
// test-only uint8 index dumps (a uniform branch per four elements per quantiser even when off): general kernel.
// SRC32: fp32 storage read directly, as in the one-pass kernel (oeh_attn_flash.inl): the K-then-V tile stream comes
// through registers (32 B of fp32 per lane and piece), one tile ahead - committed at the top of an iteration, the next tile's
// loads issued right after the barrier - and is written to LDS as TWO fp16 images (hi, lo: oeh_common.h split8) in the layout
// the DMA would produce; every product is accumulated from the operand pairs (3 MFMAs per score k-step, 2 per context k-step),
// i.e. with fp32 accuracy.  Two ring slots shared by the K-then-V stream (a slot's readers are all behind the barrier that
// precedes its next commit); Q goes global -> registers directly.
// O32: 16-bit storage with the output taken from the fp32 accumulators (include/oeh.h: o_dtype = OEH_F32) - the same loops, only the
// epilogue's store differs (a runtime switch in the epilogue measured +2 ... +8 % on the production launches, round 4).
template <int NT, int D, int IN, bool CLIP, bool GATE, int FQ = 0, bool SRC32 = false, bool O32 = false>
__global__ __launch_bounds__(256, (fast_occupancy<NT, D, SRC32, GATE>())) void oeh_attn_fast_kernel(const AttnParams P) {
  static_assert(!SRC32 || IN == IN_F16, "fp32 storage: fp16 operand pairs, fp32 output");
  static_assert(!O32 || (!SRC32 && FQ == 0 && !(GATE && CLIP)), "fp32 output of 16-bit storage: the plain and clipped forms, the plain form + in-kernel gate");
  constexpr bool OUT32 = SRC32 || O32;
  static_assert(!FQ || !GATE, "the fake-quant variant has no in-kernel gate predictor");
  constexpr bool GRID = (FQ == 1 || FQ == 3), GRIDPAD = (FQ == 3);  // FQ == 3: the grid chain with a key-padding vector of 0 / <= -1e4 entries
  static_assert(IN == IN_F16 || IN == IN_BF16, "16-bit storage only");
  constexpr int KT = NT / 4;
  constexpr int ROWB = 2 * D;
  constexpr int TILEB = 64 * ROWB;
  constexpr int CPR = D / 8;          // 16-B chunks per row
  constexpr int RPP = 64 / CPR;       // rows per 1-KiB LDS-DMA piece
  constexpr int G = D / 32;           // pieces per wave per tile (tile = 4*G pieces)
  constexpr int KS = D / 32;
  constexpr int DT = D / 16;
  constexpr int R = 3;                // slots per ring (K ring, V ring)
  // (Measured and dropped, round 3: for rows of <= 128 keys - at most two K and two V tiles, a free ring slot each - ALL tiles requested in
  // the prologue instead of V tile 0 only once K tile 0 has landed: BERT-base 8.2 us against 7.65.  The launch is one burst of every
  // workgroup's requests against HBM; V tiles requested early only delay the K tiles everyone needs first.)
  constexpr float NEG = -3.0e38f;

  constexpr int SLOT32 = 2 * TILEB;   // SRC32: one ring slot = hi image + lo image
  constexpr int RINGB = SRC32 ? 2 * SLOT32 : 2 * R * TILEB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RINGB + NT * 16 * 4];
  float* lds_pad = reinterpret_cast<float*>(lds + RINGB);

  // The kernel arguments the prologue needs, requested in ONE round of scalar loads at entry (the compiler loads an argument where it
  // is first used: several dependent rounds of ~300 cycles each in front of the first LDS-DMA request, on every wave), and the
  // block id decoded without integer divisions (oeh_common.h: div_magic): 15.43 -> 14.80 us on the headline launch of the one-pass
  // kernel, same treatment here.
  {
    asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.nBHpad), "s"(P.nQT), "s"(P.nBH), "s"(P.H), "s"(P.Sq), "s"(P.Sk), "s"(P.skip_ok),
                 "s"(P.magic_nbh), "s"(P.magic_h), "s"(P.qs_b), "s"(P.qs_h), "s"(P.qs_s), "s"(P.ks_b), "s"(P.ks_h), "s"(P.ks_s), "s"(P.vs_b), "s"(P.vs_h),
                 "s"(P.vs_s), "s"(P.pad));
  }
  const int bid = blockIdx.x;  // (snake_block_id measured 3-4 % slower here: 8 q tiles per head, not all workgroups resident)
  int qt_rev, bh;
  if (SRC32 && P.head_major) block_to_tile(bid, P.nBHpad, P.nQT, P.head_major, qt_rev, bh);
  else div_magic((unsigned)bid, (unsigned)P.nBHpad, P.magic_nbh, qt_rev, bh);
  if (bh >= P.nBH) return;
  const int qt = P.nQT - 1 - qt_rev;
  int b, h;
  div_magic((unsigned)bh, (unsigned)P.H, P.magic_h, b, h);

  if constexpr (SRC32) fp16_overflow_clamp();  // out-of-range fp32 operands saturate (oeh_common.h)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int q0 = qt * 64 + wave * 16;
  const int qrow = q0 + c;
  const bool qvalid = qrow < P.Sq;
  const int off = P.Sk - P.Sq;

  int kend_wg = P.Sk;
  if (P.skip_ok) {
    kend_wg = min(P.Sk, max(0, qt * 64 + 64 + off));
  }
  const int n_kt = (kend_wg + 63) >> 6;
  const int T = 2 * n_kt;  // tiles in the K-then-V stream

  // ---- LDS-DMA source addressing: lane -> (row inside the piece, swizzled chunk)
  const unsigned short* kbase = reinterpret_cast<const unsigned short*>(P.k) + bh_offset(b, P.ks_b, h, P.ks_h);
  const unsigned short* vbase = reinterpret_cast<const unsigned short*>(P.v) + bh_offset(b, P.vs_b, h, P.vs_h);
  // Per-lane source offsets of this wave's G pieces inside a tile (row of the piece, swizzled 16-B chunk); a tile
  // further on is +64 rows.  Rows past Sk (last tile only) are redirected to row Sk-1.
  const int prow = lane / CPR, pch = lane % CPR;
  const unsigned lds_base = lds_offset(lds);
  auto piece_row = [&](int j) { return (wave * G + j) * RPP + prow; };
  // The K-then-V tile stream is issued strictly in order by `issue_next()`.  Round 5 (from the disassembly: every request cost ~12 vector
  // instructions of 64-bit per-lane address arithmetic - v_mad_u64_u32, v_lshl_add_u64, exec masking for the ragged rows - a fifth of the
  // clipped kernel's vector instructions): the scalar-base form the one-pass kernel has had since round 3.  A request is a wave-uniform
  // 64-bit base in scalar registers (it advances by 64 rows per tile) + a CONSTANT 32-bit byte offset per lane (row of the piece, swizzled
  // 16-B chunk; oeh_api.hip: fast_eligible bounds the row strides), two constant sets: K and V.  Only the ragged last tile adjusts offsets.
  unsigned noff[G], voff[G];   // the stream's current offsets (K first), and V's for the switch
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int row = piece_row(j);
    noff[j] = 2u * (unsigned)(row * (int)P.ks_s + (pch ^ swz_k<D>(row)) * 8);
    voff[j] = 2u * (unsigned)(row * (int)P.vs_s + ((((pch >> 1) ^ swz_v<D>(row)) << 1) | (pch & 1)) * 8);
  }
  const unsigned char* ncur = reinterpret_cast<const unsigned char*>(kbase);
  long nstep = 128 * P.ks_s;      // bytes per 64 rows
  int nrow2 = 2 * (int)P.ks_s;    // bytes per row (the ragged tile's redirection)
  int nx_tile = 0, nx_slot = 0;
  bool nx_isv = false;
  auto issue_next = [&]() {
    const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(((nx_isv ? R : 0) + nx_slot) * TILEB + wave * G * 1024));
    if (nx_tile * 64 + 64 > P.Sk) {  // wave-uniform: the ragged last tile - rows past Sk are redirected to row Sk-1 (finite data, masked later)
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int over = nx_tile * 64 + piece_row(j) - (P.Sk - 1);
        glds16_s(ncur, noff[j] - (over > 0 ? (unsigned)(over * nrow2) : 0u), slot + j * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(ncur, noff[j], slot + j * 1024);
    }
    ncur += nstep;
    ++nx_tile;
    nx_slot = (nx_slot == R - 1) ? 0 : nx_slot + 1;
    if (!nx_isv && nx_tile == n_kt) {  // K exhausted: continue with V tile 0 in the V ring
      nx_isv = true;
      nx_tile = 0;
      nx_slot = 0;
      ncur = reinterpret_cast<const unsigned char*>(vbase);
      nstep = 128 * P.vs_s;
      nrow2 = 2 * (int)P.vs_s;
#pragma unroll
      for (int j = 0; j < G; ++j) noff[j] = voff[j];
    }
  };

#ifdef OEH_TIMELINE  // diagnostic build of tools/timeline.py only (see oeh_attn_flash.inl)
  unsigned long long* stamp = nullptr;
  if (P.stamps != nullptr) stamp = P.stamps + ((long)bid * 4 + wave) * 32;
#define OEH_STAMP(slot)                                                                               \
  do {                                                                                                \
    if (stamp != nullptr && lane == 0) stamp[(slot)] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
#else
#define OEH_STAMP(slot) do { } while (0)
#endif
  OEH_STAMP(0);
#ifdef OEH_TIMELINE
  if (stamp != nullptr && lane == 0) stamp[30] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- Q rides the LDS-DMA stream, first, as a K-shaped tile in the V ring's last slot (first used by V tile R-1, long
  // after the operands below are in registers): the bytes in front of the first MFMA are Q + K tile 0, requested together,
  // instead of a register load of Q that had to land before the first transfer could even be issued.
  f4 treg[SRC32 ? G : 1][2];  // SRC32: the staged tile
  auto load_tile = [&](const int i) {  // tile i of the K-then-V stream -> registers
    if constexpr (SRC32) {
      const bool isv = i >= n_kt;
      const int t = isv ? i - n_kt : i;
      const float* src = isv ? reinterpret_cast<const float*>(P.v) + bh_offset(b, P.vs_b, h, P.vs_h)
                             : reinterpret_cast<const float*>(P.k) + bh_offset(b, P.ks_b, h, P.ks_h);
      const long srow = isv ? P.vs_s : P.ks_s;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int row = piece_row(j);
        const int kr = min(t * 64 + row, P.Sk - 1);  // rows past Sk: finite data, masked later
        const int ch = isv ? ((((pch >> 1) ^ swz_v<D>(row)) << 1) | (pch & 1)) : (pch ^ swz_k<D>(row));
        const float* p = src + (long)kr * srow + ch * 8;
        treg[j][0] = *reinterpret_cast<const f4*>(p);
        treg[j][1] = *reinterpret_cast<const f4*>(p + 4);
      }
    }
  };
  auto commit_tile = [&](const int slot) {  // registers -> ring slot (stream tile i -> slot i & 1), split into the two images
    if constexpr (SRC32) {
      unsigned char* base = lds + slot * SLOT32 + (wave * G) * 1024 + lane * 16;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        u4 hi, lo;
        split8(treg[j][0], treg[j][1], hi, lo);
        *reinterpret_cast<u4*>(base + j * 1024) = hi;
        *reinterpret_cast<u4*>(base + TILEB + j * 1024) = lo;
      }
    }
  };
  u4 qf[KS], ql[SRC32 ? KS : 1];  // Q^T operands of this lane's query row (SRC32: the hi / lo pair)
  if constexpr (SRC32) {  // Q: global -> registers in the operand layout (row q0 + c, elements 32 ks + 8 g ..), no LDS round trip
    const int qr = min(qrow, P.Sq - 1);  // rows past Sq: finite data, never stored
    const float* qp = reinterpret_cast<const float*>(P.q) + bh_offset(b, P.qs_b, h, P.qs_h) + (long)qr * P.qs_s + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      split8(__builtin_nontemporal_load(reinterpret_cast<const f4*>(qp + 32 * ks)), __builtin_nontemporal_load(reinterpret_cast<const f4*>(qp + 32 * ks + 4)), qf[ks], ql[ks]);
    load_tile(0);
  } else {
    const unsigned short* qbase = reinterpret_cast<const unsigned short*>(P.q) + bh_offset(b, P.qs_b, h, P.qs_h);
    const unsigned qslot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((2 * R - 1) * TILEB + wave * G * 1024));
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int row = piece_row(j);
      int qr = qt * 64 + row;
      qr = qr < P.Sq ? qr : P.Sq - 1;  // rows past Sq: finite data, never stored
      // (scalar base = the q tile's first row, lane offset = the row inside the tile: 32 bits whatever Sq)
      glds16_s(qbase + (long)qt * 64 * P.qs_s, 2u * (unsigned)((qr - qt * 64) * (int)P.qs_s + (pch ^ swz_k<D>(row)) * 8), qslot + j * 1024);
    }
  }
  // GATE: this workgroup's 64 rows of the layer input, head h's slice, as a K-shaped tile in K ring slot R-1 (first used by
  // K tile R-1, issued after the gate has been formed); and the lane's share of the predictor weights (hidden unit c,
  // inputs 8g.. of each 32-wide k-step; first-layer bias and second-layer weights of units 4g..4g+3), requested in the
  // same round as everything else
  constexpr int GT = GATE ? 4 : 1;       // 16-unit MFMA tiles of predictor hidden units (<= 64 units: attn_gate_mlp2 has head_dim of them)
  u4 gwf[GT][KS];                         // GATE: the lane's share of the first-layer weights, rounded to the storage dtype
  u4 gwl[(GATE && SRC32) ? GT : 1][KS];   // ... fp32 storage: the weights' and the input rows' low halves (operand pairs: the logits are fp32-accurate)
  u4 xf32[(GATE && SRC32) ? KS : 1], xl32[(GATE && SRC32) ? KS : 1];
  f4 gb1v[GT], gw2v[GT];
  int g_mt = 1;
  if constexpr (GATE && SRC32) {  // the lane's query row of the layer input, head h's slice: global -> registers like Q
    const int xr = min(qrow, P.Sq - 1);
    const float* xp = reinterpret_cast<const float*>(P.gh) + (long)b * P.ghs_b + (long)xr * P.ghs_t + (long)h * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      split8(*reinterpret_cast<const f4*>(xp + 32 * ks), *reinterpret_cast<const f4*>(xp + 32 * ks + 4), xf32[ks], xl32[ks]);
  }
  if constexpr (GATE && !SRC32) {
    const unsigned short* xbase = reinterpret_cast<const unsigned short*>(P.gh) + (long)b * P.ghs_b + (long)h * D;
    const unsigned xslot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((R - 1) * TILEB + wave * G * 1024));
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int row = piece_row(j);
      int xr = qt * 64 + row;
      xr = xr < P.Sq ? xr : P.Sq - 1;
      glds16_s(xbase + (long)qt * 64 * P.ghs_t, 2u * (unsigned)((xr - qt * 64) * (int)P.ghs_t + (pch ^ swz_k<D>(row)) * 8), xslot + j * 1024);
    }
  }
  // stream prologue: two tiles in flight
  if constexpr (!SRC32) {
    issue_next();
    if (1 < T) issue_next();
  }
  if constexpr (GATE) {
    const int mm = P.g_units > 0 ? P.g_units : 1;  // <= 64 (host)
    g_mt = (mm + 15) >> 4;
#pragma unroll
    for (int tau = 0; tau < GT; ++tau) {
      gb1v[tau] = gw2v[tau] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        gwf[tau][ks] = u4{0u, 0u, 0u, 0u};
        if constexpr (SRC32) gwl[tau][ks] = u4{0u, 0u, 0u, 0u};
      }
      if (tau < g_mt) {  // hidden unit 16 tau + c, inputs 8g.. of each 32-wide k-step; first-layer bias and second-layer weights of units 16 tau + 4g..4g+3
        const int u = 16 * tau + c;
        const bool uv = u < mm;
        const float* wr = P.gw1 + ((long)h * mm + (uv ? u : 0)) * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          f4 w0 = *reinterpret_cast<const f4*>(wr + 32 * ks), w1 = *reinterpret_cast<const f4*>(wr + 32 * ks + 4);
          if (!uv) w0 = w1 = f4{0.f, 0.f, 0.f, 0.f};
          if constexpr (SRC32) split8(w0, w1, gwf[tau][ks], gwl[tau][ks]);
          else if constexpr (IN == IN_BF16) gwf[tau][ks] = u4{pack2_bf16(w0[0], w0[1]), pack2_bf16(w0[2], w0[3]), pack2_bf16(w1[0], w1[1]), pack2_bf16(w1[2], w1[3])};
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
  const bool has_pad = P.pad != nullptr;
  if (has_pad) {  // (compiler-visible loads: its wait for them also covers the transfers above, which the next wait needs anyway)
    for (int i = tid; i < NT * 16; i += 256) {
      float f = (i < P.Sk) ? load_mask(P.pad, P.pad_f16, (long)b * P.pad_sb + i) : 0.0f;
      if constexpr (GRIDPAD) {  // the grid chain (key_pad_boolean): per key +big (visible), the sentinel (padded) or -inf (past Sk: not even
        // a masked key - a row without a visible key is uniform over the Sk keys under the vanilla softmax, as in the reference)
        f = (i < P.Sk) ? (f <= -1.0e4f ? -1.0e30f : 3.0e38f) : -__builtin_inff();
      }
      lds_pad[i] = f;
    }
  }
  if constexpr (!SRC32) wait_tiles_in_flight<G>(min(2, T));  // Q landed; the one or two tiles behind it may still be in flight
  barrier_mem();
  OEH_STAMP(1);
  const unsigned char* kaddr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kaddr[ks] = lds + c * ROWB + (((ks * 4 + g) ^ swz_k<D>(c)) << 4);  // swz_k(16*sub + c) == swz_k(c)
  if constexpr (!SRC32) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + (2 * R - 1) * TILEB + wave * 16 * ROWB);
  }
  float gate_row = 1.0f;  // GATE: sigmoid(logit) * scaling of this lane's query row
  if constexpr (GATE) {
    u4 xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (SRC32) xf[ks] = xf32[ks];
      else xf[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + (R - 1) * TILEB + wave * 16 * ROWB);
    }
    float a = 0.0f;
#pragma unroll
    for (int tau = 0; tau < GT; ++tau) {
      if (tau < g_mt) {
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma16<IN>(gwf[tau][ks], xf[ks], acc);  // rows = hidden units 16 tau + 4g + r, column = token c
        if constexpr (SRC32) {  // + (W_hi X_lo + W_lo X_hi) 2^-11: the first layer to fp32 accuracy, as the reference's fp32 Linear
          f4 accx = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            accx = mfma16<IN>(gwf[tau][ks], xl32[ks], accx);
            accx = mfma16<IN>(gwl[tau][ks], xf[ks], accx);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(accx[r], kSplitDown, acc[r]);
        }
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
    gate_row = a * P.g_scaling;
    if (P.g_out != nullptr && g == 0 && qvalid) P.g_out[((long)b * P.H + h) * P.Sq + qrow] = a;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // Instruction stream: the 64-key LDS tile is also the unit of control flow.  Every wave of the workgroup runs all
  // four 16-key sub-tiles of every tile it has waited for as straight-line code with compile-time LDS offsets and
  // register indices - one uniform branch per tile and pass instead of one per sub-tile (the per-sub-tile guards of
  // the first version cost ~100 scalar/branch/select instructions per 16-key tile: the kernel was ISSUE-bound).
  // Surplus sub-tiles (above the diagonal in the last tile, or past Sk) are computed and masked; masking only runs
  // on tiles at or after the first masked key.
  // =========================== phase 1: S^T = K Q^T ===========================
  f4 s[NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      const int i = kt;
      if constexpr (SRC32) {
        commit_tile(kt & 1);                       // tile i (loaded an iteration ago) -> its ring slot; last readers: tile i-2, behind the previous barrier
        barrier_mem();
        if (i + 1 < T) load_tile(i + 1);           // lands while tile i is computed (after the last K tile: V tile 0, awaited after the softmax phase)
      } else {
      wait_tiles_in_flight<G>(min(1, T - 1 - i));  // tile i landed; tile i+1 may still be in flight
      barrier_mem();
      OEH_STAMP(2 + kt);
      if (i + 2 < T) issue_next();                 // stream tile i+2, into the slot every wave finished reading one iteration ago
      }
      const int koff = SRC32 ? (kt & 1) * SLOT32 : (kt % R) * TILEB;  // compile-time after unrolling
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SRC32) {
          f4 accx = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const u4 kh = *reinterpret_cast<const u4*>(kaddr[ks] + koff + sub * 16 * ROWB);
            const u4 kl = *reinterpret_cast<const u4*>(kaddr[ks] + koff + TILEB + sub * 16 * ROWB);
            acc = mfma16<IN>(kh, qf[ks], acc);
            accx = mfma16<IN>(kh, ql[ks], accx);
            accx = mfma16<IN>(kl, qf[ks], accx);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(accx[r], kSplitDown, acc[r]);
        } else {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc = mfma16<IN>(*reinterpret_cast<const u4*>(kaddr[ks] + koff + sub * 16 * ROWB), qf[ks], acc);
        }
        s[kt * 4 + sub] = acc;
      }
    }
  }

  OEH_STAMP(10);
  // =========================== phase 2: row statistics and exponentials ===========================
  const float sc = P.scale;
  const int causal = P.causal;
  const int Sk = P.Sk;
  const int klim = causal ? min(qrow + off, Sk - 1) : Sk - 1;            // last admissible key of this lane's row
  const int gm0 = ((causal ? min(q0 + off, Sk - 1) : Sk - 1) + 1) >> 6;  // first 64-key tile holding a masked key (wave-uniform)
  float m = -__builtin_inff();
  float inv_fq = 1.0f;
  if constexpr (FQ) {
    const float mask_min = P.mask_min;
    const int klimc = qrow + off;                                          // last key a causal row may see
    const int kt_causal = causal ? (max(0, q0 + off + 1) >> 6) : KT;       // first 64-key tile with a key the wave's first row must not see
    const int kt_tail = Sk >> 6;                                           // first 64-key tile with a key >= Sk
    if constexpr (GRID) {
      // ---- the chain on the quantiser GRID (no clipping; key padding only as a vector of 0 / <= -1e4 entries - include/oeh.h
      // key_pad_boolean: a flag per key in LDS, rel = min(rel, flag), one instruction per element - OPT's and BERT's configuration).  Once a score is on its
      // grid only the integer rel = idx - zp matters: the row maximum is scale * max(rel), x - m is scale * (rel - rel_max)
      // - formed here without the reference's two roundings of scale * rel - and exp(x - m) = exp2((rel - rel_max) * c2)
      // needs no range reduction: the argument's rounding error is 2^-24 |t|, i.e. below one ulp of the result wherever the
      // probability is not negligible.  The score that reaches the quantiser is a matrix-core dot product (its summation
      // order differs from the reference's bmm by more than an ulp anyway), so the quotient is one multiply by RN(sc / scale)
      // and the probability's index comes from e * RN(1 / (den scale_p)): ~2 ulp in front of rint() instead of 1.5, twelve
      // vector operations per score element instead of twenty-five.  A key the row must not see carries RELMASK: its
      // exponential is exactly 0, as in the reference (a causal row always sees key 0, so the row maximum is a real score).
      // Instruction count is the cost of this phase (the launch is VALU-issue bound): the index comes from ONE fused multiply-add
      // against M = 1.5 * 2^23 - RN(s k1 + M) = M + rint(s k1), ties to even on the exact product - clamped in that domain
      // (a product too large for the trick lies beyond the clamp on the same side); rel is carried as M + rel, whose
      // differences are the exact integer differences the exponent needs; row maximum by v_max3 chains; the causal / tail
      // test of a diagonal tile is one compare + select per element against a per-lane limit, and only there.
      constexpr float RELMASK = -1.0e30f, MAGIC = kGridMagic;  // (oeh_common.h: grid_rel_m)
      const float k1 = sc * P.fq_s.rscale, slo = MAGIC + P.fq_s.lo, shi = MAGIC + P.fq_s.hi;
      const int klime = causal ? min(klimc, Sk - 1) : Sk - 1;                 // last key of this lane's row
      const int klim_g = klime - 4 * g;                                       // ... relative to the lane's first key of a tile
      float mr = RELMASK;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt < n_kt) {
          const bool open_tile = !GRIDPAD && kt < kt_causal && kt < kt_tail;   // no mask touches this 64-key tile (wave-uniform)
          const int lim = klim_g - 64 * kt;                                    // element (sub, r) is masked when 16 sub + r > lim
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {  // (as ONE body: an open and a masked copy of it cost 37 spilled registers at NT = 32)
            const int t = kt * 4 + sub;
            f4 rel;
#pragma unroll
            for (int r = 0; r < 4; ++r) rel[r] = grid_rel_m(s[t][r], k1, slo, shi);
            if (!open_tile && (!GRIDPAD || kt >= kt_causal || kt >= kt_tail)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) rel[r] = (16 * sub + r > lim) ? RELMASK : rel[r];
            }
            if constexpr (GRIDPAD) {
              const f4 flag = *reinterpret_cast<const f4*>(&lds_pad[16 * t + 4 * g]);
#pragma unroll
              for (int r = 0; r < 4; ++r) rel[r] = __builtin_fminf(rel[r], flag[r]);
            }
            s[t] = rel;
          }
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {  // (v_max3 from the plain builtins: the operands are v_med3 / select results, known canonical)
            mr = __builtin_fmaxf(__builtin_fmaxf(mr, s[kt * 4 + sub][0]), s[kt * 4 + sub][1]);
            mr = __builtin_fmaxf(__builtin_fmaxf(mr, s[kt * 4 + sub][2]), s[kt * 4 + sub][3]);
          }
        }
      }
      mr = row4_max(mr);
      m = (mr - MAGIC) * P.fq_s.scale;                                        // the reference's row maximum, fl(scale * rel_max)
      const float c2 = P.fq_s.c2;
      f4 sum4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt < n_kt) {
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            const int t = kt * 4 + sub;
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
      }
      float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
      sum = row4_sum(sum);
      float den = sum;
      if (P.base != 0) den = sum + exp_acc(m * -1.0f);                       // softmax_1: + 1*exp(-max)  (softmax_1.py:18-20)
      inv_fq = 1.0f / den;
      const float cinv = inv_fq * P.fq_p.rscale, plo = P.fq_p.lo, phi = P.fq_p.hi;
      const float clip_iw = inv_fq * P.clip_w, clip_g = P.clip_g;  // (CLIP)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt < n_kt) {
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            const int t = kt * 4 + sub;
            float pv[4];  // integer valued (idx - zp): exact in f16/bf16; the scale is applied after the product
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (CLIP) {  // clip(p (eta - gamma) + gamma, 0, 1) as one clamped fma (a masked key: e = 0, gamma <= 0 -> 0), then the index
                const float pc = __builtin_amdgcn_fmed3f(__builtin_fmaf(s[t][r], clip_iw, clip_g), 0.0f, 1.0f);
                pv[r] = __builtin_amdgcn_fmed3f(__builtin_rintf(pc * P.fq_p.rscale), plo, phi);
              } else {
                pv[r] = __builtin_amdgcn_fmed3f(__builtin_rintf(s[t][r] * cinv), plo, phi);
              }
            }
            const unsigned lo = (IN == IN_BF16) ? pack2_bf16(pv[0], pv[1]) : pack2_f16(pv[0], pv[1]);
            const unsigned hi = (IN == IN_BF16) ? pack2_bf16(pv[2], pv[3]) : pack2_f16(pv[2], pv[3]);
            s[t][0] = bits_f32(lo);
            s[t][1] = bits_f32(hi);
          }
        }
      }
    } else {
      // ---- the reference's op order literally (key padding adds arbitrary values to the dequantised scores; the clip needs
      // the probability itself)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt < n_kt && !has_pad && kt < kt_causal && kt < kt_tail) {
        // no mask touches this 64-key tile: the chain without its four per-sub-tile uniform tests (one test per tile)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
          const f4 x = fq_rel4(s[t] * sc, P.fq_s) * P.fq_s.scale;
          s[t] = x;
          m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(x[0], x[1])), __builtin_fmaxf(x[2], x[3]));
        }
      } else if (kt < n_kt) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
          const int key0 = 16 * t + 4 * g;
          f4 x = s[t];
          x = x * sc;
          {
            const f4 rel = fq_rel4(x, P.fq_s);
            x = rel * P.fq_s.scale;
          }
          if (has_pad) {
            const f4 padv = *reinterpret_cast<const f4*>(&lds_pad[key0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = x[r] + padv[r];
          }
          if (kt >= kt_causal) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (key0 + r > klimc) x[r] = x[r] + mask_min;
          }
          if (P.clamp_min && (has_pad || kt >= kt_causal)) {  // elsewhere x is a finite product of 16-bit data: the clamp is dead
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = __builtin_fmaxf(x[r], mask_min);
          }
          if (kt >= kt_tail) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (key0 + r >= Sk) x[r] = -__builtin_inff();
          }
          s[t] = x;
          m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(x[0], x[1])), __builtin_fmaxf(x[2], x[3]));
        }
      }
    }
    m = row4_max(m);
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt < n_kt) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
          const f2 m2 = f2{m, m};
          const f2 e01 = exp_acc_nonpos2(f2{s[t][0], s[t][1]} - m2), e23 = exp_acc_nonpos2(f2{s[t][2], s[t][3]} - m2);
          s[t] = f4{e01[0], e01[1], e23[0], e23[1]};
          sum += e01[0];
          sum += e01[1];
          sum += e23[0];
          sum += e23[1];
        }
      }
    }
    sum = row4_sum(sum);
    float den = sum;
    if (P.base != 0) den = sum + exp_acc(m * -1.0f);  // softmax_1: + 1*exp(-max)  (softmax_1.py:18-20)
    inv_fq = 1.0f / den;
    // probabilities -> [clip] -> [fq] -> packed 16-bit P^T operand in the first two registers of the score tile
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt < n_kt) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
          const int key0 = 16 * t + 4 * g;
          f4 pv;
          pv = s[t] * inv_fq;
          if constexpr (CLIP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = pv[r] * P.clip_w;
              p = p + P.clip_g;
              pv[r] = __builtin_fminf(__builtin_fmaxf(p, 0.0f), 1.0f);
            }
          }
          pv = fq_rel4(pv, P.fq_p);  // integer valued (idx - zp): exact in f16/bf16; the scale is applied after the product
          if (kt >= kt_tail) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (key0 + r >= Sk) pv[r] = 0.0f;
          }
          const unsigned lo = (IN == IN_BF16) ? pack2_bf16(pv[0], pv[1]) : pack2_f16(pv[0], pv[1]);
          const unsigned hi = (IN == IN_BF16) ? pack2_bf16(pv[2], pv[3]) : pack2_f16(pv[2], pv[3]);
          s[t][0] = bits_f32(lo);
          s[t][1] = bits_f32(hi);
        }
      }
    }
  
    }
  }
  float inv = inv_fq;
  if constexpr (!FQ) {
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      if (has_pad) {  // scores become scale*s + pad (BERT order); otherwise raw dot products are kept, scale folded below
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
          const f4 padv = *reinterpret_cast<const f4*>(&lds_pad[16 * t + 4 * g]);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][r] = __builtin_fmaf(s[t][r], sc, padv[r]);
        }
      }
      if (kt >= gm0) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * g + r > klim) s[t][r] = NEG;
        }
      }
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int t = kt * 4 + sub;
        m = __builtin_fmaxf(__builtin_fmaxf(m, s[t][0]), s[t][1]);
        m = __builtin_fmaxf(__builtin_fmaxf(m, s[t][2]), s[t][3]);
      }
    }
  }
  m = row4_max(m);

  OEH_STAMP(11);
  // in pad mode the registers hold scaled+masked scores, otherwise raw dot products
  const float c1 = has_pad ? kLog2e : sc * kLog2e;
  // A row whose every key is masked (only possible with a padding mask) has exp(x - m) == 1 for every key in
  // the reference; zeroing the lane's exponent constants gives exp2(fma(x, 0, 0)) = 1 without any select.
  const bool dead = m < -1.0e30f;
  const float c1l = dead ? 0.0f : c1;
  const float mcl = dead ? 0.0f : m * c1;
  const float m_true = has_pad ? m : m * sc;
  float sum = 0.0f;
  // the packed 16-bit P^T operand overwrites the first two registers of its own score tile (no second array)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int t = kt * 4 + sub;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c1l, -mcl));
      }
      if (dead && kt >= gm0) {  // dead rows gave exp == 1 to keys past Sk as well: remove them
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * g + r >= Sk) s[t][r] = 0.0f;
        }
      }
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int t = kt * 4 + sub;
        sum += (s[t][0] + s[t][1]) + (s[t][2] + s[t][3]);
        if constexpr (!CLIP) {
          const unsigned lo = (IN == IN_BF16) ? pack2_bf16(s[t][0], s[t][1]) : pack2_f16(s[t][0], s[t][1]);
          const unsigned hi = (IN == IN_BF16) ? pack2_bf16(s[t][2], s[t][3]) : pack2_f16(s[t][2], s[t][3]);
          s[t][0] = bits_f32(lo);
          s[t][1] = bits_f32(hi);
        }
      }
    }
  }
  sum = row4_sum(sum);
  float den = sum;
  if (P.base != 0) den = sum + exp_acc(m_true * -1.0f);
  inv = 1.0f / den;

  if constexpr (CLIP) {  // clip(p*(eta-gamma)+gamma, 0, 1); masked keys have p == 0 and stay 0 (this path requires gamma <= 0)
    // one fused multiply-add per element, clamped by the instruction's own clamp bit: e * (w / den) + gamma instead of the
    // reference's ((e / den) * w) + gamma with its three roundings - 1e-7 relative, on a path checked to 1e-3 (the clipped
    // INT8 chain, where the rounding decides an index, keeps the literal order: FQ == 2)
    const float clip_iw = inv * P.clip_w, clip_g = P.clip_g;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt < n_kt) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int t = kt * 4 + sub;
          float pv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_fmed3f(__builtin_fmaf(s[t][r], clip_iw, clip_g), 0.0f, 1.0f);
          const unsigned lo = (IN == IN_BF16) ? pack2_bf16(pv[0], pv[1]) : pack2_f16(pv[0], pv[1]);
          const unsigned hi = (IN == IN_BF16) ? pack2_bf16(pv[2], pv[3]) : pack2_f16(pv[2], pv[3]);
          s[t][0] = bits_f32(lo);
          s[t][1] = bits_f32(hi);
        }
      }
    }
  }

  }  // !FQ
  OEH_STAMP(12);
  // =========================== phase 3: O^T = V^T P^T ===========================
  f4 o[DT], ox[SRC32 ? DT : 1];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f4{0.f, 0.f, 0.f, 0.f};
  if constexpr (SRC32) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ox[dt] = f4{0.f, 0.f, 0.f, 0.f};
  }
  const int vrow = 4 * g + (c >> 2);
  const unsigned char* vaddr[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) vaddr[dt] = lds + vrow * ROWB + ((dt ^ swz_v<D>(vrow)) << 5) + ((c & 3) << 3);  // swz_v(32u+vrow) == swz_v(vrow)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt < n_kt) {
      const int i = n_kt + kt;
      if constexpr (SRC32) {
        commit_tile(i & 1);
        barrier_mem();
        if (i + 1 < T) load_tile(i + 1);
      } else {
      wait_tiles_in_flight<G>(min(1, T - 1 - i));
      barrier_mem();
      OEH_STAMP(13 + kt);
      if (i + 2 < T) issue_next();
      }
      const int slot_off = SRC32 ? (i & 1) * SLOT32 : R * TILEB + (kt % R) * TILEB;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t0 = kt * 4 + 2 * u;
        const u4 pb = u4{f32_bits(s[t0][0]), f32_bits(s[t0][1]), f32_bits(s[t0 + 1][0]), f32_bits(s[t0 + 1][1])};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const unsigned char* a0 = vaddr[dt] + slot_off + u * 32 * ROWB;
          const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
          const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 16 * ROWB));
          const u2 l2 = __builtin_bit_cast(u2, lo), h2 = __builtin_bit_cast(u2, hi);
          o[dt] = mfma16<IN>(u4{l2.x, l2.y, h2.x, h2.y}, pb, o[dt]);
          if constexpr (SRC32) {  // the lo image of V
            const s4 lol = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + TILEB));
            const s4 hil = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + TILEB + 16 * ROWB));
            const u2 l3 = __builtin_bit_cast(u2, lol), h3 = __builtin_bit_cast(u2, hil);
            ox[dt] = mfma16<IN>(u4{l3.x, l3.y, h3.x, h3.y}, pb, ox[dt]);
          }
        }
      }
    }
  }
  if constexpr (SRC32) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] = __builtin_fmaf(ox[dt][r], kSplitDown, o[dt][r]);
  }

  OEH_STAMP(21);
  // =========================== epilogue ===========================
  // O^T is scaled in registers, staged through the (idle) K ring - every wave has left the K phase once any wave is
  // past the V-phase barriers - and stored as whole rows, 16 B per lane, write-through (oeh_common.h: store_wt16).
  // Each wave owns the 16 rows of its query block in K slot 0: no workgroup barrier.
  // (addresses are derived from an opaque copy of the lane id so that they are formed HERE, not hoisted above the loops
  // where the NT=32 variant has no register to spare)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int ce = lane_e & 15, ge = lane_e >> 4;
  float rowscale = (CLIP || FQ) ? 1.0f : inv;
  if (P.gate != nullptr && qvalid) rowscale = rowscale * P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qrow * P.gs_s];
  if constexpr (GATE) rowscale = rowscale * gate_row;
  constexpr int XM = (CPR < 8 ? CPR : 8) - 1;
  unsigned char* ebase = lds + wave * (16 * ROWB);
  float xs[FQ ? DT * 4 : 1];
  if constexpr (FQ) {  // [scale of the quantised P] [fq] gate [fq]: the general kernel's epilogue chain, in whole passes (oeh_common.h: ctx_chain)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xs[dt * 4 + r] = P.fq_p.scale * o[dt][r];
    ctx_chain<DT * 4>(xs, P.fq_c, P.ctx_before_gate, P.gate != nullptr, rowscale);
  }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    float ov[4];
    if constexpr (FQ) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = xs[dt * 4 + r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = o[dt][r] * rowscale;
    }
    if constexpr (OUT32) {  // fp32 output straight from the accumulators (fp32 storage; O32: 16-bit storage with o_dtype = OEH_F32 -
                            // the kernel's arithmetic before the output rounding, include/oeh.h)
      if (q0 + ce < P.Sq)
        store_wt16(reinterpret_cast<float*>(P.o) + bh_offset(b, P.os_b, h, P.os_h) + (long)(q0 + ce) * P.os_s + 16 * dt + 4 * ge,
                   u4{f32_bits(ov[0]), f32_bits(ov[1]), f32_bits(ov[2]), f32_bits(ov[3])});
    } else {
    u2 w;
    if constexpr (IN == IN_BF16) {
      w.x = pack2_bf16(ov[0], ov[1]);
      w.y = pack2_bf16(ov[2], ov[3]);
    } else {
      w.x = pack2_f16(ov[0], ov[1]);
      w.y = pack2_f16(ov[2], ov[3]);
    }
    *reinterpret_cast<u2*>(ebase + ce * ROWB + ((((2 * dt + (ge >> 1)) ^ (ce & XM)) << 4) | ((ge & 1) << 3))) = w;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before it reads them back
  if constexpr (!OUT32) {
    unsigned short* obase = reinterpret_cast<unsigned short*>(P.o) + bh_offset(b, P.os_b, h, P.os_h);
    const int lr = lane_e / CPR, lc = lane_e % CPR;
    static_assert(16 % RPP == 0 || RPP % 16 == 0, "store passes tile the 16-row block");
#pragma unroll
    for (int pass = 0; pass < (16 + RPP - 1) / RPP; ++pass) {
      const int row = pass * RPP + lr;
      if (row < 16) {
        const u4 w = *reinterpret_cast<const u4*>(ebase + row * ROWB + ((lc ^ (row & XM)) << 4));
        const int grow = q0 + row;
        if (grow < P.Sq) store_wt16(obase + (long)grow * P.os_s + lc * 8, w);
      }
    }
  }
  OEH_STAMP(22);
#ifdef OEH_TIMELINE
  if (stamp != nullptr && lane == 0) stamp[31] = __builtin_amdgcn_s_memrealtime();
#endif
}
#undef OEH_STAMP

template <int NT, int D, int IN>
static void launch_fast_nt_d_in(const AttnParams& P, unsigned grid, hipStream_t st) {
  const bool gate = P.gh != nullptr;
  const bool fqon = P.fq_s.en && P.fq_p.en;
  const bool grid_chain = fqon && (P.pad == nullptr || P.pad_bool);  // FQ == 1, or 3 with a key-padding vector (include/oeh.h: key_pad_boolean)
  const bool grid_pad = grid_chain && P.pad != nullptr;
  if (P.src32) {  // fp32 storage read directly, fp32 output
    if constexpr (IN == IN_F16) {
      if (grid_pad && P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 3, true>), dim3(grid), dim3(256), 0, st, P);
      else if (grid_pad) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 3, true>), dim3(grid), dim3(256), 0, st, P);
      else if (grid_chain && P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 1, true>), dim3(grid), dim3(256), 0, st, P);
      else if (grid_chain) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 1, true>), dim3(grid), dim3(256), 0, st, P);
      else if (fqon && P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 2, true>), dim3(grid), dim3(256), 0, st, P);
      else if (fqon) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 2, true>), dim3(grid), dim3(256), 0, st, P);
      else if (P.clip && gate) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, true, 0, true>), dim3(grid), dim3(256), 0, st, P);
      else if (P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 0, true>), dim3(grid), dim3(256), 0, st, P);
      else if (gate) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, true, 0, true>), dim3(grid), dim3(256), 0, st, P);
      else hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 0, true>), dim3(grid), dim3(256), 0, st, P);
    }
    return;
  }
  if (fqon) {
    if (grid_pad && P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 3>), dim3(grid), dim3(256), 0, st, P);
    else if (grid_pad) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 3>), dim3(grid), dim3(256), 0, st, P);
    else if (grid_chain && P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 1>), dim3(grid), dim3(256), 0, st, P);
    else if (grid_chain) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 1>), dim3(grid), dim3(256), 0, st, P);
    else if (P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 2>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 2>), dim3(grid), dim3(256), 0, st, P);
    return;
  }
  if constexpr (D == 64) {  // (oeh_api.hip: the out32 rule - head dim 64: plain, clipped, plain + the in-kernel gate predictor)
    if (P.out32) {
      if (P.clip) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false, 0, false, true>), dim3(grid), dim3(256), 0, st, P);
      else if (gate) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, true, 0, false, true>), dim3(grid), dim3(256), 0, st, P);
      else hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 0, false, true>), dim3(grid), dim3(256), 0, st, P);
      return;
    }
  }
  if constexpr (D == 128) {  // (head dim 128: the plain form)
    if (P.out32) {
      hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false, 0, false, true>), dim3(grid), dim3(256), 0, st, P);
      return;
    }
  }
  if (P.clip) {
    if (gate) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, true>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, true, false>), dim3(grid), dim3(256), 0, st, P);
  } else {
    if (gate) hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, true>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((oeh_attn_fast_kernel<NT, D, IN, false, false>), dim3(grid), dim3(256), 0, st, P);
  }
}

template <int NT, int D>
static int launch_fast_nt_d(const AttnParams& P, int in, hipStream_t st) {
  const unsigned grid = (unsigned)(P.nQT * P.nBHpad);
  if (in == IN_BF16) launch_fast_nt_d_in<NT, D, IN_BF16>(P, grid, st);
  else launch_fast_nt_d_in<NT, D, IN_F16>(P, grid, st);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

template <int D>
static int launch_fast_d(const AttnParams& P, int in, hipStream_t st) {
  if (P.Sk <= 128) return launch_fast_nt_d<8, D>(P, in, st);
  if (P.Sk <= 256) return launch_fast_nt_d<16, D>(P, in, st);
  return launch_fast_nt_d<32, D>(P, in, st);
}

}  // namespace oeh
