// A QuantLinear projection - or the three of an attention layer side by side - as ONE matrix-core GEMM with the output quantisers in its
// epilogue (include/oeh.h: oeh_proj_quant_i8).  Activations (M, K) fp16, or an fp32 model's activations
// as fp16 operand pairs [hi | lo 2^11] (oeh_split_pairs' output, or the fp32 matrix itself: its rows go to LDS as they are and a wave
// forms (hi, lo) from the 8 values of a fragment when it reads them - split8, the split pass's own arithmetic); the QuantLinear weights as their 8-bit integers in fp16 (N rows of K,
// exact); fp32 accumulation:
//     acc[m][n] = sum_k hi[m][k] W[n][k] + sum_k lo[m][k] (W[n][k] 2^-11)
// and, in the epilogue, per column segment (q | k | v): value = alpha acc + bias[n], the centred 8-bit index of the segment's quantiser in
// the layout the INT8-storage attention core reads (q, k: (B, S, E) int8; v: (B, H, 64, S) int8) and / or the dequantised values (the
// decoder's (k, v) cache; out_proj's output) - what the library GEMM + one `oeh_quantize_heads_i8` pass per segment did, without the
// (M, N) fp32 accumulator ever reaching memory.
//
// Shape of the kernel (gfx950).  A workgroup of 4 waves (2 x 2) owns an output tile of 32 MI x 32 NJ; a wave MI x NJ accumulator tiles
// of v_mfma_f32_16x16x32_f16.  Two instantiations: MI x NJ = 4 x 9 (128 x 288, 144 accumulator registers, 232 VGPRs, two workgroups
// per CU: at M = 8192, N = 2304 every workgroup of the launch is resident at once, 64 x 8 = 512) and 2 x 6 (64 x 192, four per CU: narrow
// outputs such as out_proj's N = 768 and problems of fewer than 512 large tiles).  K in steps of 32: one LDS slot holds hi (BM rows x
// 64 B), lo (the same) and W (BN rows x 64 B) - 34 KB for the large tile - filled by LDS-DMA (global_load_lds_dwordx4, pieces of 1 KB =
// 16 rows, spread over the 4 waves, scalar base + constant lane offsets); two slots, one counted-to-zero wait + one barrier per step; the
// next step's pieces are issued two at a time behind the first MFMA groups of the current one.  Per step and wave (large tile): 17
// ds_read_b128, all requested before the first MFMA (4 hi + 4 lo + 9 W fragments), 36 v_pk_mul_f16 (W 2^-11: exact, the integers are
// >= 1 in magnitude) and 72 MFMAs.  LDS image: rows of 64 B, 16-byte chunk c of row r stored at chunk c ^ (-(r >> 2) & 3): ds_read_b128
// is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md), i.e. rows 0-3 and 12-15 of chunk c
// together with rows 4-11 of chunk c + 1, and with this swizzle the 16 lanes of every group hit 16 different 16-byte bank groups (the
// first version swizzled by (r >> 2) & 3, right for groups of 16 consecutive lanes: SQ_LDS_BANK_CONFLICT was half of the LDS cycles);
// the DMA writes whole kilobytes and applies the swizzle to its SOURCE address.  Workgroup -> tile: the eight XCDs take contiguous ranges of row tiles, all column tiles of a row tile on
// one XCD (an activation tile is fetched once per XCD and hit in its L2 by the other column tiles).
// Epilogue: 7 vector instructions per output (fma, the three-instruction exact quotient, rint, + zero point, v_cvt_pk_u8_f32 whose
// saturation is the clamp; 9 with values), index bytes through LDS images (4 x 4 byte transposes inside lane quads for the row-major
// one) so that they leave as whole 16-byte pieces of contiguous output; values straight from the accumulator layout, write-through.
// Measured (one MI355X, M = 8192, K = 768): q/k/v (N = 2304) from operand pairs 74 us with the (k, v) values, 63 us without, against 107 /
// 91 us for the library GEMM (hipBLASLt, 70 us) + three quantiser passes; from the fp32 activations 76 / 70 us against 116 / 100 with the
// 8.9-us split pass in front; out_proj (N = 768, integers) 20.5 against 31.8 us.  The loop alone
// is 52-55 us = 1.1 PFLOP/s (40 us with the DMA and the barriers knocked out: the matrix core at the clock it holds under this load).
#include "../../include/oeh.h"
#include "oeh_common.h"
#include "oeh_gemm.h"

#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace oeh {

// The outputs leave WRITE-THROUGH (as the attention kernels' do, oeh_common.h: store_wt16): a plain store leaves the line dirty in the XCD's L2
// and what is still there at the end of the kernel goes to memory after the last wave, where nothing overlaps it (out_proj, 25 MB of
// values: 24.4 -> 20.7 us; the q/k/v launch with its 69 MB is bound by the write itself either way).
__device__ __forceinline__ void store_wt4_s(const void* sbase, unsigned voff, float v) {  // scalar base + 32-bit lane byte offset
  asm volatile("global_store_dword %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}

constexpr int GBK = kGemmBK, GROWB = 64;
#define GLDS glds16_s   // (oeh_common.h: without the save / restore of M0 unless -DOEH_KEEP_M0)
constexpr int kPer = 2;  // LDS-DMA pieces issued behind each of the first MFMA groups of a step (3 and 5 measured the same or slower)

// activation forms: fp16 values; fp16 operand pairs [hi | lo] (oeh_split_pairs); fp32 values, split into (hi, lo) when a wave reads its
// fragments (the split pass folded into the kernel: same arithmetic, oeh_common.h: split8); int8 values against int8 weights on
// v_mfma_i32_16x16x64_i8 (a K step is 64 elements = the same 64-byte rows; exact int32 sums, + a per-column integer in the epilogue)
enum { A_F16 = 0, A_PAIRS = 1, A_F32 = 2, A_I8 = 3 };

// tile geometry for MI x NJ accumulator tiles (16 x 16) per wave, waves 2 x 2
template <int MI, int NJ>
struct Geo {
  static constexpr int BM = 32 * MI, BN = 32 * NJ;
  static constexpr int AHI = 0, ALO = BM * GROWB, W = 2 * BM * GROWB, SLOT = W + BN * GROWB;
  static constexpr int PITCH_C = BM + 16, IMG_C = BM * BN, IMGS = IMG_C + BN * PITCH_C;   // the epilogue's byte images
  static constexpr int LDS = 2 * SLOT > IMGS ? 2 * SLOT : IMGS;
  static constexpr int LDS3 = 3 * SLOT > IMGS ? 3 * SLOT : IMGS;   // the one-workgroup-per-CU loop's ring of three (LOOP == 1)
};

// Diagnostic knock-outs (OEH_GEMM_DBG, tools/exp/proj_time.py) exist only in a build with -DOEH_GEMM_EXPERIMENT (`make experiment`): in the
// production kernel they are the constant 0 and compile out (a leftover environment variable must not make the shipping
// library return wrong numbers).
#ifdef OEH_GEMM_EXPERIMENT
#define GEMM_DBG(P) ((P).dbg)
#else
#define GEMM_DBG(P) 0
#endif

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

// swizzle of the fp32 activation image (rows of 128 B = eight 16-byte chunks): chunk c of row r is stored at c ^ swz32(r), swz32 = g[(r >> 1) & 7]
// with g = (0, 2, 4, 5, 6, 7, 1, 3) - under ds_read_b128's lane groups (rows 0-3 and 12-15 of chunk c together with rows 4-11 of chunk c + 2) the
// sixteen lanes of a group then hit sixteen different 16-byte bank groups, for both reads of a fragment
__device__ __forceinline__ int swz32(int row) { return (int)((0x31765420u >> (4 * ((row >> 1) & 7))) & 7u); }

// LOOP != 0 (round 5; fp32 activations, 128 x 288 tile): the pipelined K loop.  LOOP == 1: launches that give a CU ONE workgroup - 160-256 tiles, BERT-base's
// M = 4096: 256 - where each SIMD has a single wave and nothing hides what that wave waits for: ring of THREE slots (tile t + 2 requested during step t, counted
// wait).  LOOP == 2: the same body on the ring of two with two workgroups per CU (the OPT shape's 512 tiles: 0.97 of the round-4 loop, same box).  Fragment-major groups (A fragment i against the NJ W fragments: 2 NJ MFMAs) with the LDS-DMA requests, the reads of fragment i + 2 and the split
// of fragment i + 1 - one v_fma_mix instruction at a time - placed BETWEEN the MFMAs by construction (inline-asm MFMAs: volatile statements keep their
// order); one wait + barrier per step.  tools/exp/big_gemm (profiles/r05_big_tile_gemm_experiment.txt) is where the structure was measured first.
template <int AM, int MI, int NJ, int LOOP = 0>
__global__ __launch_bounds__(256, (LOOP == 1 ? 1 : MI * NJ > 16 ? 2 : 4)) void oeh_gemm_kernel(const GemmParams P) {   // (LOOP == 2: A/B form - the LOOP == 1 body on a ring of two, two workgroups per CU)
  typedef Geo<MI, NJ> G;
  static_assert(LOOP == 0 || (AM == A_F32 && MI == 4), "the one-workgroup-per-CU loop: fp32 activations, four row blocks per wave");
  constexpr int RING = LOOP == 1 ? 3 : 2;
  constexpr bool PIPE = LOOP != 0;
  constexpr bool PAIRS = AM == A_PAIRS || AM == A_F32;   // two MFMA products per term: (hi, lo) of fp32 activations
  constexpr int EB = AM == A_I8 ? 1 : 2;               // bytes per element of a and w (a K step is 64 bytes of a row either way)
#ifndef OEH_GEMM_KPRE
#define OEH_GEMM_KPRE 32   // (A/B builds: 0 = the plain unscaled residual of x itself, split8_raw)
#endif
#if !defined(OEH_SCALED_LO) && OEH_GEMM_KPRE
  constexpr float kApre = (float)OEH_GEMM_KPRE, kApreInv = AM == A_F32 ? 1.0f / kApre : 1.0f;   // fp32 activations are split as kApre x (below)
#else
  constexpr float kApreInv = 1.0f;
#endif
  constexpr int GBM = G::BM, GBN = G::BN, G_AHI = G::AHI, G_ALO = G::ALO, G_W = G::W, G_SLOT = G::SLOT, G_PITCH_C = G::PITCH_C, G_IMG_C = G::IMG_C;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  // tile of this workgroup: the eight XCDs (block id % 8) take contiguous ranges of row tiles, all column tiles of a row tile on one XCD
  int mi, ni;
  {
    const int id = blockIdx.x;
    if ((P.MT & 7) == 0) {
      const int xcd = id & 7, s = id >> 3;
      mi = xcd * (P.MT >> 3) + s / P.NT;
      ni = s % P.NT;
    } else {
      mi = id / P.NT;
      ni = id % P.NT;
    }
  }
  const int m0 = mi * GBM, n0 = ni * GBN;
  const int T = (GEMM_DBG(P) & 2) ? 2 : P.K * EB / (GBK * 2);

  // ---- LDS-DMA: piece p (1 KB = 16 rows x 64 B); lane -> row p * 16 + (lane >> 2), stored chunk lane & 3 = logical chunk ^ (-(row >> 2) & 3)
  const unsigned lds_base = lds_offset(lds);
  const int prow = lane >> 2;
  const int pchunk = (lane & 3) ^ ((-(lane >> 4)) & 3);
  const unsigned char* ab = reinterpret_cast<const unsigned char*>(P.a);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(P.w);
  constexpr int NPA = GBM / 16, NPW = GBN / 16;                 // 8, 18
  constexpr int NP = (PAIRS ? 2 * NPA : NPA) + NPW;             // 34 | 26
  // piece p = 4 q + wave: q < QA -> hi rows, q < QL -> lo rows, else W rows (the kind of a piece depends on q only: 8 | 8 | 18 pieces).
  // Lane byte offsets are constant over the K loop (rows clamped to the matrix: tails compute on repeated rows and are not stored); the
  // scalar bases advance by 64 B per step.
  constexpr int QA = NPA / 4, QL = PAIRS ? 2 * QA : QA, NQ = (NP + 3) / 4;
  unsigned voff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int p = PIPE ? min(4 * q + wave, NP - 1) : 4 * q + wave;   // (LOOP == 1: the last round repeats piece NP - 1 - every wave issues NQ requests per tile)
    if (q < QL) {
      if constexpr (AM == A_F32) {
        // fp32 rows of 128 B (32 values = one K step): a piece = 8 rows, lane -> row 8 p + (lane >> 3), stored chunk lane & 7 = logical
        // chunk ^ swz32(row); the q < QL rounds carry the 2 NPA pieces of the fp32 image (the footprint of hi + lo)
        const int rl = 8 * p + (lane >> 3);
        const int r = min(m0 + rl, P.M - 1);
        voff[q] = (unsigned)(((long)r * P.lda) * 4 + ((lane & 7) ^ swz32(rl)) * 16);
      } else {
        const int r = min(m0 + (p - (q < QA ? 0 : NPA)) * 16 + prow, P.M - 1);
        voff[q] = (unsigned)(((long)r * P.lda) * EB + pchunk * 16);
      }
    } else {
      const int r = min(n0 + (p - QL * 4) * 16 + prow, P.N - 1);
      voff[q] = (unsigned)(((long)r * P.ldw) * EB + pchunk * 16);
    }
  }
  auto issue_q = [&](int t, int q) {
    const unsigned slot = lds_base + (unsigned)((RING == 2 ? (t & 1) : t % 3) * G_SLOT);
    const long kb = (long)t * (GBK * 2);
    const int p = PIPE ? min(4 * q + wave, NP - 1) : 4 * q + wave;
    if (AM == A_F32 && q < QL) GLDS(ab + 2 * kb, voff[q], slot + G_AHI + p * 1024);
    else if (q < QA) GLDS(ab + kb, voff[q], slot + G_AHI + p * 1024);
    else if (q < QL) GLDS(ab + (long)P.K * 2 + kb, voff[q], slot + G_ALO + (p - NPA) * 1024);
    else if (PIPE || q < NQ - 1 || wave < NP - 4 * (NQ - 1)) GLDS(wb + kb, voff[q], __builtin_amdgcn_readfirstlane(slot + G_W + (p - QL * 4) * 1024));
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) issue_q(t, q);
  };

  // ---- fragment addresses (constant per lane up to the slot)
  const unsigned swz = (unsigned)((lq ^ ((-(l15 >> 2)) & 3)) << 4);
  const unsigned a_off = (unsigned)((16 * MI * wm + l15) * GROWB) + swz;
  const unsigned w_off = (unsigned)(G_W + (16 * NJ * wn + l15) * GROWB) + swz;
  // fp32 image: the lane's 8 values of a fragment are chunks 2 lq and 2 lq + 1 of its row (two ds_read_b128, 16 B apart after the swizzle)
  const unsigned a32_off = (unsigned)((16 * MI * wm + l15) * 128) + (unsigned)(((2 * lq) ^ swz32(l15)) << 4);
  if constexpr (AM == A_F32) fp16_overflow_clamp();  // the split saturates beyond the fp16 range, as oeh_split_pairs does

  f4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  if constexpr (PIPE) {
    constexpr int PPG = (NQ + MI - 1) / MI;   // requests per group, the first groups
    f4 raw[2][2];
    unsigned hi[2][4], lo[2][4];
    h8v fw[NJ];
    const float kpre = kApre;
    auto read_raw = [&](const unsigned char* sl, int i, f4 (&d)[2]) {
      d[0] = *reinterpret_cast<const f4*>(sl + G_AHI + a32_off + i * 16 * 128);
      d[1] = *reinterpret_cast<const f4*>(sl + G_AHI + (a32_off ^ 16u) + i * 16 * 128);
    };
    // one instruction of split8_raw_scaled (oeh_common.h) on the fragment's 8 values: k = 0..7 the hi halves, 8..15 the residuals
    auto split_op = [&](int k, const f4 (&x)[2], unsigned (&h)[4], unsigned (&l)[4]) {
      const int e = k & 7;
      const float xe = x[e >> 2][e & 3];
      unsigned& hr = h[e >> 1];
      unsigned& lr = l[e >> 1];
      if (k < 8) {
        if (!(e & 1)) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hr) : "v"(xe), "s"(kpre));
        else asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hr) : "v"(xe), "s"(kpre));
      } else {
        if (!(e & 1)) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lr) : "v"(xe), "s"(kpre), "v"(hr));
        else asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lr) : "v"(xe), "s"(kpre), "v"(hr));
      }
    };
    issue(0);
    if (RING == 3 && T > 1) issue(1);
    auto step = [&](int t, auto dma_, auto w1_) {
      constexpr bool DMA = decltype(dma_)::value;
      // tile t has landed (the NQ requests of tile t + 1 may be in flight) - for every wave, and every wave has left tile t - 1: its slot takes tile t + 2
      if constexpr (decltype(w1_)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      barrier_mem();
      const unsigned char* sl = lds + (RING == 2 ? (t & 1) : t % 3) * G_SLOT;
      read_raw(sl, 0, raw[0]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fw[j] = *reinterpret_cast<const h8v*>(sl + w_off + j * 16 * GROWB);
      read_raw(sl, 1, raw[1]);
#pragma unroll
      for (int k = 0; k < 16; ++k) split_op(k, raw[0], hi[0], lo[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        // the group's LDS-DMA requests: with two waves per SIMD (LOOP == 2) in front of its MFMAs - the other wave keeps the matrix core busy; with ONE wave
        // per SIMD (LOOP == 1) one at a time behind MFMAs 1, 7 and 13 (same box: 59.9 vs 61.0 us the first way at the OPT shape, 34.9 vs 35.1 the second at BERT-base's)
        if constexpr (DMA && LOOP == 2) {
#pragma unroll
          for (int u = 0; u < PPG; ++u)
            if (PPG * i + u < NQ) issue_q(t + RING - 1, PPG * i + u);
        }
        if (i + 2 < MI) read_raw(sl, i + 2, raw[i & 1]);
        const h8v ahv = __builtin_bit_cast(h8v, u4{hi[i & 1][0], hi[i & 1][1], hi[i & 1][2], hi[i & 1][3]});
        const h8v alv = __builtin_bit_cast(h8v, u4{lo[i & 1][0], lo[i & 1][1], lo[i & 1][2], lo[i & 1][3]});
#pragma unroll
        for (int k = 0; k < 2 * NJ; ++k) {
          const int j = k % NJ;
          const h8v av = k < NJ ? ahv : alv;
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(av), "v"(fw[j]));
          if (i + 1 < MI && k >= 1 && k - 1 < 16) split_op(k - 1, raw[(i + 1) & 1], hi[(i + 1) & 1], lo[(i + 1) & 1]);
          if constexpr (DMA && LOOP == 1) {
            if (k % 6 == 1 && k / 6 < PPG && PPG * i + k / 6 < NQ) issue_q(t + RING - 1, PPG * i + k / 6);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    {
      int t = 0;
      if constexpr (RING == 3) {
        for (; t + 2 < T; ++t) step(t, std::true_type{}, std::true_type{});
        step(t, std::false_type{}, std::true_type{});       // t = T - 2 (T >= 2: the host's rule)
        step(t + 1, std::false_type{}, std::false_type{});
      } else {
        for (; t + 1 < T; ++t) step(t, std::true_type{}, std::false_type{});
        step(t, std::false_type{}, std::false_type{});
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results before the epilogue reads them: inline-asm MFMAs are outside the compiler's hazard model)
  } else {
  issue(0);
  for (int t = 0; t < T; ++t) {
    if (!(GEMM_DBG(P) & 8)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      barrier_mem();
    }
    const bool more = t + 1 < T && !((GEMM_DBG(P) & 4) && t >= 1);
    if ((GEMM_DBG(P) & 64) && more) issue(t + 1);
    const unsigned char* sl = lds + (t & 1) * G_SLOT;
    h8v ah[MI], al[MI];
    f4 x32[AM == A_F32 ? MI : 1][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (AM == A_F32) {
        x32[i][0] = *reinterpret_cast<const f4*>(sl + G_AHI + a32_off + i * 16 * 128);
        x32[i][1] = *reinterpret_cast<const f4*>(sl + G_AHI + (a32_off ^ 16u) + i * 16 * 128);
      } else {
        ah[i] = *reinterpret_cast<const h8v*>(sl + G_AHI + a_off + i * 16 * GROWB);
        if constexpr (PAIRS) al[i] = *reinterpret_cast<const h8v*>(sl + G_ALO + a_off + i * 16 * GROWB);
      }
    }
    // every fragment of the step is requested before the first MFMA (17 ds_read_b128 in flight: the matrix core never waits for
    // LDS behind the first group); the compiler's counted lgkmcnt waits release the groups in order
    h8v bf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const h8v*>(sl + w_off + j * 16 * GROWB);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (AM == A_F32) {
      // fragment-major first sweep: (hi, lo) of fragment i + 1 are formed (split8: oeh_split_pairs' arithmetic, ~25 vector instructions)
      // while the matrix core works through the NJ hi products of fragment i; then the lo products column-major as in the other forms
      constexpr int PI = (NQ + MI - 1) / MI;   // LDS-DMA pieces of the next tile issued behind each fragment's group
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        u4 hi, lo;
#ifndef OEH_SCALED_LO   // (round 5; -DOEH_SCALED_LO: the round-4 form, [hi | lo 2^11] against W and W 2^-11, for A/B builds)
        // 32 x = hi + lo', lo' = RN16(32 x - hi) unscaled (oeh_common.h: why the pre-scale): the second product runs against W itself (below), and the
        // epilogue folds the 2^-5 into the segment's alpha (exact).  22 bits of x for 2^-8 <= |x| <= 2 047 (an absolute 2^-30 below, 11 bits up to 4 094)
#if OEH_GEMM_KPRE
        split8_raw_scaled(x32[i][0], x32[i][1], kApre, hi, lo);
#else
        split8_raw(x32[i][0], x32[i][1], hi, lo);
#endif
#else
        split8_scaled(x32[i][0], x32[i][1], hi, lo);
#endif
        ah[i] = __builtin_bit_cast(h8v, hi);
        al[i] = __builtin_bit_cast(h8v, lo);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(GEMM_DBG(P) & 64) && more) {
#pragma unroll
          for (int u = 0; u < PI; ++u)
            if (PI * i + u < NQ) issue_q(t + 1, PI * i + u);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#ifndef OEH_SCALED_LO
        const h8v bs = bf[j];
#else
        const h8v bs = bf[j] * (_Float16)0.00048828125f;  // 2^-11: exact on the 8-bit integers
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bs, acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if constexpr (AM == A_I8)
            acc[i][j] = __builtin_bit_cast(f4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i4, ah[i]), __builtin_bit_cast(i4, bf[j]),
                                                                                      __builtin_bit_cast(i4, acc[i][j]), 0, 0, 0));
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (PAIRS) {
          const h8v bs = bf[j] * (_Float16)0.00048828125f;  // 2^-11: exact on the 8-bit integers
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bs, acc[i][j], 0, 0, 0);
        }
        // the next tile's LDS-DMA pieces go out between the first MFMA groups (two per group): the matrix core has work queued while the
        // wave spends its issue slots on them, and every piece is under way before the middle of the step
        if (kPer * j < NQ) {
          __builtin_amdgcn_sched_barrier(0);
          if (!(GEMM_DBG(P) & 64) && more) {
#pragma unroll
            for (int u = 0; u < kPer; ++u)
              if (kPer * j + u < NQ) issue_q(t + 1, kPer * j + u);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  }  // LOOP == 0

  // ---- epilogue.  C[row 16 i + 4 lq + r][col 16 j + l15]
  if (GEMM_DBG(P) & 1) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  {
    // per accumulator column tile (16 columns: inside one segment and one head): value -> index byte (+ dequantised value).
    // The index bytes go through LDS so that they leave as whole 16-byte pieces of contiguous output: columns of a plain segment
    // (q, k) as a [row][288] image, columns of a transposed segment (v) as a [column][128 rows] image (pitch 144 B).
    barrier_mem();  // the slots are free: every wave is past its last fragment read
    unsigned char* img_r = lds;
    unsigned char* img_c = lds + G_IMG_C;
    const int c4 = l15 & 3, a4 = l15 >> 2;
    float biav[NJ];  // (all of the lane's bias values in one round trip)
#pragma unroll
    for (int j = 0; j < NJ; ++j) biav[j] = P.bias[min(n0 + 16 * NJ * wn + 16 * j + l15, P.N - 1)];
    const unsigned sel_t = (unsigned)c4 | ((unsigned)(4 + c4) << 8);   // v_perm_b32 selector: byte c4 of the second / of the first source
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nl = 16 * NJ * wn + 16 * j;          // tile-local first column (wave-uniform)
      const int n = n0 + nl;
      if (n < P.N) {
        const int sg = (n >= P.E) + (n >= 2 * P.E);
        const GemmSeg& g = P.seg[sg];
        const FqP f = g.f;
        const float alpha = g.alpha * kApreInv;
        const float bia = biav[j];
        int iadd = 0;  // int8 form: the accumulator is an int32 sum of centred indices; + (128 - zero_point) * column sum of w = the sum over idx - zp
        if constexpr (AM == A_I8) iadd = g.acc_add != nullptr ? g.acc_add[(n - sg * P.E) + l15] : 0;
        auto accv = [&](int i, int r) -> float {
          if constexpr (AM == A_I8) return (float)((int)f32_bits(acc[i][j][r]) + iadd);  // (f32_bits: not __builtin_bit_cast of a vector element - oeh_common.h)
          else return acc[i][j][r];
        };
        // values: wave-uniform row base (scalar registers) + one lane offset for the whole tile column: no vector address arithmetic per store
        const long y_ld = g.y_ld;
        const char* ybase = g.y != nullptr ? reinterpret_cast<const char*>(g.y + (long)(m0 + 16 * MI * wm) * y_ld + (n - sg * P.E)) : nullptr;
        const unsigned y_voff = (unsigned)((4 * lq) * y_ld + l15) * 4u;
        const bool idx_r = g.out != nullptr && !g.transpose, idx_c = g.out != nullptr && g.transpose;
        // the column tile's forms - values or not, index bytes row-major / transposed / none - are wave-uniform: ONE dispatch per column tile into a body
        // compiled for the form (round 5: the tests sat inside the row-block loop - ~290 scalar branches per wave in an epilogue that a single wave per
        // SIMD runs at the latency of its branches)
        auto body = [&](auto values_, auto idxr_, auto idxc_) {
          const bool VALUES = values_, IDXR = idxr_, IDXC = idxc_;   // (std::true_type / false_type: folded at compile time; plain bools: the int8 form keeps ONE body)
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            unsigned word = 0;
            const int rl = 16 * MI * wm + 16 * i + 4 * lq;  // tile-local first row of the lane's four
            if (VALUES) {
              const bool rows_in = m0 + 16 * MI * wm + 16 * i < P.M && !(GEMM_DBG(P) & 16);  // (M % 16 == 0: a 16-row tile is inside or outside as a whole)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float rel = fq_rel(__builtin_fmaf(accv(i, r), alpha, bia), f);
                word = __builtin_amdgcn_cvt_pk_u8_f32(rel + f.zp, r, word);
                if (rows_in) store_wt4_s(ybase + (long)(16 * i + r) * y_ld * 4, y_voff, f.scale * rel);
              }
            } else {
              // (no values wanted: the conversion's saturation to [0, 255] is the clamp)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                word = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fq_quot(__builtin_fmaf(accv(i, r), alpha, bia), f)) + f.zp, r, word);
            }
            word ^= 0x80808080u;
            if (IDXC) *reinterpret_cast<unsigned*>(img_c + (nl + l15) * G_PITCH_C + rl) = word;
            if (IDXR) {
              // 4 x 4 byte transpose inside the quad of lanes (columns 4 a4 .. 4 a4 + 3): lane c4 ends with row rl + c4, four columns
              const unsigned t0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x00, 0xf, 0xf, false);
              const unsigned t1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x55, 0xf, 0xf, false);
              const unsigned t2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0xaa, 0xf, 0xf, false);
              const unsigned t3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0xff, 0xf, 0xf, false);
              const unsigned lo2 = __builtin_amdgcn_perm(t1, t0, sel_t), hi2 = __builtin_amdgcn_perm(t3, t2, sel_t);
              *reinterpret_cast<unsigned*>(img_r + (rl + c4) * GBN + nl + 4 * a4) = __builtin_amdgcn_perm(hi2, lo2, 0x05040100u);
            }
          }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if constexpr (AM == A_I8) body(ybase != nullptr, idx_r, idx_c);   // (five bodies per column tile spill the 128 x 288 int8 form's scalar registers)
        else if (ybase != nullptr) {
          if (idx_r) body(T_{}, T_{}, F_{});
          else if (idx_c) body(T_{}, F_{}, T_{});
          else body(T_{}, F_{}, F_{});
        } else if (idx_r) body(F_{}, T_{}, F_{});
        else if (idx_c) body(F_{}, F_{}, T_{});
      }
    }
    barrier_mem();
    if (GEMM_DBG(P) & 32) return;
    // The three segments' output pointers and layouts in scalar registers, selected per piece (round 5: the loops below indexed P.seg[] with a per-lane
    // segment number - a dependent vector load of the descriptor from the kernel-argument segment in front of every store: 18 round trips per thread,
    // 4.6-5.9 us of the launch).
    signed char* const so0 = P.seg[0].out; signed char* const so1 = P.seg[1].out; signed char* const so2 = P.seg[2].out;
    const bool st0 = P.seg[0].transpose != 0, st1 = P.seg[1].transpose != 0, st2 = P.seg[2].transpose != 0;
    // plain segments: BM rows x BN / 16 pieces, consecutive threads on consecutive pieces of a row
    if ((so0 != nullptr && !st0) || (so1 != nullptr && !st1) || (so2 != nullptr && !st2)) {
#pragma unroll
      for (int it = 0; it < (GBM * (GBN / 16) + 255) / 256; ++it) {
        const int e = tid + 256 * it;
        const int row = e / (GBN / 16), c16 = e - row * (GBN / 16);
        const int n = n0 + 16 * c16, m = m0 + row;
        const int sg = (n >= P.E) + (n >= 2 * P.E);
        signed char* const so = sg == 0 ? so0 : sg == 1 ? so1 : so2;
        const bool tr = sg == 0 ? st0 : sg == 1 ? st1 : st2;
        if (e < GBM * (GBN / 16) && n < P.N && m < P.M && so != nullptr && !tr)
          store_wt16(so + (long)m * P.E + (n - sg * P.E), *reinterpret_cast<const u4*>(img_r + row * GBN + 16 * c16));   // (write-through, as the values: see store_wt4_s)
      }
    }
    // transposed segments: BN columns x BM / 16 pieces of 16 rows (= 16 keys of one batch element: S % 16 == 0)
    if ((so0 != nullptr && st0) || (so1 != nullptr && st1) || (so2 != nullptr && st2)) {
#pragma unroll
      for (int it = 0; it < (GBN * (GBM / 16) + 255) / 256; ++it) {
        const int e = tid + 256 * it;
        const int col = e / (GBM / 16), pc = e - col * (GBM / 16);
        const int n = n0 + col, m = m0 + 16 * pc;
        const int sg = (n >= P.E) + (n >= 2 * P.E);
        signed char* const so = sg == 0 ? so0 : sg == 1 ? so1 : so2;
        const bool tr = sg == 0 ? st0 : sg == 1 ? st1 : st2;
        if (e < GBN * (GBM / 16) && n < P.N && m < P.M && so != nullptr && tr) {
          const int ns = n - sg * P.E;
          int bidx, srow;
          div_magic((unsigned)m, (unsigned)P.S, P.magic_s, bidx, srow);
          store_wt16(so + (((long)bidx * P.H + (ns >> 6)) * 64 + (ns & 63)) * P.S + srow, *reinterpret_cast<const u4*>(img_c + col * G_PITCH_C + 16 * pc));
        }
      }
    }
  }
}

template <int AM, int MI, int NJ, int LOOP = 0>
static int launch_gemm_t(const GemmParams& P, hipStream_t st) {
  static bool attr[64] = {};   // (the LDS opt-in is per device: a process that drives several GPUs sets it on each)
  const int ldsb = LOOP == 1 ? Geo<MI, NJ>::LDS3 : Geo<MI, NJ>::LDS;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -5;
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&oeh_gemm_kernel<AM, MI, NJ, LOOP>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess) return -5;
    attr[dev] = true;
  }
  hipLaunchKernelGGL((oeh_gemm_kernel<AM, MI, NJ, LOOP>), dim3(P.MT * P.NT), dim3(256), ldsb, st, P);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// Tile choice.  128 x 288 (two workgroups per CU, 144 accumulator registers) when the problem fills the chip with such tiles and the
// columns divide evenly enough; 64 x 192 (four workgroups per CU) otherwise: narrow outputs (out_proj: N = 768 is 4 x 192) and
// problems of fewer than 512 large tiles.
int launch_gemm(const GemmParams& P0, hipStream_t st) {
  GemmParams P = P0;
  // diagnostic switches (include/oeh_debug.h), inert unless OEH_DEBUG_HOOKS=1: OEH_GEMM_DBG knocks parts of the kernel out for timing
  // (results are then wrong), OEH_GEMM_TILE = 1 | 2 forces the 128 x 288 | 64 x 192 tile
  static const bool hooks = [] { const char* e = getenv("OEH_DEBUG_HOOKS"); return e != nullptr && e[0] == '1'; }();
  static const int dbg = [] { const char* e = getenv("OEH_GEMM_DBG"); return e ? atoi(e) : 0; }() * (hooks ? 1 : 0);
  static const int force = [] { const char* e = getenv("OEH_GEMM_TILE"); return e ? atoi(e) : 0; }() * (hooks ? 1 : 0);
#ifdef OEH_GEMM_EXPERIMENT
  P.dbg = dbg;
#else
  P.dbg = 0;
  if (dbg != 0) {
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "liboeh_hip: OEH_GEMM_DBG is set but this build has no knock-outs (make experiment): ignored\n"); }
  }
#endif
  P.magic_s = P.S > 1 ? (unsigned)(0x100000000ULL / (unsigned long long)P.S) : 0xffffffffu;
  const long t0 = (long)((P.M + 127) / 128) * ((P.N + 287) / 288);
  const double waste0 = (double)t0 * 128.0 * 288.0 / ((double)P.M * (double)P.N);
  const bool big = force ? force == 1 : (t0 >= 512 && waste0 <= 1.06);
  // fp32 activations on the 128 x 288 tile: the pipelined body (LOOP == 2: ring of two, two workgroups per CU; OEH_GEMM_LOOP0 = 1 keeps the round-4 loop for A/B)
  static const int loop0 = [] { const char* e = getenv("OEH_GEMM_LOOP0"); return e ? atoi(e) : 0; }() * (hooks ? 1 : 0);
  if (big) {
    P.MT = (P.M + 127) / 128; P.NT = (P.N + 287) / 288;
    return P.pairs == 3 ? launch_gemm_t<A_I8, 4, 9>(P, st) : P.pairs == 2 ? (loop0 ? launch_gemm_t<A_F32, 4, 9>(P, st) : launch_gemm_t<A_F32, 4, 9, 2>(P, st)) : P.pairs ? launch_gemm_t<A_PAIRS, 4, 9>(P, st) : launch_gemm_t<A_F16, 4, 9>(P, st);
  }
  // fp32 activations, 128 x 288 tiles for at most one workgroup per CU (BERT-base: M = 4096 -> 256 tiles): the one-workgroup-per-CU loop (LOOP == 1):
  // 64 x 192 tiles would re-read A twelve and W sixty-four times; OEH_GEMM_TILE = 2 keeps the small tile (A/B)
  if (P.pairs == 2 && !force && t0 >= 160 && t0 <= 256 && waste0 <= 1.06 && P.K >= 64) {
    P.MT = (P.M + 127) / 128; P.NT = (P.N + 287) / 288;
    return launch_gemm_t<A_F32, 4, 9, 1>(P, st);
  }
  P.MT = (P.M + 63) / 64; P.NT = (P.N + 191) / 192;
  return P.pairs == 3 ? launch_gemm_t<A_I8, 2, 6>(P, st) : P.pairs == 2 ? launch_gemm_t<A_F32, 2, 6>(P, st) : P.pairs ? launch_gemm_t<A_PAIRS, 2, 6>(P, st) : launch_gemm_t<A_F16, 2, 6>(P, st);
}

}  // namespace oeh
