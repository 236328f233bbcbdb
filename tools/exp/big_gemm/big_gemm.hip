// EXPERIMENT (round 5): a plain 16-bit Linear, C = A W^T + bias, as ONE workgroup per CU - 4 waves, one per SIMD, each with a 128 x 144 accumulator
// tile (288 accumulator registers; the 512-register budget of a single wave per SIMD), K in steps of 32 through a 4-slot LDS-DMA ring, the fragments
// of step t + 1 read from LDS while the matrix core works through step t.  Prototype for the projections of the fp16 / bf16 modules (library GEMMs:
// 40 us for M = 8192, N = 2304, K = 768 = 0.72 PFLOP/s).  hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libbig_gemm.so big_gemm.hip
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../../outeffhop_amd/csrc/oeh_common.h"

namespace oeh {
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

struct BigP {
  const void* a; const void* w; const float* bias; void* c;
  long lda, ldw, ldc;
  int M, N, K, MT, NT, bf16, dbg;
  float alpha;
};

#ifndef BIG_DBG
#define BIG_DBG 0
#endif
template <int MI, int NJ, int R>
__global__ __launch_bounds__(256, 1) void big_gemm_kernel(const BigP P) {
  constexpr int BM = 32 * MI, BN = 32 * NJ, ROWB = 64;
  constexpr int SLOT = (BM + BN) * ROWB, W0 = BM * ROWB;
  constexpr int NPA = BM / 16, NP = (BM + BN) / 16, NQ = (NP + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
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
  const int m0 = mi * BM, n0 = ni * BN;
  const int T = P.K / 32;

  const unsigned lds_base = lds_offset(lds);
  const int prow = lane >> 2;
  const int pchunk = (lane & 3) ^ ((-(lane >> 4)) & 3);
  const unsigned char* ab = reinterpret_cast<const unsigned char*>(P.a);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(P.w);
  unsigned voff[NQ];
  unsigned dst[NQ];
  bool isw[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int p = 4 * q + wave;
    if (p > NP - 1) p = NP - 1;   // (the last round repeats a piece: every wave issues the same number of requests per step)
    if (p < NPA) {
      const int r = min(m0 + p * 16 + prow, P.M - 1);
      voff[q] = (unsigned)(((long)r * P.lda) * 2 + pchunk * 16);
      isw[q] = false;
    } else {
      const int r = min(n0 + (p - NPA) * 16 + prow, P.N - 1);
      voff[q] = (unsigned)(((long)r * P.ldw) * 2 + pchunk * 16);
      isw[q] = true;
    }
    dst[q] = (unsigned)(p * 1024);
  }
  auto issue = [&](int t) {
    const unsigned slot = lds_base + (unsigned)((t % R) * SLOT);
    const long kb = (long)t * 64;
#pragma unroll
    for (int q = 0; q < NQ; ++q) glds16_s((isw[q] ? wb : ab) + kb, voff[q], __builtin_amdgcn_readfirstlane(slot + dst[q]));
  };
  const unsigned swz = (unsigned)((lq ^ ((-(l15 >> 2)) & 3)) << 4);
  const unsigned a_off = (unsigned)((16 * MI * wm + l15) * ROWB) + swz;
  const unsigned w_off = (unsigned)(W0 + (16 * NJ * wn + l15) * ROWB) + swz;

  f4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  h8v fa[2][MI], fw[2][NJ];
  auto read_frags = [&](int t, auto par) {
    constexpr int PAR = decltype(par)::value;
    const unsigned char* sl = lds + (t % R) * SLOT;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[PAR][i] = *reinterpret_cast<const h8v*>(sl + a_off + i * 16 * ROWB);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fw[PAR][j] = *reinterpret_cast<const h8v*>(sl + w_off + j * 16 * ROWB);
  };
  auto wait_tiles = [&](int newer) {   // all but the `newer` youngest tiles of this wave's requests have landed
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ) : "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(R == 4 && 2 * NQ <= 63, "ring of four; vmcnt immediates");

  // prologue: tiles 0, 1, 2 requested; fragments of tile 0 in registers
  issue(0);
  if (T > 1 && !(BIG_DBG & 1)) issue(1);
  if (T > 2 && !(BIG_DBG & 1)) issue(2);
  if (BIG_DBG & 1) wait_tiles(0); else wait_tiles(min(T - 1, 2));
  barrier_mem();
  read_frags(0, std::integral_constant<int, 0>{});

  auto issue_piece = [&](int t, int q) {
    const unsigned slot = lds_base + (unsigned)((t % R) * SLOT);
    glds16_s((isw[q] ? wb : ab) + (long)t * 64, voff[q], __builtin_amdgcn_readfirstlane(slot + dst[q]));
  };
  static_assert(NQ <= NJ, "one LDS-DMA piece behind each group of MI MFMAs");
#ifndef BIG_DBG
#define BIG_DBG 0
#endif
  // step t: [tile t + 1 landed: wait + barrier] then NJ groups of { one request of tile t + 3 | two fragment reads of tile t + 1 | MI MFMAs of tile t }
  // - the memory instructions never queue up in front of the matrix core.  NXT / DMA: compile-time (the last three steps are peeled).
  auto step = [&](int t, auto par, auto nxt_, auto dma_, auto w1_) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool NXT = decltype(nxt_)::value, DMA = decltype(dma_)::value && !(BIG_DBG & 1);
    if constexpr (NXT) {
      if (!(BIG_DBG & 1)) {   // tile t + 1 has landed: all but tile t + 2's requests (none in the last two steps)
        if constexpr (decltype(w1_)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!(BIG_DBG & 4)) barrier_mem();
    }
    const unsigned char* sl = lds + ((t + 1) % R) * SLOT;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if constexpr (DMA) { if (j < NQ) issue_piece(t + 3, j); }
      if constexpr (NXT) {
        if (j < MI) fa[1 - PAR][j] = *reinterpret_cast<const h8v*>(sl + a_off + j * 16 * ROWB);
        fw[1 - PAR][j] = *reinterpret_cast<const h8v*>(sl + w_off + j * 16 * ROWB);
      }
      if constexpr (!(BIG_DBG & 2)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          // (explicit register classes: 64 accumulator tiles in the 256 AGPRs, the ninth column's 8 in VGPRs - left to the allocator the 288 registers
          // shuffle between the files around every MFMA)
          if (j < NJ - 1 || NJ * MI * 4 <= 256) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[PAR][j]), "v"(fa[PAR][i]));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[PAR][j]), "v"(fa[PAR][i]));
        }
      } else {
        if (j == 0) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[PAR][0], fa[PAR][0], acc[0][0], 0, 0, 0);
        asm volatile("" ::"v"(fw[PAR][j]));
        if (j < MI) asm volatile("" ::"v"(fa[PAR][j]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef BIG_VACC
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(acc[i][NJ - 1]));   // the ninth column's accumulators live in VGPRs (288 > the 256 AGPRs)
#endif
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  int t = 0;
  for (; t + 4 < T; t += 2) {
    step(t, std::integral_constant<int, 0>{}, T_{}, T_{}, T_{});
    step(t + 1, std::integral_constant<int, 1>{}, T_{}, T_{}, T_{});
  }
  // T even, T >= 4: four steps left (t = T - 4): the first still requests tile T - 1
  step(t, std::integral_constant<int, 0>{}, T_{}, T_{}, T_{});
  step(t + 1, std::integral_constant<int, 1>{}, T_{}, F_{}, T_{});
  step(t + 2, std::integral_constant<int, 0>{}, T_{}, F_{}, F_{});
  step(t + 3, std::integral_constant<int, 1>{}, F_{}, F_{}, F_{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results before the epilogue reads them: the inline-asm MFMAs are outside the compiler's hazard model)

  // epilogue: acc[i][j][r] = C[m0 + 16 MI wm + 16 i + l15][n0 + 16 NJ wn + 16 j + 4 lq + r].  Through LDS, per wave (no barrier: a wave's LDS operations
  // execute in order), in halves of MI / 2 row blocks: the lane's 4 consecutive columns as 8 B into the wave's image [16 MI / 2 rows][PITCH], then whole
  // 16-byte chunks of contiguous output rows (16 NJ columns = 18 chunks per row) leave write-through.
  constexpr int PITCH = 32 * NJ + 16, HALF = MI / 2, HROWS = 16 * HALF, CPRW = 2 * NJ;   // 304 B; 64 rows; 18 chunks of 16 B per row
  static_assert(4 * HROWS * PITCH <= R * SLOT, "the four images fit the ring");
  barrier_mem();   // every wave has read its last fragments
  unsigned char* img = lds + wave * (HROWS * PITCH);
  unsigned char* cb = reinterpret_cast<unsigned char*>(P.c);
  const int nw0 = n0 + 16 * NJ * wn;
  f4 bv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nw0 + 16 * j + 4 * lq;
    bv[j] = f4{0.f, 0.f, 0.f, 0.f};
    if (P.bias != nullptr && n < P.N) bv[j] = *reinterpret_cast<const f4*>(P.bias + n);
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int ii = 0; ii < HALF; ++ii) {
        const f4 v = acc[hf * HALF + ii][j] + bv[j];
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        const h4v o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<h4v*>(img + (16 * ii + l15) * PITCH + (16 * j + 4 * lq) * 2) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mw0 = m0 + 16 * MI * wm + HROWS * hf;
#pragma unroll
    for (int s_ = 0; s_ < (HROWS * CPRW + 63) / 64; ++s_) {
      const int q = lane + 64 * s_;
      const int row = q / CPRW, cc = q - row * CPRW;
      if (q < HROWS * CPRW) {
        const u4 w = *reinterpret_cast<const u4*>(img + row * PITCH + cc * 16);
        const int m = mw0 + row, n = nw0 + 8 * cc;
        if (m < P.M && n + 8 <= P.N) store_wt16(cb + ((long)m * P.ldc + n) * 2, w);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the image is rewritten by the second half)
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same structure for fp32 ACTIVATIONS against integer-valued fp16 weights (the QuantLinear projections of an fp32 model, oeh_gemm.hip A_F32):
// the fp32 rows go to LDS as they are (128 B per row and K step, 8-row pieces, swz32), a wave reads a fragment's 8 values (two ds_read_b128),
// splits them into the fp16 pair 32 x = hi + lo' (oeh_common.h: split8_raw_scaled, one instruction at a time between the MFMAs) and multiplies both
// against W.  Fragment-major: group i = A fragment i against the 9 W fragments (18 MFMAs), while fragment i + 1 is converted and fragment i + 2 read.
// Ring of 3 slots of 50 KB; tile t + 2 is requested during step t; one wait + barrier per step.
__device__ __forceinline__ int swz32_(int row) { return (int)((0x31765420u >> (4 * ((row >> 1) & 7))) & 7u); }

template <int MI, int NJ, int R>
__global__ __launch_bounds__(256, 1) void big_gemm_pairs_kernel(const BigP P) {
  constexpr int BM = 32 * MI, BN = 32 * NJ;
  constexpr int W0 = BM * 128, SLOT = W0 + BN * 64;
  constexpr int NPA = BM / 8, NPW = BN / 16, NP = NPA + NPW, NQ = (NP + 3) / 4;   // 32 + 18 = 50 pieces, 13 per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
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
  const int m0 = mi * BM, n0 = ni * BN;
  const int T = P.K / 32;
  fp16_overflow_clamp();

  const unsigned lds_base = lds_offset(lds);
  const unsigned char* ab = reinterpret_cast<const unsigned char*>(P.a);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(P.w);
  unsigned voff[NQ], dst[NQ];
  bool isw[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int p = 4 * q + wave;
    if (p > NP - 1) p = NP - 1;
    if (p < NPA) {
      const int rl = 8 * p + (lane >> 3);
      const int r = min(m0 + rl, P.M - 1);
      voff[q] = (unsigned)(((long)r * P.lda) * 4 + ((lane & 7) ^ swz32_(rl)) * 16);
      dst[q] = (unsigned)(p * 1024);
      isw[q] = false;
    } else {
      const int pw = p - NPA;
      const int r = min(n0 + pw * 16 + (lane >> 2), P.N - 1);
      voff[q] = (unsigned)(((long)r * P.ldw) * 2 + ((lane & 3) ^ ((-(lane >> 4)) & 3)) * 16);
      dst[q] = (unsigned)(W0 + pw * 1024);
      isw[q] = true;
    }
  }
  auto issue_piece = [&](int t, int q) {
    const unsigned slot = lds_base + (unsigned)((t % R) * SLOT);
    glds16_s(isw[q] ? wb + (long)t * 64 : ab + (long)t * 128, voff[q], __builtin_amdgcn_readfirstlane(slot + dst[q]));
  };
  const unsigned a32_off = (unsigned)((16 * MI * wm + l15) * 128) + (unsigned)(((2 * lq) ^ swz32_(l15)) << 4);
  const unsigned w_off = (unsigned)(W0 + (16 * NJ * wn + l15) * 64) + (unsigned)((lq ^ ((-(l15 >> 2)) & 3)) << 4);

  f4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const float kpre = 32.0f;
  f4 raw[2][2];
  unsigned hi[2][4], lo[2][4];
  h8v fw[NJ];
  static_assert(R == 3 && NQ <= 16, "ring of three; pieces over the MI groups");
  constexpr int PPG = (NQ + MI - 1) / MI;   // pieces per group (2), the first groups

  issue_piece(0, 0);
#pragma unroll
  for (int q = 1; q < NQ; ++q) issue_piece(0, q);
  if (T > 1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) issue_piece(1, q);
  }

  auto read_raw = [&](const unsigned char* sl, int i, f4 (&dstr)[2]) {
    dstr[0] = *reinterpret_cast<const f4*>(sl + a32_off + i * 16 * 128);
    dstr[1] = *reinterpret_cast<const f4*>(sl + (a32_off ^ 16u) + i * 16 * 128);
  };
  // one instruction of the split of 8 values (oeh_common.h: split8_raw_scaled, unrolled so that each can sit between two MFMAs): k = 0..7 the hi halves, 8..15 the lo halves
  auto split_op = [&](int k, const f4 (&x)[2], unsigned (&h)[4], unsigned (&l)[4]) {
    const int e = k & 7;                       // element 0..7
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

  auto step = [&](int t, auto dma_, auto w1_) {
    constexpr bool DMA = decltype(dma_)::value;
    // tile t has landed (tile t + 1 may be in flight), every wave has left tile t - 1: its slot takes tile t + 2
    if constexpr (decltype(w1_)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier_mem();
    const unsigned char* sl = lds + (t % R) * SLOT;
    read_raw(sl, 0, raw[0]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fw[j] = *reinterpret_cast<const h8v*>(sl + w_off + j * 16 * 64);
    read_raw(sl, 1, raw[1]);
#pragma unroll
    for (int k = 0; k < 16; ++k) split_op(k, raw[0], hi[0], lo[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (DMA) {
#pragma unroll
        for (int u = 0; u < PPG; ++u)
          if (PPG * i + u < NQ) issue_piece(t + 2, PPG * i + u);
      }
      if (i + 2 < MI) read_raw(sl, i + 2, raw[i & 1]);
      const h8v ahv = __builtin_bit_cast(h8v, u4{hi[i & 1][0], hi[i & 1][1], hi[i & 1][2], hi[i & 1][3]});
      const h8v alv = __builtin_bit_cast(h8v, u4{lo[i & 1][0], lo[i & 1][1], lo[i & 1][2], lo[i & 1][3]});
#pragma unroll
      for (int k = 0; k < 2 * NJ; ++k) {
        const int j = k % NJ;
        const h8v av = k < NJ ? ahv : alv;
        if (j < NJ - 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[j]), "v"(av));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[j]), "v"(av));
        if (i + 1 < MI && k >= 1 && k - 1 < 16) split_op(k - 1, raw[(i + 1) & 1], hi[(i + 1) & 1], lo[(i + 1) & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    int t = 0;
    for (; t + 2 < T; ++t) step(t, std::true_type{}, std::true_type{});
    step(t, std::false_type{}, std::true_type{});
    step(t + 1, std::false_type{}, std::false_type{});
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // epilogue (prototype): fp16 of alpha * acc + bias through the per-wave LDS images, as in the fp16 kernel
  constexpr int PITCH = 32 * NJ + 16, HALF = MI / 2, HROWS = 16 * HALF, CPRW = 2 * NJ;
  static_assert(4 * HROWS * PITCH <= R * SLOT, "the four images fit the ring");
  barrier_mem();
  unsigned char* img = lds + wave * (HROWS * PITCH);
  unsigned char* cb = reinterpret_cast<unsigned char*>(P.c);
  const int nw0 = n0 + 16 * NJ * wn;
  const float alpha = P.alpha * (1.0f / 32.0f);
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = nw0 + 16 * j + 4 * lq;
      f4 bv = f4{0.f, 0.f, 0.f, 0.f};
      if (P.bias != nullptr && n < P.N) bv = *reinterpret_cast<const f4*>(P.bias + n);
#pragma unroll
      for (int ii = 0; ii < HALF; ++ii) {
        const f4 v = acc[hf * HALF + ii][j] * alpha + bv;
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        const h4v o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<h4v*>(img + (16 * ii + l15) * PITCH + (16 * j + 4 * lq) * 2) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mw0 = m0 + 16 * MI * wm + HROWS * hf;
#pragma unroll
    for (int s_ = 0; s_ < (HROWS * CPRW + 63) / 64; ++s_) {
      const int q = lane + 64 * s_;
      const int row = q / CPRW, cc = q - row * CPRW;
      if (q < HROWS * CPRW) {
        const u4 w = *reinterpret_cast<const u4*>(img + row * PITCH + cc * 16);
        const int m = mw0 + row, n = nw0 + 8 * cc;
        if (m < P.M && n + 8 <= P.N) store_wt16(cb + ((long)m * P.ldc + n) * 2, w);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}
}  // namespace oeh

extern "C" int big_gemm_f16(const void* a, const void* w, const float* bias, void* c, int M, int N, int K, long lda, long ldw, long ldc, void* stream) {
  using namespace oeh;
  constexpr int MI = 8, NJ = 9, R = 4;
  BigP P;
  P.a = a; P.w = w; P.bias = bias; P.c = c; P.lda = lda; P.ldw = ldw; P.ldc = ldc; P.M = M; P.N = N; P.K = K; P.bf16 = 0;
  { const char* e = getenv("BIG_DBG"); P.dbg = e ? atoi(e) : 0; }
  P.MT = (M + 32 * MI - 1) / (32 * MI); P.NT = (N + 32 * NJ - 1) / (32 * NJ);
  if (K % 64 != 0 || K < 128) return -1;
  const size_t ldsb = (size_t)R * (32 * MI + 32 * NJ) * 64;
  static bool once = false;
  if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(&big_gemm_kernel<MI, NJ, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); once = true; }
  hipLaunchKernelGGL((big_gemm_kernel<MI, NJ, R>), dim3(P.MT * P.NT), dim3(256), ldsb, reinterpret_cast<hipStream_t>(stream), P);
  return (int)hipGetLastError();
}

template <int MI>
static int big_gemm_pairs_t(const float* a, const void* w, const float* bias, void* c, int M, int N, int K, long lda, long ldw, long ldc, float alpha, void* stream) {
  using namespace oeh;
  constexpr int NJ = 9, R = 3;
  BigP P;
  P.a = a; P.w = w; P.bias = bias; P.c = c; P.lda = lda; P.ldw = ldw; P.ldc = ldc; P.M = M; P.N = N; P.K = K; P.bf16 = 0; P.dbg = 0; P.alpha = alpha;
  P.MT = (M + 32 * MI - 1) / (32 * MI); P.NT = (N + 32 * NJ - 1) / (32 * NJ);
  if (K % 32 != 0 || K < 64) return -1;
  const size_t ldsb = (size_t)R * (32 * MI * 128 + 32 * NJ * 64);
  static bool once = false;
  if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(&big_gemm_pairs_kernel<MI, NJ, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); once = true; }
  hipLaunchKernelGGL((big_gemm_pairs_kernel<MI, NJ, R>), dim3(P.MT * P.NT), dim3(256), ldsb, reinterpret_cast<hipStream_t>(stream), P);
  return (int)hipGetLastError();
}
extern "C" int big_gemm_pairs(const float* a, const void* w, const float* bias, void* c, int M, int N, int K, long lda, long ldw, long ldc, float alpha, void* stream) {
  return big_gemm_pairs_t<8>(a, w, bias, c, M, N, K, lda, ldw, ldc, alpha, stream);
}
extern "C" int big_gemm_pairs_m4(const float* a, const void* w, const float* bias, void* c, int M, int N, int K, long lda, long ldw, long ldc, float alpha, void* stream) {
  return big_gemm_pairs_t<4>(a, w, bias, c, M, N, K, lda, ldw, ldc, alpha, stream);
}
