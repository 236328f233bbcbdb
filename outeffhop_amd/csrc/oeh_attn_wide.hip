// One-pass fused attention on the 32x32x16 matrix-core shape ("wide" form of oeh_attn_flash.inl) - VERDICT r3 next #1's second
// candidate for the headline launch (OPT-125m softmax1: plain softmax / softmax_1, 16-bit storage, head dim 64, masks none | causal).
//
// Why: the one-pass kernel's loop is bound by the SIMD's vector issue port, and every MFMA holds that port for 8 cycles
// whatever its shape (MI355X_MICROARCH.md, cycle constants).  v_mfma_f32_32x32x16 does twice the work per instruction: a wave's
// 32-row x 64-key tile takes 8 + 8 product MFMAs (+ 4 for the row sums) instead of 16 + 16 + 4 - 160 instead of 288 issue cycles per
// tile, at the same matrix-pipe time - and one lane holds 32 scores of ONE query (its partner lane l ^ 32 the other 32), so the row
// statistics are one register each and the cross-lane step is a single v_permlane32_swap.
//
// Same skeleton as the one-pass kernel: 128 query rows per workgroup, 4 waves, K / V tiles of 64 keys through a 3-stage LDS-DMA
// ring (Q first, into the stage the ring does not use yet), one counted wait + one barrier per tile, lazy reference in the
// exponent domain, row sums on the matrix core, O staged through LDS and written as whole rows, write-through.
// Layout: swapped products S^T = K Q^T, O^T = V^T P^T.  32x32x16 operands: lane l holds A[row l & 31][k = 8 (l >> 5) + j] and
// B[k = 8 (l >> 5) + j][col l & 31], j = 0..7; C[row 8 (r >> 2) + 4 (l >> 5) + (r & 3)][col l & 31], r = 0..15.  A wave owns 32
// CONTIGUOUS query rows (columns of both products); which 32 of the workgroup's 128 rotates with the block id so that the waves
// with one causal tile more do not always sit on the same SIMDs.  P^T for the k-step (32-key tile kt, half hh) is the lane's own
// score registers 8 hh .. 8 hh + 7 of tile kt (keys 32 kt + 16 hh + 8 (j >> 2) + 4 hi + (j & 3)); V^T's operand gathers the same
// keys with two ds_read_b64_tr_b16.  LDS images: K rows XOR-swizzled by (row >> 1) & 7 on 16-byte chunks, V rows by bit 1 of the row
// on 64-byte segments: both fragment reads are conflict free under the lane-group rules of tools/lds_bank_sim.py.
#include "../../include/oeh.h"
#include "oeh_attn_params.h"

#include <type_traits>

namespace oeh {

typedef float f16v __attribute__((ext_vector_type(16)));

template <int IN>
__device__ __forceinline__ f16v mfma32(u4 a, u4 b, f16v c) {
  if constexpr (IN == IN_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float pair_allreduce_max(float x) {  // over the two lanes (l, l ^ 32) that hold one query
  auto b = __builtin_amdgcn_permlane32_swap(f32_bits(x), f32_bits(x), false, false);
  return __builtin_fmaxf(bits_f32(b[0]), bits_f32(b[1]));
}

template <int IN>
__global__ __launch_bounds__(256, 3) void oeh_attn_wide_kernel(const AttnParams P) {
  constexpr int ROWB = 128, TILEB = 64 * ROWB, STAGEB = 2 * TILEB, R = 3, G = 2, RPP = 8;
  constexpr float NEGT = -1.0e30f;  // exponent argument of a masked key: exp2 -> 0 exactly
  constexpr float kThr = 8.0f;      // lazy reference: P stays <= 2^8
  __shared__ __attribute__((aligned(16))) unsigned char lds[R * STAGEB];
  {
    asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.nBHpad), "s"(P.nQT), "s"(P.nBH), "s"(P.H), "s"(P.Sq), "s"(P.Sk), "s"(P.causal), "s"(P.snake),
                 "s"(P.magic_nbh), "s"(P.magic_h), "s"(P.qs_b), "s"(P.qs_h), "s"(P.qs_s), "s"(P.ks_b), "s"(P.ks_h), "s"(P.ks_s), "s"(P.vs_b), "s"(P.vs_h),
                 "s"(P.vs_s));
  }
  const int bid = P.snake ? snake_block_id(blockIdx.x, P.nQT * P.nBHpad) : (int)blockIdx.x;
  int qt_rev, bh;
  div_magic((unsigned)bid, (unsigned)P.nBHpad, P.magic_nbh, qt_rev, bh);
  if (bh >= P.nBH) return;
  int b, h;
  div_magic((unsigned)bh, (unsigned)P.H, P.magic_h, b, h);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31, hi = lane >> 5;
  const int Sk = P.Sk, Sq = P.Sq, off = Sk - Sq, causal = P.causal;
  const int qt = P.nQT - 1 - qt_rev;                 // heaviest causal tiles first
  const int rg = (wave + bid) & 3;                    // which 32 rows of the workgroup's 128 this wave owns
  const int q0 = 128 * qt + 32 * rg;
  const int last_row_wg = 128 * qt + 127;
  const int n_kt = ((causal ? min(Sk, max(0, last_row_wg + 1 + off)) : Sk) + 63) >> 6;   // tiles the workgroup streams
  const int nkw = ((causal ? min(Sk, max(0, q0 + 32 + off)) : Sk) + 63) >> 6;             // tiles this wave's rows need
  const int tm0 = ((causal ? min(q0 + off, Sk - 1) : Sk - 1) + 1) >> 6;                   // first tile that holds a masked key for the wave's first row

  // ---- LDS-DMA stream of (K tile, V tile) stages: scalar bases advance by 64 rows per stage, lane byte offsets are constant
  const unsigned char* kcur = reinterpret_cast<const unsigned char*>(P.k) + 2 * (bh_offset(b, P.ks_b, h, P.ks_h));
  const unsigned char* vcur = reinterpret_cast<const unsigned char*>(P.v) + 2 * (bh_offset(b, P.vs_b, h, P.vs_h));
  const int prow = lane >> 3, pch = lane & 7;
  const unsigned lds_base = lds_offset(lds);
  auto piece_row = [&](int j) { return (wave * G + j) * RPP + prow; };
  auto kchunk = [&](int row) { return pch ^ ((row >> 1) & 7); };
  auto vchunk = [&](int row) { return ((((pch >> 1) ^ (((row >> 1) & 1) << 1)) << 1) | (pch & 1)); };
  unsigned koff[G], voff[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int row = piece_row(j);
    koff[j] = 2u * (unsigned)(row * P.ks_s + kchunk(row) * 8);
    voff[j] = 2u * (unsigned)(row * P.vs_s + vchunk(row) * 8);
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
    } else {
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(kcur, koff[j], slot + j * 1024);
#pragma unroll
      for (int j = 0; j < G; ++j) glds16_s(vcur, voff[j], slot + TILEB + j * 1024);
    }
    kcur += kstep;
    vcur += vstep;
    ++nx_tile;
    nx_slot = (nx_slot == R - 1) ? 0 : nx_slot + 1;
  };
  {  // Q first, as two K-shaped tiles (rows 0..63, 64..127 of the workgroup) in the stage the ring does not use yet
    const unsigned char* qbase = reinterpret_cast<const unsigned char*>(P.q) + 2 * (bh_offset(b, P.qs_b, h, P.qs_h));
    const unsigned qslot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((R - 1) * STAGEB + wave * G * 1024));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int row = piece_row(j);
        int qrow = 128 * qt + 64 * t + row;
        qrow = qrow < Sq ? qrow : Sq - 1;  // rows past Sq: finite data, never stored
        glds16_s_nt(qbase, 2u * (unsigned)(qrow * P.qs_s + kchunk(row) * 8), qslot + t * TILEB + j * 1024);
      }
    }
  }
  issue_next();
  if (1 < n_kt) issue_next();
  auto wait_vm = [&](auto nc) {  // s_waitcnt vmcnt(N * G)
    constexpr int N = decltype(nc)::value * G;
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  };
  // Q and K tile 0 landed, for every wave: all but the G (V tile 0) + 2G (stage 1) younger transfers
  if (1 < n_kt) wait_vm(std::integral_constant<int, 3>{});
  else wait_vm(std::integral_constant<int, 1>{});
  barrier_mem();

  // lane-constant parts of the LDS fragment addresses
  const unsigned char* kaddr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kaddr[ks] = lds + c32 * ROWB + (((2 * ks + hi) ^ ((c32 >> 1) & 7)) << 4);
  const int c16 = lane & 15, dhalf = (lane >> 4) & 1;
  const unsigned char* vaddr[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    vaddr[mt] = lds + TILEB + (4 * hi + (c16 >> 2)) * ROWB + ((((2 * mt + dhalf) ^ (((c16 >> 3) & 1) << 1))) << 5) + ((c16 & 3) << 3);

  // Q^T operands of the wave's 32 rows (rows 32 rg .. of the workgroup: tile rg >> 1, rows 32 (rg & 1) ..)
  u4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const u4*>(kaddr[ks] + (R - 1) * STAGEB + (rg >> 1) * TILEB + (rg & 1) * 32 * ROWB);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const float c1 = P.scale * kLog2e;
  float mcneg = 0.0f;  // -(reference score) * c1: t = fma(s, c1, mcneg) is the exponent argument
  float lsum = 0.0f;   // this lane's share of the row sum (its 32 keys per tile; the partner lane l ^ 32 holds the other 32)
  f16v o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  const int klim = causal ? min(q0 + c32 + off, Sk - 1) : Sk - 1;  // last visible key of this lane's query

  auto tile = [&](auto firstc, const int i, const int soff) {
    constexpr bool FIRST = decltype(firstc)::value;
    __builtin_amdgcn_s_setprio(1);
    f16v s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = mfma32<IN>(*reinterpret_cast<const u4*>(kaddr[ks] + soff + kt * 32 * ROWB), qf[ks], acc);
      s[kt] = acc;
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = __builtin_fmaf(s[kt][r], c1, mcneg);
    if (i >= tm0) {  // causal / tail mask (tiles that hold a masked key for some row of the wave): a compare + select per element
      // against the lane's last visible key, relative to the tile's first key
      const int rel = klim - 64 * i - 4 * hi;   // key 32 kt + 8 (r >> 2) + (r & 3) of the tile is masked iff it is > rel
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = (32 * kt + 8 * (r >> 2) + (r & 3) > rel) ? NEGT : s[kt][r];
    }
    // row maximum of the lane's 32 exponent arguments: four independent chains (a single chain of 16 dependent v_max3 is pure latency)
    float m4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kt = u >> 1, r0 = 8 * (u & 1);
      m4[u] = __builtin_fmaxf(__builtin_fmaxf(s[kt][r0], s[kt][r0 + 1]), s[kt][r0 + 2]);
      m4[u] = __builtin_fmaxf(__builtin_fmaxf(m4[u], s[kt][r0 + 3]), s[kt][r0 + 4]);
      m4[u] = __builtin_fmaxf(__builtin_fmaxf(m4[u], s[kt][r0 + 5]), s[kt][r0 + 6]);
      m4[u] = __builtin_fmaxf(m4[u], s[kt][r0 + 7]);
    }
    float mt_ = __builtin_fmaxf(__builtin_fmaxf(m4[0], m4[1]), __builtin_fmaxf(m4[2], m4[3]));
    const float thr = (i == 0) ? -1.0e20f : kThr;
    // the exponentials are formed against the CURRENT reference straight away (they do not wait for the maximum chain and the branch);
    // when some row has to move its reference - always on tile 0, rare afterwards - they are redone below
    f16v e[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) e[kt][r] = __builtin_amdgcn_exp2f(s[kt][r]);
    if (__builtin_amdgcn_ballot_w64(mt_ > thr) != 0) {  // some row moves its reference (always on the first tile; later only beyond 2^8)
      mt_ = pair_allreduce_max(mt_);
      const float delta = (mt_ > thr) ? mt_ : 0.0f;
      mcneg -= delta;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) e[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - delta);
      if (i != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        lsum *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[0][r] *= alpha;
          o[1][r] *= alpha;
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) s[kt] = e[kt];
    {  // row sums as fp32 adds of the exponentials, four independent chains (a 32x32x16 ones-operand MFMA per 16 keys costs the matrix
       // pipe 128 cycles per tile and 16 accumulator registers; measured 3 - 9 % slower)
      float a4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kt = u >> 1, r0 = 8 * (u & 1);
        a4[u] = (s[kt][r0] + s[kt][r0 + 1]) + (s[kt][r0 + 2] + s[kt][r0 + 3]);
        a4[u] += (s[kt][r0 + 4] + s[kt][r0 + 5]) + (s[kt][r0 + 6] + s[kt][r0 + 7]);
      }
      lsum += (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    u4 pb[2][2];  // P^T operands: k-step (kt, hh) = the lane's registers 8 hh .. 8 hh + 7 of tile kt
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int r0 = 8 * hh;
        if constexpr (IN == IN_BF16)
          pb[kt][hh] = u4{pack2_bf16(s[kt][r0], s[kt][r0 + 1]), pack2_bf16(s[kt][r0 + 2], s[kt][r0 + 3]), pack2_bf16(s[kt][r0 + 4], s[kt][r0 + 5]), pack2_bf16(s[kt][r0 + 6], s[kt][r0 + 7])};
        else
          pb[kt][hh] = u4{pack2_f16(s[kt][r0], s[kt][r0 + 1]), pack2_f16(s[kt][r0 + 2], s[kt][r0 + 3]), pack2_f16(s[kt][r0 + 4], s[kt][r0 + 5]), pack2_f16(s[kt][r0 + 6], s[kt][r0 + 7])};
      }
    if constexpr (FIRST) {  // V tile 0 landed for every wave; every wave has its Q operands: the Q stage can be refilled with stage 2
      if (1 < n_kt) wait_vm(std::integral_constant<int, 2>{});
      else wait_vm(std::integral_constant<int, 0>{});
      barrier_mem();
      if (2 < n_kt) issue_next();
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const unsigned char* a0 = vaddr[mt] + soff + (32 * kt + 16 * hh) * ROWB;
          const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
          const s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 8 * ROWB));
          const u2 l2 = __builtin_bit_cast(u2, lo), h2 = __builtin_bit_cast(u2, hi4);
          o[mt] = mfma32<IN>(u4{l2.x, l2.y, h2.x, h2.y}, pb[kt][hh], o[mt]);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };

  tile(std::true_type{}, 0, 0);  // every row sees key 0: tile 0 is computed by every wave
  int slot_i = 1;
  for (int i = 1; i < n_kt; ++i) {
    if (i + 1 < n_kt) wait_vm(std::integral_constant<int, 2>{});
    else wait_vm(std::integral_constant<int, 0>{});
    barrier_mem();
    if (i + 2 < n_kt) issue_next();
    const int soff = slot_i * STAGEB;
    slot_i = (slot_i == R - 1) ? 0 : slot_i + 1;
    if (i >= nkw) continue;  // this wave's rows end before this tile (causal)
    tile(std::false_type{}, i, soff);
  }

  // ---- epilogue: 1 / denominator (and the gate), O^T staged through the free stage so that global stores are whole rows.  Stage
  // n_kt % R is free (its last reader was tile n_kt - 3, no DMA is in flight); a wave owns 32 rows x 128 B of it: no barrier.
  unsigned char* ebase = lds + (n_kt % R) * STAGEB + wave * (32 * ROWB);
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int ce = lane_e & 31, he = lane_e >> 5;
  {
    const int qrow = q0 + ce;
    float den = lsum;
    {  // the partner lane's share
      auto sw = __builtin_amdgcn_permlane32_swap(f32_bits(den), f32_bits(den), false, false);
      den = bits_f32(sw[0]) + bits_f32(sw[1]);
    }
    if (P.base != 0) den = den + __builtin_amdgcn_exp2f(mcneg);  // softmax_1: + 1*exp(-reference)  (vutils/softmax_1.py:18-20)
    float rowscale = 1.0f / den;
    if (P.gate != nullptr && qrow < Sq) rowscale = rowscale * P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qrow * P.gs_s];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {  // d = 32 mt + 8 rr + 4 he + 0..3: 8 bytes at chunk (4 mt + rr), half he
        u2 w;
        if constexpr (IN == IN_BF16) {
          w.x = pack2_bf16(o[mt][4 * rr] * rowscale, o[mt][4 * rr + 1] * rowscale);
          w.y = pack2_bf16(o[mt][4 * rr + 2] * rowscale, o[mt][4 * rr + 3] * rowscale);
        } else {
          w.x = pack2_f16(o[mt][4 * rr] * rowscale, o[mt][4 * rr + 1] * rowscale);
          w.y = pack2_f16(o[mt][4 * rr + 2] * rowscale, o[mt][4 * rr + 3] * rowscale);
        }
        *reinterpret_cast<u2*>(ebase + ce * ROWB + ((((4 * mt + rr) ^ (ce & 7)) << 4) | (he << 3))) = w;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before it reads them back
  {
    unsigned short* obase = reinterpret_cast<unsigned short*>(P.o) + bh_offset(b, P.os_b, h, P.os_h);
    const int lr = lane_e >> 3, lc = lane_e & 7;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 8 + lr;
      const int grow = q0 + row;
      const u4 w = *reinterpret_cast<const u4*>(ebase + row * ROWB + ((lc ^ (row & 7)) << 4));
      if (grow < Sq) store_wt16(obase + (long)grow * P.os_s + lc * 8, w);
    }
  }
}

int launch_attn_wide(const AttnParams& P, int in, hipStream_t st) {
  const unsigned grid = (unsigned)(P.nQT * P.nBHpad);
  if (in == IN_BF16) hipLaunchKernelGGL((oeh_attn_wide_kernel<IN_BF16>), dim3(grid), dim3(256), 0, st, P);
  else hipLaunchKernelGGL((oeh_attn_wide_kernel<IN_F16>), dim3(grid), dim3(256), 0, st, P);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace oeh
