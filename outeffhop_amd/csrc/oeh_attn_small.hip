// Small-shape attention: MANY tiny problems, ONE WAVE per (batch, head) - the STanHop / theory_verification Hopfield
// `Association` (STanHop_time_seeries/cross_models/hopfield.py:42-51: L, S ~ 28 time segments or data dimensions, H = 4,
// E in {16, 32, 64}, batch = B * data_dim, fp32 models; theory_verification/layers.py:107-123).
//
// At these sizes the 64-row workgroups of the big kernels are mostly padding and the any-shape kernel spends one workgroup
// per query ROW.  Here a wave keeps its problem's whole K and V in registers, already in matrix-core operand layout,
// loaded straight from global memory (no LDS, no barrier, nothing shared between the four waves of a workgroup), and walks
// the query rows 16 at a time:
//   * products on v_mfma_f32_16x16x4_f32 - fp32 operands, exactly an fmaf chain: the reference's fp32 arithmetic, no operand
//     rounding at all.  It runs at 1/16 of the fp16 rate, which is irrelevant here: a 28 x 28 x 64 problem is 128 MFMAs.
//   * the swapped orientation of the big kernels: S^T = K Q^T puts the query on the lane (col = lane & 15) and the keys
//     of a 16-key tile on (lane >> 4, register r), so the softmax row statistics are in-lane plus two cross-lane steps and
//     P^T - the lane's own four registers - IS the B operand of O^T = V^T P^T for the four k-steps r = 0..3, with
//     A = V[16 t + 4 g + r][16 dt + c] a coalesced 64-byte row segment per 16 lanes.  The k index of an MFMA step is a
//     free relabelling as long as both operands agree, which is what makes these direct loads line up.
//   * the reference's op order on the elements: scale (multiply or true division), max, 1-ulp exp, sum, [+exp(-m)], divide,
//     [clip]; all four Association modes (softmax, softmax1, clip, clip_softmax1) and an optional gate.
// Masks, fake-quant and the in-kernel gate predictor are not part of this path (oeh_api.hip: small_eligible).
#include "oeh_attn_params.h"

namespace oeh {

template <int IN>
__device__ __forceinline__ f4 load4_f32(const void* base, long elem_off) {
  if constexpr (IN == IN_F32) {
    return *reinterpret_cast<const f4*>(reinterpret_cast<const float*>(base) + elem_off);
  } else {
    const u2 w = *reinterpret_cast<const u2*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
    return f4{In<IN>::to_f32((unsigned short)(w.x & 0xffffu)), In<IN>::to_f32((unsigned short)(w.x >> 16)),
              In<IN>::to_f32((unsigned short)(w.y & 0xffffu)), In<IN>::to_f32((unsigned short)(w.y >> 16))};
  }
}
template <int IN>
__device__ __forceinline__ float load1_f32(const void* base, long elem_off) {
  return In<IN>::to_f32(reinterpret_cast<const typename In<IN>::elem*>(base)[elem_off]);
}

// ET = D / 16 (1, 2, 4); ST = 16-key tiles held (2: Sk <= 32, 4: Sk <= 64)
template <int ET, int ST, int IN>
__global__ __launch_bounds__(256, 2) void oeh_attn_small_kernel(const AttnParams P) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // P.nQT query blocks of 16 rows per wave: all of them (one wave per problem), or ONE (P.nQT == 1: a wave per problem and
  // 16-row block, the waves of a problem next to each other in a workgroup so that their K / V loads meet in the CU's cache) -
  // chosen by the launcher when whole problems would leave most SIMDs without a wave
  // (the arguments of the prologue in one round of scalar loads, no integer divisions: oeh_attn_fast.inl / oeh_common.h: div_magic;
  // magic_nbh carries floor(2^32 / waves-per-problem) here, set by the launcher)
  asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.o), "s"(P.nQT), "s"(P.nBH), "s"(P.H), "s"(P.Sq), "s"(P.Sk), "s"(P.magic_nbh), "s"(P.magic_h),
               "s"(P.qs_b), "s"(P.qs_h), "s"(P.qs_s), "s"(P.ks_b), "s"(P.ks_h), "s"(P.ks_s), "s"(P.vs_b), "s"(P.vs_h), "s"(P.vs_s));
  const int nqb = (P.Sq + 15) >> 4;
  const int per = (P.nQT == 1) ? nqb : 1;       // waves per problem
  const int wg = blockIdx.x * 4 + wave;
  int bh, wrem;
  div_magic((unsigned)wg, (unsigned)per, P.magic_nbh, bh, wrem);
  if (bh >= P.nBH) return;
  const int qb0 = wrem * 16;
  const int q_end = (P.nQT == 1) ? min(P.Sq, qb0 + 16) : P.Sq;
  int b, h;
  div_magic((unsigned)bh, (unsigned)P.H, P.magic_h, b, h);
  const int c = lane & 15, g = lane >> 4;
  const int Sk = P.Sk, Sq = P.Sq;

  // ---- the problem's K (A operand of S^T = K Q^T: row = key 16t + c, k-slot g <-> elements 16j + 4g + i) and
  //      V (A operand of O^T = V^T P^T: row = d 16dt + c, k-slot g <-> key 16t + 4g + r), once, in registers
  f4 kf[ST][ET], vf[ST][ET];
  {
    const long kb = bh_offset(b, P.ks_b, h, P.ks_h), vb = bh_offset(b, P.vs_b, h, P.vs_h);
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      const int key = min(16 * t + c, Sk - 1);  // rows past Sk: finite data, masked below
#pragma unroll
      for (int j = 0; j < ET; ++j) kf[t][j] = load4_f32<IN>(P.k, kb + (long)key * P.ks_s + 16 * j + 4 * g);
#pragma unroll
      for (int dt = 0; dt < ET; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vf[t][dt][r] = load1_f32<IN>(P.v, vb + (long)min(16 * t + 4 * g + r, Sk - 1) * P.vs_s + 16 * dt + c);
    }
  }
  const bool use_div = P.scale_div != 0.0f;

  for (int q0 = qb0; q0 < q_end; q0 += 16) {
    const int qrow = q0 + c;
    const long qoff = bh_offset(b, P.qs_b, h, P.qs_h) + (long)min(qrow, Sq - 1) * P.qs_s + 4 * g;
    f4 qf[ET];
#pragma unroll
    for (int j = 0; j < ET; ++j) qf[j] = load4_f32<IN>(P.q, qoff + 16 * j);
    // ---- scores: lane (c, g) gets S[q0 + c][16t + 4g + r]
    f4 s[ST];
    float m = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < ET; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t][j][i], qf[j][i], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = use_div ? acc[r] / P.scale_div : acc[r] * P.scale;
        if (16 * t + 4 * g + r >= Sk) x = -__builtin_inff();
        acc[r] = x;
      }
      s[t] = acc;
      m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(acc[0], acc[1])), __builtin_fmaxf(acc[2], acc[3]));
    }
    m = row4_max(m);
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = exp_acc(s[t][r] - m);  // keys past Sk: exp(-inf) = 0
        s[t][r] = e;
        sum += e;
      }
    sum = row4_sum(sum);
    float den = sum;
    if (P.base != 0) den = sum + exp_acc(m * -1.0f);  // softmax_1: + 1*exp(-max)  (softmax_1.py:18-20)
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = s[t][r] / den;
        if (P.clip) {
          p = p * P.clip_w;
          p = p + P.clip_g;
          p = __builtin_fminf(__builtin_fmaxf(p, 0.0f), 1.0f);
          if (16 * t + 4 * g + r >= Sk) p = 0.0f;  // (gamma > 0 would lift a padded key off zero)
        }
        s[t][r] = p;
      }
    // ---- context: lane (c, g) gets O[q0 + c][16dt + 4g + r]
    float gatev = 1.0f;
    if (P.gate != nullptr && qrow < Sq) gatev = P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qrow * P.gs_s];
#pragma unroll
    for (int dt = 0; dt < ET; ++dt) {
      f4 o = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[t][dt][r], s[t][r], o, 0, 0, 0);
      if (P.gate != nullptr) o = o * gatev;
      if (qrow < Sq) {
        const long ooff = bh_offset(b, P.os_b, h, P.os_h) + (long)qrow * P.os_s + 16 * dt + 4 * g;
        if constexpr (IN == IN_F32) {
          *reinterpret_cast<f4*>(reinterpret_cast<float*>(P.o) + ooff) = o;
        } else {
          u2 w;
          if constexpr (IN == IN_BF16) { w.x = pack2_bf16(o[0], o[1]); w.y = pack2_bf16(o[2], o[3]); }
          else { w.x = pack2_f16(o[0], o[1]); w.y = pack2_f16(o[2], o[3]); }
          *reinterpret_cast<u2*>(reinterpret_cast<unsigned short*>(P.o) + ooff) = w;
        }
      }
    }
  }
}

template <int ET, int ST>
static int launch_small_et_st(const AttnParams& P, int in, hipStream_t st) {
  // fewer than two waves per SIMD of the device with one wave per problem: a wave per 16-row block instead (STanHop: 896
  // problems of 28 rows on 1024 SIMDs -> 1792 waves; 12.1 -> see DESIGN 4.8)
  AttnParams Q = P;
  const long nqb = (P.Sq + 15) / 16;
  Q.nQT = (nqb > 1 && (long)P.nBH < 2048) ? 1 : 0;
  const long waves = Q.nQT == 1 ? (long)P.nBH * nqb : (long)P.nBH;
  Q.magic_nbh = Q.nQT == 1 ? (unsigned)(0x100000000ULL / (unsigned long long)nqb) : 0xffffffffu;  // floor(2^32 / waves per problem)
  const unsigned grid = (unsigned)((waves + 3) / 4);
  switch (in) {
    case IN_F16: hipLaunchKernelGGL((oeh_attn_small_kernel<ET, ST, IN_F16>), dim3(grid), dim3(256), 0, st, Q); break;
    case IN_BF16: hipLaunchKernelGGL((oeh_attn_small_kernel<ET, ST, IN_BF16>), dim3(grid), dim3(256), 0, st, Q); break;
    default: hipLaunchKernelGGL((oeh_attn_small_kernel<ET, ST, IN_F32>), dim3(grid), dim3(256), 0, st, Q); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

template <int ET>
static int launch_small_et(const AttnParams& P, int in, hipStream_t st) {
  return P.Sk <= 32 ? launch_small_et_st<ET, 2>(P, in, st) : launch_small_et_st<ET, 4>(P, in, st);
}

// D in {16, 32, 64}, Sk <= 64 (oeh_api.hip: small_eligible)
int launch_attn_small(const AttnParams& P, int in, hipStream_t st) {
  switch (P.D) {
    case 16: return launch_small_et<1>(P, in, st);
    case 32: return launch_small_et<2>(P, in, st);
    case 64: return launch_small_et<4>(P, in, st);
    default: return -95;
  }
}

}  // namespace oeh
