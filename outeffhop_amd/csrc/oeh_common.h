// Shared device helpers for liboeh_hip.so (gfx950 only).  Built with -ffp-contract=off: every
// fp32 multiply/add below is a separately rounded IEEE op exactly as in the reference's eager
// torch chain; fused multiply-adds appear only where written explicitly (__builtin_fmaf).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oeh {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

enum { IN_F16 = 0, IN_BF16 = 1, IN_F32 = 2 };

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLog2eHi = 0x1.715476p+0f;   // fl32(log2 e)
constexpr float kLog2eLo = 0x1.4ae0bep-26f;  // log2 e - kLog2eHi

// One fixed-range asymmetric fake-quantiser (uniform_quantizers.py:72-82,114-115,146).
struct FqP {
  int en;
  float scale, rscale, zp, qmax;  // rscale = RN(1/scale), formed by the host (oeh_api.hip: make_fq)
  float lo, hi;                   // -zp and qmax - zp: the grid relative to the zero point
  float c2;                       // RN(scale * log2(e)), formed in double by the host: exp(scale * d) = exp2(d * c2) for grid differences d
  float oscale;                   // what a quantised value is written out as: oscale * (idx - zp); = scale, or 1 for the context quantiser
                                  // with oeh_fq_desc.ctx_emit_index (the integers themselves: include/oeh.h)
  unsigned char* dump;
};

// x / scale, correctly rounded, in three instructions: the product with the correctly rounded reciprocal is within
// 1.5 ulp of the quotient, its residual r = x - q0*scale is then exact in one fma, and q0 + r*rscale rounds to RN(x/scale)
// (Markstein's correction step).  The reference divides (uniform_quantizers.py:114), and rint() of a quotient that is one
// ulp off flips the index next to a half-integer, so the quotient has to be the IEEE one: tools/div_check.hip compares
// this sequence with the hardware division on every float within 4 ulp of every half-integer |h| <= 520 for 2^20 scales
// (incl. all-ones mantissas) and on 8.6e9 random pairs - 0 differences in 2.8e10.  (The first version multiplied and fell
// back to a real division inside a guard band around the half-integers: one more instruction per element, a wave-uniform
// branch per four, and 1.1 k instructions of division code per kernel; INT8 OPT shape 46.3 -> 40 us without it.)
__device__ __forceinline__ float fq_quot(float x, const FqP& f) {
  const float q0 = x * f.rscale;
  const float r = __builtin_fmaf(-q0, f.scale, x);
  return __builtin_fmaf(r, f.rscale, q0);
}
// Index relative to the zero point, idx - zp = clamp(rint(x/scale), -zp, qmax - zp): the same integer as
// clamp(rint(x/scale) + zp, 0, qmax) - zp (all operands are integers far below 2^24, or the clamp saturates either way)
// for two instructions less per element; x_q = scale * rel, and rel itself is the integer the P operand carries.
__device__ __forceinline__ float fq_rel(float x, const FqP& f) { return __builtin_amdgcn_fmed3f(__builtin_rintf(fq_quot(x, f)), f.lo, f.hi); }
// (A packed form of the quotient - v_pk_mul_f32 / v_pk_fma_f32 on element pairs - measured much slower in the full-row
// kernel: 55 vs 39 us on the INT8 OPT shape, 25.5 vs 13.9 us on BERT-base.  Packed fp32 buys no throughput on this chip
// anyway: tools/pk_bench.hip times 16 scalar VALU operations at 37.7 cycles per wave and the same arithmetic as 8 packed
// ones at 37.1 - a wave64 fp32 operation issues in ~2.3 cycles, a packed one in twice that.)
// The grid chain's score index (oeh_attn_fast.inl FQ == 1, oeh_attn_i8.hip, the two-pass one-pass form, and the general kernel
// that serves their index dumps - one formula, so that all of them give the same bits): idx - zp = clamp(rint(s k1)) by ONE
// fused multiply-add against M = 1.5 * 2^23.  RN(s k1 + M) = M + (s k1 rounded to an integer, ties to even, from the EXACT
// product) while |s k1| < 2^22; beyond that the sum is still on the same side of the clamp bounds M + lo, M + hi.  The result
// is M + rel: exactly representable, differences of two such values are the exact integer differences.
constexpr float kGridMagic = 12582912.0f;
__device__ __forceinline__ float grid_rel_m(float s, float k1, float lo_m, float hi_m) {
  return __builtin_amdgcn_fmed3f(__builtin_fmaf(s, k1, kGridMagic), lo_m, hi_m);
}
// The epilogue chain [context quantiser] gate [context quantiser] over a lane's N context values, in WHOLE passes under
// wave-uniform branches.  Written per element with the three conditions inline the compiler turns them into selects and
// evaluates BOTH quantisers for every element (21 vector instructions per element where 10 do: a quarter of the int8 core's
// vector instructions were this epilogue).  The opaque asm in a pass keeps it a branch.  rel (tests' index dumps): idx - zp.
template <int N, bool WANT_REL = false>
__device__ __forceinline__ void ctx_chain(float (&x)[N], const FqP& fc, const int before_gate, const bool gated, const float gatev, float* rel = nullptr) {
  const bool first = fc.en && (before_gate || !gated), last = fc.en && !before_gate && gated;
  if (first) {
    asm volatile("");
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float r = fq_rel(x[i], fc);
      if constexpr (WANT_REL) rel[i] = r;
      x[i] = fc.oscale * r;
    }
  }
  if (gated) {
    asm volatile("");
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] * gatev;
  }
  if (last) {
    asm volatile("");
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float r = fq_rel(x[i], fc);
      if constexpr (WANT_REL) rel[i] = r;
      x[i] = fc.oscale * r;
    }
  }
}
__device__ __forceinline__ f4 fq_rel4(f4 x, const FqP& f) { return f4{fq_rel(x[0], f), fq_rel(x[1], f), fq_rel(x[2], f), fq_rel(x[3], f)}; }
__device__ __forceinline__ float fq_index(float x, const FqP& f) { return fq_rel(x, f) + f.zp; }
__device__ __forceinline__ float fq_dequant(float idx, const FqP& f) { return f.scale * (idx - f.zp); }
__device__ __forceinline__ float fq_out(float idx, const FqP& f) { return f.oscale * (idx - f.zp); }  // (the context quantiser's written value)
__device__ __forceinline__ unsigned int fq_dump_word(f4 rel, const FqP& f) {  // four uint8 indices (test dumps)
  return (unsigned int)(rel[0] + f.zp) | ((unsigned int)(rel[1] + f.zp) << 8) | ((unsigned int)(rel[2] + f.zp) << 16) | ((unsigned int)(rel[3] + f.zp) << 24);
}

// exp(y) to ~1 ulp: n = rint(y*log2e), f = y*log2e - n in two fma steps, 2^f by v_exp_f32, ldexp.
// Used wherever a value feeds a quantiser (index parity with an IEEE-accurate expf).
__device__ __forceinline__ float exp_acc(float y) {
  y = __builtin_fminf(__builtin_fmaxf(y, -110.0f), 90.0f);
  float t = y * kLog2eHi;
  float n = __builtin_rintf(t);
  float f = __builtin_fmaf(y, kLog2eHi, -n);
  f = __builtin_fmaf(y, kLog2eLo, f);
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
// the same for y <= 0 (softmax arguments): the upper clamp is dead
__device__ __forceinline__ float exp_acc_nonpos(float y) {
  y = __builtin_fmaxf(y, -110.0f);
  float t = y * kLog2eHi;
  float n = __builtin_rintf(t);
  float f = __builtin_fmaf(y, kLog2eHi, -n);
  f = __builtin_fmaf(y, kLog2eLo, f);
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
// ... two at a time (packed multiply / fma for the range reduction; rint, v_exp_f32, ldexp stay per element)
__device__ __forceinline__ f2 exp_acc_nonpos2(f2 y) {
  y = f2{__builtin_fmaxf(y[0], -110.0f), __builtin_fmaxf(y[1], -110.0f)};
  const f2 hi = f2{kLog2eHi, kLog2eHi}, lo = f2{kLog2eLo, kLog2eLo};
  const f2 t = y * hi;
  const f2 n = f2{__builtin_rintf(t[0]), __builtin_rintf(t[1])};
  f2 f = __builtin_elementwise_fma(y, hi, -n);
  f = __builtin_elementwise_fma(y, lo, f);
  return f2{__builtin_ldexpf(__builtin_amdgcn_exp2f(f[0]), (int)n[0]), __builtin_ldexpf(__builtin_amdgcn_exp2f(f[1]), (int)n[1])};
}
// exp(y) by one multiply + v_exp_f32 (relative error ~|y| * 1e-7): the fp16/bf16 path.
__device__ __forceinline__ float exp_fast(float y) { return __builtin_amdgcn_exp2f(y * kLog2e); }

template <int IN>
struct In;  // storage element helpers
template <>
struct In<IN_F16> {
  typedef unsigned short elem;
  static constexpr int bytes = 2;
  static __device__ __forceinline__ float to_f32(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
};
template <>
struct In<IN_BF16> {
  typedef unsigned short elem;
  static constexpr int bytes = 2;
  static __device__ __forceinline__ float to_f32(unsigned short v) { return __builtin_bit_cast(float, (unsigned int)v << 16); }
  static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
};
template <>
struct In<IN_F32> {
  typedef float elem;
  static constexpr int bytes = 4;
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float f) { return f; }
};

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
// one v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 (round to nearest even) per pair
__device__ __forceinline__ unsigned int pack2_f16(float a, float b) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((f2{a, b}), h2));
}
__device__ __forceinline__ unsigned int pack2_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((f2{a, b}), b2));
}

// Bit views of a float BY VALUE.  Never write __builtin_bit_cast(unsigned, vec[i]) on an ext_vector element: clang
// (ROCm 7.2, clang 22) lowers that lvalue form to a load from the vector's BASE address, i.e. element 0 for every i
// (seen as every attention row wrong with row sums of 2*(e0+e1)/den; minimal repro: tools/clang_bitcast_repro.hip).
__device__ __forceinline__ unsigned f32_bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bits_f32(unsigned u) { return __builtin_bit_cast(float, u); }

// 8 consecutive storage elements -> 8 x 16-bit MFMA operand elements (f32 storage is rounded to f16).
template <int IN>
__device__ __forceinline__ u4 load8_as16(const void* base, long elem_off) {
  if constexpr (IN == IN_F32) {
    const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(base) + elem_off);
    f4 a = p[0], b = p[1];
    u4 r;
    r.x = pack2_f16(a.x, a.y);
    r.y = pack2_f16(a.z, a.w);
    r.z = pack2_f16(b.x, b.y);
    r.w = pack2_f16(b.z, b.w);
    return r;
  } else {
    return *reinterpret_cast<const u4*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
  }
}

// fp32 storage on the 16-bit matrix cores WITHOUT giving up fp32 accuracy: every fp32 value x is carried as two fp16
// operands, x = hi + lo * 2^-11 with hi = RN16(x) and lo = RN16((x - hi) * 2^11) (the residual x - hi is exact in fp32,
// the power-of-two scaling keeps lo out of the fp16 subnormals), so |x - (hi + lo 2^-11)| <= 2^-22 |x|, and a product of two
// such values is accumulated as  a.b = ah.bh + 2^-11 (ah.bl + al.bh)  - three MFMAs into two fp32 accumulators, the
// dropped al.bl term being 2^-22 relative.  That is the reference's fp32 `bmm` (quantized_opt.py:151, validate_clm.py runs
// fp32 models) to ~3e-7 relative instead of the 5e-4 of operands rounded to fp16, which flipped 0.5 % of the score
// quantiser's indices.  The pair is exact to 2^-22 for |x| <= 65504 + 32 (q / k / v of a transformer are
// far inside).  Kernels that split call fp16_overflow_clamp() first: with MODE.FP16_OVFL set the conversions SATURATE at the
// fp16 range instead of producing inf, so that a stray larger value costs accuracy (the excess is dropped) and never turns a
// whole row into NaN - at no instruction cost (two v_med3 per element in the load path measured +15 % on the fp32 kernels).
// OEH_PAIR_RAW (round 5 experiment -> see profiles/r05_pair_raw_ab.txt): the attention kernels' operand pairs with the UNSCALED residual (split8_raw below) -
// the factor that brings the lo products back is then 1.
#ifdef OEH_PAIR_RAW
constexpr float kSplitUp = 1.0f, kSplitDown = 1.0f;
#else
constexpr float kSplitUp = 2048.0f, kSplitDown = 1.0f / 2048.0f;
#endif
__device__ __forceinline__ void fp16_overflow_clamp() {
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // hwreg(HW_REG_MODE, 23, 1): FP16_OVFL
}
__device__ __forceinline__ h2 sat_h2(float x, float y) { return __builtin_convertvector((f2{x, y}), h2); }
__device__ __forceinline__ void split8_ref(const f4 a, const f4 b, u4& hi, u4& lo) {
  const h2 h0 = sat_h2(a[0], a[1]), h1 = sat_h2(a[2], a[3]);
  const h2 h2_ = sat_h2(b[0], b[1]), h3 = sat_h2(b[2], b[3]);
  hi = u4{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2_), __builtin_bit_cast(unsigned, h3)};
  auto lo2 = [](float x0, _Float16 hh0, float x1, _Float16 hh1) {
    return __builtin_bit_cast(unsigned, sat_h2((x0 - (float)hh0) * kSplitUp, (x1 - (float)hh1) * kSplitUp));
  };
  lo = u4{lo2(a[0], h0[0], a[1], h0[1]), lo2(a[2], h1[0], a[3], h1[1]), lo2(b[0], h2_[0], b[1], h2_[1]), lo2(b[2], h3[0], b[3], h3[1])};
}
// split8 in 20 vector instructions instead of ~37 (round 5): hi by four v_cvt_pk_f16_f32 as above; the residual x - hi by v_fma_mix_f32 reading the
// fp16 half straight out of the packed register (fma(hi, -1, x): exact, no separate conversion), and lo = RN16(residual * 2^11) by
// v_fma_mixlo_f16 / v_fma_mixhi_f16 (the product is exact, ONE rounding, written into its half of the packed register: no multiply, no pack).
// Bit-identical to split8 for |x| inside the fp16 range (tests/test_proj_gpu.py compares the two paths bit for bit).  One asm statement per
// fragment: the compiler pads every asm statement that writes vector registers with an s_nop.  k2048: 2048.0f in a scalar register.
__device__ __forceinline__ void split8_mix(const f4 a, const f4 b, const float k2048, u4& hi, u4& lo) {
  const h2 h0 = sat_h2(a[0], a[1]), h1 = sat_h2(a[2], a[3]);
  const h2 h2_ = sat_h2(b[0], b[1]), h3 = sat_h2(b[2], b[3]);
  hi = u4{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2_), __builtin_bit_cast(unsigned, h3)};
  float d0, d1, d2, d3, d4, d5, d6, d7;
  unsigned l0, l1, l2, l3;
  asm("v_fma_mix_f32 %0, %12, -1.0, %16 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %12, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %2, %13, -1.0, %18 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %13, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %4, %14, -1.0, %20 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %5, %14, -1.0, %21 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %6, %15, -1.0, %22 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %7, %15, -1.0, %23 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %8, %0, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %8, %1, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %9, %2, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %9, %3, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %10, %4, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %10, %5, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %11, %6, %24, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %11, %7, %24, 0 op_sel_hi:[0,0,0]"
      : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
      : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "s"(k2048));
  lo = u4{l0, l1, l2, l3};
}
// The UNSCALED operand pair (round 5): x = hi + lo' with lo' = RN16(x - hi), no 2^11.  tools/probe/mix_probe.hip: v_mfma_f32_16x16x32_f16 takes
// fp16 SUBNORMAL operands exactly (2^-24 * 1 and 2^-24 * 2^-24 come out exact), so the scaling that kept lo out of the subnormals is not needed by
// the matrix core; what it bought is precision of lo itself: unscaled, a residual below 2^-14 (|x| < ~0.25) lands on the subnormal grid of 2^-24,
// i.e. |x - (hi + lo')| <= 2^-25 absolute instead of 2^-22 |x| relative - the same size for |x| ~ 0.1 ... 1, and far below the fp32 accumulation's
// own rounding in a dot product.  It costs ONE instruction per element (v_fma_mixlo/hi_f16: fma(hi, -1, x) rounded once into its half of the packed
// register) and the second product needs no 2^-11 on the other operand and no second accumulator: a.b = ah.bh + (ah.bl' + al'.bh) in one register set.
__device__ __forceinline__ void split8_raw(const f4 a, const f4 b, u4& hi, u4& lo) {
  const h2 h0 = sat_h2(a[0], a[1]), h1 = sat_h2(a[2], a[3]);
  const h2 h2_ = sat_h2(b[0], b[1]), h3 = sat_h2(b[2], b[3]);
  hi = u4{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2_), __builtin_bit_cast(unsigned, h3)};
  unsigned l0, l1, l2, l3;
  asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
      : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  lo = u4{l0, l1, l2, l3};
}
// split8_raw with a power-of-two PRE-SCALE k of the values (k in a scalar register): k x = hi + lo', hi = RN16(k x), lo' = RN16(k x - hi), both by
// v_fma_mixlo/hi_f16 (16 instructions per 8 values, no conversion, no pack).  Why: the unscaled residual is a fp16 subnormal once |k x| < ~2^-3, i.e. the pair's
// RELATIVE precision falls below 2^-22 for small values (2^-15 at |k x| = 2^-10) - harmless in a dot product dominated by O(1) elements, a real loss if a
// whole tensor is tiny.  With k = 64 the 22 bits hold down to |x| = 2^-9 and degrade gracefully below; the range is |x| <= 2 047 (hi saturates at 65 504 under
// MODE.FP16_OVFL and lo' carries the excess up to another 65 504; beyond that the value saturates).  The caller folds 1 / k into its output scale (exact).
__device__ __forceinline__ void split8_raw_scaled(const f4 a, const f4 b, const float k, u4& hi, u4& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  asm("v_fma_mixlo_f16 %0, %8, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %9, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %10, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %11, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %12, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %13, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %14, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %15, %16, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %4, %8, %16, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %4, %9, %16, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %5, %10, %16, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %5, %11, %16, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %6, %12, %16, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %6, %13, %16, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %7, %14, %16, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %7, %15, %16, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "s"(k));
  hi = u4{h0, h1, h2, h3};
  lo = u4{l0, l1, l2, l3};
}
// the documented [hi | lo 2^11] pair (oeh_split_pairs' output format, the projection GEMM's `pairs` = 1 input)
__device__ __forceinline__ void split8_scaled(const f4 a, const f4 b, u4& hi, u4& lo) {
#ifdef OEH_SPLIT_REF   // (the round-4 form, for A/B builds: make alt NAME=ref DEFS=-DOEH_SPLIT_REF)
  split8_ref(a, b, hi, lo);
#else
  split8_mix(a, b, 2048.0f, hi, lo);
#endif
}
// the kernels' own operand pairs (never leave a kernel)
__device__ __forceinline__ void split8(const f4 a, const f4 b, u4& hi, u4& lo) {
#ifdef OEH_PAIR_RAW
  split8_raw(a, b, hi, lo);
#else
  split8_scaled(a, b, hi, lo);
#endif
}
// 8 consecutive fp32 storage elements -> the (hi, lo) operand pair
__device__ __forceinline__ void load8_split(const void* base, long elem_off, u4& hi, u4& lo) {
  const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(base) + elem_off);
  split8(p[0], p[1], hi, lo);
}

// LDS-DMA of 16 B per lane: LDS[lds_addr + lane*16] <- *gsrc (global_load_lds_dwordx4; 1 KiB per wave-instruction).
// Issued through inline asm on purpose: hipcc tracks the __builtin_amdgcn_global_load_lds form as a pending LDS write
// and puts `s_waitcnt vmcnt(0)` in front of every later LDS read, which drains a multi-tile ring right after it was
// issued (measured: SQ_WAIT_ANY 54 % of wave cycles).  Invisible to the compiler, the transfers are ordered only by
// the caller's counted `s_waitcnt vmcnt(N)` + s_barrier, as intended.  M0 (the LDS destination base) is saved and
// restored inside the statement; `lds_addr` must be wave-uniform.
__device__ __forceinline__ void glds16_keep(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
// The same transfer in scalar-base form: address = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit BYTE offset.
// A tile stream then advances the scalar base and keeps the lane offsets constant: no per-tile vector address math and
// half the address registers.  Lane offsets must stay below 4 GiB from the base (checked on the host).
__device__ __forceinline__ void glds16_s_keep(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}
// ... without saving / restoring M0 around the request (two scalar instructions less per request): for kernels whose code the compiler never gives M0
// to (no indirect register indexing, no GWS / message instructions: the projection GEMM's loops) - M0 is declared clobbered
__device__ __forceinline__ void glds16_s_m0(const void* sbase, unsigned voff, unsigned lds_addr) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void glds16_s_nt_keep(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}
// Round 5: the requests WITHOUT the save / restore of M0 (glds16_s_m0 above: M0 declared clobbered, two scalar instructions less per request) in every kernel;
// -DOEH_KEEP_M0 builds keep the guarded forms (A/B: profiles/r05_m0_ab.txt).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_m0(const void* gsrc, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void glds16_s_nt_m0(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
#ifdef OEH_KEEP_M0
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) { glds16_keep(gsrc, lds_addr); }
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) { glds16_s_keep(sbase, voff, lds_addr); }
__device__ __forceinline__ void glds16_s_nt(const void* sbase, unsigned voff, unsigned lds_addr) { glds16_s_nt_keep(sbase, voff, lds_addr); }
#else
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) { glds16_m0(gsrc, lds_addr); }
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) { glds16_s_m0(sbase, voff, lds_addr); }
__device__ __forceinline__ void glds16_s_nt(const void* sbase, unsigned voff, unsigned lds_addr) { glds16_s_nt_m0(sbase, voff, lds_addr); }
#endif
// Output stores are WRITE-THROUGH (sc0 sc1).  A plain store leaves the line dirty in the XCD's L2; the whole output
// (12.6 MB per OPT-125m launch, 1.6 MB per XCD: it all fits) then goes to memory in the end-of-kernel release, after
// the last wave, where nothing overlaps it: measured 18.97 -> 16.2 us per launch on the headline workload (`nt` alone:
// no change).  Written through, the bytes leave while other workgroups still compute.
// The trailing s_nop is part of the instruction: a store of more than 64 bits reads its data registers over several
// cycles and the next VALU write to them needs a wait state that the compiler inserts for its own stores but not after
// inline asm (seen: every fp32-output row corrupted when the scheduler put the next v_cndmask right behind the store).
__device__ __forceinline__ void store_wt16(void* dst, u4 w) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
}
// Workgroup barrier that is ALSO a compiler barrier for memory operations and waits for this wave's own LDS traffic.
// __builtin_amdgcn_s_barrier() is "no memory, has side effects" to LLVM: with the LDS-DMA hidden in inline asm the compiler
// may hoist LDS reads of a freshly landed tile above it (seen as wholesale wrong results after an unrelated scheduling
// change), and - unlike __syncthreads(), which carries a fence - it emits NO `s_waitcnt lgkmcnt(0)` in front of the raw
// s_barrier, so a ds_write issued just before it (padding row, scan results) could still be in flight when another wave
// read the location after the barrier: 1 launch in ~1000 wrong on padded inputs under a concurrent stream (tools stress
// run).  The explicit wait costs nothing in the tile loops: a wave's LDS reads are consumed before it arrives here.
__device__ __forceinline__ void barrier_mem() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// max(a, b, c) as the one instruction it is (inputs are arithmetic results, never signalling NaNs)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Lane maximum of the 16 exponent arguments of one 16 x 64 score block, as ONE statement: behind every single-instruction asm the compiler
// puts an `s_nop 0` (8 per block and tile in the chain form above), and a chain of eight dependent v_max3 pays the dependent-issue latency
// eight times.  Five independent v_max3, then depth 2 more (round 6; max is exact, so the value is the chain's bit for bit).
__device__ __forceinline__ float max16_tree(const f4 (&s)[4]) {
  float r, t1, t2, t3, t4;
  asm("v_max3_f32 %0, %5, %6, %7\n\t"
      "v_max3_f32 %1, %8, %9, %10\n\t"
      "v_max3_f32 %2, %11, %12, %13\n\t"
      "v_max3_f32 %3, %14, %15, %16\n\t"
      "v_max3_f32 %4, %17, %18, %19\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_max3_f32 %3, %3, %4, %20\n\t"
      "v_max_f32_e32 %0, %0, %3"
      : "=&v"(r), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
      : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]), "v"(s[2][0]), "v"(s[2][1]),
        "v"(s[2][2]), "v"(s[2][3]), "v"(s[3][0]), "v"(s[3][1]), "v"(s[3][2]), "v"(s[3][3]));
  return r;
}
// ... of the first 4 n (n = 1, 2, 3) of them (a tile in which only the first n 16-key sub-tiles of the block hold a visible key)
template <int N4>
__device__ __forceinline__ float max_first(const f4 (&s)[4]) {
  static_assert(N4 >= 1 && N4 <= 3, "1..3 sub-tiles");
  float r, t1, t2;
  if constexpr (N4 == 1) {
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32_e32 %0, %0, %4" : "=&v"(r) : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]));
  } else if constexpr (N4 == 2) {
    asm("v_max3_f32 %0, %3, %4, %5\n\t"
        "v_max3_f32 %1, %6, %7, %8\n\t"
        "v_max3_f32 %0, %0, %9, %10\n\t"
        "v_max_f32_e32 %0, %0, %1"
        : "=&v"(r), "=&v"(t1), "=&v"(t2)
        : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]));
  } else {
    asm("v_max3_f32 %0, %3, %4, %5\n\t"
        "v_max3_f32 %1, %6, %7, %8\n\t"
        "v_max3_f32 %2, %9, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %1, %1, %2, %14\n\t"
        "v_max_f32_e32 %0, %0, %1"
        : "=&v"(r), "=&v"(t1), "=&v"(t2)
        : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]), "v"(s[2][0]), "v"(s[2][1]),
          "v"(s[2][2]), "v"(s[2][3]));
  }
  return r;
}

// Workgroup placement.  With every workgroup of a launch resident at once, CU c is given the block ids c, c + 256,
// c + 512 (tools/timeline.py reads HW_ID), so with the causal q tiles in heaviest-first order a quarter of the CUs get
// 8 + 6 + 4 key tiles and a quarter 6 + 4 + 2.  Walking every second row of 256 ids backwards (in units of 8, so that a
// head keeps its XCD and its K/V stay L2 hits for all its q tiles) gives 16 / 14 / 16 / 14 instead.
__device__ __forceinline__ int snake_block_id(int bid, const int nblocks) {
  const int row = bid >> 8;
  if ((row & 1) && ((row + 1) << 8) <= nblocks) {
    const int col = bid & 255;
    bid = (row << 8) | ((31 - (col >> 3)) << 3) | (col & 7);
  }
  return bid;
}

// b * stride_b + h * stride_h in elements, for strides the host has checked to lie in [0, 2^32) (oeh_api.hip: validate): two
// 32 x 32 -> 64-bit products (s_mul_i32 + s_mul_hi_u32 each) instead of two 64 x 64-bit ones with their sign handling - six of
// these sit in front of a kernel's first load.
__device__ __forceinline__ long bh_offset(const int b, const long sb, const int h, const long sh) {
  return (long)((unsigned long)(unsigned)b * (unsigned long)(unsigned)sb + (unsigned long)(unsigned)h * (unsigned long)(unsigned)sh);
}

// n / d and n % d for a wave-uniform n < 2^31 with m = floor(2^32 / d) from the host: the product's high word is the quotient or one
// below it, one correction step makes it exact (an integer division is ~30 scalar instructions and a v_rcp round trip, and the
// kernels do two of them before they can issue their first load).
__device__ __forceinline__ void div_magic(const unsigned n, const unsigned d, const unsigned m, int& q, int& r) {
  unsigned qq = __umulhi(n, m);
  unsigned rr = n - qq * d;
  if (rr >= d) { qq += 1u; rr -= d; }
  q = (int)qq;
  r = (int)rr;
}

// Block id -> (position of the q tile in the head's heaviest-first list, head).  Plain (group = 0): all heads' heaviest q
// tiles first, id = qt_rev * nBHpad + head.  Grouped: the heads in groups of `group` (a multiple of 8: blocks b and b + 8 share
// an XCD, so a head keeps its XCD), heaviest q tiles first INSIDE a group - id = (head / group) * group * nQT + qt_rev * group
// + head % group - so that all q tiles of a head are dispatched within group * nQT ids of each other.
__device__ __forceinline__ void block_to_tile(const int bid, const int nBHpad, const int nQT, const int group, int& qt_rev, int& bh) {
  if (group > 0) {
    const int per = group * nQT;
    const int grp = bid / per, rem = bid - grp * per;
    const int gsz = min(group, nBHpad - grp * group);  // the last group may be smaller
    qt_rev = rem / gsz;
    bh = grp * group + (rem - qt_rev * gsz);
  } else {
    qt_rev = bid / nBHpad;
    bh = bid - qt_rev * nBHpad;
  }
}

// byte offset of an LDS object inside the workgroup's allocation (what M0 / ds_* addresses are made of)
__device__ __forceinline__ unsigned lds_offset(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

// all-reduce over the 4 lanes (c, c+16, c+32, c+48) that hold one query row: v_permlane16_swap + v_permlane32_swap (VALU, no LDS
// crossbar round trip, no lane-index registers as ds_bpermute needs)
__device__ __forceinline__ float row4_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(f32_bits(x), f32_bits(x), false, false);
  x = bits_f32(a[0]) + bits_f32(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(f32_bits(x), f32_bits(x), false, false);
  return bits_f32(b[0]) + bits_f32(b[1]);
}
__device__ __forceinline__ int row4_sum(int x) {
  auto a = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
  x = (int)a[0] + (int)a[1];
  auto b = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  return (int)b[0] + (int)b[1];
}
__device__ __forceinline__ float row4_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(f32_bits(x), f32_bits(x), false, false);
  x = __builtin_fmaxf(bits_f32(a[0]), bits_f32(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(f32_bits(x), f32_bits(x), false, false);
  return __builtin_fmaxf(bits_f32(b[0]), bits_f32(b[1]));
}

__device__ __forceinline__ float load_mask(const void* base, int is_f16, long off) {
  return is_f16 ? (float)reinterpret_cast<const _Float16*>(base)[off] : reinterpret_cast<const float*>(base)[off];
}

}  // namespace oeh
