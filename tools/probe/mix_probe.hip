// Hardware probe behind oeh_common.h: split8_mix (round 5).  (1) split8_mix == split8 bit for bit on random fp32 values, on values around the fp16
// range's end with MODE.FP16_OVFL set (the clamp), on tiny values (fp16 subnormals) and on special values.  (2) Does v_mfma_f32_16x16x32_f16 keep
// fp16 SUBNORMAL operands (an unscaled residual x - hi would be one for |x| < 0.25)?
//   /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probe/mix_probe.hip -o tools/probe/mix_probe && tools/probe/mix_probe
#include "../../outeffhop_amd/csrc/oeh_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace oeh;

__global__ void k_split_scaled(const float* x, unsigned* out, float k, int clampmode) {   // split8_raw_scaled against its definition in plain C
  if (clampmode) fp16_overflow_clamp();
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  f4 a = *reinterpret_cast<const f4*>(x + t * 8), b = *reinterpret_cast<const f4*>(x + t * 8 + 4);
  u4 hi, lo;
  split8_raw_scaled(a, b, k, hi, lo);
  const float xs[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  for (int i = 0; i < 4; ++i) {
    const h2 h = sat_h2(xs[2 * i] * k, xs[2 * i + 1] * k);                         // RN16(k x) (the product by a power of two is exact)
    const h2 l = sat_h2(__builtin_fmaf(xs[2 * i], k, -(float)h[0]), __builtin_fmaf(xs[2 * i + 1], k, -(float)h[1]));   // RN16(k x - hi): the fma is exact
    out[t * 16 + i] = hi[i]; out[t * 16 + 4 + i] = lo[i]; out[t * 16 + 8 + i] = __builtin_bit_cast(unsigned, h); out[t * 16 + 12 + i] = __builtin_bit_cast(unsigned, l);
  }
}

__global__ void k_split(const float* x, unsigned* out, float k2048, int clampmode) {
  if (clampmode) fp16_overflow_clamp();
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  f4 a = *reinterpret_cast<const f4*>(x + t * 8), b = *reinterpret_cast<const f4*>(x + t * 8 + 4);
  u4 hi, lo, hi2, lo2;
  split8_ref(a, b, hi, lo);
  split8_mix(a, b, k2048, hi2, lo2);
  for (int i = 0; i < 4; ++i) {
    out[t * 16 + i] = hi[i]; out[t * 16 + 4 + i] = lo[i]; out[t * 16 + 8 + i] = hi2[i]; out[t * 16 + 12 + i] = lo2[i];
  }
}

typedef _Float16 h8t __attribute__((ext_vector_type(8)));
__global__ void k_mfma(float* out, unsigned short abits, unsigned short bbits) {
  // A[row][k] = a for every element, B[k][col] = b: C = 32 a b in every element
  h8t a, b;
  const _Float16 av = __builtin_bit_cast(_Float16, abits), bv = __builtin_bit_cast(_Float16, bbits);
  for (int i = 0; i < 8; ++i) { a[i] = av; b[i] = bv; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

int main() {
  const long n = 1 << 22;  // floats
  std::vector<float> h(n);
  srand(1);
  for (long i = 0; i < n; ++i) {
    const int kind = i % 8;
    float v = ((float)rand() / RAND_MAX * 2.0f - 1.0f);
    if (kind == 0) v *= 1e-6f; else if (kind == 1) v *= 1e-3f; else if (kind == 2) v *= 100.0f; else if (kind == 3) v *= 70000.0f; else if (kind == 4) v *= 6e-5f; else if (kind == 5) v *= 4.0f;
    h[i] = v;
  }
  h[0] = 0.0f; h[1] = -0.0f; h[2] = 65504.0f; h[3] = 65519.9f; h[4] = 65520.0f; h[5] = 1e30f; h[6] = -1e30f; h[7] = 5.96e-8f;
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 2 * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> o(n * 2);
  for (int clampmode = 0; clampmode < 2; ++clampmode) {
    hipLaunchKernelGGL(k_split, dim3(n / 8 / 256), dim3(256), 0, 0, dx, dout, 2048.0f, clampmode);
    hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
    long bad_hi = 0, bad_lo = 0, bad_lo_in_range = 0; long first = -1;
    for (long t = 0; t < n / 8; ++t)
      for (int i = 0; i < 4; ++i) {
        if (o[t * 16 + i] != o[t * 16 + 8 + i]) ++bad_hi;
        if (o[t * 16 + 4 + i] != o[t * 16 + 12 + i]) {
          ++bad_lo;
          const float x0 = h[t * 8 + 2 * i], x1 = h[t * 8 + 2 * i + 1];
          if (fabsf(x0) <= 65504.0f && fabsf(x1) <= 65504.0f) { ++bad_lo_in_range; if (first < 0) first = t * 8 + 2 * i; }
        }
      }
    printf("FP16_OVFL=%d: %ld register pairs compared; hi differs %ld, lo differs %ld (of those with both values inside the fp16 range: %ld)\n", clampmode, n / 2, bad_hi, bad_lo, bad_lo_in_range);
    if (first >= 0) printf("  first in-range difference at x = %.9g, %.9g\n", h[first], h[first + 1]);
  }
  for (int clampmode = 0; clampmode < 2; ++clampmode) {
    hipLaunchKernelGGL(k_split_scaled, dim3(n / 8 / 256), dim3(256), 0, 0, dx, dout, 32.0f, clampmode);
    hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
    long bad_hi = 0, bad_lo = 0, bad_in = 0;
    for (long t = 0; t < n / 8; ++t)
      for (int i = 0; i < 4; ++i) {
        const bool inr = fabsf(h[t * 8 + 2 * i]) <= 1000.0f && fabsf(h[t * 8 + 2 * i + 1]) <= 1000.0f;
        // (a zero's sign is not compared: fma(-0, k, +0) = +0 where the plain product keeps -0)
        auto half_differs = [](unsigned p_, unsigned q_) { return (p_ & 0xffffu) != (q_ & 0xffffu) && ((p_ | q_) & 0x7fffu) != 0; };
        auto differ = [&](unsigned p_, unsigned q_) { return half_differs(p_, q_) || half_differs(p_ >> 16, q_ >> 16); };
        if (differ(o[t * 16 + i], o[t * 16 + 8 + i])) { ++bad_hi; if (inr) ++bad_in; }
        if (differ(o[t * 16 + 4 + i], o[t * 16 + 12 + i])) { ++bad_lo; if (inr) ++bad_in; }
      }
    printf("split8_raw_scaled(k = 32), FP16_OVFL=%d: hi differs %ld, lo differs %ld of %ld register pairs (with both |x| <= 1000: %ld)\n", clampmode, bad_hi, bad_lo, n / 2, bad_in);
  }
  float* dc; hipMalloc(&dc, 4);
  struct { unsigned short a, b; const char* what; } cases[] = {
    {0x0001, 0x3C00, "a = 2^-24 (smallest fp16 subnormal), b = 1: exact 32 * 2^-24 = 1.9073486e-06"},
    {0x03FF, 0x3C00, "a = largest subnormal (1023 * 2^-24), b = 1: exact 32 * 1023 * 2^-24 = 1.9512177e-03"},
    {0x0001, 0x0001, "a = b = 2^-24: exact 32 * 2^-48 = 1.1368684e-13"},
    {0x0400, 0x3C00, "a = 2^-14 (smallest normal), b = 1: 32 * 2^-14 = 1.953125e-03"},
  };
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dc, cs.a, cs.b);
    float r; hipMemcpy(&r, dc, 4, hipMemcpyDeviceToHost);
    printf("mfma_f32_16x16x32_f16 %s -> %.8g\n", cs.what, r);
  }
  return 0;
}
