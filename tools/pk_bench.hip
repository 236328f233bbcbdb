// Micro-benchmark (GPU box): the correctly rounded quotient chain q0 = x*y; r = fma(-q0, s, x); q = fma(r, y, q0) as
// scalar VALU operations against packed fp32 forms (v_pk_mul_f32 / v_pk_fma_f32), to see why the packed form slowed
// the INT8 kernel down.  build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/pk_bench.hip -o tools/pk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s, float y, int iters) {
  float x0 = threadIdx.x * 0.37f + 1.0f, x1 = x0 + 0.5f, x2 = x0 + 1.25f, x3 = x0 + 2.0f;
  const float ns = -s;
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {  // scalar, 4 independent chains
      float q0 = x0 * y, q1 = x1 * y, q2 = x2 * y, q3 = x3 * y;
      float r0 = __builtin_fmaf(-q0, s, x0), r1 = __builtin_fmaf(-q1, s, x1), r2 = __builtin_fmaf(-q2, s, x2), r3 = __builtin_fmaf(-q3, s, x3);
      x0 = __builtin_fmaf(r0, y, q0) + 1.0f; x1 = __builtin_fmaf(r1, y, q1) + 1.0f; x2 = __builtin_fmaf(r2, y, q2) + 1.0f; x3 = __builtin_fmaf(r3, y, q3) + 1.0f;
    } else if constexpr (MODE == 1) {  // packed, neg modifier on the product
      f2 a = f2{x0, x1}, b = f2{x2, x3};
      const f2 yy = f2{y, y}, ss = f2{s, s};
      f2 qa = a * yy, qb = b * yy;
      f2 ra = __builtin_elementwise_fma(-qa, ss, a), rb = __builtin_elementwise_fma(-qb, ss, b);
      a = __builtin_elementwise_fma(ra, yy, qa) + f2{1.0f, 1.0f}; b = __builtin_elementwise_fma(rb, yy, qb) + f2{1.0f, 1.0f};
      x0 = a[0]; x1 = a[1]; x2 = b[0]; x3 = b[1];
    } else {  // packed, negated constant
      f2 a = f2{x0, x1}, b = f2{x2, x3};
      const f2 yy = f2{y, y}, nss = f2{ns, ns};
      f2 qa = a * yy, qb = b * yy;
      f2 ra = __builtin_elementwise_fma(qa, nss, a), rb = __builtin_elementwise_fma(qb, nss, b);
      a = __builtin_elementwise_fma(ra, yy, qa) + f2{1.0f, 1.0f}; b = __builtin_elementwise_fma(rb, yy, qb) + f2{1.0f, 1.0f};
      x0 = a[0]; x1 = a[1]; x2 = b[0]; x3 = b[1];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<4096, 256>>>(out, 0.05f, 20.0f, iters);
      else if (mode == 1) k<1><<<4096, 256>>>(out, 0.05f, 20.0f, iters);
      else k<2><<<4096, 256>>>(out, 0.05f, 20.0f, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per iteration and wave: 4 quotients = 16 scalar VALU (mode 0) or 8 packed (modes 1, 2)
      const double wave_iters = 4096.0 * 4 * iters;  // waves * iterations
      printf("mode %d (%s): %8.3f ms  -> %6.2f cycles per wave-iteration per SIMD at 2.4 GHz\n", mode,
             mode == 0 ? "scalar" : mode == 1 ? "packed, neg modifier" : "packed, negated constant", ms, ms * 1e-3 * 2.4e9 / (wave_iters / 1024.0));
    }
  return 0;
}
