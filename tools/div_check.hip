// Checks on the GPU that the three-instruction quotient  q0 = x*y; r = fma(-q0, s, x); q = fma(r, y, q0)  with
// y = RN(1/s) equals the correctly rounded IEEE quotient x/s bit for bit (Markstein's correction step), on
//   (1) every float within +-4 ulp of h*s for every half-integer |h| <= 520 and NS scales (the only place where the
//       quantiser index rint(x/s) can change), and
//   (2) NR random (x, s) pairs with |x/s| <= 600.
// build: hipcc -O2 --offload-arch=gfx950 tools/div_check.hip -o tools/div_check ; run on the GPU box: tools/div_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ inline uint32_t rng(uint64_t& st) {
  st = st * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(st >> 32);
}
__device__ inline float scale_from(uint32_t u) {  // log-uniform in [2^-20, 2^4), any mantissa
  const uint32_t e = 107u + (u >> 23) % 24u;
  return __uint_as_float((e << 23) | (u & 0x7fffffu));
}
__device__ unsigned long long g_q0_differs;  // sanity: the uncorrected product must differ now and then
__device__ inline bool same(float x, float s, float y) {
  const float q0 = x * y;
  if (__float_as_uint(q0) != __float_as_uint(x / s)) atomicAdd(&g_q0_differs, 1ull);
  const float r = __builtin_fmaf(-q0, s, x);
  const float q = __builtin_fmaf(r, y, q0);
  const float d = x / s;  // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
  return __float_as_uint(q) == __float_as_uint(d);
}

__global__ void check_half_integers(unsigned long long* bad, float* bad_x, float* bad_s, int ns, int all_ones) {
  const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= ns) return;
  uint64_t st = 0x9E3779B97F4A7C15ull * (sidx + 1);
  float s = scale_from(rng(st));
  if (all_ones) s = __uint_as_float((__float_as_uint(s) & 0xff800000u) | (0x7fffffu >> (sidx % 3)));  // mantissa 1...1
  const float y = 1.0f / s;
  for (int k = -1041; k <= 1041; k += 2) {
    const float h = 0.5f * (float)k;
    const float c = h * s;
    const uint32_t cb = __float_as_uint(c);
    for (int d = -4; d <= 4; ++d) {
      const float x = __uint_as_float(cb + d);
      if (!same(x, s, y)) {
        const unsigned long long n = atomicAdd(bad, 1ull);
        if (n < 16) { bad_x[n] = x; bad_s[n] = s; }
      }
    }
  }
}

__global__ void check_random(unsigned long long* bad, float* bad_x, float* bad_s, int per_thread) {
  uint64_t st = 0xD1B54A32D192ED03ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
  for (int i = 0; i < per_thread; ++i) {
    const float s = scale_from(rng(st));
    const float y = 1.0f / s;
    const float qt = ((float)(int32_t)rng(st)) * (600.0f / 2147483648.0f);
    const float x = __uint_as_float(__float_as_uint(qt * s) + (rng(st) & 7u));
    if (!same(x, s, y)) {
      const unsigned long long n = atomicAdd(bad, 1ull);
      if (n < 16) { bad_x[n] = x; bad_s[n] = s; }
    }
  }
}

int main() {
  unsigned long long* bad; float *bx, *bs;
  hipMallocManaged(&bad, 8); hipMallocManaged(&bx, 64); hipMallocManaged(&bs, 64);
  for (int mode = 0; mode < 3; ++mode) {
    *bad = 0;
    unsigned long long total;
    if (mode < 2) {
      const int ns = 1 << 20;
      check_half_integers<<<ns / 256, 256>>>(bad, bx, bs, ns, mode);
      total = (unsigned long long)ns * 1042ull * 9ull;
    } else {
      check_random<<<4096, 256>>>(bad, bx, bs, 8192);
      total = 4096ull * 256ull * 8192ull;
    }
    hipDeviceSynchronize();
    printf("%s: %llu checks, %llu differ\n", mode == 0 ? "half-integer neighbourhoods" : mode == 1 ? "half-integer neighbourhoods, all-ones mantissas" : "random", total, *bad);
    for (unsigned long long i = 0; i < (*bad < 16 ? *bad : 16); ++i) printf("   x=%a s=%a\n", bx[i], bs[i]);
  }
  unsigned long long q0d = 0;
  hipMemcpyFromSymbol(&q0d, HIP_SYMBOL(g_q0_differs), 8);
  printf("(sanity: the uncorrected product x*RN(1/s) differed from x/s in %llu of those checks)\n", q0d);
  return 0;
}
