#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2{a, b}), h2)); }
__global__ void k(const f4* in, u4* out) {
  f4 s[2];
  s[0] = in[threadIdx.x]; s[1] = in[threadIdx.x + 64];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const unsigned lo = pk(s[t][0], s[t][1]);
    const unsigned hi = pk(s[t][2], s[t][3]);
    s[t][0] = __builtin_bit_cast(float, lo);
    s[t][1] = __builtin_bit_cast(float, hi);
  }
  out[threadIdx.x] = u4{__builtin_bit_cast(unsigned, s[0][0]), __builtin_bit_cast(unsigned, s[0][1]), __builtin_bit_cast(unsigned, s[1][0]), __builtin_bit_cast(unsigned, s[1][1])};
}
