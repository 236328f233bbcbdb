// Shape-generic attention kernel (any D, any Sk that fits LDS, fp32 FMA arithmetic): one 128-thread
// workgroup per (batch, head, query row).  It exists so that EVERY shape the reference accepts has a HIP
// path (STanHop's tiny (L,S<=64, E=16) problems hopfield.py:42-51, ViT head dims, Sk > 512); the MFMA
// kernel in oeh_attn_mfma.hip is the fast path for D in {32,64,128}, Sk <= 512.  Same op order as the
// reference chain; the score row lives in LDS as fp32.
#include "oeh_attn_params.h"

namespace oeh {

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = __builtin_fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += red[i];
  return r;
}

template <int IN>
__global__ __launch_bounds__(128) void oeh_attn_generic_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float* qrow_f = dyn;             // D
  float* sc = dyn + P.D;           // Sk
  __shared__ float red[4];
  typedef typename In<IN>::elem E;

  const long bid = blockIdx.x;
  const int qi = (int)(bid % P.Sq);
  const int h = (int)((bid / P.Sq) % P.H);
  const int b = (int)(bid / ((long)P.Sq * P.H));
  const int tid = threadIdx.x, nth = blockDim.x;
  const int off = P.Sk - P.Sq;

  const E* qp = reinterpret_cast<const E*>(P.q) + (long)b * P.qs_b + (long)h * P.qs_h + (long)qi * P.qs_s;
  for (int d = tid; d < P.D; d += nth) qrow_f[d] = In<IN>::to_f32(qp[d]);
  __syncthreads();

  float lmax = -__builtin_inff();
  for (int j = tid; j < P.Sk; j += nth) {
    const E* kp = reinterpret_cast<const E*>(P.k) + (long)b * P.ks_b + (long)h * P.ks_h + (long)j * P.ks_s;
    float acc = 0.0f;
    for (int d = 0; d < P.D; ++d) acc = __builtin_fmaf(qrow_f[d], In<IN>::to_f32(kp[d]), acc);
    float x = (P.scale_div != 0.0f) ? acc / P.scale_div : acc * P.scale;
    if (P.fq_s.en) {
      const float idx = fq_index(x, P.fq_s);
      if (P.fq_s.dump) P.fq_s.dump[(((long)b * P.H + h) * P.Sq + qi) * P.Sk + j] = (unsigned char)idx;
      x = fq_dequant(idx, P.fq_s);
    }
    if (P.pad) x = x + load_mask(P.pad, P.pad_f16, (long)b * P.pad_sb + j);
    if (P.full) x = x + load_mask(P.full, P.full_f16, (long)b * P.full_sb + (long)qi * P.full_sq + j);
    if (P.causal && j > qi + off) x = x + P.mask_min;
    if (P.clamp_min) x = __builtin_fmaxf(x, P.mask_min);
    sc[j] = x;
    lmax = __builtin_fmaxf(lmax, x);
  }
  const float m = block_reduce_max(lmax, red);
  float lsum = 0.0f;
  for (int j = tid; j < P.Sk; j += nth) {
    const float e = exp_acc(sc[j] - m);
    sc[j] = e;
    lsum += e;
  }
  const float sum = block_reduce_sum(lsum, red);
  float den = sum;
  if (P.base != 0) den = sum + exp_acc(m * -1.0f);
  for (int j = tid; j < P.Sk; j += nth) {
    float p = sc[j] / den;
    if (P.clip) {
      p = p * P.clip_w;
      p = p + P.clip_g;
      p = __builtin_fminf(__builtin_fmaxf(p, 0.0f), 1.0f);
    }
    if (P.fq_p.en) {
      const float idx = fq_index(p, P.fq_p);
      if (P.fq_p.dump) P.fq_p.dump[(((long)b * P.H + h) * P.Sq + qi) * P.Sk + j] = (unsigned char)idx;
      p = fq_dequant(idx, P.fq_p);
    }
    sc[j] = p;
  }
  __syncthreads();
  float gatev = 1.0f;
  if (P.gate) gatev = P.gate[(long)b * P.gs_b + (long)h * P.gs_h + (long)qi * P.gs_s];
  E* op = reinterpret_cast<E*>(P.o) + (long)b * P.os_b + (long)h * P.os_h + (long)qi * P.os_s;
  for (int d = tid; d < P.D; d += nth) {
    const E* vp = reinterpret_cast<const E*>(P.v) + (long)b * P.vs_b + (long)h * P.vs_h + d;
    float acc = 0.0f;
    for (int j = 0; j < P.Sk; ++j) acc = __builtin_fmaf(sc[j], In<IN>::to_f32(vp[(long)j * P.vs_s]), acc);
    float x = acc;
    if (P.fq_c.en && P.ctx_before_gate) {
      const float idx = fq_index(x, P.fq_c);
      if (P.fq_c.dump) P.fq_c.dump[(((long)b * P.H + h) * P.Sq + qi) * P.D + d] = (unsigned char)idx;
      x = fq_out(idx, P.fq_c);
    }
    if (P.gate) x = x * gatev;
    if (P.fq_c.en && !P.ctx_before_gate) {
      const float idx = fq_index(x, P.fq_c);
      if (P.fq_c.dump) P.fq_c.dump[(((long)b * P.H + h) * P.Sq + qi) * P.D + d] = (unsigned char)idx;
      x = fq_out(idx, P.fq_c);
    }
    op[d] = In<IN>::from_f32(x);
  }
}

int launch_attn_generic(const AttnParams& P, int in, hipStream_t st) {
  const size_t shmem = (size_t)(P.D + P.Sk) * sizeof(float);
  if (shmem > 64 * 1024) return -95;  // default dynamic-LDS limit; Sk up to ~16k
  const long nblk = (long)P.B * P.H * P.Sq;
  if (nblk > 0x7fffffffL) return -95;
  switch (in) {
    case IN_F16: hipLaunchKernelGGL(oeh_attn_generic_kernel<IN_F16>, dim3((unsigned)nblk), dim3(128), shmem, st, P); break;
    case IN_BF16: hipLaunchKernelGGL(oeh_attn_generic_kernel<IN_BF16>, dim3((unsigned)nblk), dim3(128), shmem, st, P); break;
    default: hipLaunchKernelGGL(oeh_attn_generic_kernel<IN_F32>, dim3((unsigned)nblk), dim3(128), shmem, st, P); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace oeh
