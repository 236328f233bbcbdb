// Stand-alone row / elementwise kernels behind the C ABI: the SOFTMAX_MAPPING callables
// (models/softmax.py:22-64), QuantizedActivation in fixed-range mode (base_quantized_classes.py:182-199),
// the per-head gate predictors (bert_attention.py:301-327) and min/max range statistics
// (range_estimators.py:71-72,96-97).  All HBM-bound: 16-B accesses, one pass where the row fits LDS.
#include "oeh_common.h"

namespace oeh {

// ---------------------------------------------------------------------------------------------------------
// softmax rows: one 256-thread workgroup per row; the row is staged once into LDS as fp32 (cols <= 15872),
// otherwise re-read from HBM in the three passes.
template <int IN, bool STAGED>
__global__ __launch_bounds__(256) void oeh_softmax_rows_kernel(const void* __restrict__ xin, void* yout, long rows, int cols,
                                                              int base, int clip, float clip_w, float clip_g) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  __shared__ float red[4];
  typedef typename In<IN>::elem E;
  const int tid = threadIdx.x;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const E* x = reinterpret_cast<const E*>(xin) + row * cols;
    E* y = reinterpret_cast<E*>(yout) + row * cols;
    float lmax = -__builtin_inff();
    for (int j = tid; j < cols; j += 256) {
      const float v = In<IN>::to_f32(x[j]);
      if (STAGED) rowbuf[j] = v;
      lmax = __builtin_fmaxf(lmax, v);
    }
    for (int o = 32; o > 0; o >>= 1) lmax = __builtin_fmaxf(lmax, __shfl_xor(lmax, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float m = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
    float lsum = 0.0f;
    for (int j = tid; j < cols; j += 256) {
      const float v = STAGED ? rowbuf[j] : In<IN>::to_f32(x[j]);
      const float e = exp_acc(v - m);
      if (STAGED) rowbuf[j] = e;
      lsum += e;
    }
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    const float sum = ((red[0] + red[1]) + red[2]) + red[3];
    float den = sum;
    if (base != 0) den = sum + exp_acc(m * -1.0f);
    for (int j = tid; j < cols; j += 256) {
      const float e = STAGED ? rowbuf[j] : exp_acc(In<IN>::to_f32(x[j]) - m);
      float p = e / den;
      if (clip) {
        p = p * clip_w;
        p = p + clip_g;
        p = __builtin_fminf(__builtin_fmaxf(p, 0.0f), 1.0f);
      }
      y[j] = In<IN>::from_f32(p);
    }
    __syncthreads();
  }
}

// ---- 16-byte vector access for the streaming kernels: VEC elements per access (4 fp32 / 8 x 16-bit)
template <int IN>
struct Vec16 {
  static constexpr int N = 16 / In<IN>::bytes;
  static __device__ __forceinline__ void load(const void* p, float* out) {
    const u4 w = *reinterpret_cast<const u4*>(p);
    if constexpr (IN == IN_F32) {
      out[0] = bits_f32(w.x); out[1] = bits_f32(w.y); out[2] = bits_f32(w.z); out[3] = bits_f32(w.w);
    } else {
      const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        out[2 * i] = In<IN>::to_f32((unsigned short)(ws[i] & 0xffffu));
        out[2 * i + 1] = In<IN>::to_f32((unsigned short)(ws[i] >> 16));
      }
    }
  }
  static __device__ __forceinline__ void store(void* p, const float* v) {
    u4 w;
    if constexpr (IN == IN_F32) {
      w = u4{f32_bits(v[0]), f32_bits(v[1]), f32_bits(v[2]), f32_bits(v[3])};
    } else if constexpr (IN == IN_BF16) {
      w = u4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
    } else {
      w = u4{pack2_f16(v[0], v[1]), pack2_f16(v[2], v[3]), pack2_f16(v[4], v[5]), pack2_f16(v[6], v[7])};
    }
    *reinterpret_cast<u4*>(p) = w;
  }
};

// softmax rows of up to 64*VEC*CH elements: ONE WAVE per row, the row in registers (16-B loads, lane-strided chunks), wave
// reductions by shuffles - no LDS, no workgroup barrier; four rows per 256-thread workgroup.  Same per-element arithmetic as
// the staged kernel above (1-ulp exp, true division).
template <int IN, int CH>
__global__ __launch_bounds__(256) void oeh_softmax_rows_wave_kernel(const void* __restrict__ xin, void* yout, long rows, int cols,
                                                                   int base, int clip, float clip_w, float clip_g) {
  constexpr int N = Vec16<IN>::N;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const unsigned char* x = reinterpret_cast<const unsigned char*>(xin) + row * cols * In<IN>::bytes;
  unsigned char* y = reinterpret_cast<unsigned char*>(yout) + row * cols * In<IN>::bytes;
  const int nchunk = cols / N;
  float v[CH][N];
  float m = -__builtin_inff();
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) {
      Vec16<IN>::load(x + (long)ch * 16, v[c]);
#pragma unroll
      for (int i = 0; i < N; ++i) m = __builtin_fmaxf(m, v[c][i]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
  float sum = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (lane + 64 * c < nchunk) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        v[c][i] = exp_acc(v[c][i] - m);
        sum += v[c][i];
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  float den = sum;
  if (base != 0) den = sum + exp_acc(m * -1.0f);
  const float rden = 1.0f / den;  // RN(1/den): e/den below is the three-instruction correctly rounded quotient (oeh_common.h: fq_quot)
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float q0 = v[c][i] * rden;
        float p = __builtin_fmaf(__builtin_fmaf(-q0, den, v[c][i]), rden, q0);
        if (!(den < 3.0e38f)) p = v[c][i] / den;  // softmax_1 of a fully masked row: den = inf, p = 0 (row-uniform test)
        if (clip) {
          p = p * clip_w;
          p = p + clip_g;
          p = __builtin_fminf(__builtin_fmaxf(p, 0.0f), 1.0f);
        }
        v[c][i] = p;
      }
      Vec16<IN>::store(y + (long)ch * 16, v[c]);
    }
  }
}

template <int IN>
static bool launch_softmax_rows_wave(const void* x, void* y, long rows, int cols, int base, int clip, float w, float g, hipStream_t st) {
  constexpr int N = Vec16<IN>::N;
  if (cols % N != 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) return false;
  const int nchunk = cols / N;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (nchunk <= 64) hipLaunchKernelGGL((oeh_softmax_rows_wave_kernel<IN, 1>), dim3(grid), dim3(256), 0, st, x, y, rows, cols, base, clip, w, g);
  else if (nchunk <= 128) hipLaunchKernelGGL((oeh_softmax_rows_wave_kernel<IN, 2>), dim3(grid), dim3(256), 0, st, x, y, rows, cols, base, clip, w, g);
  else if (nchunk <= 256) hipLaunchKernelGGL((oeh_softmax_rows_wave_kernel<IN, 4>), dim3(grid), dim3(256), 0, st, x, y, rows, cols, base, clip, w, g);
  else if (nchunk <= 512) hipLaunchKernelGGL((oeh_softmax_rows_wave_kernel<IN, 8>), dim3(grid), dim3(256), 0, st, x, y, rows, cols, base, clip, w, g);
  else return false;
  return true;
}

int launch_softmax_rows(const void* x, void* y, long rows, int cols, int in, int base, int clip, float w, float g, hipStream_t st) {
  {  // rows that fit a wave's registers (<= 2048 fp32 / 4096 16-bit elements, multiple of the 16-B vector)
    bool done;
    switch (in) {
      case IN_F16: done = launch_softmax_rows_wave<IN_F16>(x, y, rows, cols, base, clip, w, g, st); break;
      case IN_BF16: done = launch_softmax_rows_wave<IN_BF16>(x, y, rows, cols, base, clip, w, g, st); break;
      default: done = launch_softmax_rows_wave<IN_F32>(x, y, rows, cols, base, clip, w, g, st); break;
    }
    if (done) return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  const bool staged = cols <= 15872;
  const size_t shmem = staged ? (size_t)cols * 4 : 0;
  const unsigned grid = (unsigned)(rows < 65536 ? rows : 65536);
#define OEH_SMX(INV)                                                                                                   \
  if (staged) hipLaunchKernelGGL((oeh_softmax_rows_kernel<INV, true>), dim3(grid), dim3(256), shmem, st, x, y, rows, cols, base, clip, w, g); \
  else hipLaunchKernelGGL((oeh_softmax_rows_kernel<INV, false>), dim3(grid), dim3(256), 0, st, x, y, rows, cols, base, clip, w, g);
  switch (in) {
    case IN_F16: OEH_SMX(IN_F16) break;
    case IN_BF16: OEH_SMX(IN_BF16) break;
    default: OEH_SMX(IN_F32) break;
  }
#undef OEH_SMX
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------
// fake-quant: grid-stride, 4 elements per thread-iteration.
template <int IN>
__global__ __launch_bounds__(256) void oeh_fake_quant_kernel(const void* __restrict__ xin, void* yout, unsigned char* idx_out, long n,
                                                             FqP f) {
  typedef typename In<IN>::elem E;
  const E* x = reinterpret_cast<const E*>(xin);
  E* y = reinterpret_cast<E*>(yout);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = In<IN>::to_f32(x[i]);
    const float r = __builtin_rintf(v / f.scale) + f.zp;  // true IEEE division (uniform_quantizers.py:114)
    const float idx = __builtin_fminf(__builtin_fmaxf(r, 0.0f), f.qmax);
    if (idx_out) idx_out[i] = (unsigned char)idx;
    if (y) y[i] = In<IN>::from_f32(f.scale * (idx - f.zp));
  }
}

// 16-B accesses, the three-instruction correctly rounded quotient (oeh_common.h: fq_quot) instead of a division; the
// (n % VEC) tail elements are done by the last thread with the scalar formula
template <int IN>
__global__ __launch_bounds__(256) void oeh_fake_quant_vec_kernel(const void* __restrict__ xin, void* yout, unsigned char* idx_out, long n, FqP f) {
  constexpr int N = Vec16<IN>::N;
  const long nvec = n / N;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float v[N];
    Vec16<IN>::load(reinterpret_cast<const unsigned char*>(xin) + i * 16, v);
    unsigned int packed[N / 4];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float rel = fq_rel(v[k], f);
      if (k % 4 == 0) packed[k / 4] = 0;
      packed[k / 4] |= ((unsigned int)(rel + f.zp)) << (8 * (k % 4));
      v[k] = f.scale * rel;
    }
    if (yout) Vec16<IN>::store(reinterpret_cast<unsigned char*>(yout) + i * 16, v);
    if (idx_out) {
#pragma unroll
      for (int k = 0; k < N / 4; ++k) reinterpret_cast<unsigned int*>(idx_out + i * N)[k] = packed[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    typedef typename In<IN>::elem E;
    for (long i = nvec * N; i < n; ++i) {
      const float rel = fq_rel(In<IN>::to_f32(reinterpret_cast<const E*>(xin)[i]), f);
      if (idx_out) idx_out[i] = (unsigned char)(rel + f.zp);
      if (yout) reinterpret_cast<E*>(yout)[i] = In<IN>::from_f32(f.scale * rel);
    }
  }
}

int launch_fake_quant(const void* x, void* y, unsigned char* idx, long n, int in, FqP f, hipStream_t st) {
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && (reinterpret_cast<uintptr_t>(idx) & 3) == 0 && (idx == nullptr || f.qmax <= 255.0f)) {
    const long nvec = n / (in == IN_F32 ? 4 : 8);
    long blocks = (nvec + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    switch (in) {
      case IN_F16: hipLaunchKernelGGL(oeh_fake_quant_vec_kernel<IN_F16>, dim3((unsigned)blocks), dim3(256), 0, st, x, y, idx, n, f); break;
      case IN_BF16: hipLaunchKernelGGL(oeh_fake_quant_vec_kernel<IN_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, x, y, idx, n, f); break;
      default: hipLaunchKernelGGL(oeh_fake_quant_vec_kernel<IN_F32>, dim3((unsigned)blocks), dim3(256), 0, st, x, y, idx, n, f); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  long blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  switch (in) {
    case IN_F16: hipLaunchKernelGGL(oeh_fake_quant_kernel<IN_F16>, dim3((unsigned)blocks), dim3(256), 0, st, x, y, idx, n, f); break;
    case IN_BF16: hipLaunchKernelGGL(oeh_fake_quant_kernel<IN_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, x, y, idx, n, f); break;
    default: hipLaunchKernelGGL(oeh_fake_quant_kernel<IN_F32>, dim3((unsigned)blocks), dim3(256), 0, st, x, y, idx, n, f); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------------------------------
// gate: thread per (b,t,h); consecutive threads read consecutive 2d-byte slices of one hidden row.
template <int IN>
__global__ __launch_bounds__(256) void oeh_gate_logit_kernel(const void* __restrict__ hidden, int B, int T, int H, int d, long hs_b, long hs_t,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2, int m_units,
                                                             int apply_sigmoid, float scaling, float* out) {
  typedef typename In<IN>::elem E;
  const long n = (long)B * T * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const int t = (int)((i / H) % T);
    const int b = (int)(i / ((long)H * T));
    const E* x = reinterpret_cast<const E*>(hidden) + (long)b * hs_b + (long)t * hs_t + (long)h * d;
    float a;
    if (m_units == 0) {
      a = 0.0f;
      for (int k = 0; k < d; ++k) a = __builtin_fmaf(In<IN>::to_f32(x[k]), w1[(long)h * d + k], a);
      a = a + b1[h];
    } else {
      a = 0.0f;
      for (int j = 0; j < m_units; ++j) {
        const float* wr = w1 + ((long)h * m_units + j) * d;
        float u = 0.0f;
        for (int k = 0; k < d; ++k) u = __builtin_fmaf(In<IN>::to_f32(x[k]), wr[k], u);
        u = __builtin_fmaxf(u + b1[(long)h * m_units + j], 0.0f);
        a = __builtin_fmaf(u, w2[(long)h * m_units + j], a);
      }
      a = a + b2[h];
    }
    if (apply_sigmoid) a = (1.0f / (1.0f + exp_acc(-a))) * scaling;
    out[((long)b * H + h) * T + t] = a;
  }
}

// conditional_per_head: gate[b,h,0] = sigmoid(mean_t logit[b,h,t]) * scaling  (bert_attention.py:110,321-325)
__global__ __launch_bounds__(64) void oeh_gate_pool_kernel(float* io, int T, float scaling) {
  float* row = io + (long)blockIdx.x * T;
  float s = 0.0f;
  for (int t = threadIdx.x; t < T; t += 64) s += row[t];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) {
    const float a = s / (float)T;
    row[0] = (1.0f / (1.0f + exp_acc(-a))) * scaling;
  }
}

// The same arithmetic (same fma order inside a hidden unit) for the head dims the attention kernels take, without the
// per-element global traffic of the kernel above (204 us for BERT-base B=32 S=128 with 64->16->1 predictors).  One
// workgroup = 64 consecutive tokens of ONE head x NW waves; wave w computes hidden units w, w+NW, w+2NW, ... for all 64 tokens
// (lane = token), so the unit index is wave-uniform and the predictor weights are SCALAR loads feeding v_fmac directly: no
// LDS traffic for weights (broadcast ds_reads made an earlier version LDS-bound at 91 us for B=512), four independent
// accumulator chains per lane.  A token's d inputs are loaded once (16-B loads) and held in registers as fp32; the four
// waves' partial second-layer sums meet in LDS.
template <int IN, int D, int NW>
__global__ __launch_bounds__(64 * NW) void oeh_gate_logit_fast_kernel(const void* __restrict__ hidden, long ntok, int T, int H, long hs_b, long hs_t,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                                  const float* __restrict__ w2, const float* __restrict__ b2, int m_units,
                                                                  int apply_sigmoid, float scaling, float* out) {
  typedef typename In<IN>::elem E;
  __shared__ float part[NW][64];
  const int h = blockIdx.y;
  const int mm = m_units > 0 ? m_units : 1;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  long tok = (long)blockIdx.x * 64 + lane;
  const bool live = tok < ntok;
  if (!live) tok = ntok - 1;
  const long b = tok / T;
  const int t = (int)(tok - b * T);
  const E* xp = reinterpret_cast<const E*>(hidden) + b * hs_b + (long)t * hs_t + (long)h * D;
  float x[D];
  if constexpr (IN == IN_F32) {
#pragma unroll
    for (int k = 0; k < D; k += 4) {
      const f4 v = *reinterpret_cast<const f4*>(xp + k);
      x[k] = v[0]; x[k + 1] = v[1]; x[k + 2] = v[2]; x[k + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int k = 0; k < D; k += 8) {
      const u4 v = *reinterpret_cast<const u4*>(xp + k);
      const unsigned int wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        x[k + 2 * q] = In<IN>::to_f32((unsigned short)(wds[q] & 0xffffu));
        x[k + 2 * q + 1] = In<IN>::to_f32((unsigned short)(wds[q] >> 16));
      }
    }
  }
  const float* wh = w1 + (long)h * mm * D;  // wave-uniform addresses from here on: scalar loads
  const float* b1h = b1 + (long)h * mm;
  const float* w2h = m_units > 0 ? w2 + (long)h * mm : nullptr;
  float a = 0.0f;
  auto finish = [&](float u, int j) {
    u = u + b1h[j];
    if (m_units > 0) {
      u = __builtin_fmaxf(u, 0.0f);
      a = __builtin_fmaf(u, w2h[j], a);
    } else {
      a = u;
    }
  };
  int j = wave;
  for (; j + 3 * NW < mm; j += 4 * NW) {  // units j, j+NW, j+2NW, j+3NW: four independent chains, each in the element kernel's k order
    const float* r0 = wh + (long)j * D;
    const float* r1 = r0 + NW * D;
    const float* r2 = r0 + 2 * NW * D;
    const float* r3 = r0 + 3 * NW * D;
    float u0 = 0.0f, u1 = 0.0f, u2 = 0.0f, u3 = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      if ((k & 15) == 0) asm volatile("" ::: "memory");  // 4 x 16 weights in flight: keeps the scalar loads within the SGPR file
      u0 = __builtin_fmaf(x[k], r0[k], u0);
      u1 = __builtin_fmaf(x[k], r1[k], u1);
      u2 = __builtin_fmaf(x[k], r2[k], u2);
      u3 = __builtin_fmaf(x[k], r3[k], u3);
    }
    finish(u0, j);
    finish(u1, j + NW);
    finish(u2, j + 2 * NW);
    finish(u3, j + 3 * NW);
  }
  for (; j < mm; j += NW) {
    const float* r0 = wh + (long)j * D;
    float u = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) u = __builtin_fmaf(x[k], r0[k], u);
    finish(u, j);
  }
  if constexpr (NW > 1) {
    part[wave][lane] = a;
    __syncthreads();
  }
  if (wave == 0) {
    if constexpr (NW > 1) a = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (m_units > 0) a = a + b2[h];
    if (apply_sigmoid) a = (1.0f / (1.0f + exp_acc(-a))) * scaling;
    if (live) out[(b * H + h) * T + t] = a;
  }
}

// The predictor's first layer on the MATRIX CORES (fp16 and fp32 layer inputs; head dims 32 / 64 / 128): logits^T = W1 X^T per
// wave of 16 tokens, as inside the attention kernels (oeh_attn_fast.inl, GATE) - but with the fp32 weights as fp16 operand PAIRS
// (and, for an fp32 input, the input rows too), i.e. to 2^-22 of the fp32 Linear: 2 (3) MFMAs per 32 inputs and 16 hidden units
// instead of 512 scalar-operand FMAs per lane.  A workgroup = 64 consecutive tokens of one head, a wave 16 of them (token on the
// lane's column, hidden units over registers).  BERT-base B=32 S=128 fp32, 64 -> 16 -> 1: 9.3 -> 6.2 us, 64 -> 64 -> 1: 18.6 -> 11.5 us,
// B=512: 95 -> 66 us; the kernel above stays for
// bf16 (a bf16 pair carries 16 mantissa bits) and the element kernel for everything else.
// Round 5: a workgroup takes TG groups of 64 consecutive tokens of its head (grid x = ceil(ntok / (64 TG))) with the first layer's weight fragments - loaded,
// split into the fp16 pair and kept in registers ONCE (up to 64 hidden units: 4 x KS x 2 fragments) - instead of once per 64 tokens: the launch was made of
// 3 072 short workgroups per 25 MB of layer input and ran at 1.0-1.4 TB/s (B = 128, S = 128 fp32: 17.7 us).
template <int IN, int D, int TG>
__global__ __launch_bounds__(256) void oeh_gate_mfma_kernel(const void* __restrict__ hidden, long ntok, int T, int H, long hs_b, long hs_t,
                                                            const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, int m_units, int apply_sigmoid, float scaling, float* out) {
  static_assert(IN == IN_F16 || IN == IN_F32, "fp16 pairs");
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  constexpr int KS = D / 32, MT = 4;   // up to MT 16-unit tiles of hidden units held in registers (the host sends wider predictors to the kernel above)
  const int h = blockIdx.y;
  const int mm = m_units > 0 ? m_units : 1;
  const int ntau = (mm + 15) >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  fp16_overflow_clamp();
  auto mma = [](u4 a, u4 bb, f4 acc) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, bb), acc, 0, 0, 0); };
  // the weights of this head: unit 16 tau + c supplies the A rows; b1 / w2 of the units 16 tau + 4 g + r that come back in the lane's accumulator registers
  u4 wh[MT][KS], wl[MT][KS];
  float b1r[MT][4], w2r[MT][4];
#pragma unroll
  for (int tau = 0; tau < MT; ++tau) {
    if (tau < ntau) {
      const int u = 16 * tau + c;
      const bool uv = u < mm;
      const float* wr = w1 + ((long)h * mm + (uv ? u : 0)) * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f4 w0 = *reinterpret_cast<const f4*>(wr + 32 * ks), w1v = *reinterpret_cast<const f4*>(wr + 32 * ks + 4);
        if (!uv) w0 = w1v = f4{0.f, 0.f, 0.f, 0.f};
        split8(w0, w1v, wh[tau][ks], wl[tau][ks]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ur = 16 * tau + 4 * g + r;
        b1r[tau][r] = (m_units > 0 ? ur < mm : ur == 0) ? b1[(long)h * mm + ur] : 0.0f;
        w2r[tau][r] = (m_units > 0 && ur < mm) ? w2[(long)h * mm + ur] : 0.0f;
      }
    }
  }
  const float b2h = m_units > 0 ? b2[h] : 0.0f;
  for (int tg = 0; tg < TG; ++tg) {
    long tok = ((long)blockIdx.x * TG + tg) * 64 + wave * 16 + c;
    if ((long)(blockIdx.x * TG + tg) * 64 >= ntok) break;   // (workgroup-uniform)
    const bool live = tok < ntok;
    if (!live) tok = ntok - 1;
    const long b = tok / T;
    const int t = (int)(tok - b * T);
    u4 xf[KS], xl[IN == IN_F32 ? KS : 1];
    if constexpr (IN == IN_F32) {
      const float* xp = reinterpret_cast<const float*>(hidden) + b * hs_b + (long)t * hs_t + (long)h * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) split8(*reinterpret_cast<const f4*>(xp + 32 * ks), *reinterpret_cast<const f4*>(xp + 32 * ks + 4), xf[ks], xl[ks]);
    } else {
      const unsigned short* xp = reinterpret_cast<const unsigned short*>(hidden) + b * hs_b + (long)t * hs_t + (long)h * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const u4*>(xp + 32 * ks);
    }
    float a = 0.0f;
#pragma unroll
    for (int tau = 0; tau < MT; ++tau) {
      if (tau < ntau) {
        f4 acc = f4{0.f, 0.f, 0.f, 0.f}, accx = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc = mma(wh[tau][ks], xf[ks], acc);
          accx = mma(wl[tau][ks], xf[ks], accx);
          if constexpr (IN == IN_F32) accx = mma(wh[tau][ks], xl[ks], accx);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ur = 16 * tau + 4 * g + r;
          const float lg = __builtin_fmaf(accx[r], kSplitDown, acc[r]);
          if (m_units > 0) {
            if (ur < mm) a = __builtin_fmaf(__builtin_fmaxf(lg + b1r[tau][r], 0.0f), w2r[tau][r], a);
          } else if (ur == 0) {
            a = lg + b1r[tau][r];  // Linear(D,1): unit 0 only
          }
        }
      }
    }
    a += __shfl_xor(a, 16);  // the row's four lanes hold disjoint hidden units
    a += __shfl_xor(a, 32);
    if (m_units > 0) a = a + b2h;
    if (apply_sigmoid) a = (1.0f / (1.0f + exp_acc(-a))) * scaling;
    if (live && g == 0) out[(b * H + h) * T + t] = a;
  }
}

template <int IN, int D>
static void launch_gate_fast_d(const void* hidden, long ntok, int T, int H, long hs_b, long hs_t, const float* w1, const float* b1,
                               const float* w2, const float* b2, int m_units, int apply_sigmoid, float scaling, float* out, hipStream_t st) {
  const dim3 grid((unsigned)((ntok + 63) / 64), (unsigned)H);
  if (m_units >= 4)  // MLP predictor: the hidden units are dealt over 4 waves
    hipLaunchKernelGGL((oeh_gate_logit_fast_kernel<IN, D, 4>), grid, dim3(256), 0, st, hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out);
  else
    hipLaunchKernelGGL((oeh_gate_logit_fast_kernel<IN, D, 1>), grid, dim3(64), 0, st, hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out);
}

template <int IN>
static bool launch_gate_fast(const void* hidden, int B, int T, int H, int d, long hs_b, long hs_t, const float* w1, const float* b1,
                             const float* w2, const float* b2, int m_units, int apply_sigmoid, float scaling, float* out, hipStream_t st) {
  const int eb = IN == IN_F32 ? 4 : 2;
  if (((reinterpret_cast<uintptr_t>(hidden) | (uintptr_t)(hs_b * eb) | (uintptr_t)(hs_t * eb)) & 15) != 0) return false;
  const long ntok = (long)B * T;
  if constexpr (IN != IN_BF16) {  // the first layer on the matrix cores (fp16 operand pairs: fp32-accurate)
    if ((reinterpret_cast<uintptr_t>(w1) & 15) == 0 && (d == 32 || d == 64 || d == 128)) {
      // (token groups per workgroup: 4 once that still leaves every CU four workgroups)
      const long groups = (ntok + 63) / 64;
      const bool tg4 = groups * H >= 4096 && m_units <= 64;
      const dim3 grid((unsigned)(tg4 ? (groups + 3) / 4 : groups), (unsigned)H);
#define OEH_GM(D_) \
      if (m_units > 64) goto element_kernel; \
      if (tg4) hipLaunchKernelGGL((oeh_gate_mfma_kernel<IN, D_, 4>), grid, dim3(256), 0, st, hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out); \
      else hipLaunchKernelGGL((oeh_gate_mfma_kernel<IN, D_, 1>), grid, dim3(256), 0, st, hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out)
      if (d == 32) { OEH_GM(32); }
      else if (d == 64) { OEH_GM(64); }
      else { OEH_GM(128); }
#undef OEH_GM
      return true;
    }
  }
element_kernel:
  switch (d) {
    case 32: launch_gate_fast_d<IN, 32>(hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out, st); return true;
    case 64: launch_gate_fast_d<IN, 64>(hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out, st); return true;
    case 128: launch_gate_fast_d<IN, 128>(hidden, ntok, T, H, hs_b, hs_t, w1, b1, w2, b2, m_units, apply_sigmoid, scaling, out, st); return true;
    default: return false;
  }
}

int launch_gate(const void* hidden, int in, int B, int T, int H, int d, long hs_b, long hs_t, const float* w1, const float* b1,
                const float* w2, const float* b2, int m_units, int pool, float scaling, float* out, hipStream_t st) {
  bool done = false;
  switch (in) {
    case IN_F16: done = launch_gate_fast<IN_F16>(hidden, B, T, H, d, hs_b, hs_t, w1, b1, w2, b2, m_units, !pool, scaling, out, st); break;
    case IN_BF16: done = launch_gate_fast<IN_BF16>(hidden, B, T, H, d, hs_b, hs_t, w1, b1, w2, b2, m_units, !pool, scaling, out, st); break;
    default: done = launch_gate_fast<IN_F32>(hidden, B, T, H, d, hs_b, hs_t, w1, b1, w2, b2, m_units, !pool, scaling, out, st); break;
  }
  if (!done) {  // any head dim / alignment: one thread per (b,t,h), element loads
    const long n = (long)B * T * H;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    switch (in) {
      case IN_F16: hipLaunchKernelGGL(oeh_gate_logit_kernel<IN_F16>, dim3((unsigned)blocks), dim3(256), 0, st, hidden, B, T, H, d, hs_b, hs_t, w1, b1, w2, b2, m_units, !pool, scaling, out); break;
      case IN_BF16: hipLaunchKernelGGL(oeh_gate_logit_kernel<IN_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, hidden, B, T, H, d, hs_b, hs_t, w1, b1, w2, b2, m_units, !pool, scaling, out); break;
      default: hipLaunchKernelGGL(oeh_gate_logit_kernel<IN_F32>, dim3((unsigned)blocks), dim3(256), 0, st, hidden, B, T, H, d, hs_b, hs_t, w1, b1, w2, b2, m_units, !pool, scaling, out); break;
    }
  }
  if (pool) hipLaunchKernelGGL(oeh_gate_pool_kernel, dim3((unsigned)(B * H)), dim3(64), 0, st, out, T, scaling);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}


// ---------------------------------------------------------------------------------------------------------
// min / max: order-preserving int keys + atomics, decoded in place by a 1-thread kernel.
__device__ __forceinline__ int f2key(float f) { int b = __builtin_bit_cast(int, f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float key2f(int k) { return __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void oeh_minmax_init_kernel(int* keys) { keys[0] = 0x7fffffff; keys[1] = (int)0x80000000; }
__global__ void oeh_minmax_fin_kernel(int* keys) {
  const float lo = key2f(keys[0]), hi = key2f(keys[1]);
  reinterpret_cast<float*>(keys)[0] = lo;
  reinterpret_cast<float*>(keys)[1] = hi;
}
template <int IN>
__global__ __launch_bounds__(256) void oeh_minmax_kernel(const void* __restrict__ xin, long n, int* keys) {
  typedef typename In<IN>::elem E;
  const E* x = reinterpret_cast<const E*>(xin);
  float lo = __builtin_inff(), hi = -__builtin_inff();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = In<IN>::to_f32(x[i]);
    lo = __builtin_fminf(lo, v);
    hi = __builtin_fmaxf(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = __builtin_fminf(lo, __shfl_xor(lo, o));
    hi = __builtin_fmaxf(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&keys[0], f2key(lo));
    atomicMax(&keys[1], f2key(hi));
  }
}

// 16-B loads, four in flight per thread; one atomic pair per workgroup
template <int IN>
__global__ __launch_bounds__(256) void oeh_minmax_vec_kernel(const void* __restrict__ xin, long n, int* keys) {
  constexpr int N = Vec16<IN>::N;
  __shared__ float red[8];
  const long nvec = n / N;
  const long stride = (long)gridDim.x * blockDim.x;
  float lo = __builtin_inff(), hi = -__builtin_inff();
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    float v[4][N];
#pragma unroll
    for (int u = 0; u < 4; ++u) Vec16<IN>::load(reinterpret_cast<const unsigned char*>(xin) + (i + u * stride) * 16, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < N; ++k) {
        lo = __builtin_fminf(lo, v[u][k]);
        hi = __builtin_fmaxf(hi, v[u][k]);
      }
  }
  for (; i < nvec; i += stride) {
    float v[N];
    Vec16<IN>::load(reinterpret_cast<const unsigned char*>(xin) + i * 16, v);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      lo = __builtin_fminf(lo, v[k]);
      hi = __builtin_fmaxf(hi, v[k]);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    typedef typename In<IN>::elem E;
    for (long t = nvec * N; t < n; ++t) {
      const float v = In<IN>::to_f32(reinterpret_cast<const E*>(xin)[t]);
      lo = __builtin_fminf(lo, v);
      hi = __builtin_fmaxf(hi, v);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = __builtin_fminf(lo, __shfl_xor(lo, o));
    hi = __builtin_fmaxf(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lo; red[4 + (threadIdx.x >> 6)] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&keys[0], f2key(__builtin_fminf(__builtin_fminf(red[0], red[1]), __builtin_fminf(red[2], red[3]))));
    atomicMax(&keys[1], f2key(__builtin_fmaxf(__builtin_fmaxf(red[4], red[5]), __builtin_fmaxf(red[6], red[7]))));
  }
}

int launch_minmax(const void* x, long n, int in, float* out2, hipStream_t st) {
  int* keys = reinterpret_cast<int*>(out2);
  hipLaunchKernelGGL(oeh_minmax_init_kernel, dim3(1), dim3(1), 0, st, keys);
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const long nvec = n / (in == IN_F32 ? 4 : 8);
    long vb = (nvec + 1023) / 1024;  // four vectors per thread
    if (vb > 2048) vb = 2048;
    if (vb < 1) vb = 1;
    switch (in) {
      case IN_F16: hipLaunchKernelGGL(oeh_minmax_vec_kernel<IN_F16>, dim3((unsigned)vb), dim3(256), 0, st, x, n, keys); break;
      case IN_BF16: hipLaunchKernelGGL(oeh_minmax_vec_kernel<IN_BF16>, dim3((unsigned)vb), dim3(256), 0, st, x, n, keys); break;
      default: hipLaunchKernelGGL(oeh_minmax_vec_kernel<IN_F32>, dim3((unsigned)vb), dim3(256), 0, st, x, n, keys); break;
    }
    hipLaunchKernelGGL(oeh_minmax_fin_kernel, dim3(1), dim3(1), 0, st, keys);
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  switch (in) {
    case IN_F16: hipLaunchKernelGGL(oeh_minmax_kernel<IN_F16>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, keys); break;
    case IN_BF16: hipLaunchKernelGGL(oeh_minmax_kernel<IN_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, keys); break;
    default: hipLaunchKernelGGL(oeh_minmax_kernel<IN_F32>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, keys); break;
  }
  hipLaunchKernelGGL(oeh_minmax_fin_kernel, dim3(1), dim3(1), 0, st, keys);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace oeh
