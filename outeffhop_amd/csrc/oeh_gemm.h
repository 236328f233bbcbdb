// Parameters of the projection GEMM with quantiser epilogue (oeh_gemm.hip), shared with the C-ABI layer (oeh_api.hip: oeh_proj_quant_i8).
#pragma once
#include <hip/hip_runtime.h>

#include "oeh_common.h"

namespace oeh {

struct GemmSeg {            // one column segment (a projection): columns [n0, n0 + E)
  float alpha;              // weight scale
  FqP f;                    // output quantiser
  signed char* out;         // centred indices: (B, S, E) or (B, H, 64, S) [transpose]
  float* y;                 // dequantised values (B, S, E) fp32 with row stride y_ld, or nullptr
  long y_ld;
  int transpose;
  const int* acc_add;       // int8 form: per-column integer added to the int32 accumulator (E entries), or nullptr
};

struct GemmParams {
  const void* a;            // (M, 2K) fp16 pairs, or (M, K) fp16 [pairs == 0]
  const void* w;            // (N, K) fp16
  const float* bias;        // (N) fp32
  long lda, ldw;            // row strides in elements
  int M, N, K, pairs;
  int MT, NT;               // tiles
  int dbg;                  // diagnostic knock-outs (oeh_gemm.hip: launch_gemm), 0 in production
  unsigned magic_s;         // floor(2^32 / S)
  int E, S, H;              // segment width, rows per batch element, heads per segment (E = 64 H)
  GemmSeg seg[3];
};

constexpr int kGemmBK = 32;  // K must be a multiple (64 in the int8 form: a step is 64 bytes of a row)

int launch_gemm(const GemmParams& P, hipStream_t st);

}  // namespace oeh
