// Kernel-argument block shared by the fused attention kernels (host fills it in oeh_api.hip).
#pragma once
#include "oeh_common.h"

namespace oeh {

struct AttnParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  int B, H, Sq, Sk, D;
  long qs_b, qs_h, qs_s;
  long ks_b, ks_h, ks_s;
  long vs_b, vs_h, vs_s;
  long os_b, os_h, os_s;
  float scale, scale_div;     // scale_div != 0: divide (BERT), else multiply
  int base;                   // 0 vanilla, 1 softmax_1
  int clip;
  float clip_w, clip_g;       // fl32(eta - gamma), fl32(gamma)
  const void* pad;            // (B,Sk) additive or null
  int pad_f16;
  long pad_sb;
  const void* full;           // (B,1,Sq,Sk) additive or null
  int full_f16;
  long full_sb, full_sq;
  int causal, clamp_min;
  float mask_min;
  const float* gate;          // null = none
  long gs_b, gs_h, gs_s;
  // fused gate predictor (null gh = none): layer input rows, per-head weights, optional copy of the gate (include/oeh.h)
  const void* gh;
  long ghs_b, ghs_t;
  const float *gw1, *gb1, *gw2, *gb2;
  int g_units;
  float g_scaling;
  float* g_out;
  FqP fq_s, fq_p, fq_c;
  int ctx_before_gate;
  int src32;                  // fp32 q / k / v read directly by the kernel and fp32 output (SRC32 variants)
  int out32;                  // 16-bit q / k / v, fp32 output straight from the accumulators (o_dtype = OEH_F32; one-pass and full-row kernels)
  // INT8 storage (oeh_attn_i8.hip): 128 - zero point of q, k, v and of the probabilities; RN(s_q s_k scale / s_scores), RN(s_p s_v)
  int i8_cq, i8_ck, i8_cv, i8_cp;
  float i8_k1, i8_so;
  // launch geometry
  int nQT;                    // q tiles (64 rows) per (b,h)
  int nBH, nBHpad;            // B*H and B*H rounded up to a multiple of 8 (XCD affinity of a head's q tiles)
  unsigned magic_nbh, magic_h;  // floor(2^32 / nBHpad), floor(2^32 / H): block id -> (q tile, batch, head) without integer divisions (oeh_common.h: div_magic)
  int skip_ok;                // causal tiles above the diagonal may be skipped (see oeh_api.hip)
  int pad_bool;   // key_pad_boolean (include/oeh.h): mask entries are 0 or <= -1e4
  int head_major;             // fp32-storage kernels: block order in groups of this many heads (0 = all heads' heaviest q tiles first;
                              // oeh_common.h: block_to_tile)
  int snake;                  // one-pass kernel: every second row of 256 block ids walked backwards (snake_block_id, oeh_common.h)
  unsigned long long* stamps; // diagnostic builds only: per-wave s_memtime stamps (null in production)
};

}  // namespace oeh
