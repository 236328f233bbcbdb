// Head dim 64, the FULL-ROW 16-bit / fp32-storage kernel (oeh_attn_fast.inl: NT x {f16, bf16} x {clip, gate, fake-quant forms}).
#include "oeh_attn_fast.inl"

namespace oeh {
int launch_attn_fast_d64(const AttnParams& P, int in, hipStream_t st) { return launch_fast_d<64>(P, in, st); }
}  // namespace oeh
