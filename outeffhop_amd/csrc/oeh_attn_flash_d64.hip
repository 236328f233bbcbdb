// Head dim 64, the ONE-PASS kernel (oeh_attn_flash.inl: {f16, bf16, f32 storage} x MQ x {key padding, gate, two-pass clip / INT8 forms}).
#include "oeh_attn_flash.inl"

namespace oeh {
int launch_attn_flash_d64(const AttnParams& P, int in, int mq, hipStream_t st) { return launch_flash_d<64>(P, in, mq, st); }
}  // namespace oeh
