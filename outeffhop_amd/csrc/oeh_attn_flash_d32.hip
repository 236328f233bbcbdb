// Head dim 32, the ONE-PASS kernel (oeh_attn_flash.inl: {f16, bf16, f32 storage} x MQ x {key padding, gate, two-pass clip / INT8 forms}).
#include "oeh_attn_flash.inl"

namespace oeh {
int launch_attn_flash_d32(const AttnParams& P, int in, int mq, hipStream_t st) { return launch_flash_d<32>(P, in, mq, st); }
}  // namespace oeh
