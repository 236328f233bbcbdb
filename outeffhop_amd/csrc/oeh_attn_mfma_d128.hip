// Instantiates the fused attention kernels for head dim 128: the general kernel (oeh_attn_mfma.inl: NT in {8,16,32} x
// {f16,bf16,f32 storage} x {plain, fake-quant}) the full-row 16-bit kernel (oeh_attn_fast.inl: NT x {f16,bf16} x {clip}) and the
// one-pass 16-bit kernel (oeh_attn_flash.inl: {f16,bf16} x MQ).
// One translation unit per head dim keeps the build parallel.
#include "oeh_attn_flash.inl"

namespace oeh {
int launch_attn_mfma_d128(const AttnParams& P, int in, bool fq, hipStream_t st) { return launch_d<128>(P, in, fq, st); }
int launch_attn_fast_d128(const AttnParams& P, int in, hipStream_t st) { return launch_fast_d<128>(P, in, st); }
int launch_attn_flash_d128(const AttnParams& P, int in, int mq, hipStream_t st) { return launch_flash_d<128>(P, in, mq, st); }
}  // namespace oeh
