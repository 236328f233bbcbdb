// Instantiates the fused attention kernels for head dim 64: the general kernel (oeh_attn_mfma.inl: NT in {8,16,32} x
// {f16,bf16,f32 storage} x {plain, fake-quant}) and the fast 16-bit kernel (oeh_attn_fast.inl: NT x {f16,bf16} x {clip}).
// One translation unit per head dim keeps the build parallel.
#include "oeh_attn_fast.inl"

namespace oeh {
int launch_attn_mfma_d64(const AttnParams& P, int in, bool fq, hipStream_t st) { return launch_d<64>(P, in, fq, st); }
int launch_attn_fast_d64(const AttnParams& P, int in, hipStream_t st) { return launch_fast_d<64>(P, in, st); }
}  // namespace oeh
