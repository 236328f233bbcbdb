// Instantiates the fused attention kernel (oeh_attn_mfma.inl) for head dim 64: NT in {8,16,32} x
// {f16,bf16,f32 storage} x {plain, fake-quant}.  One translation unit per head dim keeps the build parallel.
#include "oeh_attn_mfma.inl"

namespace oeh {
int launch_attn_mfma_d64(const AttnParams& P, int in, bool fq, hipStream_t st) { return launch_d<64>(P, in, fq, st); }
}  // namespace oeh
