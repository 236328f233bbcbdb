// Head dim 64, the GENERAL kernel (oeh_attn_mfma.inl: NT in {8,16,32} x {f16, bf16, f32 storage} x {plain, fake-quant}).
// One translation unit per head dim AND kernel family (round 5; one per head dim before): nine units of similar weight build in parallel,
// and an edit to one family's .inl no longer recompiles the other two.
#include "oeh_attn_mfma.inl"

namespace oeh {
int launch_attn_mfma_d64(const AttnParams& P, int in, bool fq, hipStream_t st) { return launch_d<64>(P, in, fq, st); }
}  // namespace oeh
