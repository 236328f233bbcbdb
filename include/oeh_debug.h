/*
 * oeh_debug.h - diagnostic hooks of liboeh_hip.so.  NOT part of the product ABI (include/oeh.h): process-global, not
 * thread-safe, they change which kernel variant oeh_attn_fwd launches for the whole process.  They exist for the A/B timing
 * tools under tools/ and for the tests that force a second kernel over the same problem (tests/test_attn_gpu.py).
 *
 * They do nothing unless the environment variable OEH_DEBUG_HOOKS=1 was set when the library was first used; otherwise
 * they return OEH_ENOTSUP (-95) and leave the library's behaviour untouched.
 */
#ifndef OEH_DEBUG_H_
#define OEH_DEBUG_H_
#ifdef __cplusplus
extern "C" {
#endif

/* off_mask: bit (1 << v) disables kernel variant v (1 one-pass, 2 full-row, 3 general, 4 any-shape, 5 small-shape; 6 / 7 the
 * fp32-storage forms of the one-pass / full-row kernels); bit 8 lets the one-pass kernel take rows of <= 128 keys too;
 * bit 9: plain block order in the one-pass kernel (no snake placement); bit 10: the small-shape kernel wherever it can run;
 * bit 11: the full-row kernel also for head dim 128 with clip / INT8 (otherwise the general kernel there);
 * bit 12: the 32x32x16 form of the one-pass kernel (oeh_attn_wide.hip) wherever it applies.
 *  flash_mq_force != 0 fixes the one-pass kernel's query blocks per
 * wave.  (0, 0) restores the defaults.  Returns 0, or -95 when the hooks are not enabled. */
int oeh_debug_set_variant(int off_mask, int flash_mq_force);

/* device buffer of 32 u64 per wave that the one-pass / full-row kernels fill with s_memtime / s_memrealtime stamps
 * (tools/timeline.py); NULL switches the stamps off.  Returns 0, or -95 when the hooks are not enabled. */
int oeh_debug_set_stamps(void* device_buffer);

/* Environment switches of the projection GEMM (oeh_proj_quant_i8; csrc/oeh_gemm.hip), read once, inert unless OEH_DEBUG_HOOKS=1:
 *   OEH_GEMM_TILE = 1 | 2   force the 128 x 288 | 64 x 192 output tile (2 also keeps BERT-base-sized launches off the one-workgroup-per-CU loop);
 *   OEH_GEMM_LOOP0 = 1      fp32 activations on the 128 x 288 tile: the round-4 K loop instead of the pipelined one (A/B);
 *   OEH_GEMM_DBG  = bits    knock parts of the kernel out for timing - the results are then WRONG (tools/exp/proj_time.py):
 *                           1 no epilogue, 4 no LDS-DMA after the first step, 8 no wait + barrier per step, 16 no value stores,
 *                           32 no index output stage, 64 the whole next tile's DMA issued at the top of a step (the knock-outs exist in
 *                           `make experiment` builds only; the pipelined loops honour bits 1, 16 and 32). */

#ifdef __cplusplus
}
#endif
#endif /* OEH_DEBUG_H_ */
