/*
 * Developer entry points of libtonic_hip.so — NOT part of the drop-in boundary (include/tonic_hip.h).
 * They exist for the parity tests and the profiling scripts of this repository: a process-wide
 * tuning switch between kernel variants that compute the same thing, one stand-alone GEMM of the
 * small-batch building block, and a cycle probe of the fused grad kernel.  Nothing in the product
 * path (the tonic_amd package itself, as opposed to tests, scripts and bench) may depend on the knobs being set.
 */
#ifndef TONIC_HIP_DEV_H
#define TONIC_HIP_DEV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs (process-wide atomics: a setting takes effect for launches issued after the call;
 * call before sizing workspaces).  Keys:
 *   "grad_waves" = 4       waves per workgroup of the 32x32x2-tile fused grad kernel;
 *   "grad_variant" = 0 .. 3 | -1  fused grad kernel: 0 = 32x32x2 fp32 tiles, 1 wave/SIMD;
 *                          1 = 16x16x4 fp32 tiles, 2 waves/SIMD; 2 = 1 with the two 64x64
 *                          hidden-layer products of a tile computed on bf16x3 terms (each fp32
 *                          operand = hi + mid + lo bf16 exactly; six bf16 MFMAs per product, fp32
 *                          accumulation: fp32-equivalent results, 2.4x less MFMA time); 3 (default)
 *                          = 2 with the 64x64 weight-gradient product dW2 on bf16x3 terms as well;
 *                          -1 restores the default;
 *   "grad_skew" = 0..64    start delay of half of the waves of variant 1 (experiment, default 0);
 *   "chain_fault" = 1      the NEXT chained critic step loses its first workgroup (test hook);
 *   "q_chain" = 0 | 1      off-policy iteration as chained launches (1, default) or one launch per pass;
 *   "gae_stream" = 0..4    which bit-exact GAE kernel serves small W (developer probe);
 *   "policy_tail" = 0 | 1  off-policy actors: sampling / target noise / dense copy in the tail of
 *                          the forward launch (1, default) or in their own launches (0); same bits. */
int tonic_set_tuning(const char* key, int32_t value);
int tonic_get_tuning(const char* key, int32_t* value);

/* Developer tool: per-phase cycle totals (s_memtime) of the 8 waves of workgroup 0 of the
 * 16x16x4 fused actor grad kernel; d_phase_cycles = uint64[8][12]. */
int tonic_debug_grad16_phases(const float* d_actor_params, const float* d_observations,
                              const float* d_actions, const float* d_advantages,
                              const float* d_adv_stats, const float* d_old_log_probs, int64_t n,
                              int32_t O, int32_t A, void* d_workspace, int64_t workspace_bytes,
                              uint64_t* d_phase_cycles, void* stream);

/* Developer tool: wall-clock stamps (10 ns ticks) of workgroup (0, 0) of the fused off-policy
 * forward at its phase boundaries {entry, loads issued, layer 1, epilogue + barrier, layer 2,
 * epilogue + barrier, heads / output, policy tail}; d_stamps = uint64[8 launches][16] (wall clock | shader cycles), filled in
 * rotation by the launches that follow, then uint64[8 launches][8] of the weight-gradient group
 * {entry, requests out, main loop, partials exchanged, folded, -, end}; null switches the probe off
 * (scripts/forward_stamps.py). */
int tonic_debug_forward_stamps(uint64_t* d_stamps);

/* Developer / test entry: one GEMM of the small-batch fp32 MFMA building block
 * (mode "NT" | "NN" | "TN"; act 0 none, 1 relu, 2 tanh; d_mask multiplies by (mask > 0)). */
int tonic_gemm_f32(const char* mode, const float* d_a, const float* d_b, float* d_c,
                   const float* d_bias, const float* d_mask, float* d_colsum, int32_t M,
                   int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t act,
                   int32_t accumulate, double alpha, void* stream);

/* Test tool: `workgroups` workgroups that each hold a compute unit to themselves (100 KB of LDS:
 * neither a collect nor a grad workgroup fits beside one) and spin for `milliseconds` of the
 * 100 MHz wall clock — a foreign kernel that occupies part of the chip, for the tests of what the
 * resident collect kernel does when not all of its workgroups find a compute unit. */
int tonic_debug_occupy(int32_t workgroups, double milliseconds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TONIC_HIP_DEV_H */
