/*
 * dsmgp_hip_diag.h -- entry points of the DIAGNOSTIC build only (libdsmgp_hip_diag.so, compiled with -DDSMGP_DIAG
 * by deepstructuredmixtures_amd/csrc/build.sh diag).  They exist for kernel tuning (tools/): in-kernel cycle
 * stamps, micro-benchmarks of single kernels, scheduling knobs read from the environment.  The product library
 * libdsmgp_hip.so contains none of this code and none of these symbols (tests/test_host_cpu.py checks).
 */
#ifndef DSMGP_HIP_DIAG_H
#define DSMGP_HIP_DIAG_H

#include "dsmgp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic: f64 MFMA vs f64 VALU FMA rates alone and co-issued: out[9] = {mfma TF/s, valu TF/s, ms} x {MFMA only, VALU only, both} */
int dsmgp_probe_coissue(dsmgp_ctx* ctx, double* out);
/* diagnostic: seconds per launch of the tile GEMM on a uniform batch of ntiles tiles of depth K
 * (mode 0: own A panel per tile, B panel shared by `group` tiles; mode 1: all operands shared, L2-resident) */
int dsmgp_bench_tile(dsmgp_ctx* ctx, int32_t ntiles, int32_t K, int32_t mode, int32_t group, int32_t reps,
                     double* seconds_per_launch);
/* diagnostic: seconds per launch of the eight-wave fused tile task on ntasks tasks of depth K (eight full 16-row blocks each, own A
 * rows, B panel shared by `group` tasks): the slope over K is the steady-state rate of its product loop */
int dsmgp_bench_fused8(dsmgp_ctx* ctx, int32_t ntasks, int32_t K, int32_t group, int32_t reps, double* seconds_per_launch);
/* diagnostic: the diagonal-block kernel alone on ntiles blocks (us per launch) and the wall-clock phases of one block:
 * phases_us[23] = load, first 16x16 block, (P1, P2) x 8 block steps, write-back, inverse phase, and inside step 3's P2 on
 * wave 0 the trailing product and the potrf + inverse of the next 16x16 diagonal block (us, then shader cycles) */
int dsmgp_probe_diag(dsmgp_ctx* ctx, int32_t ntiles, int32_t ld, int32_t reps, double* kernel_us, double* phases_us);

/* diagnostic: the diagonal-block task of a fused step (diag_fused_reg_kernel) alone on ntiles synthetic blocks whose tile update has
 * depth K (a multiple of 128; 0 = the first block step): us per launch.  256 / 512 / 768 blocks = one / two / three tasks per CU */
int dsmgp_probe_diag_fused(dsmgp_ctx* ctx, int32_t ntiles, int32_t K, int32_t reps, double* kernel_us);

#ifdef __cplusplus
}
#endif
#endif
