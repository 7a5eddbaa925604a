/*
 * dsmgp_hip.h -- C ABI of the MI355X (gfx950) GP-expert hot path of DeepStructuredMixtures.
 *
 * The reference (Julia) has no FFI of its own: the boundary is the set of Julia methods whose
 * bodies are the hot path.  Each entry point below names the reference
 * code it replaces; INTEGRATION.md shows the `ccall` glue a maintainer would add.
 *
 * Conventions
 *   - all matrices are column-major Float64 (Julia `Matrix{Float64}`), indices 0-based, sizes int64/int32
 *   - every function returns 0 on success, a negative DSMGP_E_* code on failure; the message is
 *     available from dsmgp_last_error(); nothing throws or longjmps across the ABI
 *   - the caller owns every host pointer; the library never keeps a host pointer after return
 *   - a context is bound to ONE GPU and must not be used from two threads at once (the reference
 *     `fit!` loop is serial, src/fit.jl:88); multi-GPU = one context per process/GPU, leaves sharded
 *     by the caller
 *   - hyper-parameters are on the reference's log scale: [logl..., logs, logNoise] with
 *     lengthscale exp(logl), signal variance exp(2 logs), noise variance exp(2 logNoise)
 *     (src/kernels.jl:68-73, src/gaussianprocess.jl:39,153-161)
 */
#ifndef DSMGP_HIP_H
#define DSMGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsmgp_ctx dsmgp_ctx;

/* kernel kinds (src/kernels.jl:59,109,174) */
#define DSMGP_KIND_ISO_SE     0
#define DSMGP_KIND_ARD_SE     1   /* additive form, src/kernels.jl:39-49 */
#define DSMGP_KIND_ISO_LINEAR 2

/* per-leaf sharing decisions of the shared-Cholesky fit! (src/fit.jl:107-117) */
#define DSMGP_SHARE_FULL   0      /* update_cholesky!               src/gaussianprocess.jl:82-108 */
#define DSMGP_SHARE_COPY   1      /* identical observation sets     src/fit.jl:132-143 */
#define DSMGP_SHARE_PREFIX 2      /* chol_continue! from column p   src/fit.jl:276-278, src/AdvancedCholeskey.jl:152-174 */

/* error codes */
#define DSMGP_OK            0
#define DSMGP_E_ARG        -1
#define DSMGP_E_STATE      -2
#define DSMGP_E_HIP        -3
#define DSMGP_E_NOMEM      -4
#define DSMGP_E_NODEVICE   -5
#define DSMGP_E_DOMAIN     -6   /* a test row outside the region of a split node (NaN included): the reference loops forever there */

/* number of doubles dsmgp_timings() fills: gram, chol_update (the update launches of the tile kernel alone),
 * chol_diag, chol_trsm, solve (forward substitution of COPY / PREFIX leaves), mll, predict_gram, predict_update,
 * predict_trsm, predict_var, gradients, total_fit, total_predict, chol_reduce (split-K reduce launches of the
 * factorisation), alpha (the backward sweep alpha = L^-T z, run on first use after a fit: gradients, download_factor),
 * grad_inverse (L^-T by blocked triangular inversion), grad_contraction (tile_graddot_kernel), grad_traces,
 * chol_fused (the tile_fused8_kernel launches of fused block steps: update + solve of the tiles below the diagonal blocks),
 * chol_update_union, chol_fused_union (with two leaf lanes the launches of a kind overlap in time: chol_update / chol_fused are the
 * SUM of their durations, these the time during which any of them ran; one lane: the same numbers) */
#define DSMGP_N_TIMINGS 21

/* kernel ids are dense small integers (one hyper-vector each; finetune! gives every leaf its own) */
#define DSMGP_MAX_KERNEL_IDS (1 << 22)

/* ---- context ---------------------------------------------------------------------------------- */
int dsmgp_create(int32_t device_id, dsmgp_ctx** out);
int dsmgp_destroy(dsmgp_ctx* ctx);
const char* dsmgp_last_error(dsmgp_ctx* ctx);     /* ctx may be NULL: error of the failed create */
int dsmgp_device_name(dsmgp_ctx* ctx, char* buf, int32_t len);

/* ---- data: replaces the GaussianProcess constructor's host copies
 *      (src/gaussianprocess.jl:50-80: x, mean-subtracted y; the distance tensor P is never stored) */
int dsmgp_set_train(dsmgp_ctx* ctx, const double* X /* N x D */, const double* y /* N */,
                    int64_t N, int32_t D);

/* leaf table = output of buildTree (src/treeStructure.jl:245-307): obs lists in CSR form
 * (ascending 0-based row indices), kernel id (0-based), ConstMean value per leaf (src/means.jl:7-14) */
int dsmgp_set_leaves(dsmgp_ctx* ctx, int32_t L, const int64_t* obs_ptr /* L+1 */,
                     const int64_t* obs_idx, const int32_t* kernel_id, const double* mean /* L */);

/* sharing schedule decided by the caller exactly as src/fit.jl:78-117 does; NULL op = all FULL.
 * The library validates COPY/PREFIX claims against the obs lists and returns DSMGP_E_ARG on a lie. */
int dsmgp_set_sharing(dsmgp_ctx* ctx, const int32_t* op /* L */, const int32_t* src /* L */,
                      const int64_t* prefix_len /* L */);

/* replaces setparams!(gp, hyper) (src/gaussianprocess.jl:153-161, src/optimize.jl:188-198) for all
 * leaves of one kernel id; n = (#lengthscales) + 2 */
int dsmgp_set_hyper(dsmgp_ctx* ctx, int32_t kernel_id, int32_t kind, const double* loghyp, int32_t n);

/* ---- fit!: Gram assembly + Cholesky + alpha for every leaf
 *      replaces fit!/fit_naive!/update_cholesky! (src/fit.jl:71-122,294-304; src/gaussianprocess.jl:82-108)
 *      and mll(gp) (src/gaussianprocess.jl:163).
 *      mll_out[l]  = -(y.alpha + logdet + n log 2pi)/2, with y.alpha evaluated as |L^-1 y|^2; gp.alpha itself is
 *                    materialised on first use (dsmgp_gradients, dsmgp_download_factor), not by fit
 *      info_out[l] = 0, or k>0 if the leading minor of order k is not positive definite (LAPACK potrf)
 *      seconds     = device time of the call (hipEvents), like the @elapsed value fit! returns */
int dsmgp_fit(dsmgp_ctx* ctx, double* mll_out /* L */, int32_t* info_out /* L */, double* seconds);

/* ---- prediction(gp, xtest) for every (leaf, routed test row): src/gaussianprocess.jl:110-137
 *      mu  = m + Knt' alpha ; var = k(x*,x*) - |L^-1 k_n*|^2 + exp(2 logNoise)   (diag only; no clamp,
 *      the caller applies src/common.jl:137).  route_ptr/route_idx: per leaf, which rows of Xt.
 *      Outputs are aligned with route_idx. */
/* While a test set is resident (dsmgp_set_test), dsmgp_fit also advances its rows through the factorisation
 * launches (V^T = K_tn L^-T rides along as extra row tiles), and dsmgp_predict_run only finishes mu and var.
 * dsmgp_set_joint(ctx, 0) switches that off (e.g. inside train!, where fit is not followed by predict). */
int dsmgp_set_joint(dsmgp_ctx* ctx, int32_t on);
/* D = number of columns of Xt as the caller holds it: DSMGP_E_ARG unless it is the D of dsmgp_set_train (n_t * D doubles are read). */
int dsmgp_set_test(dsmgp_ctx* ctx, const double* Xt /* n_t x D column-major */, int64_t n_t, int32_t D,
                   const int64_t* route_ptr /* L+1 */, const int64_t* route_idx);
/* predict(model, x) on rows the model has not seen (the reference's normal call, src/common.jl:304-307, src/plot.jl:40): the routing
 * of src/common.jl:101-122,181-196,275-292 on the device.  dsmgp_set_tree registers the model's tree once per leaf table (flat
 * arrays as dsmgp_tree_route takes them; leaf_id = index in THIS context's leaf table, -1 = a region another rank holds; a new
 * leaf table or new training data drop it).  dsmgp_set_test_routed(Xt, n_t, D) then is dsmgp_set_test with the routes made on the
 * device: one thread per row walks the tree, a bitmap per leaf turns the visits into the same CSR (rows ascending per leaf) and the
 * same per-row entry index the host path builds -- entry by entry -- and only the L + 1 per-leaf offsets come back to the host
 * (they size the K_tn arena and the sweep's task lists).  DSMGP_E_DOMAIN: a row outside the region of a split node (NaN included).
 * dsmgp_routes fetches the CSR of the registered test set (route_ptr: L + 1; route_idx: route_ptr[L] entries, may be NULL). */
int dsmgp_set_tree(dsmgp_ctx* ctx, int64_t n_nodes, const int8_t* kind, const int64_t* first_child, const int64_t* n_child,
                   const int64_t* split_dim, const double* thr, int64_t thr_ld, const int64_t* leaf_id);
int dsmgp_set_test_routed(dsmgp_ctx* ctx, const double* Xt /* n_t x D column-major */, int64_t n_t, int32_t D);
int dsmgp_routes(dsmgp_ctx* ctx, int64_t* route_ptr /* L+1 */, int64_t* route_idx /* or NULL */);
int dsmgp_predict_run(dsmgp_ctx* ctx, double* seconds);   /* device work only, inputs resident */
int dsmgp_predict_fetch(dsmgp_ctx* ctx, double* mu_out, double* var_out);
int dsmgp_predict_leaves(dsmgp_ctx* ctx, const double* Xt, int64_t n_t, int32_t D, const int64_t* route_ptr,
                         const int64_t* route_idx, double* mu_out, double* var_out);

/* ---- predict(model, x): sum/product aggregation of the leaf moments over the leaves every test row visits, on the
 *      moments the last dsmgp_predict_run left in HBM (replaces the host recursions of src/common.jl:134-149,198-302).
 *      family  DSMGP_AGG_MIXTURE  DSMGP (_predict / _minpredict, :134-143,151-196,275-302): leaf_coef[l] = W_l, the product of the
 *                                 sum-node weights exp(logweights) on leaf l's path; mu = sum W mu_l,
 *                                 var = sum W sigma2_l + sum W mu_l^2 - mu^2 with sigma2 <= 0 -> 1e-8 (:137)
 *              DSMGP_AGG_POE / DSMGP_AGG_GPOE   (:145-149,198-222): leaf_coef[l] = beta_l (1, resp. 1/#root children)
 *              DSMGP_AGG_RBCM     (:224-241): leaf_group[l] = root child of leaf l (n_groups of them);
 *                                 prior_kernel_id = kernel id of the model's first leaf (leftGP, :227)
 *      plain != 0: the root is a single GP (predict(node::GPNode), :175-179).
 *      dsmgp_aggregate = partial + finish on one context.  With leaves spread over ranks or contexts every holder
 *      calls dsmgp_aggregate_partial (partial_out: W x n_t sums, W = 3 mixture / 2 PoE, gPoE / 2 n_groups rBCM), the
 *      caller adds the partial sums (the multi-GPU exchange: one all-gather of W n_t doubles per rank) and hands the
 *      total to dsmgp_aggregate_finish.  mu_out / var_out (n_t each) may be NULL: results stay resident for dsmgp_scores. */
#define DSMGP_AGG_MIXTURE 0
#define DSMGP_AGG_POE     1
#define DSMGP_AGG_GPOE    2
#define DSMGP_AGG_RBCM    3
int dsmgp_aggregate(dsmgp_ctx* ctx, int32_t family, const double* leaf_coef /* L */, const int32_t* leaf_group /* L or NULL */,
                    int32_t n_groups, int32_t plain, int32_t prior_kernel_id, double* mu_out, double* var_out);
int dsmgp_aggregate_partial(dsmgp_ctx* ctx, int32_t family, const double* leaf_coef, const int32_t* leaf_group,
                            int32_t n_groups, double* partial_out /* W x n_t or NULL */);
int dsmgp_aggregate_finish(dsmgp_ctx* ctx, const double* partial_in /* NULL: the context's own sums */, int32_t plain,
                           int32_t prior_kernel_id, double* mu_out, double* var_out);
/* ---- score functions of src/scorefunctions.jl:6-16 on the aggregated prediction still in HBM:
 *      out[5] = { mse, sse (std(se)/sqrt(n)), mae, sae, nlpd } */
int dsmgp_scores(dsmgp_ctx* ctx, const double* y_test /* n_t */, double* out /* 5 */);

/* ---- updategradients!(gp) + grad vector of src/gaussianprocess.jl:165-178,185-217 per leaf.
 *      grad_out[l*stride + j], j over [dl..., ds, dnoise] (reference order, src/gaussianprocess.jl:212-214),
 *      reproducing the reference's scaling and ArdSE dl == 0. */
int dsmgp_gradients(dsmgp_ctx* ctx, double* grad_out, int32_t stride);
/* Restricts dsmgp_gradients to the leaves with active[l] != 0 (NULL: every leaf again; a new leaf table resets it): the rows
 * of the others come back as zeros, and neither L^-T nor the contraction tiles of leaves nobody asked for are computed (a
 * COPY leaf's source and every active leaf's factor owner are included as needed).  finetune! weights leaf l's gradient by
 * the overlap D[j, l] while it moves leaf j's vector (src/optimize.jl:101, src/finetuning.jl:34-57): all but the overlapping
 * leaves are multiplied by zero there.  Changing the set rebuilds the task lists of the gradient pass (the L^-T arena stays). */
int dsmgp_set_gradient_leaves(dsmgp_ctx* ctx, const int32_t* active /* L flags, or NULL */);
/* Options.  DSMGP_OPT_ARD_LENGTHSCALE_GRADIENT: 0 (default) = ArdSE length-scale gradients exactly as the reference
 * computes them, i.e. identically zero (`precomp * K .* (p/ls[d])` parses as `(precomp*K) .* (p/ls[d])` and p has a zero
 * diagonal, src/kernels.jl:161: train!/finetune! never move ARD length-scales); 1 = the true derivative of the
 * log-marginal, dl_d = 0.5 tr((alpha alpha^T - K_y^-1) dK/dlog l_d) with dK/dlog l_d = sigma^2 exp(-u_d^2/2l_d^2) u_d^2/l_d^2
 * of the additive kernel (no extra factor sigma; costs the contraction pass, n^3/3 flops per leaf; D <= 35). */
#define DSMGP_OPT_ARD_LENGTHSCALE_GRADIENT 1
/* DSMGP_OPT_FUSED_GRAM: 1 (default) = the update tasks of fit! evaluate the kernel function for their tile themselves
 * (same operations as the Gram launch, bit-identical values) instead of reading what a Gram launch wrote, which then
 * covers block column 0 only (D <= 32; above that the option has no effect); 0 = every lower tile of K_y goes through
 * memory first.  Changing it discards the leaf plan and a registered test set: set it before dsmgp_set_test. */
#define DSMGP_OPT_FUSED_GRAM 2
/* DSMGP_OPT_FUSED_STEPS: 1 (default) = a block step of the factorisation whose diagonal blocks alone fill the chip (more
 * leaves in the step than CUs: depth >= 3 trees, large PoE models), and every shallow step (K <= 512), runs as two
 * launches -- the diagonal block's task also updates its tile, the tasks of the tiles below update AND solve them, each
 * tile written once (the per-step order of src/AdvancedCholeskey.jl:161-171, batched over leaves); 0 = every step as
 * update / diagonal block / panel solve launches.  Diagonal blocks come out bit-identical either way; the tiles below
 * agree to rounding (a fused task accumulates the product on -k(row, col) where the classic update subtracts the finished
 * product from k(row, col)).  Needs DSMGP_OPT_FUSED_GRAM (D <= 32); changing it discards the leaf plan and a registered
 * test set. */
#define DSMGP_OPT_FUSED_STEPS 3
/* DSMGP_OPT_DIAG_IN_UPDATE: 1 (default) = where two consecutive block steps are classic, the diagonal block runs one step
 * ahead of the panel below it (the lookahead of a blocked factorisation, src/AdvancedCholeskey.jl:161-171 per step): the update
 * launch of step k - 1 also updates tile (k, k) over the columns it covers, and the update launch of step k carries a task
 * per leaf that applies the last block column, factorises the block and inverts it beside the updates of the tiles below,
 * so the panel-solve launch finds L_kk, Dinv_k and z_k ready and the per-step chain loses a dependent launch (35 us x every
 * block step of the deepest leaf); 0 = update / reduce / diagonal block / panel solve launches one after the other.
 * Results agree to rounding (the diagonal tile's update is summed in two parts); a fit is bit-reproducible either way.
 * Needs DSMGP_OPT_FUSED_GRAM (D <= 32); changing it discards the leaf plan and a registered test set. */
#define DSMGP_OPT_DIAG_IN_UPDATE 4
/* DSMGP_OPT_FIT_GRAPH: 1 = while per-launch timing is off (dsmgp_set_profile 0) dsmgp_fit replays its launch sequence as a
 * captured hipGraph (captured on the first such fit of a plan, dropped with the plan, the test set or a re-allocated
 * kernel-parameter table).  Same kernels, same arguments, same results to the bit.  0 (default) = plain launches: what the
 * graph buys is the host-side cost between dependent launches, measured at 2 % of a single GP's 32-step chain (config 2: 2.50
 * -> 2.47 ms) and nothing elsewhere, and a capture does not tolerate other contexts being driven from concurrent host threads
 * in the same process. */
#define DSMGP_OPT_FIT_GRAPH 5
/* DSMGP_OPT_LANES: leaf lanes of a fit (src/fit.jl:88-119: the leaves are independent): 0 (default) = automatic -- two lanes from 8
 * sharing groups on (a source leaf with its COPY / PREFIX leaves is one group), one below (a single GP) -- 1 = one lane, 2 = two.
 * With two lanes the leaves are dealt longest-processing-time first into two halves with step lists, split-K workspace and HIP
 * stream of their own, joined at the end of the factorisation: one half's latency-bound launches (diagonal blocks, panel solves,
 * reduces) run under the other's update launches.  Per-leaf results agree to rounding with the one-lane schedule (a launch of half
 * the tiles cuts its tail along K differently: the order of a few additions per entry), a fit is bit-reproducible either way.
 * Changing it discards the leaf plan and a registered test set. */
#define DSMGP_OPT_LANES 6
int dsmgp_set_option(dsmgp_ctx* ctx, int32_t option, int32_t value);
/* leaf lanes of the current plan (DSMGP_OPT_LANES: 1 or 2; 0 = no plan yet: it is made by the first fit / set_test of a leaf table) */
int dsmgp_lanes(dsmgp_ctx* ctx, int32_t* lanes);

/* ---- inspection ------------------------------------------------------------------------------- */
/* kernelmatrix(kernel, x1, x2) (src/kernels.jl:15-18) through the same device code as the fit path */
int dsmgp_kernel_matrix(dsmgp_ctx* ctx, int32_t kernel_id, const double* x1, int64_t n1,
                        const double* x2, int64_t n2, double* K_out /* n1 x n2 */);
/* gp.cK.factors (lower triangle = L, strict upper = 0) and gp.alpha of one leaf; either may be NULL */
int dsmgp_download_factor(dsmgp_ctx* ctx, int32_t leaf, double* F /* n x n */, double* alpha /* n */);
/* hipEvent timing per launch: level 0 = totals only, 1 = the update launches of the factorisation (the dominant
 * kernel; what bench.py's roofline uses), 2 = every kernel category (adds event records between all launches;
 * also switched on by the environment variable DSMGP_PROFILE=1 at dsmgp_create -- the only environment variable the
 * product library reads besides DSMGP_STEPLOG / DSMGP_HOSTLOG (stderr logging); none of them changes results);
 * 3 = level 1 with the launches under the kernel instantiation names of level 0 (the timed launches of bench.py run as
 * tile_gemm_kernel_v2<false, 0, *> / tile_fused8_kernel<0>, every other launch as <false, 2, *> / <1> (sweeps) / <2>: a profiler's
 * per-kernel average of the first names is exactly the timed quantity) */
int dsmgp_set_profile(dsmgp_ctx* ctx, int32_t level);
int dsmgp_timings(dsmgp_ctx* ctx, double* out /* DSMGP_N_TIMINGS, seconds of the last fit/predict */);
/* work of the dominant kernel (the f64-MFMA Cholesky update) in the last fit: algorithmic flops over
 * all its launches (2*K per lower-triangle element of every block column, unpadded sizes) and the
 * number of launches */
int dsmgp_work(dsmgp_ctx* ctx, double* alg_flops_update, int32_t* n_update_launches);
/* the same for the tile_fused8_kernel launches of fused block steps (timing slot chol_fused): algorithmic flops of their
 * update part plus the triangular solves (c_k^2 per row below a diagonal block of c_k columns), and the number of launches.
 * dsmgp_work counts only the steps that run as update launches. */
int dsmgp_work_fused(dsmgp_ctx* ctx, double* alg_flops_fused, int32_t* n_fused_launches);
/* algorithmic flops of the two matrix passes of dsmgp_gradients (n^3/3 each per leaf) and the number of contraction tiles */
int dsmgp_work_gradients(dsmgp_ctx* ctx, double* alg_flops_inverse, double* alg_flops_contraction, int32_t* n_contraction_tiles);
/* Reserve one device pool of `bytes` (0 = drop it): while a pool exists, the large arenas of every following leaf
 * table (factors, inverse blocks, K_tn rows, L^-1 for gradients, split-K slabs) are carved out of it instead of being
 * allocated and freed per table -- the driver clears memory on allocation, ~5 s per 230 GB.  Used by the
 * factor-and-discard mode (hipabi.StreamingContext).  Drops the current leaf table's device state. */
int dsmgp_reserve(dsmgp_ctx* ctx, int64_t bytes);

/* streaming ("factor and discard"): drop every buffer that scales with the leaf sizes, keep data, leaf table,
 * schedule and hyper-parameters; the next dsmgp_fit rebuilds.  dsmgp_estimate_bytes sizes a leaf group beforehand. */
int dsmgp_release(dsmgp_ctx* ctx);
int64_t dsmgp_estimate_bytes(int32_t L, const int64_t* n, const int64_t* n_test /* may be NULL */, int32_t D,
                             int32_t with_gradients);
/* bytes of device memory the current leaf table needs / the device has free */
int dsmgp_memory(dsmgp_ctx* ctx, int64_t* needed, int64_t* free_bytes);

/* f64 MFMA issue-rate probe used for the roofline peak (bench.py): returns TFLOP/s of a register-only
 * v_mfma_f64_16x16x4_f64 loop over the whole chip */
int dsmgp_probe_f64_mfma(dsmgp_ctx* ctx, double* tflops);
/* out[0] TFLOP/s, out[1] shader cycles per MFMA per wave, out[2] shader clock (GHz) held in the loop,
 * out[3] = blocks_per_cu (256-thread workgroups per CU, i.e. waves per SIMD) */
int dsmgp_probe_f64_mfma_detail(dsmgp_ctx* ctx, int32_t blocks_per_cu, double* out);

/* Shader clock the chip holds under load (bench.py prints it beside the roofline: the f64 matrix peak scales with it).
 * start: a one-wave kernel on a stream of its own sleeps for `milliseconds` (<= 5000) of wall time beside whatever the context
 * launches meanwhile and counts shader cycles; returns at once.  read: waits for it; ghz = shader cycles per nanosecond over
 * the `milliseconds` it actually covered.  One sample at a time per context. */
int dsmgp_clock_sample_start(dsmgp_ctx* ctx, double milliseconds);
int dsmgp_clock_sample_read(dsmgp_ctx* ctx, double* ghz, double* milliseconds);

/* Host-only (no device): per leaf j the "main" leaf of the sharing schedule of src/fit.jl:78-86,
 * main[j] = argmax_i D[i,j] D[j,i] over the overlap matrix D of src/fit.jl:12-39 (first maximum, 0 if leaf j overlaps
 * no other leaf), and c_main[j] = number of observations leaf j shares with it -- computed from an inverted index
 * instead of the dense L x L matrix (18k leaves at depth 4).  Leaves as in dsmgp_set_leaves; one kernel id. */
int dsmgp_overlap_main(int32_t L, const int64_t* obs_ptr, const int64_t* obs_idx, int64_t N, int64_t* main_out,
                       int64_t* c_main_out);

/* Host-only: which test rows each leaf is asked to predict (the routing of predict, src/common.jl:181-196,275-292: a sum node
 * forwards its rows to every child, a split node to the first child k with x[d] <= s_k) -- the CSR that dsmgp_set_test takes.
 * The tree as flat arrays with the children of node i at first_child[i] .. first_child[i] + n_child[i] - 1 (root = node 0):
 * kind as in dsmgp_tree_export: 0 = region, i.e. leaf (leaf_id), 1 = split node (split_dim, ascending thresholds
 * thr[i * thr_ld + 0 .. n_child[i] - 1]), 2 = sum node.
 * x: n_t rows of D columns, element (row r, dimension d) at x[r * row_stride + d * col_stride].  route_ptr: n_leaves + 1 entries; route_idx: `capacity`
 * entries for the rows of every leaf, ascending (a row reaches at most as many leaves as the tree has below sum nodes along one
 * path: n_t times that bound always suffices).  DSMGP_E_ARG: malformed tree or a split dimension >= D (a tree of any width and depth is taken: the
 * walk's stack is sized by the tree); DSMGP_E_DOMAIN: a row
 * beyond the last threshold of a split node (the reference loops forever there; a NaN coordinate is beyond every threshold); DSMGP_E_NOMEM: capacity too small -- route_ptr and *n_routes_out are valid, route_idx
 * is not.  At depth 4 (18k leaves, 10k rows to 81 leaves each) the recursion over node objects took 0.09 s on the host, four
 * times the prediction sweep it feeds. */
int dsmgp_tree_route(int64_t n_nodes, const int8_t* kind, const int64_t* first_child, const int64_t* n_child,
                     const int64_t* split_dim, const double* thr, int64_t thr_ld, const int64_t* leaf_id, int64_t n_leaves,
                     const double* x, int64_t n_t, int64_t D, int64_t row_stride, int64_t col_stride, int64_t* route_ptr,
                     int64_t* route_idx, int64_t capacity, int64_t* n_routes_out);

/* ---- multi-GPU: the one exchange step of the path.  Leaves are independent, every rank (one process
 *      per GPU, one context each) fits and predicts its own shard; what crosses GPUs is one all-gather of per-leaf
 *      log-marginals after dsmgp_fit and one of the aggregation's partial sums after dsmgp_aggregate_partial, over RCCL
 *      (xGMI inside a node) on the context's stream.  librccl.so is dlopen'ed on first use.
 *      dsmgp_comm_unique_id: rank 0 obtains the 128-byte ncclUniqueId and hands it to the other ranks by whatever
 *      channel the host has (MPI.jl bcast, a file, the Distributed stdlib); every rank then calls dsmgp_comm_init.
 *      dsmgp_allgather: `count` doubles from every rank, recv[r * count ...] = rank r's block (host buffers; blocking). */
int dsmgp_comm_unique_id(char* id_out /* 128 bytes */);
int dsmgp_comm_init(dsmgp_ctx* ctx, int32_t rank, int32_t world, const char* id /* 128 bytes */);
int dsmgp_allgather(dsmgp_ctx* ctx, const double* send, int64_t count, double* recv /* world x count */);
/*      The two exchanges of the path with the payload staying in HBM until after the collective:
 *      dsmgp_fit_exchange: per-leaf (log-marginal, info) of the last dsmgp_fit, straight from the device results, padded to
 *        `count` leaves per rank (count >= this rank's leaf count; a rank without leaves passes its context with no leaf
 *        table and contributes zeros): out[(r * count + l) * 2 + {0, 1}] = mll / info of rank r's leaf l.
 *      dsmgp_aggregate_exchange: after dsmgp_aggregate_partial on every rank, all-gathers the W x n_t partial sums and
 *        adds them in rank order on the device (the same bits on every rank); dsmgp_aggregate_finish(ctx, NULL, ...)
 *        then finishes from the total (a second exchange of the same partial sums returns DSMGP_E_STATE: they already hold
 *        the total).  A rank without leaves calls dsmgp_aggregate_exchange_empty(ctx, W, n_t) instead
 *        (it contributes zeros and receives the total in `total_out`, W x n_t doubles, may be NULL). */
int dsmgp_fit_exchange(dsmgp_ctx* ctx, int64_t count, double* out /* world x count x 2 */);
int dsmgp_aggregate_exchange(dsmgp_ctx* ctx, double* total_out /* W x n_t, may be NULL */);
int dsmgp_aggregate_exchange_empty(dsmgp_ctx* ctx, int32_t W, int64_t n_t, double* total_out /* may be NULL */);
int dsmgp_comm_destroy(dsmgp_ctx* ctx);

/* Host-only (no device): the random partition tree of buildTree (src/treeStructure.jl:4-307: getSplits, _buildSplit,
 * _buildSum and the regions _buildGP turns into leaves) as one native recursion, drawing from the portable counter stream
 * `seed` (SplitMix64 in counter mode, deepstructuredmixtures_amd/datagen.py) in the order of the interpreted builder
 * (tree.py), with which it agrees bit for bit.  n_splits = config.K (cuts per split node follow the depth^2 rule of
 * :33,69,83), n_sum_children = config.V, depth = config.depth, bnoise = the eps of :49-54, n_kernels > 0: every region is
 * a sum over n_kernels GPs and n_kernels uniforms are drawn for its Dirichlet(1) weights (:258-261).
 * Result: nodes in creation (pre-)order: kind 0 region / 1 split / 2 sum, parent, split dimension, bounds lb/ub (D per
 * node), split thresholds (CSR, last = upper bound), observation lists of the regions (CSR, ascending row indices),
 * and the Dirichlet uniforms in region order. */
typedef struct dsmgp_tree dsmgp_tree;
int dsmgp_tree_build(const double* X /* N x D */, int64_t N, int32_t D, int32_t min_data, int32_t n_splits,
                     int32_t n_sum_children, int32_t depth, double bnoise, int32_t sum_root, int32_t n_kernels, uint64_t seed,
                     dsmgp_tree** out);
int dsmgp_tree_sizes(const dsmgp_tree* t, int64_t* n_nodes, int64_t* n_thr, int64_t* n_obs, int64_t* n_dir);
int dsmgp_tree_export(const dsmgp_tree* t, int32_t* kind, int32_t* parent, int32_t* split_dim, double* lb, double* ub,
                      int64_t* thr_ptr, double* thr, int64_t* obs_ptr, int64_t* obs, double* dir_u);
/* Mean of y over the observation list of every region, in region (creation) order: the ConstMean of a leaf built
 * without a mean function (src/treeStructure.jl:253).  Summed as NumPy sums a contiguous vector (pairwise), so the
 * value equals mean(y[obs]) of the interpreted builder bit for bit.  mean_out: one double per region (kind 0 node). */
int dsmgp_tree_means(const dsmgp_tree* t, const double* y, int64_t N, double* mean_out);
int dsmgp_tree_free(dsmgp_tree* t);

#ifdef __cplusplus
}
#endif
#endif
