// Device kernels of the GP-expert hot path for gfx950 (MI355X, wave64, f64 MFMA).
//
// Everything here works on 128x128 Float64 tiles of column-major matrices whose dimensions are
// padded to multiples of 128 (padding = identity block, see gram_tile_kernel).  One leaf GP's
// factor F (npad x npad) is produced by a LEFT-LOOKING blocked Cholesky, batched over all leaves:
//   step k:  F[i,k] -= F[i,0:k] F[k,0:k]^T   (tile_gemm_kernel_v2, v_mfma_f64_16x16x4_f64; diagonal tiles optionally by
//                                              tile_syrk_body; K-split pieces summed by tile_reduce_kernel)
//            F[k,k]  = chol(F[k,k]), Dinv_k = F[k,k]^-1   (chol_diag_packed_kernel: 75 KB LDS image, two per CU)
//            F[i,k]  = F[i,k] Dinv_k^T       (tile_trsm_kernel: triangular product; fused forward solve and, for test
//                                              rows, the predictive moments)
// which is update_cholesky!/potrf! of the reference (src/gaussianprocess.jl:82-108) and, started at
// a later column with the leading block copied, chol_continue! (src/AdvancedCholeskey.jl:152-174).
// prediction() (src/gaussianprocess.jl:110-137) appends the test rows below the factor: V^T = K_tn L^-T
// is the same two tile kernels run on the rows of K_tn.  Aggregation over the leaves of a test row and the score
// functions (src/common.jl:134-302, src/scorefunctions.jl) are the agg_* kernels at the end of this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "route_walk.hpp"

namespace dsmgp {

constexpr int TB = 128;    // tile edge = Cholesky block size
constexpr int KC = 16;     // K-chunk staged through LDS per iteration
constexpr int LDP = 144;   // LDS leading dimension (doubles): 128 + 16 keeps the f64 MFMA operand reads conflict-free
constexpr int DCH = 8;     // input dimensions staged per pass in the Gram kernel

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
// pointers read from task structs are generic to the compiler; the matrices live in global memory, and
// saying so turns flat_load/flat_store into global_load/global_store with counted vmcnt waits
typedef const d2 __attribute__((address_space(1)))* gd2_cptr;
typedef double __attribute__((address_space(1)))* gf64_ptr;
#define AS_GLOBAL_D2(p) (reinterpret_cast<gd2_cptr>(reinterpret_cast<uintptr_t>(p)))
#define AS_GLOBAL_F64(p) (reinterpret_cast<gf64_ptr>(reinterpret_cast<uintptr_t>(p)))
// the hyper-parameter tables (KParam.l2 / nh) are written by the host between launches only and indexed wave-uniformly: in the
// constant address space the compiler reads them with scalar loads (a flat load per dimension kept an address pair and a result
// pair of VECTOR registers alive across the loop over the dimensions of the additive ArdSE kernel)
typedef const double __attribute__((address_space(4)))* cf64_ptr;
#define AS_CONST_F64(p) (reinterpret_cast<cf64_ptr>(reinterpret_cast<uintptr_t>(p)))

// ---------------------------------------------------------------------------------------------
// kernel-function parameters (one per kernel id), derived on the host from the log-scale vector
// [logl..., logs, logNoise]  (src/kernels.jl:68-73, src/gaussianprocess.jl:39)
struct KParam {
    int kind;           // DSMGP_KIND_*
    int nl;             // number of lengthscales (1 for Iso, D for Ard)
    double sigma2;      // exp(2 logs)  (1.0 for IsoLinear, src/kernels.jl:181)
    double sigma;       // exp(logs)
    double noise;       // exp(2 logNoise)
    const double* l2;   // device: lengthscale^2 per slot
    const double* nh;   // device: -0.5 / lengthscale^2 per slot (the factor of the exponent)
    double nh0;         // nh[0]
    double il2;         // 1 / l2[0]
};

// exp(x) for finite x <= 0: the argument reduction and degree-12 polynomial of the device library's exp
// (n = rint(x log2 e), r = x - n ln2 in two pieces, Horner, ldexp) without its overflow / underflow selects --
// the argument of a squared-exponential kernel is never positive, and ldexp underflows to 0 by itself.
// Same bits as exp() on (-745, 0].
__device__ __forceinline__ double exp_nonpos(double x) {
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(n, -0x1.62e42fefa39efp-1, x);
    r = fma(n, -0x1.abc9e3b39803fp-56, r);
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma(r, p, 0x1.71dee623fde64p-19);
    p = fma(r, p, 0x1.a01997c89e6b0p-16);
    p = fma(r, p, 0x1.a01a014761f6ep-13);
    p = fma(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma(r, p, 0x1.1111111122322p-7);
    p = fma(r, p, 0x1.55555555502a1p-5);
    p = fma(r, p, 0x1.5555555555511p-3);
    p = fma(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)n);
}


// ---------------------------------------------------------------------------------------------
// Gram tiles.  out(r,c) = k(a_r, b_c); rows/cols beyond the valid counts are 0, and with `sym` the
// tile belongs to K_nn: the global diagonal gets + noise + 1e-8 (src/gaussianprocess.jl:94-98,
// eps = src/DeepStructuredMixtures.jl:27) and padded diagonal entries are 1 (identity padding).
struct GramTask {
    const double* xa;   // a-points, column-major [lda x D], already offset to the tile's first row
    const double* xb;
    double* out;        // tile origin
    int lda, ldb, ldo;
    int na, nb;         // valid rows / cols in this tile (<= 128)
    int sym;            // 1: tile of a symmetric K_nn
    int diag;           // 1: tile sits on the block diagonal (global row == global col possible)
    int kid;
};

// One 256-thread workgroup per 128x64 half tile (blockIdx = 2*task + half): thread t owns rows
// 4*(t&31)..+3 and columns 64*half + (t>>5) + 8q, q<8, so every column is written as 32 threads x 32 B =
// 1 KiB contiguous, and the 32 accumulators keep the kernel at 4 waves per SIMD.
// IsoSE follows src/kernels.jl:21-27,78-83 as exp(z * (-0.5/l^2)) then * sigma^2 (the reference divides by l^2 per entry;
// one rounding of the exponent's argument apart: <= 4e-15 relative on the kernel value), with z accumulated
// from direct differences (the reference's Distances.pairwise uses |a|^2+|b|^2-2a.b; same value up to
// rounding).  ArdSE is the additive form sigma^2 * sum_d exp(-0.5 (a_d-b_d)^2 / l_d^2)
// (src/kernels.jl:39-49).  IsoLinear is a.b / l^2 (src/kernels.jl:189-194).
template <int KIND>
__device__ __forceinline__ void gram_half_tile(const GramTask& tk, const KParam& p, int D, int half,
                                               double (*sa)[TB], double (*sb)[TB / 2]) {
    const int t = threadIdx.x;
    const int r0 = (t & 31) * 4;
    const int cb = t >> 5;
    const int c0 = 64 * half;
    // K_tn tiles (sym == 0): the four rows of this thread are all padding -> nothing to compute or store.  What the padding rows
    // of the K_tn arena hold never reaches a result (a row of a tile product depends on its own operand row only, the riders sum
    // per row, nothing reads beyond a leaf's routed rows): since round 5 the arena is not even cleared (register_test)
    const bool rows_live = tk.sym != 0 || r0 < tk.na;
    double acc[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = 0.0;

    for (int d0 = 0; d0 < D; d0 += DCH) {
        const int dn = min(DCH, D - d0);
        __syncthreads();
        for (int e = t; e < dn * TB; e += 256) {
            const int d = e / TB, r = e % TB;
            sa[d][r] = (r < tk.na) ? tk.xa[r + (size_t)(d0 + d) * tk.lda] : 0.0;
        }
        for (int e = t; e < dn * (TB / 2); e += 256) {
            const int d = e / (TB / 2), r = e % (TB / 2);
            sb[d][r] = (c0 + r < tk.nb) ? tk.xb[c0 + r + (size_t)(d0 + d) * tk.ldb] : 0.0;
        }
        __syncthreads();
        if (!rows_live) continue;
        for (int d = 0; d < dn; ++d) {
            const double a0 = sa[d][r0], a1 = sa[d][r0 + 1], a2 = sa[d][r0 + 2], a3 = sa[d][r0 + 3];
            const double nhd = (KIND == 1) ? p.nh[d0 + d] : 0.0;   // -0.5 / l_d^2
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double b = sb[d][cb + 8 * q];
                if (KIND == 0) {
                    double u;
                    u = a0 - b; acc[q][0] = fma(u, u, acc[q][0]);
                    u = a1 - b; acc[q][1] = fma(u, u, acc[q][1]);
                    u = a2 - b; acc[q][2] = fma(u, u, acc[q][2]);
                    u = a3 - b; acc[q][3] = fma(u, u, acc[q][3]);
                } else if (KIND == 1) {
                    double u;
                    u = a0 - b; acc[q][0] += exp_nonpos((u * u) * nhd);
                    u = a1 - b; acc[q][1] += exp_nonpos((u * u) * nhd);
                    u = a2 - b; acc[q][2] += exp_nonpos((u * u) * nhd);
                    u = a3 - b; acc[q][3] += exp_nonpos((u * u) * nhd);
                } else {
                    acc[q][0] = fma(a0, b, acc[q][0]);
                    acc[q][1] = fma(a1, b, acc[q][1]);
                    acc[q][2] = fma(a2, b, acc[q][2]);
                    acc[q][3] = fma(a3, b, acc[q][3]);
                }
            }
        }
    }
    if (!rows_live) return;
    const double nh = p.nh0, il2 = p.il2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = c0 + cb + 8 * q;
        d4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + j;
            double kv;
            if (KIND == 0) kv = p.sigma2 * exp_nonpos(acc[q][j] * nh);
            else if (KIND == 1) kv = p.sigma2 * acc[q][j];
            else kv = acc[q][j] * il2;
            const bool valid = (r < tk.na) && (c < tk.nb);
            if (!valid) kv = 0.0;
            if (tk.sym && tk.diag && r == c) kv = valid ? kv + (p.noise + 1e-8) : 1.0;
            v[j] = kv;
        }
        *reinterpret_cast<d4*>(tk.out + r0 + (size_t)c * tk.ldo) = v;
    }
}

__global__ __launch_bounds__(256) void gram_tile_kernel(const GramTask* __restrict__ tasks,
                                                        const KParam* __restrict__ kp, int D) {
    __shared__ double sa[DCH][TB];
    __shared__ double sb[DCH][TB / 2];
    const GramTask tk = tasks[blockIdx.x >> 1];
    const KParam p = kp[tk.kid];
    const int half = blockIdx.x & 1;
    if (p.kind == 0) gram_half_tile<0>(tk, p, D, half, sa, sb);
    else if (p.kind == 1) gram_half_tile<1>(tk, p, D, half, sa, sb);
    else gram_half_tile<2>(tk, p, D, half, sa, sb);
}

// ---------------------------------------------------------------------------------------------
// 128x128 tile GEMM on the f64 matrix cores:  C = (update ? C : 0) -/+ sum_{kk in [k0,k1)} A(:,kk) B(:,kk)^T
//   A(r,kk) = A[r + kk*lda]   (tile rows, r < 128)      B(c,kk) = B[c + kk*ldb]   (tile columns)
// 4 waves in a 2x2 grid, each owning a 64x64 sub-tile = 4x4 MFMA tiles of 16x16.
// v_mfma_f64_16x16x4_f64: lane l supplies Aop[i=l&15][k=l>>4], Bop[k=l>>4][j=l&15] and receives
// D[(l>>4)+4r][l&15], r<4.  The tile COLUMN index is put on the MFMA row (Aop <- B matrix) and the tile
// ROW index on the MFMA column (Bop <- A matrix), so that the 16 lanes l&15 of a result register are 16
// consecutive rows = 128 contiguous bytes of the column-major tile.
// One task = one tile and one K range.  Three uses:
//   update : F[i,k] -= F[i,0:K] F[k,0:K]^T                       (update = 1, whole K range)
//   partial: slab    = F[i,Ka:Kb] F[k,Ka:Kb]^T                    (update = 0; split-K, summed by tile_reduce_kernel)
//   solve  : F[i,k]  = F[i,k] Dinv_k^T                            (update = 0, K = 128)
struct TileTask {
    const double* A;
    const double* B;
    double* C;
    const double* zk;   // panel-solve tasks: z_k = L_kk^-1 w_k (128) ...
    double* wi;         // ... train rows: right-hand side block of this row tile, w_i -= X z_k (fused forward solve);
                        //     test rows (sq set): running predictive mean, wi += X z_k.  NULL = no rider
    double* sq;         // test rows: running sum of squares of the solved row, sq += rowsumsq(X); else NULL
    int lda, ldb, ldc;
    int k0, k1;         // K range, multiples of 8
    int update;         // 0 = store the product, 1 = C - product, 2 = -product (C is not read)
    int sym;            // 1: diagonal tile of the factorisation, B == A (C -= A A^T): only the lower 16x16 blocks are
                        //    computed and written (tile_syrk_body); the strictly upper blocks of C are left alone
    int mrows;          // rows of the tile that hold data (rows >= mrows are padding: zero rows of A, whose product is
                        //    zero); 0 = all 128.  Update tasks with mrows <= 96 run in the column-split form (tile_rows_body):
                        //    a test-row tile of a small leaf (44 routed rows padded to 128) costs 3/8 of a tile; panel solves
                        //    skip the waves whose 32 rows are all padding
    // Gram matrix fused into the update (gram != 0): the tile is NOT read -- the kernel evaluates k(row, column) where
    // it would have loaded C(row, column), with the operations of gram_half_tile in the same order (bit-identical
    // values), and stores k - product (update = 1) or product - k (update = 0: first piece of a split tile, whose
    // ReduceTask is `fresh`).  The Gram launch of fit! then covers only the tiles no update task writes (block column 0).
    int rev;            // 1: the K range is streamed from its END (chunks k1-8, k1-16, ...).  Tiles that share a B panel
                        //    but start at different columns (the blocks of L^-T: row tile t starts at column 128 t, all end
                        //    at the step's column) then read the same chunks of it at the same time and share them through
                        //    L2; streamed from their own first columns they never meet
    int gram;           // bit 0: on; bit 1: tile on the block diagonal of K_nn (noise + eps on the diagonal, identity padding);
                        //    bit 2: short tiles (tile_rows_body) must write the padding rows of C too (tiles of the factor: zeros,
                        //    ones on the diagonal of a diagonal tile; the rows of K_tn already hold theirs)
    int kid;            // kernel id of the leaf (index into the KParam table)
    int pad;
    const double* gxa;  // coordinates of the tile's rows, column-major [glda x D], offset to the tile's first row
    const double* gxb;  // ... of its columns
    int glda, gldb;
    int gna, gnb;       // valid rows / columns (<= 128)
};
static_assert(sizeof(TileTask) == 128, "TileTask is read with scalar loads: keep it two cache lines");
constexpr int GRAM_FUSE_MAX_D = 32;   // the coordinates of 128 rows and 128 columns go through the ring's LDS

// diagnostic builds (-DDSMGP_DIAG, tools/bench_tile.py) stamp shader cycles; never executed by fit!/predict
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Coordinates of a fused-Gram tile -> LDS: sa[d * TB + r] = x_row(r)[d], sb likewise for the columns (zeros beyond the
// valid counts).  The ring is free once the main loop has passed its last barrier.
// Four passes of the block at a time, loads first: written as one load and one LDS store per pass the loop waits for every
// pass's memory round trip on its own (D / 2 of them per task at D = 8).
#ifndef DSMGP_COORDS_BATCH
#define DSMGP_COORDS_BATCH 1
#endif
template <int NT = 256>
__device__ __forceinline__ void stage_coords(const double* gxa, int glda, int gna, const double* gxb, int gldb, int gnb, int D,
                                             double* sa, double* sb, bool cols) {
    const int t = threadIdx.x;
    const int n = D * TB;
#if DSMGP_COORDS_BATCH
    for (int e0 = t; e0 < n; e0 += 4 * NT) {
        double va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + NT * u;
            const int d = e >> 7, r = e & (TB - 1);
            va[u] = (e < n && r < gna) ? gxa[r + (size_t)d * glda] : 0.0;
            vb[u] = (cols && e < n && r < gnb) ? gxb[r + (size_t)d * gldb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + NT * u;
            if (e < n) {
                sa[e] = va[u];
                if (cols) sb[e] = vb[u];
            }
        }
    }
#else
    for (int e = t; e < n; e += NT) {
        const int d = e >> 7, r = e & (TB - 1);
        sa[e] = (r < gna) ? gxa[r + (size_t)d * glda] : 0.0;
        if (cols) sb[e] = (r < gnb) ? gxb[r + (size_t)d * gldb] : 0.0;
    }
#endif
}
__device__ __forceinline__ void gram_stage_coords(const TileTask& tk, int D, double* sa, double* sb, bool cols) {
    stage_coords(tk.gxa, tk.glda, tk.gna, tk.gxb, tk.gldb, tk.gnb, D, sa, sb, cols);
    __syncthreads();
}

// sum over the dimensions for one entry pair list: z[i][j] over rows a[i], columns b[j] (gram_half_tile's inner loop)
template <int KIND, int NA, int NB>
__device__ __forceinline__ void gram_accumulate(double (&z)[NA][NB], const double (&a)[NA], const double (&b)[NB], double nhd) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (KIND == 0) {
                const double u = a[i] - b[j];
                z[i][j] = fma(u, u, z[i][j]);
            } else if (KIND == 1) {
                const double u = a[i] - b[j];
                z[i][j] += exp_nonpos((u * u) * nhd);
            } else {
                z[i][j] = fma(a[i], b[j], z[i][j]);
            }
        }
}

// kernel value from the accumulated sum, with the padding / diagonal rules of gram_half_tile (EDGE = false: a tile of
// 128 valid rows and columns off the block diagonal -- the value as it is)
template <int KIND, bool EDGE = true>
__device__ __forceinline__ double gram_finish(double z, const KParam& p, int row, int col, int na, int nb, bool diag_tile) {
    double kv;
    if (KIND == 0) kv = p.sigma2 * exp_nonpos(z * p.nh0);
    else if (KIND == 1) kv = p.sigma2 * z;
    else kv = z * p.il2;
    if (!EDGE) return kv;
    const bool valid = (row < na) && (col < nb);
    if (!valid) kv = 0.0;
    if (diag_tile && row == col) kv = valid ? kv + (p.noise + 1e-8) : 1.0;
    return kv;
}

// Full tile, accumulator layout of gemm_mainloop_v2: lane rows wr*64 + 16 rn + l15 (4), columns wc*64 + 16 cm + l4 + 4 r
// (16), one cm (16 entries) at a time.
template <int KIND, bool EDGE>
__device__ __forceinline__ void gram_tile_epilogue(const TileTask& tk, const KParam& p, int D, d4 (&acc)[4][4], double* sa,
                                                   double* sb) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w & 1, wc = w >> 1, l15 = lane & 15, l4 = lane >> 4;
    gram_stage_coords(tk, D, sa, sb, true);
    const unsigned lofs = (unsigned)(wr * 64 + l15) + (unsigned)(wc * 64 + l4) * (unsigned)tk.ldc;
    const size_t ldc = (size_t)tk.ldc;
    const bool diag_tile = (tk.gram & 2) != 0;
    const double* pa = sa + wr * 64 + l15;
#pragma unroll          // (acc is indexed by cm: a rolled loop would put it in scratch memory)
    for (int cm = 0; cm < 4; ++cm) {
        const double* pb = sb + wc * 64 + 16 * cm + l4;
        double z[4][4];
#pragma unroll
        for (int rn = 0; rn < 4; ++rn)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[rn][r] = 0.0;
        for (int d = 0; d < D; ++d) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = pa[d * TB + 16 * i];
                b[i] = pb[d * TB + 4 * i];
            }
            gram_accumulate<KIND, 4, 4>(z, a, b, (KIND == 1) ? p.nh[d] : 0.0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(16 * cm + 4 * r) * ldc);
            const int cidx = wc * 64 + 16 * cm + l4 + 4 * r;
#pragma unroll
            for (int rn = 0; rn < 4; ++rn) {
                const double kv = gram_finish<KIND, EDGE>(z[rn][r], p, wr * 64 + 16 * rn + l15, cidx, tk.gna, tk.gnb, diag_tile);
                col[lofs + 16 * rn] = (tk.update == 1) ? kv - acc[cm][rn][r] : acc[cm][rn][r] - kv;
            }
        }
    }
}

// Epilogue shared by the tile kernels: register r of acc[cm][rn] is C(row = wr*64+16rn+l15, col = wc*64+16cm+l4+4r).
// With tk.wi set (panel solve of the factorisation) the forward substitution y -> L^-1 y rides along:
// w_i -= X z_k, summed per row over the 4 lanes l4, then over the two column halves through LDS (fixed order).
__device__ __forceinline__ void tile_epilogue(const TileTask& tk, d4 (&acc)[4][4], double* red /* >= 512 doubles of LDS */,
                                              const KParam* __restrict__ kp = nullptr, int D = 0, double* sb = nullptr) {
    if (tk.gram != 0 && kp != nullptr) {      // Gram values instead of the tile read (update launches of fit!; no rider there)
        const KParam p = kp[tk.kid];
        const bool edge = tk.gna < TB || tk.gnb < TB || (tk.gram & 2) != 0;
        if (p.kind == 0) {
            if (edge) gram_tile_epilogue<0, true>(tk, p, D, acc, red, sb);
            else gram_tile_epilogue<0, false>(tk, p, D, acc, red, sb);
        } else if (p.kind == 1) {
            if (edge) gram_tile_epilogue<1, true>(tk, p, D, acc, red, sb);
            else gram_tile_epilogue<1, false>(tk, p, D, acc, red, sb);
        } else {
            if (edge) gram_tile_epilogue<2, true>(tk, p, D, acc, red, sb);
            else gram_tile_epilogue<2, false>(tk, p, D, acc, red, sb);
        }
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w & 1, wc = w >> 1, l15 = lane & 15, l4 = lane >> 4;
    // element (cm, rn, r) of this lane sits at C[lofs + 16 rn + (16 cm + 4 r) ldc]: uniform column base + 32-bit lane
    // offset.  C -= acc reads the tile in four quarters of 16 values per lane, all loads of a quarter in flight
    // before its first store: a load placed after a store to the same array waits for the whole round trip, and
    // 64 of those in a row cost ~20 us per task.
    const unsigned lofs = (unsigned)(wr * 64 + l15) + (unsigned)(wc * 64 + l4) * (unsigned)tk.ldc;
    const size_t ldc = (size_t)tk.ldc;
    if (tk.update == 1) {
#pragma unroll
        for (int cm = 0; cm < 4; ++cm) {
            double cv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(16 * cm + 4 * r) * ldc);
#pragma unroll
                for (int rn = 0; rn < 4; ++rn) cv[rn][r] = col[lofs + 16 * rn];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(16 * cm + 4 * r) * ldc);
#pragma unroll
                for (int rn = 0; rn < 4; ++rn) col[lofs + 16 * rn] = cv[rn][r] - acc[cm][rn][r];
            }
        }
    } else if (tk.update == 2) {      // C = -product: the tile holds nothing yet (blocks of L^-T), no read
#pragma unroll
        for (int cm = 0; cm < 4; ++cm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(16 * cm + 4 * r) * ldc);
#pragma unroll
                for (int rn = 0; rn < 4; ++rn) col[lofs + 16 * rn] = -acc[cm][rn][r];
            }
    } else {
#pragma unroll
        for (int cm = 0; cm < 4; ++cm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(16 * cm + 4 * r) * ldc);
#pragma unroll
                for (int rn = 0; rn < 4; ++rn) col[lofs + 16 * rn] = acc[cm][rn][r];
            }
    }
    if (tk.wi != nullptr) {
        double p[4] = {0.0, 0.0, 0.0, 0.0}, q2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int cm = 0; cm < 4; ++cm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double z = tk.zk[wc * 64 + 16 * cm + l4 + 4 * r];
#pragma unroll
                for (int rn = 0; rn < 4; ++rn) {
                    p[rn] = fma(acc[cm][rn][r], z, p[rn]);
                    q2[rn] = fma(acc[cm][rn][r], acc[cm][rn][r], q2[rn]);
                }
            }
#pragma unroll
        for (int rn = 0; rn < 4; ++rn) {
            p[rn] += __shfl_xor(p[rn], 16);
            p[rn] += __shfl_xor(p[rn], 32);
            q2[rn] += __shfl_xor(q2[rn], 16);
            q2[rn] += __shfl_xor(q2[rn], 32);
        }
        __syncthreads();   // the ring is no longer read
        if (l4 == 0) {
#pragma unroll
            for (int rn = 0; rn < 4; ++rn) {
                red[wc * TB + wr * 64 + 16 * rn + l15] = p[rn];
                red[2 * TB + wc * TB + wr * 64 + 16 * rn + l15] = q2[rn];
            }
        }
        __syncthreads();
        if (threadIdx.x < TB) {
            const double s = red[threadIdx.x] + red[TB + threadIdx.x];
            if (tk.sq == nullptr) tk.wi[threadIdx.x] -= s;
            else {
                tk.wi[threadIdx.x] += s;
                tk.sq[threadIdx.x] += red[2 * TB + threadIdx.x] + red[3 * TB + threadIdx.x];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined tile GEMM.
//   - K is consumed in chunks of 8 columns (two 16-MFMA groups per wave) held in a 4-deep LDS ring
//   - the MFMA operand fragments of the NEXT group are read from LDS while the current group's 16 MFMAs
//     issue, also across the chunk boundary (the next chunk is already complete and visible in the ring)
//   - global loads run two chunks ahead of their LDS write, in two alternating register sets
//   - one barrier per chunk, with nothing else waiting at it
constexpr int KC2 = 8;
constexpr int NRING = 4;

// STAMP (diagnostic build only, -DDSMGP_DIAG): lane 0 of every wave records shader cycles of the loop, of its MFMA
// spans and of its chunk boundaries (tools/bench_tile.py).
template <bool STAMP>
__device__ __forceinline__ void gemm_mainloop_v2(const TileTask& tk, d4 (&acc)[4][4], double (*sA)[KC2 * LDP],
                                                 double (*sB)[KC2 * LDP], unsigned long long* __restrict__ stamps) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wr = w & 1, wc = w >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

#pragma unroll
    for (int cm = 0; cm < 4; ++cm)
#pragma unroll
        for (int rn = 0; rn < 4; ++rn) acc[cm][rn] = (d4){0.0, 0.0, 0.0, 0.0};

    // staging: thread t moves column (t>>5) of the chunk, rows 2*(t&31) + 64j, j<2 (512 B per half wave)
    const int scol = t >> 5, srow = 2 * (t & 31);
    // chunk CH holds columns kfirst + CH * kstep .. + 7: ascending from k0, or (TileTask.rev) descending from k1 - 8
    const int kfirst = tk.rev ? tk.k1 - KC2 : tk.k0;
    const ptrdiff_t stepA = (tk.rev ? -(ptrdiff_t)KC2 : (ptrdiff_t)KC2) * tk.lda;
    const ptrdiff_t stepB = (tk.rev ? -(ptrdiff_t)KC2 : (ptrdiff_t)KC2) * tk.ldb;
    const double* gA = tk.A + srow + (size_t)(kfirst + scol) * tk.lda;
    const double* gB = tk.B + srow + (size_t)(kfirst + scol) * tk.ldb;
    const int sOff = scol * LDP + srow;
    d2 ra0[2], rb0[2], ra1[2], rb1[2];

#define GLOAD(RA, RB, CH)                                                                        \
    do {                                                                                         \
        const ptrdiff_t oa_ = (ptrdiff_t)(CH) * stepA, ob_ = (ptrdiff_t)(CH) * stepB;            \
        RA[0] = *AS_GLOBAL_D2(gA + oa_);                                                         \
        RA[1] = *AS_GLOBAL_D2(gA + oa_ + 64);                                                    \
        RB[0] = *AS_GLOBAL_D2(gB + ob_);                                                         \
        RB[1] = *AS_GLOBAL_D2(gB + ob_ + 64);                                                    \
    } while (0)
#define SWRITE(RA, RB, BUF)                                                                      \
    do {                                                                                         \
        *reinterpret_cast<d2*>(&sA[BUF][sOff]) = RA[0];                                          \
        *reinterpret_cast<d2*>(&sA[BUF][sOff + 64]) = RA[1];                                     \
        *reinterpret_cast<d2*>(&sB[BUF][sOff]) = RB[0];                                          \
        *reinterpret_cast<d2*>(&sB[BUF][sOff + 64]) = RB[1];                                     \
    } while (0)
#define FRAGS(FA, FB, BUF, G)                                                                    \
    do {                                                                                         \
        const double* pa_ = &sB[BUF][((G) * 4 + l4) * LDP + wc * 64 + l15];                      \
        const double* pb_ = &sA[BUF][((G) * 4 + l4) * LDP + wr * 64 + l15];                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                       \
            FA[i_] = pa_[16 * i_];                                                               \
            FB[i_] = pb_[16 * i_];                                                               \
        }                                                                                        \
    } while (0)
#define MFMA16(FA, FB)                                                                           \
    do {                                                                                         \
        _Pragma("unroll") for (int cm_ = 0; cm_ < 4; ++cm_)                                      \
            _Pragma("unroll") for (int rn_ = 0; rn_ < 4; ++rn_)                                  \
                acc[cm_][rn_] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[cm_], FB[rn_], acc[cm_][rn_], 0, 0, 0); \
    } while (0)

    const int nch = (tk.k1 - tk.k0) / KC2;
    // prologue: chunks 0..2 into the ring with the loads overlapped (two memory latencies, not three),
    // chunk 3 in flight in set 1
    if (nch > 0) {
        GLOAD(ra0, rb0, 0);
        GLOAD(ra1, rb1, min(1, nch - 1));
        SWRITE(ra0, rb0, 0);
        GLOAD(ra0, rb0, min(2, nch - 1));
        SWRITE(ra1, rb1, 1);
        GLOAD(ra1, rb1, min(3, nch - 1));
        SWRITE(ra0, rb0, 2);
    }
    __syncthreads();

    double fa0[4], fb0[4], fa1[4], fb1[4];
    if (nch > 0) FRAGS(fa0, fb0, 0, 0);
    unsigned long long tin = 0, tmf = 0, tbd = 0, ta = 0, tb = 0, rin = 0;
    if (STAMP) {
        tin = stamp_now();
        rin = __builtin_amdgcn_s_memrealtime();
    }

    // one chunk: LOADSET receives chunk c+4, WRITESET (holding chunk c+3) goes to the ring
    // Issue order inside a chunk: every memory instruction sits in the shadow of an MFMA (64 cycles in the
    // matrix pipe, during which the wave may issue other work); clustering them ahead of the MFMAs leaves the
    // pipe idle while they issue.  Group 0: 4 global loads + 4 ds_read2 between the first 8 MFMAs;
    // group 1: 4 ds_write + 4 ds_read2 likewise.
#define INTERLEAVE(MASK_A, MASK_B)                                                               \
    do {                                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(MASK_A, 1, 0);                                  \
        }                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(MASK_B, 1, 0);                                  \
        }                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                       \
    } while (0)
#define CHUNK(C, LRA, LRB, WRA, WRB)                                                             \
    do {                                                                                         \
        const int c_ = (C);                                                                      \
        const int buf_ = c_ & (NRING - 1);                                                       \
        if (STAMP) { __builtin_amdgcn_sched_barrier(0); ta = stamp_now(); }                      \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        GLOAD(LRA, LRB, min(c_ + 4, nch - 1)); /* clamped: static vmcnt counts */                \
        FRAGS(fa1, fb1, buf_, 1);                                                                \
        MFMA16(fa0, fb0);                                                                        \
        INTERLEAVE(0x020, 0x100);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        SWRITE(WRA, WRB, (c_ + 3) & (NRING - 1)); /* past the end: unread slot */                \
        FRAGS(fa0, fb0, (c_ + 1) & (NRING - 1), 0);                                              \
        MFMA16(fa1, fb1);                                                                        \
        INTERLEAVE(0x200, 0x100);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (STAMP) { tb = stamp_now(); __builtin_amdgcn_sched_barrier(0); tmf += tb - ta; }      \
        __syncthreads();                                                                         \
        if (STAMP) { __builtin_amdgcn_sched_barrier(0); tbd += stamp_now() - tb; __builtin_amdgcn_sched_barrier(0); } \
    } while (0)

    int c = 0;
    for (; c + 1 < nch; c += 2) {
        CHUNK(c, ra0, rb0, ra1, rb1);
        CHUNK(c + 1, ra1, rb1, ra0, rb0);
    }
    if (c < nch) CHUNK(c, ra0, rb0, ra1, rb1);
#undef CHUNK
#undef INTERLEAVE
#undef MFMA16
#undef FRAGS
#undef SWRITE
#undef GLOAD

    if (STAMP && (t & 63) == 0) {
        const unsigned long long tout = stamp_now();
        unsigned long long* s = stamps + ((size_t)blockIdx.x * 4 + w) * 8;
        s[0] = tout - tin;
        s[1] = tmf;
        s[2] = tbd;
        s[3] = nch / 2;
        s[4] = rin;                                   // 100 MHz wall ticks: loop start / loop end
        s[5] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------
// Diagonal tiles of the factorisation: F[k,k] -= F[k,0:K] F[k,0:K]^T is symmetric, and chol_diag_*_kernel reads only
// the lower 16x16 blocks of the tile (whole blocks on the diagonal).  Of the 8 x 8 grid of 16x16 MFMA tiles the 36
// lower ones are computed, 9 per wave:
//   wave 0: block rows 5..7 x block columns 0..2      wave 1: rows 5..7 x columns 3..5
//   wave 2: block rows 2..4 x block columns 0..2      wave 3: the three 2x2 lower triangles at blocks 0, 3 and 6
// Every wave reads 6 operand fragments per 4-column group (one LDS image: B == A) for its 9 MFMAs: 9/16 of the matrix
// work, half the global and LDS staging traffic of a full tile.  Same ring / prefetch protocol as gemm_mainloop_v2.
template <int SHAPE>   // 0: 3x3 square (F[0..2] rows, F[3..5] columns), 1: three 2x2 lower triangles (F[2q], F[2q+1])
__device__ __forceinline__ void syrk_mainloop(const TileTask& tk, d4 (&acc)[9], double (*sA)[KC2 * LDP], const int (&blk)[6]) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    const int scol = t >> 5, srow = 2 * (t & 31);
    const double* gA = tk.A + srow + (size_t)(tk.k0 + scol) * tk.lda;
    const int sOff = scol * LDP + srow;
    d2 ra0[2], ra1[2];
#define SGLOAD(RA, CH)                                                                           \
    do {                                                                                         \
        const size_t oa_ = (size_t)(CH) * KC2 * tk.lda;                                          \
        RA[0] = *AS_GLOBAL_D2(gA + oa_);                                                         \
        RA[1] = *AS_GLOBAL_D2(gA + oa_ + 64);                                                    \
    } while (0)
#define SSWRITE(RA, BUF)                                                                         \
    do {                                                                                         \
        *reinterpret_cast<d2*>(&sA[BUF][sOff]) = RA[0];                                          \
        *reinterpret_cast<d2*>(&sA[BUF][sOff + 64]) = RA[1];                                     \
    } while (0)
#define SFRAGS(F, BUF, G)                                                                        \
    do {                                                                                         \
        const double* p_ = &sA[BUF][((G) * 4 + l4) * LDP + l15];                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) F[i_] = p_[16 * blk[i_]];               \
    } while (0)
#define SMFMA9(F)                                                                                \
    do {                                                                                         \
        if (SHAPE == 0) {                                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                     \
                _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                 \
                    acc[3 * i_ + j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[3 + j_], F[i_], acc[3 * i_ + j_], 0, 0, 0); \
        } else {                                                                                 \
            _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                   \
                acc[3 * q_] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2 * q_], F[2 * q_], acc[3 * q_], 0, 0, 0); \
                acc[3 * q_ + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2 * q_], F[2 * q_ + 1], acc[3 * q_ + 1], 0, 0, 0); \
                acc[3 * q_ + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2 * q_ + 1], F[2 * q_ + 1], acc[3 * q_ + 2], 0, 0, 0); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    const int nch = (tk.k1 - tk.k0) / KC2;
    if (nch > 0) {
        SGLOAD(ra0, 0);
        SGLOAD(ra1, min(1, nch - 1));
        SSWRITE(ra0, 0);
        SGLOAD(ra0, min(2, nch - 1));
        SSWRITE(ra1, 1);
        SGLOAD(ra1, min(3, nch - 1));
        SSWRITE(ra0, 2);
    }
    __syncthreads();
    double f0[6], f1[6];
    if (nch > 0) SFRAGS(f0, 0, 0);
#define SCHUNK(C, LRA, WRA)                                                                      \
    do {                                                                                         \
        const int c_ = (C);                                                                      \
        const int buf_ = c_ & (NRING - 1);                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        SGLOAD(LRA, min(c_ + 4, nch - 1));                                                       \
        SFRAGS(f1, buf_, 1);                                                                     \
        SMFMA9(f0);                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                   \
        }                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
        }                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        SSWRITE(WRA, (c_ + 3) & (NRING - 1));                                                    \
        SFRAGS(f0, (c_ + 1) & (NRING - 1), 0);                                                   \
        SMFMA9(f1);                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                   \
        }                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
        }                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __syncthreads();                                                                         \
    } while (0)
    int c = 0;
    for (; c + 1 < nch; c += 2) {
        SCHUNK(c, ra0, ra1);
        SCHUNK(c + 1, ra1, ra0);
    }
    if (c < nch) SCHUNK(c, ra0, ra1);
#undef SCHUNK
#undef SMFMA9
#undef SFRAGS
#undef SSWRITE
#undef SGLOAD
}

// store / subtract the 9 tiles of a wave: register r of the tile at (block row rb, block column cb) is
// C(16 rb + l15, 16 cb + l4 + 4 r)
template <int SHAPE>
__device__ __forceinline__ void syrk_epilogue(const TileTask& tk, d4 (&acc)[9], const int (&blk)[6]) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const size_t ldc = (size_t)tk.ldc;
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {       // three tiles at a time: loads of a group in flight before its stores
        int rb[3], cb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (SHAPE == 0) {
                rb[j] = blk[g3];
                cb[j] = blk[3 + j];
            } else {
                rb[j] = blk[2 * g3 + (j > 0 ? 1 : 0)];
                cb[j] = blk[2 * g3 + (j > 1 ? 1 : 0)];
            }
        }
        double cv[3][4];
        if (tk.update == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cv[j][r] = AS_GLOBAL_F64(tk.C)[(size_t)(16 * rb[j] + l15) + (size_t)(16 * cb[j] + l4 + 4 * r) * ldc];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[3 * g3 + j][r];
                AS_GLOBAL_F64(tk.C)[(size_t)(16 * rb[j] + l15) + (size_t)(16 * cb[j] + l4 + 4 * r) * ldc] =
                    (tk.update == 1) ? cv[j][r] - v : v;
            }
    }
}

// The same with the Gram values in place of the tile read (TileTask.gram; rows and columns of a diagonal tile are the
// same 128 points: one coordinate image).
template <int SHAPE, int KIND>
__device__ __forceinline__ void syrk_gram_epilogue(const TileTask& tk, const KParam& p, int D, d4 (&acc)[9], const int (&blk)[6],
                                                   const double* sa) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const size_t ldc = (size_t)tk.ldc;
    const bool diag_tile = (tk.gram & 2) != 0;
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {
        int rb[3], cb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (SHAPE == 0) {
                rb[j] = blk[g3];
                cb[j] = blk[3 + j];
            } else {
                rb[j] = blk[2 * g3 + (j > 0 ? 1 : 0)];
                cb[j] = blk[2 * g3 + (j > 1 ? 1 : 0)];
            }
        }
        double z[3][1][4];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[j][0][r] = 0.0;
        for (int d = 0; d < D; ++d) {
            const double nhd = (KIND == 1) ? p.nh[d] : 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double a[1], b[4];
                a[0] = sa[d * TB + 16 * rb[j] + l15];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = sa[d * TB + 16 * cb[j] + l4 + 4 * r];
                gram_accumulate<KIND, 1, 4>(z[j], a, b, nhd);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb[j] + l15, col = 16 * cb[j] + l4 + 4 * r;
                const double kv = gram_finish<KIND>(z[j][0][r], p, row, col, tk.gna, tk.gnb, diag_tile);
                const double v = acc[3 * g3 + j][r];
                AS_GLOBAL_F64(tk.C)[(size_t)row + (size_t)col * ldc] = (tk.update == 1) ? kv - v : v - kv;
            }
    }
}

template <int SHAPE>
__device__ __forceinline__ void syrk_finish(const TileTask& tk, d4 (&acc)[9], const int (&blk)[6], double* sa,
                                            const KParam* __restrict__ kp, int D) {
    if (tk.gram != 0 && kp != nullptr) {
        const KParam p = kp[tk.kid];
        gram_stage_coords(tk, D, sa, nullptr, false);
        if (p.kind == 0) syrk_gram_epilogue<SHAPE, 0>(tk, p, D, acc, blk, sa);
        else if (p.kind == 1) syrk_gram_epilogue<SHAPE, 1>(tk, p, D, acc, blk, sa);
        else syrk_gram_epilogue<SHAPE, 2>(tk, p, D, acc, blk, sa);
    } else {
        syrk_epilogue<SHAPE>(tk, acc, blk);
    }
}

__device__ __forceinline__ void tile_syrk_body(const TileTask& tk, double (*sA)[KC2 * LDP], const KParam* __restrict__ kp = nullptr,
                                               int D = 0) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d4 acc[9];
    if (w == 3) {
        const int blk[6] = {0, 1, 3, 4, 6, 7};
        syrk_mainloop<1>(tk, acc, sA, blk);
        syrk_finish<1>(tk, acc, blk, &sA[0][0], kp, D);
    } else {
        const int rbase = (w == 2) ? 2 : 5, cbase = (w == 1) ? 3 : 0;
        const int blk[6] = {rbase, rbase + 1, rbase + 2, cbase, cbase + 1, cbase + 2};
        syrk_mainloop<0>(tk, acc, sA, blk);
        syrk_finish<0>(tk, acc, blk, &sA[0][0], kp, D);
    }
}

// ---------------------------------------------------------------------------------------------
// Short tiles: update tasks whose rows 16 NR.. are padding (TileTask.mrows <= 16 NR; NR = 2, 4, 6).  The small-leaf regime
// is made of them: the last row tile of a leaf holds n mod 128 rows, a test-row tile of a depth-4 leaf 44 routed rows on
// average, and at depth 4 three of four update tiles are one or the other.  In the 2x2 wave grid of gemm_mainloop_v2 such a
// tile leaves the waves of its lower half without work while the other two still take a full tile's time (and the
// co-resident workgroup still shares ITS SIMDs with them): the tile is no faster.  Here the four waves split the COLUMNS
// instead -- wave w owns columns 32w..32w+31 and all NR row blocks, acc[2][NR] -- so every SIMD carries NR/8 of a full
// tile's matrix work and the task ends in that fraction of the time.  Same ring, prefetch and barrier protocol as
// gemm_mainloop_v2; rows 64.. of the A panel are neither loaded nor staged for NR <= 4.
template <int NR>
__device__ __forceinline__ void gemm_mainloop_rows(const TileTask& tk, d4 (&acc)[2][NR], double (*sA)[KC2 * LDP],
                                                   double (*sB)[KC2 * LDP]) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    constexpr bool AHI = NR > 4;                  // rows 64.. of the A panel hold data
    constexpr int NMEM_G = AHI ? 4 : 3;           // global loads (and LDS writes) per chunk and thread
    constexpr int NFRAG = (2 + NR + 1) / 2;       // fragment reads per 4-column group, as ds_read2
    constexpr int NMFMA = 2 * NR;
#pragma unroll
    for (int cm = 0; cm < 2; ++cm)
#pragma unroll
        for (int rn = 0; rn < NR; ++rn) acc[cm][rn] = (d4){0.0, 0.0, 0.0, 0.0};
    const int scol = t >> 5, srow = 2 * (t & 31);
    const double* gA = tk.A + srow + (size_t)(tk.k0 + scol) * tk.lda;
    const double* gB = tk.B + srow + (size_t)(tk.k0 + scol) * tk.ldb;
    const int sOff = scol * LDP + srow;
    d2 ra0[2], rb0[2], ra1[2], rb1[2];
#define RGLOAD(RA, RB, CH)                                                                       \
    do {                                                                                         \
        const size_t oa_ = (size_t)(CH) * KC2 * tk.lda, ob_ = (size_t)(CH) * KC2 * tk.ldb;       \
        RA[0] = *AS_GLOBAL_D2(gA + oa_);                                                         \
        if (AHI) RA[1] = *AS_GLOBAL_D2(gA + oa_ + 64);                                           \
        RB[0] = *AS_GLOBAL_D2(gB + ob_);                                                         \
        RB[1] = *AS_GLOBAL_D2(gB + ob_ + 64);                                                    \
    } while (0)
#define RSWRITE(RA, RB, BUF)                                                                     \
    do {                                                                                         \
        *reinterpret_cast<d2*>(&sA[BUF][sOff]) = RA[0];                                          \
        if (AHI) *reinterpret_cast<d2*>(&sA[BUF][sOff + 64]) = RA[1];                            \
        *reinterpret_cast<d2*>(&sB[BUF][sOff]) = RB[0];                                          \
        *reinterpret_cast<d2*>(&sB[BUF][sOff + 64]) = RB[1];                                     \
    } while (0)
#define RFRAGS(FA, FB, BUF, G)                                                                   \
    do {                                                                                         \
        const double* pa_ = &sB[BUF][((G) * 4 + l4) * LDP + w * 32 + l15];                       \
        const double* pb_ = &sA[BUF][((G) * 4 + l4) * LDP + l15];                                \
        FA[0] = pa_[0];                                                                          \
        FA[1] = pa_[16];                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < NR; ++i_) FB[i_] = pb_[16 * i_];                 \
    } while (0)
#define RMFMA(FA, FB)                                                                            \
    do {                                                                                         \
        _Pragma("unroll") for (int cm_ = 0; cm_ < 2; ++cm_)                                      \
            _Pragma("unroll") for (int rn_ = 0; rn_ < NR; ++rn_)                                 \
                acc[cm_][rn_] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[cm_], FB[rn_], acc[cm_][rn_], 0, 0, 0); \
    } while (0)
    // one memory instruction in the shadow of each MFMA, as far as the MFMAs go (NR = 2: four of them per group)
    constexpr int NI_A = NMEM_G < NMFMA ? NMEM_G : NMFMA;
    constexpr int NI_B = (NMFMA - NI_A) < NFRAG ? (NMFMA - NI_A) : NFRAG;
    constexpr int NI_REST = NMFMA - NI_A - NI_B;
#define RINTERLEAVE(MASK_A)                                                                      \
    do {                                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < NI_A; ++i_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(MASK_A, 1, 0);                                  \
        }                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < NI_B; ++i_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
        }                                                                                        \
        if constexpr (NI_REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, NI_REST, 0);      \
    } while (0)
    const int nch = (tk.k1 - tk.k0) / KC2;
    if (nch > 0) {
        RGLOAD(ra0, rb0, 0);
        RGLOAD(ra1, rb1, min(1, nch - 1));
        RSWRITE(ra0, rb0, 0);
        RGLOAD(ra0, rb0, min(2, nch - 1));
        RSWRITE(ra1, rb1, 1);
        RGLOAD(ra1, rb1, min(3, nch - 1));
        RSWRITE(ra0, rb0, 2);
    }
    __syncthreads();
    double fa0[2], fb0[NR], fa1[2], fb1[NR];
    if (nch > 0) RFRAGS(fa0, fb0, 0, 0);
#define RCHUNK(C, LRA, LRB, WRA, WRB)                                                            \
    do {                                                                                         \
        const int c_ = (C);                                                                      \
        const int buf_ = c_ & (NRING - 1);                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        RGLOAD(LRA, LRB, min(c_ + 4, nch - 1));                                                  \
        RFRAGS(fa1, fb1, buf_, 1);                                                               \
        RMFMA(fa0, fb0);                                                                         \
        RINTERLEAVE(0x020);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        RSWRITE(WRA, WRB, (c_ + 3) & (NRING - 1));                                               \
        RFRAGS(fa0, fb0, (c_ + 1) & (NRING - 1), 0);                                             \
        RMFMA(fa1, fb1);                                                                         \
        RINTERLEAVE(0x200);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __syncthreads();                                                                         \
    } while (0)
    int c = 0;
    for (; c + 1 < nch; c += 2) {
        RCHUNK(c, ra0, rb0, ra1, rb1);
        RCHUNK(c + 1, ra1, rb1, ra0, rb0);
    }
    if (c < nch) RCHUNK(c, ra0, rb0, ra1, rb1);
#undef RCHUNK
#undef RINTERLEAVE
#undef RMFMA
#undef RFRAGS
#undef RSWRITE
#undef RGLOAD
}

// Epilogue of a short tile: register r of acc[cm][rn] is C(row = 16 rn + l15, col = 32 w + 16 cm + l4 + 4 r).
//   update = 1 without Gram: C(rows < 16 NR) -= product; the rows below are not touched (padding: they hold what they held)
//   update = 1 with the Gram fused: C(rows < 16 NR) = k(row, col) - product with gram_finish's padding rules, and -- the
//   tile has never been written -- the rows 16 NR.. get their padding values when the task says so (TileTask.gram bit 2: tiles
//   of the factor, whose padding is zero, one on the diagonal; the rows of K_tn are zeroed once by dsmgp_set_test)
template <int NR, int KIND>
__device__ __forceinline__ void rows_gram_epilogue(const TileTask& tk, const KParam& p, int D, d4 (&acc)[2][NR], const double* sa,
                                                   const double* sb) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const size_t ldc = (size_t)tk.ldc;
    const bool diag_tile = (tk.gram & 2) != 0;
    const double* pa = sa + l15;
#pragma unroll
    for (int cm = 0; cm < 2; ++cm) {
        const double* pb = sb + w * 32 + 16 * cm + l4;
        double z[NR][4];
#pragma unroll
        for (int rn = 0; rn < NR; ++rn)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[rn][r] = 0.0;
        for (int d = 0; d < D; ++d) {
            double a[NR], b[4];
#pragma unroll
            for (int i = 0; i < NR; ++i) a[i] = pa[d * TB + 16 * i];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = pb[d * TB + 4 * i];
            gram_accumulate<KIND, NR, 4>(z, a, b, (KIND == 1) ? p.nh[d] : 0.0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cidx = w * 32 + 16 * cm + l4 + 4 * r;
            const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)cidx * ldc);
#pragma unroll
            for (int rn = 0; rn < NR; ++rn) {
                const double kv = gram_finish<KIND, true>(z[rn][r], p, 16 * rn + l15, cidx, tk.gna, tk.gnb, diag_tile);
                col[16 * rn + l15] = kv - acc[cm][rn][r];
            }
            if (tk.gram & 4) {
#pragma unroll
                for (int rn = NR; rn < 8; ++rn) {
                    const int row = 16 * rn + l15;
                    col[row] = (diag_tile && row == cidx) ? 1.0 : 0.0;
                }
            }
        }
    }
}

template <int NR>
__device__ __forceinline__ void tile_rows_body(const TileTask& tk, double (*sA)[KC2 * LDP], double (*sB)[KC2 * LDP],
                                               const KParam* __restrict__ kp, int D) {
    d4 acc[2][NR];
    gemm_mainloop_rows<NR>(tk, acc, sA, sB);      // ends on a barrier: the ring is free
    if (tk.gram != 0 && kp != nullptr) {
        const KParam p = kp[tk.kid];
        gram_stage_coords(tk, D, &sA[0][0], &sB[0][0], true);
        if (p.kind == 0) rows_gram_epilogue<NR, 0>(tk, p, D, acc, &sA[0][0], &sB[0][0]);
        else if (p.kind == 1) rows_gram_epilogue<NR, 1>(tk, p, D, acc, &sA[0][0], &sB[0][0]);
        else rows_gram_epilogue<NR, 2>(tk, p, D, acc, &sA[0][0], &sB[0][0]);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const size_t ldc = (size_t)tk.ldc;
#pragma unroll
    for (int cm = 0; cm < 2; ++cm) {       // C -= product, the loads of a column group in flight before its stores
        double cv[NR][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(w * 32 + 16 * cm + l4 + 4 * r) * ldc);
#pragma unroll
            for (int rn = 0; rn < NR; ++rn) cv[rn][r] = col[16 * rn + l15];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(w * 32 + 16 * cm + l4 + 4 * r) * ldc);
#pragma unroll
            for (int rn = 0; rn < NR; ++rn) col[16 * rn + l15] = cv[rn][r] - acc[cm][rn][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Panel solve X = T Dinv^T  (T = tk.A: one 128x128 tile; Dinv = tk.B: inverse of the step's diagonal block, LOWER
// triangular, ld 128; K = 128).  X(r,c) = sum_{j <= c} T(r,j) Dinv(c,j): the 16-column block cb of X needs only
// j < 16 (cb + 1), so 36 of the 64 (column block, 16-wide j range) products are structurally zero-free and the rest
// is skipped.  To give every wave the same work the waves split the ROWS (wave w: rows 32w..32w+31, all 128 columns,
// acc[cb][rn] = 8 x 2 MFMA tiles) instead of the 2x2 quadrants of the update kernel; with the 16 chunks of 8 columns
// fully unrolled, which products exist is known at compile time: 288 MFMAs per wave instead of 512.
// A wave holds whole rows of X, so the riders of the epilogue (forward substitution w_i -= X z_k for train rows,
// predictive moments mu += X z_k, sum of squares for test rows) reduce inside the wave.
template <bool PAD>
__global__ __launch_bounds__(256, 2) void tile_trsm_kernel(const TileTask* __restrict__ tasks) {
    __shared__ __attribute__((aligned(16))) double sA[NRING][KC2 * LDP];
    __shared__ __attribute__((aligned(16))) double sB[NRING][KC2 * LDP];
    const TileTask tk = tasks[blockIdx.x];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    d4 acc[8][2];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int rn = 0; rn < 2; ++rn) acc[cb][rn] = (d4){0.0, 0.0, 0.0, 0.0};

    // staging as in gemm_mainloop_v2: thread t moves column (t>>5) of a chunk, rows 2*(t&31) + 64j
    const int scol = t >> 5, srow = 2 * (t & 31);
    const double* gA = tk.A + srow + (size_t)scol * tk.lda;
    const double* gB = tk.B + srow + (size_t)scol * tk.ldb;
    const int sOff = scol * LDP + srow;
    d2 ra[2][2], rb[2][2];
#define TGLOAD(S, CH)                                                                            \
    do {                                                                                         \
        const size_t oa_ = (size_t)(CH) * KC2 * tk.lda, ob_ = (size_t)(CH) * KC2 * tk.ldb;       \
        ra[S][0] = *AS_GLOBAL_D2(gA + oa_);                                                      \
        ra[S][1] = *AS_GLOBAL_D2(gA + oa_ + 64);                                                 \
        rb[S][0] = *AS_GLOBAL_D2(gB + ob_);                                                      \
        rb[S][1] = *AS_GLOBAL_D2(gB + ob_ + 64);                                                 \
    } while (0)
#define TSWRITE(S, BUF)                                                                          \
    do {                                                                                         \
        *reinterpret_cast<d2*>(&sA[BUF][sOff]) = ra[S][0];                                       \
        *reinterpret_cast<d2*>(&sA[BUF][sOff + 64]) = ra[S][1];                                  \
        *reinterpret_cast<d2*>(&sB[BUF][sOff]) = rb[S][0];                                       \
        *reinterpret_cast<d2*>(&sB[BUF][sOff + 64]) = rb[S][1];                                  \
    } while (0)
    TGLOAD(0, 0);
    TGLOAD(1, 1);
    TSWRITE(0, 0);
    TGLOAD(0, 2);
    TSWRITE(1, 1);
    TGLOAD(1, 3);
    TSWRITE(0, 2);
    __syncthreads();
    constexpr int NCH = TB / KC2;   // 16
    const bool act = !PAD || tk.mrows == 0 || w * 32 < tk.mrows;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & (NRING - 1);
        if (c + 4 < NCH) TGLOAD(c & 1, c + 4);
        const int cb0 = c / 2;      // column blocks below cb0 only see zeros of Dinv in this j range
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (!act) continue;     // this wave's 32 rows are padding (zero rows of T): staging and barriers only
            double fb[2], fa[8];
            const double* pb = &sA[buf][(g * 4 + l4) * LDP + w * 32 + l15];
            const double* pa = &sB[buf][(g * 4 + l4) * LDP + l15];
            fb[0] = pb[0];
            fb[1] = pb[16];
#pragma unroll
            for (int cb = cb0; cb < 8; ++cb) fa[cb] = pa[16 * cb];
#pragma unroll
            for (int cb = cb0; cb < 8; ++cb) {
                acc[cb][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cb], fb[0], acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cb], fb[1], acc[cb][1], 0, 0, 0);
            }
        }
        if (c + 3 < NCH) TSWRITE((c + 1) & 1, (c + 3) & (NRING - 1));
        if (c + 1 < NCH) __syncthreads();
    }
#undef TSWRITE
#undef TGLOAD

    if (!act) return;   // padding rows: X = 0 is what the tile already holds there
    // store: register r of acc[cb][rn] is X(row = 32 w + 16 rn + l15, col = 16 cb + l4 + 4 r)
    const unsigned lofs = (unsigned)(w * 32 + l15) + (unsigned)l4 * (unsigned)tk.ldc;
    const size_t ldc = (size_t)tk.ldc;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const gf64_ptr col = AS_GLOBAL_F64(tk.C + (size_t)(16 * cb + 4 * r) * ldc);
            col[lofs] = acc[cb][0][r];
            col[lofs + 16] = acc[cb][1][r];
        }
    if (tk.wi != nullptr) {
        double p[2] = {0.0, 0.0}, q2[2] = {0.0, 0.0};
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double z = tk.zk[16 * cb + l4 + 4 * r];
#pragma unroll
                for (int rn = 0; rn < 2; ++rn) {
                    p[rn] = fma(acc[cb][rn][r], z, p[rn]);
                    q2[rn] = fma(acc[cb][rn][r], acc[cb][rn][r], q2[rn]);
                }
            }
#pragma unroll
        for (int rn = 0; rn < 2; ++rn) {
            p[rn] += __shfl_xor(p[rn], 16);
            p[rn] += __shfl_xor(p[rn], 32);
            q2[rn] += __shfl_xor(q2[rn], 16);
            q2[rn] += __shfl_xor(q2[rn], 32);
        }
        if (l4 == 0) {
#pragma unroll
            for (int rn = 0; rn < 2; ++rn) {
                const int row = w * 32 + 16 * rn + l15;
                if (tk.sq == nullptr) tk.wi[row] -= p[rn];
                else {
                    tk.wi[row] += p[rn];
                    tk.sq[row] += q2[rn];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Gradient contraction (updategradients!, src/gaussianprocess.jl:165-178 + src/kernels.jl:85-99):
// one tile G = (K_y^-1)[i-tile, j-tile] = sum_k Xt[i,k] Xt[j,k]^T with Xt = L^-T (rows = columns of L^-1),
// never stored: the epilogue contracts  sum_rc (alpha_r alpha_c - G_rc) * K_rc * P_rc  (IsoSE: K = kernel
// value without noise, P = squared distance) and, on diagonal tiles, trace(G).  Off-diagonal tiles count
// twice (symmetry).  out[2*task] = weighted sum, out[2*task+1] = trace part.
struct LeafDev;
struct GradTask {
    TileTask gemm;          // A = Xt row tile i, B = Xt row tile j, K range [128 i, npad)
    const double* xa;       // inputs of the rows (tile i), ld = ldx
    const double* xb;       // inputs of the columns (tile j)
    const double* alpha_a;
    const double* alpha_b;
    int ldx;
    int na, nb;             // valid rows / cols
    int diag;               // tile on the block diagonal
    int kid;
    int pad;
};

// The epilogue is a Gram tile of its own (squared distance + exp per element): the coordinates of the tile's 128 rows
// and 128 columns and the two alpha blocks are staged once through the LDS the main loop no longer needs (D <= 35;
// wider inputs read them from global memory, element by element).
constexpr int GRADDOT_STAGE_D = 35;
// ostride = doubles per task in `out`: 2, or 2 + D when the per-dimension sums of the additive ArdSE kernel are
// asked for (dsmgp_set_option DSMGP_OPT_ARD_LENGTHSCALE_GRADIENT): out[2 + d] = sum_rc (alpha_r alpha_c - G_rc) *
// sigma^2 exp(-u_d^2 / 2 l_d^2) * u_d^2 / l_d^2, u_d = x_rd - x_cd -- the contraction with dK / dlog l_d.
__global__ __launch_bounds__(256, 2) void tile_graddot_kernel(const GradTask* __restrict__ tasks,
                                                              const KParam* __restrict__ kp, int D,
                                                              double* __restrict__ out, int ostride) {
    __shared__ __attribute__((aligned(16))) double smem[2 * NRING * KC2 * LDP];
    __shared__ double red[2][4];
    double (*sA)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(smem);
    double (*sB)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(smem + NRING * KC2 * LDP);
    const GradTask g = tasks[blockIdx.x];
    const KParam p = kp[g.kid];
    d4 acc[4][4];
    gemm_mainloop_v2<false>(g.gemm, acc, sA, sB, nullptr);      // ends on a barrier: the ring is free
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wr = w & 1, wc = w >> 1, l15 = lane & 15, l4 = lane >> 4;
    const double nh = p.nh0;
    double s = 0.0, tr = 0.0;
    if (p.kind == 1) {
        // additive ArdSE (needs D <= GRADDOT_STAGE_D, checked by the host): one dimension at a time
        double* xs = smem;
        double* al = smem + (size_t)D * 256;
        for (int e = t; e < D * 256; e += 256) {
            const int d = e >> 8, rc = e & 255;
            xs[e] = (rc < TB) ? ((rc < g.na) ? g.xa[rc + (size_t)d * g.ldx] : 0.0)
                              : ((rc - TB < g.nb) ? g.xb[rc - TB + (size_t)d * g.ldx] : 0.0);
        }
        al[t] = (t < TB) ? ((t < g.na) ? g.alpha_a[t] : 0.0) : ((t - TB < g.nb) ? g.alpha_b[t - TB] : 0.0);
        __syncthreads();
        const double wgt = g.diag ? 1.0 : 2.0;
        for (int d = 0; d < D; ++d) {
            const double nhd = p.nh[d];
            double sd = 0.0;
#pragma unroll
            for (int rn = 0; rn < 4; ++rn) {
                const int r = wr * 64 + 16 * rn + l15;
                const bool rv = r < g.na;
                const double ar = al[r], a = xs[d * 256 + r];
                const double* xb = xs + d * 256 + TB + wc * 64 + l4;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = wc * 64 + 16 * (i >> 2) + l4 + 4 * (i & 3);
                    const double u = a - xb[16 * (i >> 2) + 4 * (i & 3)];
                    const double q = u * u;
                    const double pre = ar * al[TB + c] - acc[i >> 2][rn][i & 3];
                    if (rv && c < g.nb) sd = fma(pre * exp_nonpos(q * nhd), q, sd);
                }
            }
            for (int o = 32; o > 0; o >>= 1) sd += __shfl_down(sd, o);
            __syncthreads();
            if (lane == 0) red[0][w] = sd;
            __syncthreads();
            if (t == 0)   // u^2 / l_d^2 = -2 nh_d u^2
                out[(size_t)ostride * blockIdx.x + 2 + d] = wgt * p.sigma2 * (-2.0 * nhd) * (red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        }
#pragma unroll
        for (int rn = 0; rn < 4; ++rn) {
            const int r = wr * 64 + 16 * rn + l15;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = wc * 64 + 16 * (i >> 2) + l4 + 4 * (i & 3);
                if (g.diag && r == c && r < g.na) tr += acc[i >> 2][rn][i & 3];
            }
        }
        __syncthreads();
    } else if (D <= GRADDOT_STAGE_D) {
        // smem: per dimension d 256 doubles (rows' coordinate | columns' coordinate), then alpha_a | alpha_b
        double* xs = smem;
        double* al = smem + (size_t)D * 256;
        for (int e = t; e < D * 256; e += 256) {
            const int d = e >> 8, rc = e & 255;
            xs[e] = (rc < TB) ? ((rc < g.na) ? g.xa[rc + (size_t)d * g.ldx] : 0.0)
                              : ((rc - TB < g.nb) ? g.xb[rc - TB + (size_t)d * g.ldx] : 0.0);
        }
        al[t] = (t < TB) ? ((t < g.na) ? g.alpha_a[t] : 0.0) : ((t - TB < g.nb) ? g.alpha_b[t - TB] : 0.0);
        __syncthreads();
#pragma unroll
        for (int rn = 0; rn < 4; ++rn) {
            const int r = wr * 64 + 16 * rn + l15;
            const bool rv = r < g.na;
            const double ar = al[r];
            double z[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.0;
            for (int d = 0; d < D; ++d) {
                const double a = xs[d * 256 + r];
                const double* xb = xs + d * 256 + TB + wc * 64 + l4;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double u = a - xb[16 * (i >> 2) + 4 * (i & 3)];
                    z[i] = fma(u, u, z[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = wc * 64 + 16 * (i >> 2) + l4 + 4 * (i & 3);
                if (rv && c < g.nb) {
                    const double kv = p.sigma2 * exp_nonpos(z[i] * nh);
                    const double pre = ar * al[TB + c] - acc[i >> 2][rn][i & 3];
                    s = fma(pre * kv, z[i], s);
                    if (g.diag && r == c) tr += acc[i >> 2][rn][i & 3];
                }
            }
        }
    } else {
#pragma unroll
        for (int rn = 0; rn < 4; ++rn) {
            const int r = wr * 64 + 16 * rn + l15;
            const bool rv = r < g.na;
            const double ar = rv ? g.alpha_a[r] : 0.0;
#pragma unroll
            for (int cm = 0; cm < 4; ++cm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = wc * 64 + 16 * cm + l4 + 4 * q;
                    if (rv && c < g.nb) {
                        double z = 0.0;
                        for (int d = 0; d < D; ++d) {
                            const double u = g.xa[r + (size_t)d * g.ldx] - g.xb[c + (size_t)d * g.ldx];
                            z = fma(u, u, z);
                        }
                        const double kv = p.sigma2 * exp_nonpos(z * nh);
                        const double pre = ar * g.alpha_b[c] - acc[cm][rn][q];
                        s = fma(pre * kv, z, s);
                        if (g.diag && r == c) tr += acc[cm][rn][q];
                    }
                }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        tr += __shfl_down(tr, o);
    }
    if (lane == 0) {
        red[0][w] = s;
        red[1][w] = tr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double wgt = g.diag ? 1.0 : 2.0;
        out[(size_t)ostride * blockIdx.x] = wgt * (red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        out[(size_t)ostride * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// trace(K_y^-1) = |L^-1|_F^2 without forming K_y^-1: sum of squares of one row tile of Xt (columns >= its block)
struct FrobTask {
    const double* X;   // Xt + row0
    int ld;
    int col0, col1;    // column range (elements)
    int nrows;         // valid rows in this tile
    int n;             // valid columns (leaf size)
};

__global__ __launch_bounds__(256) void frob_kernel(const FrobTask* __restrict__ tasks, double* __restrict__ out) {
    __shared__ double red[256];
    const FrobTask tk = tasks[blockIdx.x];
    const int t = threadIdx.x, r = t & 127, h = t >> 7;
    double s = 0.0;
    if (r < tk.nrows)
        for (int c = tk.col0 + h; c < min(tk.col1, tk.n); c += 2) {
            const double v = tk.X[r + (size_t)c * tk.ld];
            s = fma(v, v, s);
        }
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = red[0];
}

// Xt diagonal tiles: Xt[t,t] = Dinv_t^T
struct TransTask {
    const double* src;   // 128x128, ld 128
    double* dst;
    int ldd;
    int pad;
};

__global__ __launch_bounds__(256) void transpose_tile_kernel(const TransTask* __restrict__ tasks) {
    __shared__ double tile[32][33];
    const TransTask tk = tasks[blockIdx.x >> 4];
    const int sub = blockIdx.x & 15, bi = sub >> 2, bj = sub & 3;   // 4x4 sub-tiles of 32x32
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int y = ty; y < 32; y += 8) tile[y][tx] = tk.src[(bi * 32 + tx) + (size_t)(bj * 32 + y) * TB];
    __syncthreads();
    for (int y = ty; y < 32; y += 8) tk.dst[(bj * 32 + tx) + (size_t)(bi * 32 + y) * tk.ldd] = tile[tx][y];
}

// per leaf: y.alpha and alpha.alpha (inputs of the gradient assembly)
__global__ __launch_bounds__(256) void dots_kernel(const LeafDev* __restrict__ leaves, double* __restrict__ out);

// split-K epilogue: tile -= slab_0 + slab_1 + ... (fixed order, so results are bit-reproducible)
struct ReduceTask {
    double* C;
    const double* slabs;   // nsplit consecutive 128x128 slabs, ld 128
    int ldc;
    int nsplit;
    int fresh;             // 1: the tile is not read (C = -(slab_0 + ...)): slab 0 already holds product - Gram value
    int pad;
};

// REDUCE_WGS workgroups per tile (16 columns each); a thread owns two rows of four columns and keeps the loads of
// two slabs (eight 16 B loads) in flight -- steps with few tiles and many slabs (the last block steps, multi-GPU
// shards) are a chain of dependent launches, and this kernel's latency is on it.
constexpr int REDUCE_WGS = 8;
__global__ __launch_bounds__(256) void tile_reduce_kernel(const ReduceTask* __restrict__ tasks) {
    const ReduceTask tk = tasks[blockIdx.x / REDUCE_WGS];
    const int t = threadIdx.x;
    const int r = (t & 63) * 2;
    const int c0 = (blockIdx.x % REDUCE_WGS) * (TB / REDUCE_WGS) + (t >> 6);   // columns c0, c0+4, c0+8, c0+12
    const gd2_cptr sl = AS_GLOBAL_D2(tk.slabs + r + (size_t)c0 * TB);
    d2 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = d2{0.0, 0.0};
#pragma unroll 2
    for (int q = 0; q < tk.nsplit; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += sl[((size_t)q * TB * TB + (size_t)(4 * i) * TB) >> 1];
    d2 cv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cv[i] = tk.fresh ? d2{0.0, 0.0} : *reinterpret_cast<const d2*>(tk.C + r + (size_t)(c0 + 4 * i) * tk.ldc);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<d2*>(tk.C + r + (size_t)(c0 + 4 * i) * tk.ldc) = cv[i] - s[i];
}

// ---------------------------------------------------------------------------------------------
// Diagonal block: in-LDS Cholesky of one 128x128 tile + its triangular inverse (used by the panel
// solves as a GEMM).  One workgroup per leaf and step.  potrf semantics of src/gaussianprocess.jl:101:
// lower factor; info = first non-positive pivot (1-based, global index), like LAPACK.
struct DiagTask {
    double* T;        // diagonal tile of F
    double* Dinv;     // 128x128 output, ld 128
    const double* wk; // fused forward solve: right-hand side block (NULL = not fused)
    double* zk;       // z_k = L_kk^-1 w_k
    int* info;
    int ld;
    int nvalid;       // valid rows in this tile
    int row0;         // global index of the tile's first row
    int pad;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which on this target
// waits for every outstanding GLOBAL store of the wave as well (s_waitcnt vmcnt(0)): in the diagonal-block kernel, whose
// results stream out to the tile and to Dinv while it runs, that put the store latency on the chain of all 17 barriers.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// acc[r] (+)= sum_{k<16} Aop(a,k) Bop(b,k) with a = (lane>>4)+4r, b = lane&15;
// Aop(a,k) at pa[a*saa + k*sak], Bop(b,k) at pb[b + k*sbk]
__device__ __forceinline__ d4 blk_mma(const double* pa, int saa, int sak, const double* pb, int sbk, d4 acc, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int kk = 4 * s + l4;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[l15 * saa + kk * sak], pb[l15 + kk * sbk], acc, 0, 0, 0);
    }
    return acc;
}

// 1/sqrt(d) and 1/d to full double precision from the hardware seeds (relative error ~2^-23) with one
// third-order step each (error ~ e^3): 4 resp. 3 dependent operations after the seed, no division, no sqrt call
__device__ __forceinline__ double rsqrt_nr(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);              // 1 - d y^2
    return fma(y * e, fma(e, 0.375, 0.5), y);          // y (1 + e/2 + 3e^2/8)
}
__device__ __forceinline__ double rcp_nr(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);                  // 1 - d r
    return fma(r * e, 1.0 + e, r);                     // r (1 + e + e^2)
}

// potrf + inverse of one 16x16 diagonal block on the matrix pipe, one wave, everything in accumulator layout
// (register r of lane l = element (4r + (l >> 4), l & 15)).  The block is eliminated four pivots at a time:
//   D   = the 4x4 diagonal block of group g (ten v_readlane pairs out of register g), factorised and inverted by the
//         same ~50 scalar-like VALU operations on every lane (four rsqrt chains): Linv_g
//   P^T = Linv_g M(G,:)                one MFMA: A = Linv_g in the first four rows of a 16x4 operand, B = register g of M
//                                      as it stands (rows 4g..4g+3 of the symmetric block = the operand's k-slab);
//                                      register 0 of the result holds P(l & 15, l >> 4) -- which is P as an A operand
//                                      AND P^T as a B operand
//   M  -= P P^T                        one MFMA with that register as both operands: the rank-4 trailing update of all
//                                      16 x 16 entries (rows and columns of finished groups pick up rounding residue that
//                                      nothing reads)
//   X_G = Linv_g R(G,:),  R -= P X_G   the same two MFMAs on the identity: X = L^-1 by forward substitution
// Register 0 of the P^T products, group by group, IS L^T in accumulator layout, that of the X_G products IS L^-1.
// Against the column-per-lane form it replaces (rank-1 updates, a v_readlane pair per row and pivot: ~870 VALU
// instructions, 7200 cycles measured) this is ~400 VALU instructions and 16 MFMAs.
// Returns the 1-based index of the first non-positive pivot (0 = none); lt = L^T (upper part incl. rounding residue
// below its diagonal: the caller masks), xinv = L^-1.
__device__ __forceinline__ int potrf_inv16_mfma(d4 m, d4& lt, d4& xinv, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
    d4 R;
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] = (4 * r + l4 == l15) ? 1.0 : 0.0;
    unsigned badmask = 0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const double mg = m[g];
        const double d00 = readlane_f64(mg, 4 * g);
        const double d10 = readlane_f64(mg, 16 + 4 * g), d11 = readlane_f64(mg, 16 + 4 * g + 1);
        const double d20 = readlane_f64(mg, 32 + 4 * g), d21 = readlane_f64(mg, 32 + 4 * g + 1), d22 = readlane_f64(mg, 32 + 4 * g + 2);
        const double d30 = readlane_f64(mg, 48 + 4 * g), d31 = readlane_f64(mg, 48 + 4 * g + 1), d32 = readlane_f64(mg, 48 + 4 * g + 2),
                     d33 = readlane_f64(mg, 48 + 4 * g + 3);
        const double r0 = rsqrt_nr(d00);
        const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
        const double e11 = fma(-l10, l10, d11);
        const double r1 = rsqrt_nr(e11);
        const double l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
        const double e22 = fma(-l21, l21, fma(-l20, l20, d22));
        const double r2 = rsqrt_nr(e22);
        const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
        const double e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
        const double r3 = rsqrt_nr(e33);
        badmask |= ((d00 > 0.0) ? 0u : 1u) << (4 * g) | ((e11 > 0.0) ? 0u : 2u) << (4 * g) | ((e22 > 0.0) ? 0u : 4u) << (4 * g) |
                   ((e33 > 0.0) ? 0u : 8u) << (4 * g);
        // Linv_g (lower): columns of the inverse by forward substitution
        const double i10 = -(l10 * r0) * r1;
        const double i21 = -(l21 * r1) * r2;
        const double i32 = -(l32 * r2) * r3;
        const double i20 = -fma(l21, i10, l20 * r0) * r2;
        const double i31 = -fma(l32, i21, l31 * r1) * r3;
        const double i30 = -fma(l32, i20, fma(l31, i10, l30 * r0)) * r3;
        // A operand: lane (a = l & 15, k = l >> 4) supplies Linv_g(a, k) for k <= a < 4, zero elsewhere
        double a2 = 0.0;
        a2 = (lane == 0) ? r0 : a2;
        a2 = (lane == 1) ? i10 : a2;
        a2 = (lane == 2) ? i20 : a2;
        a2 = (lane == 3) ? i30 : a2;
        a2 = (lane == 17) ? r1 : a2;
        a2 = (lane == 18) ? i21 : a2;
        a2 = (lane == 19) ? i31 : a2;
        a2 = (lane == 34) ? r2 : a2;
        a2 = (lane == 35) ? i32 : a2;
        a2 = (lane == 51) ? r3 : a2;
        // (a dependent f64 MFMA waits ~100 cycles for its predecessor: the independent product on R goes in between)
        const d4 pt = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, mg, zero, 0, 0, 0);
        const d4 xt = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, R[g], zero, 0, 0, 0);
        const double p = pt[0];
        m = __builtin_amdgcn_mfma_f64_16x16x4f64(-p, p, m, 0, 0, 0);
        R = __builtin_amdgcn_mfma_f64_16x16x4f64(-p, xt[0], R, 0, 0, 0);
        lt[g] = p;
        xinv[g] = xt[0];
    }
    return badmask ? __builtin_ctz(badmask) + 1 : 0;
}


// ---------------------------------------------------------------------------------------------
// The diagonal-block kernel: blocked right-looking Cholesky of one 128x128 tile on 16x16 sub-blocks in a 75 KB LDS image,
// so TWO workgroups fit a CU (and one fits next to a tile workgroup).  Per block step J:
//   P0  wave 0: potrf + inverse of the 16x16 diagonal block on the matrix pipe (potrf_inv16_mfma: four pivots at a time,
//       rank-4 MFMA updates, everything in accumulator layout; no LDS, no barrier inside); for J > 0 it runs inside P2 of
//       step J-1, right after wave 0 has updated that block
//   P1  all waves: panel S(I,J) = S(I,J) Linv^T (I > J); the result goes to the image AND, being final, from the
//       accumulators straight to the tile in global memory                                       [MFMA]
//   P2  waves 1..3: trailing S(I,K) -= S(I,J) S(K,J)^T (I >= K > J)                              [MFMA, two independent
//       16x16 products in flight per wave]
// Measured on one block (tools/probe_diag.py, wall-clock stamps in the diagnostic build), round 3: 46.1 us at the start
// (diagonal 16x16 blocks 3.1 us each in a column-per-lane form with a v_readlane pair per row and pivot -- 870 VALU
// instructions, and a wave alone on its SIMD issues one dependent instruction per ~8 cycles --, panel products 0.65,
// write-back 3.4, inverse phase 6.7) -> 34.4 us: diagonal blocks 1.75 us (MFMA form), the blocks known to be zero written
// at the START from all waves (the upper blocks of Dinv not at all: the arena is zeroed once), L blocks stored by P1 from
// its accumulators, whole-line block moves, barriers that wait for the LDS only (__syncthreads() also waits for the global
// stores in flight).  One wave issues an f64 MFMA about every 100 cycles whether or not it depends on the last one, so the
// inverse phase (148 MFMAs on wave 0) stays at 6.3 us.  Folding that phase into the slack waves 1..3 have in P2 (rows of
// L^-1 computed step by step, written into the slots of the L blocks they replace, a mirrored L_JJ^-1 in the diagonal slots
// for conflict-free operand reads) was built and measured: correct, but 41.6 us a block and 262 against 226 us for 2048
// blocks -- the bookkeeping costs more instructions than the overlap wins: not kept.
// Until round 3 a second, "latency" form of this kernel (L and L^-1 side by side in a 147 KB image: one workgroup per CU, the
// inverse accumulated by forward substitution on the identity during the factorisation, 42 % LDS bank conflicts) served the
// launches with fewer blocks than CUs; the packed form overtook it (a launch of 144 blocks: 52.3 against 55.0 us before
// today's work, 41.3 us after) and it is gone.
//   image : the 36 lower 16x16 blocks in a 9 x 4 block rectangle (PLD = 144 rows, 64 columns): block (I,K) with
//           K < 4 at block position (I, K); the ten blocks with K >= 4 fill the six free positions above the
//           diagonal of the first four block columns and the ninth block row (PackedMap)
//   phase 1: blocked right-looking Cholesky exactly as above.  L_JJ goes from registers straight to the tile in global
//           memory; its slot in the image keeps L_JJ^-1 (operand of the panel products and of phase 2)
//           The forward substitution z = L_kk^-1 w_k rides along: z_J = L_JJ^-1 w_J in P1 (one wave, 16 lanes), then
//           w_I -= L(I,J) z_J for the rows below in P2 (waves 1..3, one row per thread): no inverse is needed for it
//   phase 2: L^-1 = X, block column by block column WITHOUT barriers: X(K,K) = L_KK^-1 is in the image, and
//           X(I,K) = -L_II^-1 sum_{J=K..I-1} L(I,J) X(J,K) depends on the same column's earlier blocks only.  Wave w takes
//           the columns K = w and 7 - w (37, 31, 27 and 25 block products) and keeps its X blocks in REGISTERS: a 16x16
//           result in accumulator layout is the second operand of the next product as it stands (register r = rows
//           4r..4r+3 of the block = the operand's k-slab r), so the image is only read (blocks of L, conflict-free
//           column reads) and every finished block goes straight to Dinv in global memory.
constexpr int PLD = 144;
constexpr int PIMG = 64 * PLD;
constexpr int DIAGP_LDS_BYTES = (PIMG + 2 * TB + 64) * (int)sizeof(double);   // image + rhs block + z + block-offset table

struct PackedMap {
    unsigned short off[64];     // [I * 8 + K], doubles; 0xFFFF above the diagonal
    unsigned char lower[36];    // the 36 lower blocks, I << 4 | K, column by column
    constexpr PackedMap() : off{}, lower{} {
        const int freeR[10] = {0, 0, 1, 0, 1, 2, 8, 8, 8, 8}, freeC[10] = {1, 2, 2, 3, 3, 3, 0, 1, 2, 3};
        int f = 0, n = 0;
        for (int K = 0; K < 8; ++K)
            for (int I = 0; I < 8; ++I) {
                if (I < K) {
                    off[I * 8 + K] = 0xFFFF;
                    continue;
                }
                int R = I, C = K;
                if (K >= 4) {
                    R = freeR[f];
                    C = freeC[f];
                    ++f;
                }
                off[I * 8 + K] = (unsigned short)((C * 16) * PLD + R * 16);
                lower[n++] = (unsigned char)(I << 4 | K);
            }
    }
};
__constant__ const PackedMap PACKED{};

// Phase 2 of the diagonal-block kernel (see chol_diag_packed_body): L^-1 from the packed image, whose off-diagonal slots hold
// the blocks of L and whose diagonal slots hold L_JJ^-1; finished blocks go straight to Dinv.  No barrier, no image write.
__device__ __forceinline__ void packed_inverse_phase(const DiagTask& tk, const double* S, int mytab, int JN) {
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    auto off = [&](int I, int K) { return __builtin_amdgcn_readlane(mytab, I * 8 + K); };
    auto frag = [&](const double* blk, double (&f)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = blk[(4 * q + l4) * PLD + l15];
    };
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    auto mma4 = [&](const double (&fa)[4], const double (&fb)[4]) {
        d4 u = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0], zero4, 0, 0, 0);
        d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], fb[1], zero4, 0, 0, 0);
        u = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], fb[2], u, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], fb[3], v, 0, 0, 0);
        return u + v;
    };
    // L^-1 = X, block column by block column WITHOUT barriers: X(K,K) = L_KK^-1 is in the image, and
    //      X(I,K) = -L_II^-1 sum_{J=K..I-1} L(I,J) X(J,K) depends on the same column's earlier blocks only.  Wave w takes the
    //      columns K = w and 7 - w and keeps its X blocks in registers: a 16x16 result in accumulator layout is the B operand
    //      of the next product as it stands (register q = k-slab q), so the image is only read (blocks of L as A operands)
    //      and every finished block goes straight to Dinv.  The sum runs on two accumulators (even and odd J): a dependent
    //      f64 MFMA waits ~100 cycles for its predecessor where an independent one issues after 64.
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int K = half ? 7 - w : w;
        if (K >= JN) continue;      // identity padding: the diagonal block is written above, everything below it is zero
        const int m = (JN < 8 ? JN - 1 : 7) - K;     // rows I >= JN of the column are zero (and stay so in the zeroed arena)
        d4 xr[8];
        {   // X(K,K) = L_KK^-1 into accumulator layout: a product with the identity (reading it lane-per-column
            // from the image would be a 8-way bank conflict)
            double lk[4], id[4];
            frag(S + off(K, K), lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) id[q] = (l15 == 4 * q + l4) ? 1.0 : 0.0;
            xr[0] = mma4(lk, id);
        }
#pragma unroll
        for (int i = 1; i <= 7; ++i) {
            if (i > m) break;
            const int I = K + i;
            d4 acc0 = zero4, acc1 = zero4;
#pragma unroll
            for (int s2 = 0; s2 < i; ++s2) {     // sum_J L(I,J) X(J,K)
                double lb[4];
                frag(S + off(I, K + s2), lb);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (s2 & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lb[q], xr[s2][q], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lb[q], xr[s2][q], acc0, 0, 0, 0);
                }
            }
            const d4 acc = acc0 + acc1;
            double li[4], ta[4] = {acc[0], acc[1], acc[2], acc[3]};
            frag(S + off(I, I), li);
            xr[i] = -mma4(li, ta);
            // register r holds X(row 16 I + l4 + 4 r, column 16 K + l15)
            const gf64_ptr gD = AS_GLOBAL_F64(tk.Dinv + (size_t)(16 * I + l4) + (size_t)(16 * K + l15) * TB);
#pragma unroll
            for (int r = 0; r < 4; ++r) gD[4 * r] = xr[i][r];
        }
    }
}

// have_image: the caller has already put the lower blocks of the tile into the image (diag_finish_body: straight from
// the accumulators of the tile's update); the barrier below makes them visible
// (Rotating the wave roles with the workgroup index, so that the pivot chains of the two workgroups of a CU run on different
// SIMDs, measured no difference: depth 4 0.0604 / 0.0593 / 0.0594 s without, 0.0596 / 0.0597 / 0.0586 s with, same box.)
// INVERSE = false (fused block steps since round 4): phase 2 is left out -- the tile tasks of a fused step solve by block
// substitution against L_kk and the eight 16x16 inverses L_JJ^-1 this phase-1 leaves on the diagonal of Dinv_k
// (tile_fused_body), and whoever needs the whole inverse later (standalone prediction, gradients, alpha, forward solves of
// COPY / PREFIX leaves) gets it from dinv_complete_kernel: the same phase 2 on an image re-loaded from the tile and Dinv_k.
template <bool STAMP = false, bool INVERSE = true>
__device__ __forceinline__ void chol_diag_packed_body(const DiagTask& tk, double* S, bool have_image,
                                                      unsigned long long* st = nullptr) {
    const int t = threadIdx.x, lane = t & 63;
    // STAMP (diagnostic build only): wave 0 records the 100 MHz wall clock at the phase boundaries of its block
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            if (t == 0) {
                unsigned long long now;
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                st[i] = now;
            }
        }
    };
    stamp(0);
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    double* wl = S + PIMG;                 // right-hand side block w_k, updated in place by the fused forward substitution
    double* zl = S + PIMG + TB;            // z_k = L_kk^-1 w_k
    int* soff = reinterpret_cast<int*>(S + PIMG + 2 * TB);
    const bool fuse = tk.wk != nullptr;
    // block offsets: lane l keeps the offset of block (l >> 3, l & 7) and the l-th lower block; wave-uniform lookups
    // are v_readlane, per-thread lookups (the fused z at the end) go through the copy in LDS
    const int mytab = PACKED.off[lane];
    const int myblk = PACKED.lower[lane < 36 ? lane : 0];
    if (t < 64) soff[t] = mytab;
    auto off = [&](int I, int K) { return __builtin_amdgcn_readlane(mytab, I * 8 + K); };

    // Block <-> global mapping of the bulk moves: a lane moves 16 B, the 8 lanes of a column its 16 rows = one whole
    // 128-B line, an instruction 8 columns (two per block); in the image those 16 lanes of a pass hit 64 different banks
    const int mc = lane >> 3, mr = 2 * (lane & 7);
    if (!have_image) {   // lower blocks of the tile -> image: 9 blocks per wave
        d2 v[9][2];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int e = __builtin_amdgcn_readlane(myblk, w + 4 * q);
            const int I = e >> 4, K = e & 15;
            const double* src = tk.T + (size_t)(16 * I + mr) + (size_t)(16 * K + mc) * tk.ld;
            v[q][0] = *AS_GLOBAL_D2(src);
            v[q][1] = *AS_GLOBAL_D2(src + (size_t)8 * tk.ld);
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int e = __builtin_amdgcn_readlane(myblk, w + 4 * q);
            double* dst = S + off(e >> 4, e & 15) + mc * PLD + mr;
            *reinterpret_cast<d2*>(dst) = v[q][0];
            *reinterpret_cast<d2*>(dst + 8 * PLD) = v[q][1];
        }
    }
    if (tk.wk != nullptr && t < TB) {
        wl[t] = tk.wk[t];
        zl[t] = 0.0;                       // (block steps of pure padding are skipped: their z is the zero right-hand side)
    }
    // Block steps that hold data.  The last block of a leaf is padded with the identity (n mod 128 rows of data): the 16x16
    // blocks from JN on are the identity -- their own factor and inverse -- and everything below and beside them is zero,
    // so the factorisation stops after step JN - 1 (whose look at the next diagonal block still passes block JN through diag_block: it writes that
    // block's L and L^-1) and the diagonal blocks beyond are written as they are.
    const int JN = (tk.nvalid + 15) >> 4;
    // What is known to be zero in the tile goes out NOW, while the image settles: the blocks above the diagonal and the
    // off-diagonal blocks of the skipped block columns.  (At the end of the kernel these stores, then with the upper blocks
    // of Dinv, were 2 of its 3.4 us of write-back: the store queue, not the LDS, was the limit.)  The blocks of Dinv above
    // the diagonal are never written: the arena is zeroed when the leaf table is set and only this kernel writes to it.
    {
        const d2 zero = {0.0, 0.0};
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int b = w + 4 * q, I = b >> 3, K = b & 7;
            if (I < K || (I > K && K >= JN)) {
                double* gT = tk.T + (size_t)(16 * I + mr) + (size_t)(16 * K + mc) * tk.ld;
                *reinterpret_cast<d2*>(gT) = zero;
                *reinterpret_cast<d2*>(gT + (size_t)8 * tk.ld) = zero;
            }
        }
    }
    lds_barrier();
    stamp(1);
    int bad = 0;
    // diagonal block J on wave 0: L_JJ -> global tile (from registers), L_JJ^-1 -> Dinv and -> its slot in the image
    auto diag_block = [&](int J) {
        double* slot = S + off(J, J);
        d4 m, lt, xi;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = slot[(4 * r + l4) * PLD + l15];     // (c, r) for (r, c): the block is symmetric
        const int bj = potrf_inv16_mfma(m, lt, xi, lane);
        if (bj != 0 && bad == 0) bad = J * 16 + bj;
        // lt register g = L(l15, 4g + l4), xi register g = L^-1(4g + l4, l15)
        const gf64_ptr gT = AS_GLOBAL_F64(tk.T + (size_t)(16 * J + l15) + (size_t)(16 * J + l4) * tk.ld);
        const gf64_ptr gD = AS_GLOBAL_F64(tk.Dinv + (size_t)(16 * J + l4) + (size_t)(16 * J + l15) * TB);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 4 * g + l4;
            gT[(size_t)(4 * g) * tk.ld] = (l15 >= q) ? lt[g] : 0.0;
            const double x = (q >= l15) ? xi[g] : 0.0;
            gD[4 * g] = x;
            slot[l15 * PLD + q] = x;
        }
    };
    // Operand fragment of a 16x16 block for either side of the MFMA: element (l15, 4q + l4) of a column-major block is the
    // A operand A[i = l15][k = 4q + l4]; element (4q + l4, l15) of a ROW-major block is the B operand B[k][j = l15]; and the
    // B operand of the transpose of a column-major block is that same address again.  16 lanes along the contiguous index,
    // 4 along the leading dimension (144 = 16 mod 32 doubles): conflict-free.
    auto frag = [&](const double* blk, double (&f)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = blk[(4 * q + l4) * PLD + l15];
    };
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    // one block product alone: two half chains (a dependent f64 MFMA waits ~100 cycles for its predecessor where an
    // independent one issues after 64), summed
    auto mma4 = [&](const double (&fa)[4], const double (&fb)[4]) {
        d4 u = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0], zero4, 0, 0, 0);
        d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], fb[1], zero4, 0, 0, 0);
        u = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], fb[2], u, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], fb[3], v, 0, 0, 0);
        return u + v;
    };
    // two block products at once, their chains interleaved
    auto mma4x2 = [&](const double (&fa)[4], const double (&fb)[4], d4& acc0, const double (&ga)[4], const double (&gb)[4], d4& acc1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[q], fb[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[q], gb[q], acc1, 0, 0, 0);
        }
    };
    // entry `lane` of the list of trailing blocks (see P2): column K = 7, 6, ... with its blocks I = K..7
    int mytask = 0x77;
    {
        int start = 0;
#pragma unroll
        for (int K = 7; K >= 1; --K) {
            if (lane >= start && lane < start + 8 - K) mytask = (K + lane - start) << 4 | K;
            start += 8 - K;
        }
    }
    // ---- phase 1: factorisation
    if (w == 0) diag_block(0);
    lds_barrier();
    stamp(2);
    for (int J = 0; J < 8; ++J) {
        if (J >= JN) break;
        {   // P1: S(I,J) <- S(I,J) L_JJ^-T, I > J; wave w takes I = J+1+w and J+1+w+4.  register r = element (l15, l4 + 4r)
            // of the block: it goes to the image and, being final, to the tile in global memory
            const int m = 7 - J;
            if (w < m) {
                double* d0 = S + off(J + 1 + w, J);
                const bool two = w + 4 < m;
                double* d1 = two ? S + off(J + 1 + w + 4, J) : d0;
                double la[4], b0[4], b1[4];
                frag(S + off(J, J), la);            // (the slot's upper part is zero)
                frag(d0, b0);
                frag(d1, b1);
                d4 acc0 = zero4, acc1 = zero4;
                if (two) mma4x2(la, b0, acc0, la, b1, acc1);
                else acc0 = mma4(la, b0);
                const gf64_ptr g0 = AS_GLOBAL_F64(tk.T + (size_t)(16 * (J + 1 + w) + l15) + (size_t)(16 * J + l4) * tk.ld);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0[(l4 + 4 * r) * PLD + l15] = acc0[r];
                    g0[(size_t)(4 * r) * tk.ld] = acc0[r];
                }
                if (two) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d1[(l4 + 4 * r) * PLD + l15] = acc1[r];
                        g0[(size_t)(4 * r) * tk.ld + 64] = acc1[r];
                    }
                }
            }
            // z_J = L_JJ^-1 w_J (w_J had its last update in P2 of step J - 1; the slot's upper part is zero)
            if (fuse && w == 3 && lane < 16) {
                const double* linv = S + off(J, J) + lane;
                double sum = 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) sum = fma(linv[c * PLD], wl[16 * J + c], sum);
                zl[16 * J + lane] = sum;
            }
        }
        lds_barrier();
        stamp(3 + 2 * J);
        const int m = 7 - J;
        if (w == 0) {
            if (m > 0) {   // S(J+1,J+1) -= S(J+1,J) S(J+1,J)^T, then its factorisation: the chain every block step waits for
                {
                    double fa[4], c[4];
                    double* dst = S + off(J + 1, J + 1);
                    frag(S + off(J + 1, J), fa);
                    frag(dst, c);
                    const d4 acc = mma4(fa, fa);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(l4 + 4 * r) * PLD + l15] = c[r] - acc[r];
                }
                unsigned long long cyc0 = 0;
                if (J == 3) {
                    stamp(21);
                    if constexpr (STAMP) cyc0 = stamp_now();
                }
                diag_block(J + 1);
                if (J == 3) {
                    if constexpr (STAMP) {
                        if (t == 0) st[23] = stamp_now() - cyc0;     // shader cycles of the same interval
                    }
                    stamp(22);
                }
            }
        } else {
            // trailing products S(I,K) -= S(I,J) S(K,J)^T, I >= K > J, two at a time (independent chains interleave on the
            // matrix pipe).  Task list: the blocks (I,K) column by column from the LAST column backwards -- column K has 8 - K
            // blocks whatever the step, so the list of step J is the first (7-J)(8-J)/2 entries of one fixed list (lane l keeps
            // entry l, set up before the loop: no table in memory).  (J+1,J+1) -- the first block of the last column of the
            // step -- is wave 0's; the others go round-robin over waves 1..3.
            const int total = (7 - J) * (8 - J) / 2, own0 = total - (7 - J);
            auto block_of = [&](int p, const double*& pa, const double*& pb) {      // position p among the others
                const int e = __builtin_amdgcn_readlane(mytask, p < own0 ? p : p + 1);
                const int I = (e >> 4) & 15, K = e & 15;
                pa = S + off(K, J);
                pb = S + off(I, J);
                return S + off(I, K);
            };
            for (int p = w - 1; p < total - 1; p += 6) {
                const bool two = p + 3 < total - 1;
                const double *pa0, *pb0, *pa1, *pb1;
                double* d0 = block_of(p, pa0, pb0);
                double* d1 = block_of(two ? p + 3 : p, pa1, pb1);
                double fa[4], fb[4], c0[4], ga[4], gb[4], c1[4];
                frag(pa0, fa);
                frag(pb0, fb);
                frag(pa1, ga);
                frag(pb1, gb);
                frag(d0, c0);
                frag(d1, c1);
                d4 acc0 = zero4, acc1 = zero4;
                mma4x2(fa, fb, acc0, ga, gb, acc1);
#pragma unroll
                for (int r = 0; r < 4; ++r) d0[(l4 + 4 * r) * PLD + l15] = c0[r] - acc0[r];
                if (two) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) d1[(l4 + 4 * r) * PLD + l15] = c1[r] - acc1[r];
                }
            }
            // w_I -= L(I,J) z_J for the rows below block J, one row per thread of the waves 1..3
            const int row = 16 * (J + 1) + 64 * (w - 1) + lane;      // (w = role 1..3: 192 threads for at most 112 rows)
            if (fuse && row < TB) {
                const double* lrow = S + soff[(row >> 4) * 8 + J] + (row & 15);
                double sum = 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) sum = fma(lrow[c * PLD], zl[16 * J + c], sum);
                wl[row] -= sum;
            }
        }
        lds_barrier();
        stamp(4 + 2 * J);
    }
    bad = __shfl(bad, 0);
    if (fuse && t < TB) tk.zk[t] = zl[t];
    if (w == 0 && lane < 32) {             // identity diagonal blocks of the skipped steps (block JN went through diag_block)
        const int c = l15;
        const bool isb = (lane & 16) != 0;
        for (int J = JN + 1; J < 8; ++J) {
            double* gdst = isb ? tk.Dinv + (size_t)(16 * J) + (size_t)(16 * J + c) * TB
                               : tk.T + (size_t)(16 * J) + (size_t)(16 * J + c) * tk.ld;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                d2 v;
                v[0] = (r == c) ? 1.0 : 0.0;
                v[1] = (r + 1 == c) ? 1.0 : 0.0;
                *reinterpret_cast<d2*>(gdst + r) = v;
            }
        }
    }
    stamp(19);
    if constexpr (INVERSE) packed_inverse_phase(tk, S, mytab, JN);
    stamp(20);
    if (w == 0 && lane == 0 && bad != 0 && bad <= tk.nvalid && *tk.info == 0) *tk.info = tk.row0 + bad;
}

__global__ __launch_bounds__(256, 2) void chol_diag_packed_kernel(const DiagTask* __restrict__ tasks) {
    extern __shared__ __attribute__((aligned(16))) double S[];   // image [64 cols][PLD rows] + rhs[128] + z[128] + int off[64]
    const DiagTask tk = tasks[blockIdx.x];
    chol_diag_packed_body(tk, S, false);
}
// The whole inverse of diagonal blocks that were factorised without it (fused block steps): the image is re-loaded -- blocks
// of L from the tile, L_JJ^-1 from the diagonal of Dinv_k, exactly what phase 1 leaves behind -- and phase 2 runs on it.
__global__ __launch_bounds__(256, 2) void dinv_complete_kernel(const DiagTask* __restrict__ tasks) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const DiagTask tk = tasks[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mytab = PACKED.off[lane];
    const int myblk = PACKED.lower[lane < 36 ? lane : 0];
    const int mc = lane >> 3, mr = 2 * (lane & 7);
    d2 v[9][2];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int e = __builtin_amdgcn_readlane(myblk, w + 4 * q);
        const int I = e >> 4, K = e & 15;
        const double* src = (I == K) ? tk.Dinv + (size_t)(16 * I + mr) + (size_t)(16 * K + mc) * TB
                                     : tk.T + (size_t)(16 * I + mr) + (size_t)(16 * K + mc) * tk.ld;
        const size_t ld = (I == K) ? (size_t)TB : (size_t)tk.ld;
        v[q][0] = *AS_GLOBAL_D2(src);
        v[q][1] = *AS_GLOBAL_D2(src + 8 * ld);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int e = __builtin_amdgcn_readlane(myblk, w + 4 * q);
        double* dst = S + __builtin_amdgcn_readlane(mytab, (e >> 4) * 8 + (e & 15)) + mc * PLD + mr;
        *reinterpret_cast<d2*>(dst) = v[q][0];
        *reinterpret_cast<d2*>(dst + 8 * PLD) = v[q][1];
    }
    __syncthreads();
    packed_inverse_phase(tk, S, mytab, (tk.nvalid + 15) >> 4);
}

#ifdef DSMGP_DIAG
__global__ __launch_bounds__(256, 2) void chol_diag_packed_stamp_kernel(const DiagTask* __restrict__ tasks, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const DiagTask tk = tasks[blockIdx.x];
    chol_diag_packed_body<true>(tk, S, false, stamps + 24 * blockIdx.x);
}
#endif


// ---------------------------------------------------------------------------------------------
// Triangular solves alpha = L^-T (L^-1 y) (src/gaussianprocess.jl:105) as block sweeps that reuse
// the inverted diagonal blocks; one launch per block step, batched over leaves.
struct SolveTask {
    const double* T;     // off-diagonal tile (forward: L[i,k]; backward: L[k,j]); unused when self
    const double* Dk;    // inverse of the diagonal block of this step
    const double* vk;    // right-hand side block of this step (128)
    double* out_k;       // self: solution block of this step
    double* vi;          // other: block to update
    int ldt;
    int self;
};

__device__ inline void block_reduce_store(double partial, double* red, int t) {
    // 256 threads: thread (r = t&127, h = t>>7) holds half sums; result for row r in red[r]
    if (t >= TB) red[t - TB] = partial;
    __syncthreads();
    if (t < TB) red[t] += partial;
    __syncthreads();
}

// forward: z_k = Dk * w_k ; w_i -= L[i,k] z_k
__global__ __launch_bounds__(256) void solve_fwd_kernel(const SolveTask* __restrict__ tasks) {
    __shared__ double vin[TB], z[TB], red[TB];
    const SolveTask tk = tasks[blockIdx.x];
    const int t = threadIdx.x, r = t & 127, h = t >> 7;
    if (t < TB) vin[t] = tk.vk[t];
    __syncthreads();
    double s = 0.0;
    for (int c = h * 64; c < h * 64 + 64; ++c)
        if (c <= r) s = fma(tk.Dk[r + (size_t)c * TB], vin[c], s);
    block_reduce_store(s, red, t);
    if (t < TB) z[t] = red[t];
    __syncthreads();
    if (tk.self) {
        if (t < TB) tk.out_k[t] = z[t];
        return;
    }
    s = 0.0;
    for (int c = h * 64; c < h * 64 + 64; ++c) s = fma(tk.T[r + (size_t)c * tk.ldt], z[c], s);
    block_reduce_store(s, red, t);
    if (t < TB) tk.vi[t] -= red[t];
}

// backward sweep alpha = L^-T z on w (a copy of z), one launch per block step, block kb = nb-1-s of each leaf:
//   self task  (first launch only): alpha_kb = D_kb^T w_kb
//   other tasks: w_j -= L[kb,j]^T alpha_kb (alpha_kb read from memory: tk.vk), and the task of j = kb-1, whose
//   block is final after this update, goes on to alpha_j = D_j^T w_j (tk.Dk / tk.out_k set), so every step
//   reads each off-diagonal tile once and each inverse block once.
__device__ __forceinline__ void matvec_t128(const double* __restrict__ M, int ld, const double* __restrict__ x /* LDS, 128 */,
                                            double* __restrict__ y /* LDS, 128 */, int lane, int w) {
    // y = M^T x for a 128x128 column-major block: lanes over rows, 8 columns per wave in flight (the loads of a
    // group are issued together; the sweep is a chain of dependent launches, so latency is what it pays for)
    const double x0 = x[lane], x1 = x[lane + 64];
    const gf64_ptr Mg = AS_GLOBAL_F64(M);
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        double p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = w + 4 * (g * 8 + j);
            p[j] = Mg[lane + (size_t)c * ld] * x0 + Mg[lane + 64 + (size_t)c * ld] * x1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] += __shfl_down(p[j], o);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) y[w + 4 * (g * 8 + j)] = p[j];
        }
    }
}

__global__ __launch_bounds__(256) void solve_bwd_kernel(const SolveTask* __restrict__ tasks) {
    __shared__ double vin[TB], a[TB];
    const SolveTask tk = tasks[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < TB) vin[t] = tk.vk[t];
    __syncthreads();
    if (tk.self) {
        matvec_t128(tk.Dk, TB, vin, a, lane, w);
        __syncthreads();
        if (t < TB) tk.out_k[t] = a[t];
        return;
    }
    matvec_t128(tk.T, tk.ldt, vin, a, lane, w);      // a = L[kb,j]^T alpha_kb
    __syncthreads();
    if (t < TB) {
        const double v = tk.vi[t] - a[t];
        tk.vi[t] = v;
        vin[t] = v;
    }
    if (tk.Dk == nullptr) return;
    __syncthreads();
    matvec_t128(tk.Dk, TB, vin, a, lane, w);         // block j is final: alpha_j = D_j^T w_j
    __syncthreads();
    if (t < TB) tk.out_k[t] = a[t];
}

// ---------------------------------------------------------------------------------------------
// per-leaf scalars and vectors
struct LeafDev {
    double* F;            // npad x npad factor (shared by COPY leaves)
    double* Dinv;         // nb x 128 x 128
    const double* Xg;     // gathered inputs, npad x D (ld = npad), zero padded
    double* yc;           // y - mean, zero padded
    double* w;            // work vector
    double* z;            // L^-1 yc
    double* alpha;        // L^-T z
    int* info;
    double mean;
    int n, npad, nb, kid;
    // prediction
    double* Vt;           // ntpad x npad (ld = ntpad): K_tn, then K_tn L^-T
    const double* Xtg;    // gathered test inputs, ntpad x D (ld = ntpad)
    double* mu;           // nt (contiguous over leaves in route order)
    double* var;          // nt
    double* macc;         // ntpad: running V^T z of the sweep (panel-solve epilogue)
    double* sacc;         // ntpad: running rowsumsq(V^T)
    int nt, ntpad;
    int zfused;           // z is produced during the factorisation (leaf factorised in full)
    int pad1;
};

// yc = y[obs] - mean, Xg = X[obs, :]; one workgroup per (leaf, 256-row slab)
__global__ void gather_leaf_kernel(const LeafDev* __restrict__ leaves, const int64_t* __restrict__ obs_ptr,
                                   const int64_t* __restrict__ obs_idx, const double* __restrict__ X,
                                   const double* __restrict__ y, int64_t N, int D, int leaf0) {
    const LeafDev lf = leaves[leaf0 + blockIdx.y];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lf.npad) return;
    const bool valid = r < lf.n;
    const int64_t g = valid ? obs_idx[obs_ptr[leaf0 + blockIdx.y] + r] : 0;
    lf.yc[r] = valid ? y[g] - lf.mean : 0.0;
    double* xg = const_cast<double*>(lf.Xg);
    for (int d = 0; d < D; ++d) xg[r + (size_t)d * lf.npad] = valid ? X[g + (size_t)d * N] : 0.0;
}

__global__ void gather_test_kernel(const LeafDev* __restrict__ leaves, const int64_t* __restrict__ route_ptr,
                                   const int64_t* __restrict__ route_idx, const double* __restrict__ Xt,
                                   int64_t n_t, int D, int leaf0) {
    const LeafDev lf = leaves[leaf0 + blockIdx.y];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lf.ntpad) return;
    const bool valid = r < lf.nt;
    const int64_t g = valid ? route_idx[route_ptr[leaf0 + blockIdx.y] + r] : 0;
    double* xg = const_cast<double*>(lf.Xtg);
    for (int d = 0; d < D; ++d) xg[r + (size_t)d * lf.ntpad] = valid ? Xt[g + (size_t)d * n_t] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Routing of test rows on the device (src/common.jl:101-122,181-196,275-292: a sum node forwards a row to every child, a split
// node to the first child k with x[d] <= s_k): predict(model, x) on rows the model has not seen spent more host time walking
// the tree and building index lists than the device spent on the prediction sweep (depth 4: 16 + 32 ms around 16 ms).
// The tree is a few hundred KB of flat arrays (dsmgp_set_tree); one thread per test row walks it depth-first, children in
// order -- leaves are numbered depth-first, so a row meets its leaves in ascending leaf order.  Two walks and a bitmap make the
// result independent of thread timing (no sort, no order-dependent atomics):
//   walk 1   per row: number of leaves reached; bit (leaf, row) set in a bitmap of L x ceil(n_t / 32) words
//   scan     row_ptr (entries of every row, for the aggregation)
//   rank     per leaf: running bit counts of its bitmap words, its row count; scan -> route_ptr
//   fill     per leaf: the set bits in order = its rows ascending -> route_idx, ent_leaf
//   walk 2   per row, i-th leaf reached: entry position = route_ptr[leaf] + bits below the row's in that leaf's bitmap -> row_ent
// Equal, entry by entry, to what dsmgp_set_test builds on the host from dsmgp_tree_route's lists.
template <bool FILL>
__global__ __launch_bounds__(256) void route_walk_kernel(RouteTree t, const double* __restrict__ x, int64_t n_t,
                                                         int32_t* __restrict__ row_cnt, uint32_t* __restrict__ bitmap, int64_t wpl,
                                                         int* __restrict__ outside, const int64_t* __restrict__ row_ptr,
                                                         const int64_t* __restrict__ route_ptr, const uint32_t* __restrict__ wprefix,
                                                         int32_t* __restrict__ row_ent) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_t) return;
    const uint32_t bit = 1u << (r & 31);
    const int64_t w = r >> 5;
    const int64_t base = FILL ? row_ptr[r] : 0;
    const int cnt = route_walk_row(t, x, 1, n_t, r, [&](int l, int i) {      // (route_walk.hpp: the walk the host routine runs too)
        const size_t o = (size_t)l * (size_t)wpl + (size_t)w;
        if (FILL) row_ent[base + i] = (int32_t)(route_ptr[l] + (int64_t)wprefix[o] + (int64_t)__popc(bitmap[o] & (bit - 1u)));
        else atomicOr(&bitmap[o], bit);
    });
    if (!FILL) {
        if (cnt < 0) atomicExch(outside, 1);
        row_cnt[r] = cnt < 0 ? 0 : cnt;
    }
}

// out[0] = 0, out[i + 1] = in[0] + ... + in[i]: one workgroup walks the array in chunks of 1024 with a running carry (the arrays
// here are the per-row and per-leaf counts: 10^4 .. 10^6 entries, microseconds)
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    __shared__ int64_t buf[1024];
    __shared__ int64_t carry;
    const int t = threadIdx.x;
    if (t == 0) {
        carry = 0;
        out[0] = 0;
    }
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += 1024) {
        const int64_t i = i0 + t;
        buf[t] = i < n ? (int64_t)in[i] : 0;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            const int64_t v = t >= s ? buf[t - s] : 0;
            __syncthreads();
            buf[t] += v;
            __syncthreads();
        }
        if (i < n) out[i + 1] = carry + buf[t];
        __syncthreads();
        if (t == 1023) carry += buf[1023];
        __syncthreads();
    }
}

// per leaf (one workgroup): wprefix[w] = set bits in the leaf's bitmap words before word w; leaf_cnt = all of them
__global__ __launch_bounds__(256) void route_rank_kernel(const uint32_t* __restrict__ bitmap, int64_t wpl, uint32_t* __restrict__ wprefix,
                                                         int32_t* __restrict__ leaf_cnt) {
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    const int t = threadIdx.x;
    const size_t o = (size_t)blockIdx.x * (size_t)wpl;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t w0 = 0; w0 < wpl; w0 += 256) {
        const int64_t w = w0 + t;
        const uint32_t c = w < wpl ? (uint32_t)__popc(bitmap[o + w]) : 0u;
        buf[t] = c;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {
            const uint32_t v = t >= s ? buf[t - s] : 0u;
            __syncthreads();
            buf[t] += v;
            __syncthreads();
        }
        if (w < wpl) wprefix[o + w] = carry + buf[t] - c;
        __syncthreads();
        if (t == 255) carry += buf[255];
        __syncthreads();
    }
    if (t == 0) leaf_cnt[blockIdx.x] = (int32_t)carry;
}

// per leaf (one workgroup): its rows, ascending, into the CSR; the leaf of every entry for the aggregation
__global__ __launch_bounds__(256) void route_fill_kernel(const uint32_t* __restrict__ bitmap, int64_t wpl, const uint32_t* __restrict__ wprefix,
                                                         const int64_t* __restrict__ route_ptr, int64_t* __restrict__ route_idx,
                                                         int32_t* __restrict__ ent_leaf) {
    const int l = blockIdx.x;
    const size_t o = (size_t)l * (size_t)wpl;
    const int64_t p0 = route_ptr[l];
    for (int64_t w = threadIdx.x; w < wpl; w += blockDim.x) {
        uint32_t bits = bitmap[o + w];
        int64_t pos = p0 + (int64_t)wprefix[o + w];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            route_idx[pos] = w * 32 + b;
            ent_leaf[pos] = l;
            ++pos;
        }
    }
}

// mll = -(y.alpha + 2 sum log L_ii + n log 2pi)/2   (src/gaussianprocess.jl:163), with y.alpha evaluated as z.z
// (z = L^-1 y, alpha = L^-T z  =>  y.alpha = (L z).(L^-T z) = z.z): the log-marginal needs the forward substitution
// only, so fit! does not run the backward sweep; alpha is materialised on first use (ensure_alpha).
__global__ __launch_bounds__(256) void mll_kernel(const LeafDev* __restrict__ leaves, double* __restrict__ mll_out) {
    __shared__ double red[256];
    __shared__ double red2[256];
    const LeafDev lf = leaves[blockIdx.x];
    const int t = threadIdx.x;
    double s = 0.0, ld = 0.0;
    for (int i = t; i < lf.n; i += 256) {
        s = fma(lf.z[i], lf.z[i], s);
        ld += log(lf.F[i + (size_t)i * lf.npad]);
    }
    red[t] = s;
    red2[t] = ld;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            red[t] += red[t + o];
            red2[t] += red2[t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        const double log2pi = 1.8378770664093454835606594728112;
        mll_out[blockIdx.x] = -(red[0] + 2.0 * red2[0] + log2pi * (double)lf.n) / 2.0;
    }
}

// w = z (start of the backward sweep; z itself stays for the predictive mean)
__global__ void copy_z_kernel(const LeafDev* __restrict__ leaves) {
    const LeafDev lf = leaves[blockIdx.y];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < lf.npad) lf.w[r] = lf.z[r];
}

// w = yc (start of the forward sweep)
__global__ void copy_vec_kernel(const LeafDev* __restrict__ leaves) {
    const LeafDev lf = leaves[blockIdx.y];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < lf.npad) lf.w[r] = lf.yc[r];
}

// ---------------------------------------------------------------------------------------------
// prediction helpers (src/gaussianprocess.jl:117-126), one workgroup per (leaf, 128-row test tile)
struct PredTask {
    int leaf;
    int row0;
};

// mu = m + K_tn alpha = m + (K_tn L^-T)(L^-1 y) = m + V^T z, from the solved rows
__global__ __launch_bounds__(256) void pred_mu_kernel(const LeafDev* __restrict__ leaves,
                                                      const PredTask* __restrict__ tasks) {
    __shared__ double red[TB];
    const PredTask tk = tasks[blockIdx.x];
    const LeafDev lf = leaves[tk.leaf];
    const int t = threadIdx.x, r = t & 127, h = t >> 7;
    const double* V = lf.Vt + tk.row0 + r;
    double s = 0.0;
    for (int c = h; c < lf.n; c += 2) s = fma(V[(size_t)c * lf.ntpad], lf.z[c], s);
    block_reduce_store(s, red, t);
    if (t < TB && tk.row0 + t < lf.nt) lf.mu[tk.row0 + t] = lf.mean + red[t];
}

// mu = m + macc, var = k(x*,x*) + noise - sacc for the rows whose moments were accumulated by the sweep
__global__ __launch_bounds__(128) void pred_finish_kernel(const LeafDev* __restrict__ leaves, const PredTask* __restrict__ tasks,
                                                          const KParam* __restrict__ kp, int D) {
    const PredTask tk = tasks[blockIdx.x];
    const LeafDev lf = leaves[tk.leaf];
    const KParam p = kp[lf.kid];
    const int r = tk.row0 + threadIdx.x;
    if (r >= lf.nt) return;
    double kss;
    if (p.kind == 0) kss = p.sigma2;
    else if (p.kind == 1) kss = p.sigma2 * (double)D;
    else {
        double q = 0.0;
        for (int d = 0; d < D; ++d) {
            const double x = lf.Xtg[r + (size_t)d * lf.ntpad];
            q = fma(x, x, q);
        }
        kss = q / p.l2[0];
    }
    lf.mu[r] = lf.mean + lf.macc[r];
    lf.var[r] = (kss - lf.sacc[r]) + p.noise;
}

// var = k(x*,x*) + noise - sum_c V(t,c)^2
__global__ __launch_bounds__(256) void pred_var_kernel(const LeafDev* __restrict__ leaves,
                                                       const PredTask* __restrict__ tasks,
                                                       const KParam* __restrict__ kp, int D) {
    __shared__ double red[TB];
    const PredTask tk = tasks[blockIdx.x];
    const LeafDev lf = leaves[tk.leaf];
    const KParam p = kp[lf.kid];
    const int t = threadIdx.x, r = t & 127, h = t >> 7;
    const double* V = lf.Vt + tk.row0 + r;
    double s = 0.0;
    for (int c = h; c < lf.n; c += 2) {
        const double v = V[(size_t)c * lf.ntpad];
        s = fma(v, v, s);
    }
    block_reduce_store(s, red, t);
    if (t < TB && tk.row0 + t < lf.nt) {
        double kss;
        if (p.kind == 0) kss = p.sigma2;
        else if (p.kind == 1) kss = p.sigma2 * (double)D;
        else {
            double q = 0.0;
            for (int d = 0; d < D; ++d) {
                const double x = lf.Xtg[tk.row0 + t + (size_t)d * lf.ntpad];
                q = fma(x, x, q);
            }
            kss = q / p.l2[0];
        }
        lf.var[tk.row0 + t] = (kss - red[t]) + p.noise;
    }
}

// per leaf: y.alpha and alpha.alpha (inputs of the gradient assembly)
__global__ __launch_bounds__(256) void dots_kernel(const LeafDev* __restrict__ leaves, double* __restrict__ out) {
    __shared__ double r1[256], r2[256];
    const LeafDev lf = leaves[blockIdx.x];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = t; i < lf.n; i += 256) {
        a = fma(lf.yc[i], lf.alpha[i], a);
        b = fma(lf.alpha[i], lf.alpha[i], b);
    }
    r1[t] = a;
    r2[t] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            r1[t] += r1[t + o];
            r2[t] += r2[t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        out[2 * blockIdx.x] = r1[0];
        out[2 * blockIdx.x + 1] = r2[0];
    }
}

// ---------------------------------------------------------------------------------------------
// predict(model, x): sum/product aggregation of the leaf moments over the leaves every test row visits
// (src/common.jl:134-149,198-302) and the score functions (src/scorefunctions.jl:6-16), on the moments left in HBM
// by pred_finish_kernel.  All four model families reduce to per-row sums over (leaf, row) entries:
//   mixture (DSMGP, :275-302): the nested log-domain recursion is linear in (mu, mu^2, sigma^2) -- unrolled it is the
//       flat mixture with weight W_l = product of the sum-node weights on leaf l's path: S0 = sum W mu,
//       S1 = sum W mu^2, S2 = sum W sigma^2 (sigma^2 <= 0 -> 1e-8, :137); mu = S0, var = S2 + S1 - S0^2
//   PoE / gPoE (:145-149,198-222): t = 1/sigma^2; S0 = sum beta t mu, S1 = sum beta t (beta = 1 resp. 1/#root children)
//   rBCM (:224-241): per root child g the PoE sums (S_g0, S_g1) = (sum t mu, sum t); combined in agg_finish_kernel
// Partial sums are stored [k][row] (W vectors of length n_t): with leaves sharded over ranks or contexts each one
// produces its partial sums and they are added before the finish step.  One thread per test row walks that row's
// entries in ascending entry order (= leaf order): fixed summation order, no atomics.
constexpr int AGG_MIXTURE = 0, AGG_POE = 1, AGG_GPOE = 2, AGG_RBCM = 3;

struct AggArgs {
    const int64_t* row_ptr;     // n_t + 1: entries of every test row
    const int32_t* row_ent;     // entry positions (index into mu / var), ascending per row
    const int32_t* ent_leaf;    // leaf of every entry position
    const double* mu;           // per (leaf, routed row), route order
    const double* var;
    const double* coef;         // per leaf: W_l (mixture) or beta_l (PoE / gPoE); unused for rBCM
    const int32_t* group;       // per leaf: root child (rBCM); else unused
    double* part;               // W x n_t
    int64_t n_t;
    int family;
    int G;                      // rBCM: number of groups
};

__global__ __launch_bounds__(256) void agg_partial_kernel(AggArgs a) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_t) return;
    const int64_t e0 = a.row_ptr[r], e1 = a.row_ptr[r + 1];
    if (a.family == AGG_MIXTURE) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int64_t e = e0; e < e1; ++e) {
            const int pos = a.row_ent[e];
            const double w = a.coef[a.ent_leaf[pos]];
            const double m = a.mu[pos];
            double v = a.var[pos];
            if (v <= 0.0) v = 1e-8;                               // src/common.jl:137
            s0 += w * m;
            s1 += w * (m * m);
            s2 += w * v;
        }
        a.part[r] = s0;
        a.part[a.n_t + r] = s1;
        a.part[2 * a.n_t + r] = s2;
    } else if (a.family == AGG_RBCM) {
        for (int g = 0; g < a.G; ++g) {
            a.part[(size_t)(2 * g) * a.n_t + r] = 0.0;
            a.part[(size_t)(2 * g + 1) * a.n_t + r] = 0.0;
        }
        for (int64_t e = e0; e < e1; ++e) {
            const int pos = a.row_ent[e];
            const int g = a.group[a.ent_leaf[pos]];
            const double t = 1.0 / a.var[pos];                    // src/common.jl:148
            a.part[(size_t)(2 * g) * a.n_t + r] += t * a.mu[pos];
            a.part[(size_t)(2 * g + 1) * a.n_t + r] += t;
        }
    } else {
        double s0 = 0.0, s1 = 0.0;
        for (int64_t e = e0; e < e1; ++e) {
            const int pos = a.row_ent[e];
            const double bt = a.coef[a.ent_leaf[pos]] * (1.0 / a.var[pos]);
            s0 += bt * a.mu[pos];
            s1 += bt;
        }
        a.part[r] = s0;
        a.part[a.n_t + r] = s1;
    }
}

// Finish step on the (summed) partial sums: mu / var of predict(model, x), left in HBM for agg_scores_kernel.
// plain: the root is a single GP (predict(node::GPNode), src/common.jl:175-179): no mixture term.
// rBCM: s = k(x*, x*) + noise of the model's first leaf (leftGP, :227-228).
__global__ __launch_bounds__(256) void agg_finish_kernel(const double* __restrict__ part, int64_t n_t, int family, int G,
                                                         int plain, const KParam* __restrict__ kp, int prior_kid,
                                                         const double* __restrict__ Xt, int D,
                                                         double* __restrict__ mu_out, double* __restrict__ var_out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_t) return;
    if (family == AGG_MIXTURE) {
        const double m = part[r], m2 = part[n_t + r], sv = part[2 * n_t + r];
        mu_out[r] = m;
        var_out[r] = plain ? sv : sv + (m2 - m * m);              // src/common.jl:299-300
    } else if (family == AGG_RBCM) {
        const KParam p = kp[prior_kid];
        double kss;
        if (p.kind == 0) kss = p.sigma2;
        else if (p.kind == 1) kss = p.sigma2 * (double)D;
        else {
            double q = 0.0;
            for (int d = 0; d < D; ++d) {
                const double x = Xt[r + (size_t)d * n_t];
                q = fma(x, x, q);
            }
            kss = q / p.l2[0];
        }
        const double s = kss + p.noise;
        double C = 1.0 / s, m = 0.0;
        for (int g = 0; g < G; ++g) {
            const double T = part[(size_t)(2 * g + 1) * n_t + r];
            if (T == 0.0) continue;                               // no leaf of this child saw the row
            const double M = part[(size_t)(2 * g) * n_t + r] / T; // mu_ of the child (:205)
            const double beta = 0.5 * (log(s) - log(1.0 / T));    // :234-235
            C += beta * T - beta / s;                             // :236
            m += M * (beta * T);                                  // :237
        }
        mu_out[r] = m / C;
        var_out[r] = 1.0 / C;
    } else {
        const double t = part[n_t + r];
        mu_out[r] = part[r] / t;
        var_out[r] = 1.0 / t;
    }
}

// Score sums (src/scorefunctions.jl:6-16) in two passes for the standard errors: pass 0 sums se, ae and the nlpd
// terms; pass 1 sums (se - mean se)^2 and (ae - mean ae)^2.  Per-block tree reduction, blocks combined in block order
// by the host: fixed summation order.
__global__ __launch_bounds__(256) void agg_scores_kernel(const double* __restrict__ y, const double* __restrict__ mu,
                                                         const double* __restrict__ var, int64_t n_t, int pass,
                                                         double mean_se, double mean_ae, double* __restrict__ out) {
    __shared__ double red[3][256];
    const int t = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 256 + t;
    double a = 0.0, b = 0.0, c = 0.0;
    if (r < n_t) {
        const double d = y[r] - mu[r];
        const double se = d * d, ae = fabs(d);
        if (pass == 0) {
            a = se;
            b = ae;
            const double sd = sqrt(var[r]);                       // Normal(mu, sqrt(var)), :16
            const double zz = d / sd;
            c = 0.5 * (zz * zz + 1.8378770664093454835606594728112) + log(sd);
        } else {
            a = (se - mean_se) * (se - mean_se);
            b = (ae - mean_ae) * (ae - mean_ae);
        }
    }
    red[0][t] = a;
    red[1][t] = b;
    red[2][t] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            red[0][t] += red[0][t + o];
            red[1][t] += red[1][t + o];
            red[2][t] += red[2][t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        out[3 * (size_t)blockIdx.x] = red[0][0];
        out[3 * (size_t)blockIdx.x + 1] = red[1][0];
        out[3 * (size_t)blockIdx.x + 2] = red[2][0];
    }
}

// Multi-GPU exchange helpers (dsmgp_fit_exchange / dsmgp_aggregate_exchange): pack (mll, info) of the local leaves, and add
// the gathered partial sums in rank order -- a fixed order, so every rank ends with the same bits
__global__ void pack_mll_info_kernel(const double* __restrict__ mll, const int* __restrict__ info, const int* __restrict__ owner,
                                     int L, int64_t count, double* __restrict__ out) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= count) return;
    out[2 * l] = l < L ? mll[l] : 0.0;
    out[2 * l + 1] = l < L ? (double)info[owner[l]] : 0.0;
}

__global__ void sum_ranks_kernel(const double* __restrict__ gathered, int world, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = gathered[i];
    for (int r = 1; r < world; ++r) s += gathered[(size_t)r * n + i];
    out[i] = s;
}

// ---------------------------------------------------------------------------------------------
// f64 MFMA issue-rate probe: register-only chains, 4 independent accumulators per wave.  Lane 0 of
// every wave also records shader-clock cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) so the
// host can report cycles per MFMA and the clock the chip holds under this load.
__global__ __launch_bounds__(256) void mfma_probe_kernel(double* out, unsigned long long* stamps, int iters) {
    // 16 independent accumulators per wave (as in tile_gemm_kernel); inline asm keeps them in VGPRs
    d4 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (d4){0.0, 0.0, 0.0, 0.0};
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double sink = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sink += a[i][i & 3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * wv] = c1 - c0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}

// Shader clock held while other work is resident: one wave that sleeps for `ticks` of the 100 MHz wall counter and reports
// how far the shader-clock counter (s_memtime) moved meanwhile.  Launched on a stream of its own beside the launches of a fit,
// it costs one wave slot on one CU and no matrix-pipe or memory time.  The loop ends when the wall counter says so: every
// launch drains.
__global__ __launch_bounds__(64) void clock_sample_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1;
    do {
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < ticks);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

#ifdef DSMGP_DIAG
// Co-issue probe: do the f64 matrix pipe and the f64 vector pipe run at the same time?  mode 0: every wave
// issues MFMAs; mode 1: every wave issues v_fma_f64; mode 2: even waves MFMA, odd waves v_fma_f64 (two waves per
// SIMD: one of each).  out[wave] = {cycles, ticks}; flops are counted by the host.
__global__ __launch_bounds__(512) void coissue_probe_kernel(double* out, unsigned long long* stamps, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0) || (mode == 2 && (w & 4) == 0);   // waves 0..3 and 4..7 share the SIMDs pairwise
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    double sink = 0.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (do_mfma) {
        d4 a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) sink += a[i][i & 3];
    } else {
        double f[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) f[i] = 1e-3 * i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[i]) : "v"(x), "v"(y));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) sink += f[i];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * (blockDim.x >> 6) + w;
        stamps[2 * wv] = c1 - c0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}
#endif

}  // namespace dsmgp
