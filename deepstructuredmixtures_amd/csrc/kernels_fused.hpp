// Fused block step of the batched Cholesky for launches with MANY leaves (the small-leaf regime: depth >= 3 trees, PoE
// models), gfx950 only.  The classic step of kernels.hpp is three launches that pass every tile of the block column
// through memory twice more than necessary:
//     update  F[i,k] = K(i,k) - F[i,0:K] F[k,0:K]^T     (write the tile)
//     diag    L_kk, Dinv_k                               (read the diagonal tile, write it and its inverse)
//     solve   F[i,k] = F[i,k] Dinv_k^T                   (read the tile, write it again)
// which is what a launch of a few hundred tiles needs -- every tile of the step in ONE launch, the diagonal block on the
// chain of dependent launches kept as short as possible.  With thousands of leaves the diagonal tiles of a step fill the
// chip by themselves, so the step is reordered (src/AdvancedCholeskey.jl:161-171 does the same per leaf: the diagonal
// block first, then the panel below it):
//     diag_fused_reg_kernel  per leaf: S = K(k,k) - F[k,0:K] F[k,0:K]^T (lower blocks; they STAY in the accumulators: round 4),
//                         L_kk = chol(S), z_k, the 16x16 diagonal inverses  -- the tile never exists unfactorised in memory
//     tile_fused8_kernel  per eight 16-row blocks below: C = K(i,k) - F[i,0:K] F[k,0:K]^T stays in the accumulators, which ARE
//                         the second operand of the solve (register r of a 16x16 result = k-slab r of the operand);
//                         X L_kk^T = C by block forward substitution against L_kk's blocks staged once in LDS; every block
//                         is written once
// Two launches per step, no Gram launch for block column 0 (K = 0: the tasks start at the kernel function), and at depth 4
// about 60 GB of the 210 GB a fit moves are not moved.  The arithmetic is that of the classic step -- the products sum over
// K chunk by chunk, the Gram values come from gram_accumulate / gram_finish, the solve sums over j in groups of four
// ascending, the riders reduce inside a wave as in tile_trsm_kernel -- with one difference in ORDER: a block below the
// diagonal starts from -K(i,k) and accumulates the product on it (the kernel function is evaluated first, while the first
// operand loads are in flight) where the classic update subtracts the finished product from K(i,k).  Diagonal blocks are
// bit-identical to the classic step's, the blocks below agree to rounding (test_fused_steps_agree_with_the_classic_steps).
#pragma once
#include <type_traits>
#include <utility>
#include "kernels.hpp"

namespace dsmgp {

#ifndef DSMGP_DIAGR_WGS
#define DSMGP_DIAGR_WGS 3                 // workgroups per CU the register-resident diagonal-block task is compiled for
#endif
#ifndef DSMGP_DIAG_PRIO
#define DSMGP_DIAG_PRIO 0                 // s_setprio of the wave that factorises a 16x16 diagonal block (its chain of ~400 dependent
#endif                                    // f64 vector instructions shares the SIMD's f64 pipe with the other waves' MFMAs)
#ifndef DSMGP_DIAGR_SKIP
#define DSMGP_DIAGR_SKIP 0                // DIAGNOSTIC builds only (tools/probe_diag_fused.py): leave parts of the task out to see what its
#endif                                    // time is made of -- 1 kernel function, 2 the 16x16 factorisations, 4 trailing products, 8 panel solves

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct DiagFusedTask {
    DiagTask d;           // the diagonal block: T (the tile of F), Dinv, wk / zk, info, ld, nvalid, row0
    const double* A;      // F[k, 0:K]: the block row left of the tile, ld = d.ld
    const double* gx;     // coordinates of the block's points, ld glda
    int k1, glda, kid;
    int pad;
};

// The solve of a fused tile task (round 4): block forward substitution against L_kk instead of a product with Dinv_k, so that
// the diagonal-block task of a fused step can leave its whole inverse phase out:
//     X(:, cb) = ( C(:, cb) - sum_{jb < cb} X(:, jb) L(cb, jb)^T ) L(cb, cb)^-T ,      cb = 0 .. 7
// with the 28 blocks L(cb, jb), jb < cb, of the factorised diagonal tile and the eight 16x16 inverses L(cb, cb)^-1 that phase
// 1 of the diagonal-block kernel leaves on the diagonal of Dinv_k: the same 36 block products (144 MFMAs per 16 rows) as the
// product with the inverse, and one rounding-error source fewer.  The 36 blocks sit in LDS, compact, block rows from the FIRST
// one down -- block (cb, jb), jb <= cb, at 256 (lower_block_base(cb) + jb) -- each column-major with leading dimension 16: the
// MFMA operand read (16 rows per k column, four k columns per lane group) touches 64 consecutive doubles, conflict-free.
// 36 x 256 doubles = the ring's 73,728 bytes exactly.  The substitution runs from the first block column to the last, so the
// first blocks of this order are all it needs to start.
__host__ __device__ constexpr int lower_block_base(int cb) { return cb * (cb + 1) / 2; }   // sum_{c < cb} (c + 1)
struct LowerBlocks {
    unsigned char cb[36], jb[36];
    constexpr LowerBlocks() : cb{}, jb{} {
        int n = 0;
        for (int c = 0; c < 8; ++c)
            for (int j = 0; j <= c; ++j) {
                cb[n] = (unsigned char)c;
                jb[n] = (unsigned char)j;
                ++n;
            }
    }
};
// (cb, jb) of lower block b for the staging loads of L_kk.  The block index of a load is <compile-time part> + <wave-uniform
// part>; looked up in the table in memory, every one of a thread's loads waited for its own two table bytes first -- 18
// dependent round trips where one would do (found in the ISA in round 4: global_load_ubyte / s_waitcnt vmcnt(1) chains).
// Packed here into immediates, selected with scalar shifts: the loads issue back to back.
constexpr LowerBlocks LOWER_BLOCKS_CT{};
template <int STRIDE>       // blocks b0 .. b0 + STRIDE - 1 (clamped to nblk - 1), 8 bits each: cb in the low, jb in the high nibble
__host__ __device__ constexpr unsigned lower_block_pack(int b0, int nblk) {
    unsigned v = 0;
    for (int i = 0; i < STRIDE; ++i) {
        const int b = (b0 + i < nblk - 1) ? b0 + i : nblk - 1;
        v |= (unsigned)(LOWER_BLOCKS_CT.cb[b] | (LOWER_BLOCKS_CT.jb[b] << 4)) << (8 * i);
    }
    return v;
}
static_assert(lower_block_base(0) == 0 && lower_block_base(5) == 15 && lower_block_base(7) == 28, "block order");

// acc <- -k(row, col) in the accumulator layout of the fused tile task (wave w owns the rows 16 NRW w ..; register q of
// acc[cb][rn] is the entry (row = 16 NRW w + 16 rn + l15, col = 16 cb + l4 + 4 q)); the operations of gram_half_tile in its order (gram_accumulate /
// gram_finish): the values the Gram launch would have written, negated.  The product is then accumulated ON it, so the
// task starts with the kernel function -- under the memory latency of its first operand loads, with the accumulators not
// yet live -- and the accumulators end as -(K - A B^T) = -C.
template <int KIND, int NRW, int NCB = 8, bool EDGE = true>    // EDGE = false: every row and column of the wave's blocks holds data (no masks)
__device__ __forceinline__ void rowsplit_gram_init(int gna, int gnb, const KParam& p, int D, d4 (&acc)[NCB][NRW], const double* sa,
                                                      const double* sb) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int rowbase = 16 * NRW * w + l15;
    const double* pa = sa + rowbase;
    constexpr int CG = (NRW == 1 && NCB >= 2) ? 2 : 1;      // column blocks per pass: 8 entries per lane and pass where possible
    constexpr int NJ = 4 * CG;
#pragma unroll
    for (int cp = 0; cp < NCB / CG; ++cp) {
        const double* pb = sb + 16 * CG * cp + l4;
        double z[NRW][NJ];
#pragma unroll
        for (int rn = 0; rn < NRW; ++rn)
#pragma unroll
            for (int j = 0; j < NJ; ++j) z[rn][j] = 0.0;
        for (int d = 0; d < D; ++d) {
            double a[NRW], b[NJ];
#pragma unroll
            for (int i = 0; i < NRW; ++i) a[i] = pa[d * TB + 16 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = pb[d * TB + 16 * (j >> 2) + 4 * (j & 3)];
            gram_accumulate<KIND, NRW, NJ>(z, a, b, (KIND == 1) ? AS_CONST_F64(p.nh)[d] : 0.0);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cidx = 16 * CG * cp + 16 * (j >> 2) + l4 + 4 * (j & 3);
#pragma unroll
            for (int rn = 0; rn < NRW; ++rn) {
                acc[CG * cp + (j >> 2)][rn][j & 3] = -gram_finish<KIND, EDGE>(z[rn][j], p, rowbase + 16 * rn, cidx, gna, gnb, false);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused tile task, EIGHT waves (round 4).  Until then the task was one 128 x 128 tile in a workgroup of four waves (32 rows a
// wave, 16 for tiles of <= 64 data rows; 228-254 registers): two workgroups per CU, two waves per SIMD -- and one wave cannot
// keep the f64 matrix pipe busy (it issues an MFMA about every 100 cycles where the pipe takes one every 64), while every phase
// change of a task (coordinates, kernel function, product, L_kk, solve, store) left its SIMD to the one other wave: at depth 4
// the launches ran at 60 % of their matrix-pipe time.  Here a wave owns ONE 16-row block and all 128 columns (acc[8][1]): 128
// registers, so two workgroups of eight waves share a CU -- FOUR waves per SIMD.  And because a wave stages, evaluates, solves
// and stores its own 16 rows, the eight row blocks of a task need not come from one 128-row tile: a task is ANY eight 16-row
// blocks below the same diagonal block -- the short last row tile of a leaf's factor and its routed test rows share tasks,
// nothing is padded beyond 16 rows (executed / algorithmic rows of the depth-4 model 1.31 -> 1.08; 221k tile tasks -> 191k).
// Per row: -k(row, col) first, the product accumulated on it chunk by chunk, block forward substitution against L_kk with two
// accumulators (a dependent f64 MFMA waits ~100 cycles for its predecessor where an independent one issues after 64), riders
// reduced inside the wave.  Depth 4, same box: fused tile launches 36.9 -> 34.4 ms, step 0.0518 -> 0.0499 s; PoE (config 3)
// 3.8 -> 3.7 ms; the five shallow steps of the headline model, all whole tiles, 7.4 -> 7.7 ms (profiles/r04_fused8_ab.log).
struct RowBlock {
    const double* A;      // the 16 rows of the row panel (F[16 r .., 0:K] or rows of Vt), ld lda; nullptr = no block: the wave only stages
    double* C;            // where the solved 16 x 128 block goes, ld ldc
    const double* gx;     // coordinates of the rows, ld glda
    double* wi;           // rider: train rows w_i -= X z_k (forward substitution); test rows (sq set) mu += X z_k.  nullptr = none
    double* sq;           // test rows: sum of squares of the solved row
    int lda, ldc, glda;
    int nvalid;           // rows that hold data (1..16); the others are written as zeros
    int pad[2];
};
static_assert(sizeof(RowBlock) == 64, "RowBlock is read with scalar loads: one cache line");
struct FusedTask8 {
    const double* B;      // column panel F[k, 0:K], ld ldb; L_kk sits behind it (column k1)
    const double* Dinv;   // Dinv_k: its diagonal 16x16 blocks are read
    const double* zk;     // z_k for the riders
    const double* gxb;    // coordinates of the columns, ld gldb
    int ldb, gldb, gnb, k1;
    int kid, nblk;
    int pad[2];
    RowBlock rb[8];
};
static_assert(sizeof(FusedTask8) == 64 + 8 * 64, "FusedTask8 layout");

template <int NCB>
__device__ __forceinline__ void tile_fused8_body(const FusedTask8* __restrict__ task, double* smem, const KParam* __restrict__ kp, int D) {
    static_assert(NCB == 8 || NCB == 4 || NCB == 2 || NCB == 1, "column-block classes");
    double (*sA)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(smem);
    double (*sB)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(smem + NRING * KC2 * LDP);
    double* sD = smem;                          // the 36 lower blocks of L_kk take the ring's place after the product
    double* sZ = smem + 2 * NRING * KC2 * LDP;  // z_k
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);       // 0..7: the row block
    const int l15 = lane & 15, l4 = lane >> 4;
    const RowBlock rb = task->rb[w];
    const double* const tB = task->B;
    const int ldb = task->ldb, K = task->k1, gnb = task->gnb;
    const bool active = rb.A != nullptr;        // wave-uniform
    // staging: a wave moves its own 16 rows of either panel, 8 lanes (2 rows each) per column of the chunk
    const int scol = lane >> 3, srow = 2 * (lane & 7);
    // (a wave without a block, and the B rows beyond the tile's valid columns, load all the same -- from the B panel, whose 128
    // rows exist in memory: loads under a condition cost the loop its counted vmcnt waits)
    const double* gB = tB + 16 * w + srow + (size_t)scol * ldb;
    const double* gA = active ? rb.A + srow + (size_t)scol * rb.lda : gB;
    const int lda = active ? rb.lda : ldb;
    const int sOff = scol * LDP + 16 * w + srow;
    const int nch = K / KC2;
    const d2 zero2 = {0.0, 0.0};
    d2 ra0 = zero2, rb0 = zero2, ra1 = zero2, rb1 = zero2;
#define F8LOAD(RA, RB, CH)                                                                       \
    do {                                                                                         \
        const size_t ch_ = (size_t)(CH) * KC2;                                                   \
        RA = *AS_GLOBAL_D2(gA + ch_ * lda);                                                      \
        RB = *AS_GLOBAL_D2(gB + ch_ * ldb);                                                      \
    } while (0)
#define F8WRITE(RA, RB, BUF)                                                                     \
    do {                                                                                         \
        *reinterpret_cast<d2*>(&sA[BUF][sOff]) = RA;                                             \
        *reinterpret_cast<d2*>(&sB[BUF][sOff]) = RB;                                             \
    } while (0)
    if (nch > 0) {      // the first two chunks: their latency passes under the kernel function
        F8LOAD(ra0, rb0, 0);
        F8LOAD(ra1, rb1, min(1, nch - 1));
    }
    d4 acc[NCB][1];
    {   // acc = -k(row, col): coordinates through the (still unused) ring, every wave its own 16 rows and 16 columns
        const KParam p = kp[task->kid];
        double* sa = &sA[0][0];
        double* sb = &sB[0][0];
        const double* gxb = task->gxb;
        const int gldb = task->gldb;
        const int navalid = active ? rb.nvalid : 0;
        for (int d0 = l4; d0 < D; d0 += 16) {       // four dimensions per lane and pass, loads first
            double va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = d0 + 4 * u;
                va[u] = (d < D && l15 < navalid) ? rb.gx[l15 + (size_t)d * rb.glda] : 0.0;
                vb[u] = (d < D && 16 * w + l15 < gnb) ? gxb[16 * w + l15 + (size_t)d * gldb] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = d0 + 4 * u;
                if (d < D) {
                    sa[d * TB + 16 * w + l15] = va[u];
                    sb[d * TB + 16 * w + l15] = vb[u];
                }
            }
        }
        __syncthreads();
        if (active) {
            const int gna = 16 * w + rb.nvalid;     // rowsplit_gram_init counts rows from the top of the 128-row image
            if (rb.nvalid == 16 && gnb >= 16 * NCB) {       // no padding rows or columns in this wave's blocks (wave-uniform)
                if (p.kind == 0) rowsplit_gram_init<0, 1, NCB, false>(gna, gnb, p, D, acc, sa, sb);
                else if (p.kind == 1) rowsplit_gram_init<1, 1, NCB, false>(gna, gnb, p, D, acc, sa, sb);
                else rowsplit_gram_init<2, 1, NCB, false>(gna, gnb, p, D, acc, sa, sb);
            } else {
                if (p.kind == 0) rowsplit_gram_init<0, 1, NCB>(gna, gnb, p, D, acc, sa, sb);
                else if (p.kind == 1) rowsplit_gram_init<1, 1, NCB>(gna, gnb, p, D, acc, sa, sb);
                else rowsplit_gram_init<2, 1, NCB>(gna, gnb, p, D, acc, sa, sb);
            }
        }
        __syncthreads();    // the coordinates are no longer read: the ring takes the operand chunks
    }
    // product: four waves per SIMD interleave their memory and matrix instructions by themselves, so the loop is written plainly
    // (no hand-placed instruction order as in gemm_mainloop_v2); what it must not have is a load under a condition (above)
    const int rowoff = 16 * w + l15;
#define F8COMPUTE(BUF)                                                                           \
    do {                                                                                         \
        if (active) {                                                                            \
            _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                   \
                const double* pa_ = &sB[BUF][(g_ * 4 + l4) * LDP + l15];                         \
                const double fb_ = sA[BUF][(g_ * 4 + l4) * LDP + rowoff];                        \
                double fa_[NCB];                                                                 \
                _Pragma("unroll") for (int i_ = 0; i_ < NCB; ++i_) fa_[i_] = pa_[16 * i_];       \
                _Pragma("unroll") for (int i_ = 0; i_ < NCB; ++i_)                               \
                    acc[i_][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa_[i_], fb_, acc[i_][0], 0, 0, 0); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    // Two chunks per barrier: the ring as two pairs of buffers.  Pair p is computed while pair p + 1 goes from the registers to
    // its buffers (requested one pair time earlier) and pair p + 2 is requested into the registers that frees.  K is a multiple
    // of 128: eight pairs or more.  (One chunk per barrier, three chunks of LDS lookahead: depth 4 fused tile launches 34.4-34.5
    // against 33.9-34.3 ms.)
    if (nch > 0) {
        const int np = nch >> 1;
        F8WRITE(ra0, rb0, 0);
        F8WRITE(ra1, rb1, 1);
        F8LOAD(ra0, rb0, 2);
        F8LOAD(ra1, rb1, 3);
        __syncthreads();
        int pr = 0;
        for (; pr < np - 2; ++pr) {
            const int nb_ = ((pr + 1) & 1) * 2;
            F8WRITE(ra0, rb0, nb_);
            F8WRITE(ra1, rb1, nb_ + 1);
            F8LOAD(ra0, rb0, 2 * pr + 4);
            F8LOAD(ra1, rb1, 2 * pr + 5);
            F8COMPUTE((pr & 1) * 2);
            F8COMPUTE((pr & 1) * 2 + 1);
            __syncthreads();
        }
        {   // pair np - 2: the last pair goes to its buffers, nothing more is requested
            const int nb_ = ((pr + 1) & 1) * 2;
            F8WRITE(ra0, rb0, nb_);
            F8WRITE(ra1, rb1, nb_ + 1);
            F8COMPUTE((pr & 1) * 2);
            F8COMPUTE((pr & 1) * 2 + 1);
            __syncthreads();
            ++pr;
        }
        F8COMPUTE((pr & 1) * 2);
        F8COMPUTE((pr & 1) * 2 + 1);
    }
    __syncthreads();
#undef F8COMPUTE
#undef F8WRITE
#undef F8LOAD
    // the ring is free: lower blocks of L_kk (off-diagonal ones from the factorised diagonal tile behind the B panel, the
    // diagonal inverses from Dinv_k), block b at 256 b (lower_block_base); thread t moves the doubles 2 (t & 127), + 1 of
    // block 4 e + (t >> 7) -- NCB = 8: blocks 0..19 now (all that block columns 0..4 need), 20..35 under their solve
    constexpr int NBLK = lower_block_base(NCB);
    constexpr int NE = (NBLK + 3) / 4, NE1 = (NCB == 8) ? 5 : NE;
    const int dj = (t & 127) >> 3, di = 2 * (t & 7), dq = __builtin_amdgcn_readfirstlane(t >> 7);
    const double* Lkk = tB + (size_t)K * (size_t)ldb;
    const double* Dv = task->Dinv;
    auto dinv_load = [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        constexpr unsigned pk = lower_block_pack<4>(4 * e, NBLK);
        const int sel = (int)(pk >> (8 * dq)) & 0xff;            // scalar: dq is wave-uniform
        const int cb = sel & 15, jb = sel >> 4;
        const double* src = (cb == jb) ? Dv + (size_t)(16 * cb + di) + (size_t)(16 * jb + dj) * TB
                                       : Lkk + (size_t)(16 * cb + di) + (size_t)(16 * jb + dj) * (size_t)ldb;
        return *AS_GLOBAL_D2(src);
    };
    d2 dv[NE1];
    static_for<NE1>([&](auto ec) { dv[decltype(ec)::value] = dinv_load(ec); });
    d2 dw[4];
    if constexpr (NCB == 8) {
        static_for<4>([&](auto ec) { dw[decltype(ec)::value] = dinv_load(std::integral_constant<int, 5 + decltype(ec)::value>{}); });
    }
    const bool riders = active && rb.wi != nullptr;
    double wpre = 0.0, spre = 0.0;
    if (task->zk != nullptr && t < TB) sZ[t] = task->zk[t];
    if (riders) {
        wpre = rb.wi[l15];
        if (rb.sq != nullptr) spre = rb.sq[l15];
    }
#pragma unroll
    for (int e = 0; e < NE1; ++e)
        if (4 * e + dq < NBLK) *reinterpret_cast<d2*>(sD + (size_t)(4 * e + dq) * 256 + 2 * (t & 127)) = dv[e];
    __syncthreads();
    auto solve_block_column = [&](auto cbc) {
        constexpr int cb = decltype(cbc)::value;
        if constexpr (cb < NCB) {
            if (active) {
                d4 x0 = acc[cb][0], x1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int jb = 0; jb < cb; ++jb) {
                    const double* blk = sD + (lower_block_base(cb) + jb) * 256 + l15;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double a = blk[(4 * q + l4) * 16];
                        if (jb & 1) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[jb][0][q], x1, 0, 0, 0);
                        else x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[jb][0][q], x0, 0, 0, 0);
                    }
                }
                const d4 tt = (cb > 1) ? x0 + x1 : x0;
                d4 y = (d4){0.0, 0.0, 0.0, 0.0};
                const double* dblk = sD + (lower_block_base(cb) + cb) * 256 + l15;
#pragma unroll
                for (int q = 0; q < 4; ++q) y = __builtin_amdgcn_mfma_f64_16x16x4f64(dblk[(4 * q + l4) * 16], tt[q], y, 0, 0, 0);
                acc[cb][0] = -y;
            }
        }
    };
    solve_block_column(std::integral_constant<int, 0>{});
    solve_block_column(std::integral_constant<int, 1>{});
    solve_block_column(std::integral_constant<int, 2>{});
    solve_block_column(std::integral_constant<int, 3>{});
    solve_block_column(std::integral_constant<int, 4>{});
    if (NCB == 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<d2*>(sD + (size_t)(4 * (5 + e) + dq) * 256 + 2 * (t & 127)) = dw[e];
        __syncthreads();
    }
    solve_block_column(std::integral_constant<int, 5>{});
    solve_block_column(std::integral_constant<int, 6>{});
    solve_block_column(std::integral_constant<int, 7>{});
    if (!active) return;
    // store: register q of acc[cb][0] is X(row = l15, col = 16 cb + l4 + 4 q) of the wave's block; zeros beyond NCB
    {
        const unsigned lofs = (unsigned)l15 + (unsigned)l4 * (unsigned)rb.ldc;
        const size_t ldc = (size_t)rb.ldc;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const gf64_ptr col = AS_GLOBAL_F64(rb.C + (size_t)(16 * cb + 4 * q) * ldc);
                col[lofs] = (cb < NCB) ? acc[cb < NCB ? cb : 0][0][q] : 0.0;
            }
    }
    if (riders) {       // reduced inside the wave, columns ascending, then over the four lane groups
        double p = 0.0, q2 = 0.0;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double z = sZ[16 * cb + l4 + 4 * q];
                p = fma(acc[cb][0][q], z, p);
                q2 = fma(acc[cb][0][q], acc[cb][0][q], q2);
            }
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        q2 += __shfl_xor(q2, 16);
        q2 += __shfl_xor(q2, 32);
        if (l4 == 0) {
            if (rb.sq == nullptr) rb.wi[l15] = wpre - p;
            else {
                rb.wi[l15] = wpre + p;
                rb.sq[l15] = spre + q2;
            }
        }
    }
}

// ROLE only names the instantiation (same code), as for tile_gemm_kernel_v2: 0 = the fused tile launches of a fit while per-launch
// timing is on (what bench.py's roofline times where these launches dominate: a profiler's average of <0> is that quantity),
// 1 = the launches of the standalone prediction sweep, 2 = fits without per-launch timing and bench.py's single-lane extra
template <int ROLE>
__global__ __launch_bounds__(512, 4) void tile_fused8_kernel(const FusedTask8* __restrict__ tasks, const KParam* __restrict__ kp, int D) {
    __shared__ __attribute__((aligned(16))) double smem[2 * NRING * KC2 * LDP + TB];
    const FusedTask8* task = tasks + blockIdx.x;
    const int ncb = (task->gnb + 15) >> 4;      // valid columns: fewer than 128 only in the last block column of a leaf
    if (ncb <= 1) tile_fused8_body<1>(task, smem, kp, D);
    else if (ncb <= 2) tile_fused8_body<2>(task, smem, kp, D);
    else if (ncb <= 4) tile_fused8_body<4>(task, smem, kp, D);
    else tile_fused8_body<8>(task, smem, kp, D);
}

// The fused tile tasks of the standalone prediction sweep, made on the device.  predict(model, x) on rows the model has not seen
// registers a new test set per call, and at depth 4 the host spent 20 ms writing 64k of these 576-byte tasks (37 MB, then a
// copy for the XCD order, then the upload) in front of a 16 ms sweep.  Everything in a task follows from the leaf table in HBM
// (LeafDev) and the leaf's row count: the host only decides which (leaf, block step) pairs run fused and where their tasks
// start in the step's list (SweepSeg: 32 bytes per pair); one thread per pair writes the pair's tasks, straight to the
// position xcd_permute would have moved them to (tasks of a leaf share its B panel and L_kk: they sit 8 apart, on one XCD).
struct SweepSeg {
    int leaf, k;          // leaf (index in the context's table) and block step
    int src0;             // natural index, within the step's task range, of the pair's first task
    int begin, n;         // the step's range in the task list
    int pad[3];
};
static_assert(sizeof(SweepSeg) == 32, "SweepSeg layout");
__host__ __device__ inline int xcd_slot(int src, int n, bool enable) {      // where xcd_permute (dsmgp_hip.cpp) puts natural index src
    if (!enable || n < 16) return src;
    const int q = n / 8, r = n % 8;
    const int big = r * (q + 1);
    const int x = src < big ? src / (q + 1) : r + (src - big) / q;
    const int j = src < big ? src - x * (q + 1) : (src - big) - (x - r) * q;
    return x + 8 * j;
}
__global__ __launch_bounds__(128) void build_sweep8_kernel(const SweepSeg* __restrict__ segs, int nseg, const LeafDev* __restrict__ leaves,
                                                           FusedTask8* __restrict__ out, int xcd) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const SweepSeg sg = segs[s];
    const LeafDev lf = leaves[sg.leaf];
    const int k = sg.k;
    const int nblk = (lf.nt + 15) >> 4, ntask = (nblk + 7) >> 3;
    for (int t = 0; t < ntask; ++t) {
        FusedTask8 f{};
        f.B = lf.F + (size_t)k * TB;
        f.Dinv = lf.Dinv + (size_t)k * TB * TB;
        f.zk = lf.z + (size_t)k * TB;                   // every block of the sweep carries the riders (mean and variance sums)
        f.gxb = lf.Xg + (size_t)k * TB;
        f.ldb = f.gldb = lf.npad;
        f.gnb = max(0, min(TB, lf.n - k * TB));
        f.k1 = k * TB;
        f.kid = lf.kid;
        f.nblk = min(8, nblk - 8 * t);
        for (int q = 0; q < f.nblk; ++q) {
            const int r = 16 * (8 * t + q);
            RowBlock& b = f.rb[q];
            b.A = lf.Vt + r;
            b.C = lf.Vt + r + (size_t)k * TB * lf.ntpad;
            b.gx = lf.Xtg + r;
            b.lda = b.ldc = b.glda = lf.ntpad;
            b.nvalid = min(16, lf.nt - r);
            b.wi = lf.macc + r;
            b.sq = lf.sacc + r;
        }
        out[sg.begin + xcd_slot(sg.src0 + t, sg.n, xcd != 0)] = f;
    }
}

// Rows [r0, 128) of a block row of a factor over `ncols` columns <- 0 (build_plan: the padding rows below a leaf's last data rows)
struct ZeroRowsTask {
    double* p;
    int ld, r0, ncols;
};
__global__ __launch_bounds__(256) void zero_pad_rows_kernel(const ZeroRowsTask* __restrict__ tasks) {
    const ZeroRowsTask tk = tasks[blockIdx.x];
    const int r = threadIdx.x & (TB - 1);
    if (r < tk.r0) return;
    for (int col = threadIdx.x >> 7; col < tk.ncols; col += 2) tk.p[r + (size_t)col * tk.ld] = 0.0;
}

// The 28 upper 16x16 blocks of a factor's diagonal tiles <- 0, once per plan (build_plan).  No kernel reads them -- every consumer of
// a diagonal tile takes its 36 lower blocks (tile_fused8_body, dinv_complete, the packed image) or its diagonal entries (mll), the
// inverse comes from Dinv -- and until round 6 every diagonal-block task of a fused step wrote them again in every fit: 56 of the
// 128 KB a task stores (2.4 GB in the first launch of a depth-4 fit).  Now they are written here and by nobody afterwards.
struct ZeroUpperTask {
    double* p;            // origin of the leaf's first diagonal tile
    int ld, nb;           // leading dimension, number of diagonal tiles (tile k at p + k 128 (ld + 1))
};
__global__ __launch_bounds__(256) void zero_upper_blocks_kernel(const ZeroUpperTask* __restrict__ tasks) {
    const ZeroUpperTask tk = tasks[blockIdx.x];
    const int r = threadIdx.x & (TB - 1);
    for (int k = blockIdx.y; k < tk.nb; k += gridDim.y) {
        double* t = tk.p + (size_t)k * TB * ((size_t)tk.ld + 1);
        for (int col = 16 + (threadIdx.x >> 7); col < TB; col += 2)
            if (r < (col & ~15)) t[r + (size_t)col * tk.ld] = 0.0;
    }
}

// Gram values of a wave's 9 lower blocks of the diagonal tile, S = k - product in place (syrk_gram_epilogue without the
// store), then the blocks go into the packed image of chol_diag_packed_body
#ifndef DSMGP_SYRK_GRAM_GROUP
#define DSMGP_SYRK_GRAM_GROUP 2         // blocks whose kernel-function sums are in flight at once (round 4: 3 -- with 68 B of scratch)
#endif
template <int SHAPE, int KIND>
__device__ __forceinline__ void syrk_gram_inplace(const TileTask& tk, const KParam& p, int D, d4 (&acc)[9], const int (&blk)[6],
                                                  const double* sa) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    constexpr int G = (KIND == 0) ? DSMGP_SYRK_GRAM_GROUP : 1;
    int rbk[9], cbk[9];         // block row / column of accumulator i (the layout syrk_mainloop leaves)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int g3 = i / 3, j = i % 3;
        rbk[i] = (SHAPE == 0) ? blk[g3] : blk[2 * g3 + (j > 0 ? 1 : 0)];
        cbk[i] = (SHAPE == 0) ? blk[3 + j] : blk[2 * g3 + (j > 1 ? 1 : 0)];
    }
    if constexpr (KIND != 0) {
        // ArdSE / IsoLinear: one block, two of its entries at a time (the exp per dimension of the additive kernel and the
        // unrolled dot product each wanted one register more than the task has at four)
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += 2) {
                double z[1][2] = {{0.0, 0.0}};
#pragma unroll 1
                for (int d = 0; d < D; ++d) {
                    const double nhd = (KIND == 1) ? AS_CONST_F64(p.nh)[d] : 0.0;
                    double a[1], b[2];
                    a[0] = sa[d * TB + 16 * rbk[i] + l15];
#pragma unroll
                    for (int r = 0; r < 2; ++r) b[r] = sa[d * TB + 16 * cbk[i] + l4 + 4 * (r0 + r)];
                    gram_accumulate<KIND, 1, 2>(z, a, b, nhd);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = 16 * rbk[i] + l15, col = 16 * cbk[i] + l4 + 4 * (r0 + r);
                    const double kv = gram_finish<KIND>(z[0][r], p, row, col, tk.gna, tk.gnb, true);
                    acc[i][r0 + r] = kv - acc[i][r0 + r];
                }
            }
    } else {
        // IsoSE: G blocks at a time -- their 4 G exponentials overlap (one block at a time: diagonal blocks of a depth-4 fit
        // 10.12 -> 10.31 ms; three: 68 bytes of scratch per lane at three workgroups per CU)
#pragma unroll
        for (int i0 = 0; i0 < 9; i0 += G) {
            constexpr int GG = G;
            double z[GG][1][4];
#pragma unroll
            for (int j = 0; j < GG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[j][0][r] = 0.0;
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int j = 0; j < GG; ++j) {
                    if (i0 + j < 9) {
                        double a[1], b[4];
                        a[0] = sa[d * TB + 16 * rbk[i0 + j < 9 ? i0 + j : 8] + l15];
#pragma unroll
                        for (int r = 0; r < 4; ++r) b[r] = sa[d * TB + 16 * cbk[i0 + j < 9 ? i0 + j : 8] + l4 + 4 * r];
                        gram_accumulate<KIND, 1, 4>(z[j], a, b, 0.0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < GG; ++j)
                if (i0 + j < 9) {
                    const int i = i0 + j < 9 ? i0 + j : 8;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * rbk[i] + l15, col = 16 * cbk[i] + l4 + 4 * r;
                        const double kv = gram_finish<KIND>(z[j][0][r], p, row, col, tk.gna, tk.gnb, true);
                        acc[i][r] = kv - acc[i][r];
                    }
                }
        }
    }
}

// Front of a DiagFinishTask (the diagonal-block tasks that ride in the update launches): the update of the diagonal tile over
// the task's K range on the tile in memory -- which holds K(k,k) minus the product over the columns BEFORE that range: an earlier
// update launch wrote it -- straight into the packed LDS image of chol_diag_packed_body.
template <int SHAPE>
__device__ __forceinline__ void diag_finish_front(const TileTask& tt, double* S, const int (&blk)[6]) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    double (*sA)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(S);
    d4 acc[9];
    syrk_mainloop<SHAPE>(tt, acc, sA, blk);                 // ends on a barrier: the ring is free
    {   // S = tile - product over this task's columns, block by block in the accumulator layout (36 loads in flight)
        const size_t ldc = (size_t)tt.ldc;
        double cv[9][4];
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int rb = (SHAPE == 0) ? blk[g3] : blk[2 * g3 + (j > 0 ? 1 : 0)];
                const int cb = (SHAPE == 0) ? blk[3 + j] : blk[2 * g3 + (j > 1 ? 1 : 0)];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cv[3 * g3 + j][r] = AS_GLOBAL_F64(tt.C)[(size_t)(16 * rb + l15) + (size_t)(16 * cb + l4 + 4 * r) * ldc];
            }
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = cv[i][r] - acc[i][r];
    }
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int rb, cb;
            if (SHAPE == 0) {
                rb = blk[g3];
                cb = blk[3 + j];
            } else {
                rb = blk[2 * g3 + (j > 0 ? 1 : 0)];
                cb = blk[2 * g3 + (j > 1 ? 1 : 0)];
            }
            double* dst = S + PACKED.off[rb * 8 + cb] + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(l4 + 4 * r) * PLD] = acc[3 * g3 + j][r];
        }
}

// ---------------------------------------------------------------------------------------------
// Diagonal-block task of a fused step with the trailing matrix in REGISTERS (round 4).  chol_diag_packed_body keeps the 36
// lower 16x16 blocks in a 75 KB LDS image: two workgroups per CU, i.e. two latency-bound pivot chains in flight, and every
// trailing block read, updated and written back through LDS at every block step.  But the update of the tile leaves each wave
// with nine of the 36 blocks in its accumulators (syrk_mainloop), and a block in accumulator layout is BOTH operands of the
// f64 MFMA as it stands (register q as A operand: M(i = l15, k = 4q + l4); as B operand: M^T): the blocks can stay where they
// are.  Per block step J only the column's own blocks go through LDS -- L_JJ^-1 (2 KB) and the solved panel (<= 7 x 2 KB):
//   P0  the wave that owns (J,J): potrf + inverse of the 16x16 block in registers (potrf_inv16_mfma), L_JJ -> tile,
//       L_JJ^-1 -> Dinv and -> LDS                                                           | barrier
//   P1  every wave, its blocks (I,J), I > J: S(I,J) <- S(I,J) L_JJ^-T from the registers; -> registers, panel (LDS), tile;
//       z_J = L_JJ^-1 w_J                                                                      | barrier
//   P2  every wave, its blocks (I,K), I >= K > J: S(I,K) -= S(I,J) S(K,J)^T, both operands from the panel, accumulated on
//       the registers; w_I -= L(I,J) z_J
// 20 KB of LDS beside the update's 36 KB ring (they alias), registers for nine blocks: three workgroups per CU.
// No inverse phase (the fused steps' tile tasks substitute against L_kk: tile_fused8_body).
// LDS strides chosen against bank conflicts (round 6; the counters had read SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.094 for this
// kernel since round 4): a block row of the panel is 256 doubles but sits DIAGR_PB = 272 apart -- the rows-below update of the
// right-hand side reads one row per thread, 16 consecutive rows per block, so a half wave spans two block rows, which 256 apart
// fall on the same banks (272 = 16 mod 32 doubles: the second block row takes the other half of the banks); L_JJ^-1 is written
// transposed out of the accumulator layout, 16 lanes a row apart: rows DIAGR_SLD = 17 doubles apart instead of 16 (8-way -> none).
constexpr int DIAGR_PB = 272;                                // doubles between block rows of the solved panel (column-major, ld 16)
constexpr int DIAGR_SLD = 17;                                // leading dimension of L_JJ^-1 in LDS
constexpr int DIAGR_PANEL = 8 * DIAGR_PB;
constexpr int DIAGR_LDS_DOUBLES = DIAGR_PANEL + 16 * DIAGR_SLD + 2 * TB + 2;   // + L_JJ^-1 + rhs block + z + first bad pivot
constexpr int DIAGR_LDS_BYTES = ((NRING * KC2 * LDP > DIAGR_LDS_DOUBLES ? NRING * KC2 * LDP : DIAGR_LDS_DOUBLES) + 16) * (int)sizeof(double);

// Which block an accumulator of wave W holds (the layout syrk_mainloop leaves): waves 0..2 a 3x3 square of blocks, wave 3 the
// three 2x2 lower triangles on the diagonal.  Compile-time, so that with the block steps unrolled every "does this wave own
// a block of column J" folds away and a step is straight-line code: the 16x16 factorisation of the NEXT diagonal block (a
// latency-bound chain of vector instructions) and the wave's remaining trailing products (independent MFMAs) can interleave.
__host__ __device__ constexpr int diagr_rb(int w, int i) {
    return w == 3 ? (i / 3) * 3 + (i % 3 > 0 ? 1 : 0) : (w == 2 ? 2 : 5) + i / 3;
}
__host__ __device__ constexpr int diagr_cb(int w, int i) {
    return w == 3 ? (i / 3) * 3 + (i % 3 > 1 ? 1 : 0) : (w == 1 ? 3 : 0) + i % 3;
}
static_assert(diagr_rb(3, 4) == 4 && diagr_cb(3, 4) == 3 && diagr_rb(3, 8) == 7 && diagr_cb(3, 8) == 7 && diagr_rb(0, 5) == 6 &&
                  diagr_cb(0, 5) == 2 && diagr_rb(1, 4) == 6 && diagr_cb(1, 4) == 4 && diagr_rb(2, 8) == 4 && diagr_cb(2, 8) == 2,
              "accumulator -> block map of syrk_mainloop");

template <int W>
__device__ __forceinline__ void diag_reg_body(const DiagTask& tk, d4 (&acc)[9], double* S) {
    const int t = threadIdx.x, lane = t & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    double* panel = S;
    double* sLinv = S + DIAGR_PANEL;
    double* wl = sLinv + 16 * DIAGR_SLD;
    double* zl = wl + TB;
    int* sbad = reinterpret_cast<int*>(zl + TB);
    const bool fuse = tk.wk != nullptr;
    if (fuse && t < TB) {
        wl[t] = tk.wk[t];
        zl[t] = 0.0;
    }
    if (t == 0) sbad[0] = 0;
    const int JN = (tk.nvalid + 15) >> 4;
    if (JN < 8) {   // the lower blocks of the block columns without data go out now, as zeros (the upper blocks are zeroed once per
                    // plan: zero_upper_blocks_kernel)
        const int mc = lane >> 3, mr = 2 * (lane & 7);
        const d2 zero = {0.0, 0.0};
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int b = W + 4 * q, I = b >> 3, K = b & 7;
            if (I > K && K >= JN) {
                double* gT = tk.T + (size_t)(16 * I + mr) + (size_t)(16 * K + mc) * tk.ld;
                *reinterpret_cast<d2*>(gT) = zero;
                *reinterpret_cast<d2*>(gT + (size_t)8 * tk.ld) = zero;
            }
        }
    }
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    // P0 of block step J, by the wave that owns (J,J): L_JJ -> tile, L_JJ^-1 -> Dinv and LDS
    auto factor_diag = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        static_for<9>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (diagr_rb(W, i) == J && diagr_cb(W, i) == J) {
                d4 lt, xi;
                if (DSMGP_DIAG_PRIO) __builtin_amdgcn_s_setprio(DSMGP_DIAG_PRIO);
                int bj = 0;
                if (DSMGP_DIAGR_SKIP & 2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        lt[g] = (l15 == 4 * g + l4) ? 2.0 : 0.0;
                        xi[g] = (l15 == 4 * g + l4) ? 0.5 : 0.0;
                    }
                } else {
                    bj = potrf_inv16_mfma(acc[i], lt, xi, lane);
                }
                if (DSMGP_DIAG_PRIO) __builtin_amdgcn_s_setprio(0);
                if (bj != 0 && lane == 0 && sbad[0] == 0) sbad[0] = J * 16 + bj;
                // lt register g = L(l15, 4g + l4), xi register g = L^-1(4g + l4, l15)
                const gf64_ptr gT = AS_GLOBAL_F64(tk.T + (size_t)(16 * J + l15) + (size_t)(16 * J + l4) * tk.ld);
                const gf64_ptr gD = AS_GLOBAL_F64(tk.Dinv + (size_t)(16 * J + l4) + (size_t)(16 * J + l15) * TB);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q = 4 * g + l4;
                    gT[(size_t)(4 * g) * tk.ld] = (l15 >= q) ? lt[g] : 0.0;
                    const double x = (q >= l15) ? xi[g] : 0.0;
                    gD[4 * g] = x;
                    sLinv[l15 * DIAGR_SLD + q] = x;
                }
            }
        });
    };
    lds_barrier();
    factor_diag(std::integral_constant<int, 0>{});
    static_for<8>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (J < JN) {       // workgroup-uniform
            lds_barrier();  // L_JJ^-1 is in LDS; the panel of step J - 1 is no longer read
            // P1: S(I,J) <- S(I,J) L_JJ^-T for the wave's own blocks of column J
            {
                double la[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) la[q] = sLinv[(4 * q + l4) * DIAGR_SLD + l15];
                static_for<9>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int rb = diagr_rb(W, i), cb = diagr_cb(W, i);
                    if constexpr (cb == J && rb > J && !(DSMGP_DIAGR_SKIP & 8)) {
                        d4 u = __builtin_amdgcn_mfma_f64_16x16x4f64(la[0], acc[i][0], zero4, 0, 0, 0);
                        d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(la[1], acc[i][1], zero4, 0, 0, 0);
                        u = __builtin_amdgcn_mfma_f64_16x16x4f64(la[2], acc[i][2], u, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f64_16x16x4f64(la[3], acc[i][3], v, 0, 0, 0);
                        const d4 x = u + v;
                        acc[i] = x;
                        double* ps = panel + rb * DIAGR_PB + l15;
                        const gf64_ptr g0 = AS_GLOBAL_F64(tk.T + (size_t)(16 * rb + l15) + (size_t)(16 * J + l4) * tk.ld);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ps[(l4 + 4 * r) * 16] = x[r];
                            g0[(size_t)(4 * r) * tk.ld] = x[r];
                        }
                    }
                });
                if (W == 3 && fuse && lane < 16) {      // z_J = L_JJ^-1 w_J
                    const double* linv = sLinv + lane;
                    double sum = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) sum = fma(linv[c * DIAGR_SLD], wl[16 * J + c], sum);
                    zl[16 * J + lane] = sum;
                }
            }
            lds_barrier();
            // P2: S(I,K) -= S(I,J) S(K,J)^T on the registers, operands from the panel -- the next diagonal block first, and its
            // factorisation (P0 of step J + 1: registers and sLinv only, which nobody reads before the next barrier) right
            // behind it, beside the rest of this wave's products
            auto trailing = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int rb = diagr_rb(W, i), cb = diagr_cb(W, i);
                const double* pk = panel + cb * DIAGR_PB + l15;
                const double* pi = panel + rb * DIAGR_PB + l15;
                if (DSMGP_DIAGR_SKIP & 4) return;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pk[(4 * q + l4) * 16], pi[(4 * q + l4) * 16], acc[i], 0, 0, 0);
            };
            static_for<9>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (diagr_rb(W, i) == J + 1 && diagr_cb(W, i) == J + 1) trailing(ic);
            });
            if constexpr (J + 1 < 8) {
                if (J + 1 < JN) factor_diag(std::integral_constant<int, J + 1>{});
            }
            static_for<9>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (diagr_cb(W, i) > J && !(diagr_rb(W, i) == J + 1 && diagr_cb(W, i) == J + 1)) trailing(ic);
            });
            if (fuse && t < TB && t >= 16 * (J + 1)) {      // w_I -= L(I,J) z_J for the rows below block J
                const double* lrow = panel + (t >> 4) * DIAGR_PB + (t & 15);
                double sum = 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) sum = fma(lrow[c * 16], zl[16 * J + c], sum);
                wl[t] -= sum;
            }
        }
    });
    lds_barrier();
    if (fuse && t < TB) tk.zk[t] = zl[t];
    if (W == 0 && lane < 32) {             // identity diagonal blocks of the steps without data
        const int c = l15;
        const bool isb = (lane & 16) != 0;
        for (int J = JN; J < 8; ++J) {
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
    if (t == 0) {
        const int bad = sbad[0];
        if (bad != 0 && bad <= tk.nvalid && *tk.info == 0) *tk.info = tk.row0 + bad;
    }
}

template <int W>
__device__ __forceinline__ void diag_fused_reg(const TileTask& tt, const DiagTask& d, const KParam* __restrict__ kp, int D, double* S) {
    constexpr int SHAPE = W == 3 ? 1 : 0;
    constexpr int rbase = (W == 2) ? 2 : 5, cbase = (W == 1) ? 3 : 0;
    const int blk[6] = {W == 3 ? 0 : rbase, W == 3 ? 1 : rbase + 1, W == 3 ? 3 : rbase + 2,
                        W == 3 ? 4 : cbase, W == 3 ? 6 : cbase + 1, W == 3 ? 7 : cbase + 2};
    double (*sA)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(S);
    d4 acc[9];
    syrk_mainloop<SHAPE>(tt, acc, sA, blk);                 // ends on a barrier: the ring is free
    const KParam p = kp[tt.kid];
    gram_stage_coords(tt, D, S, nullptr, false);            // coordinates over the ring (barrier inside)
    if (DSMGP_DIAGR_SKIP & 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[i][r] = (diagr_rb(W, i) == diagr_cb(W, i) && (int)(threadIdx.x & 15) == (int)((threadIdx.x & 63) >> 4) + 4 * r) ? 4.0 : 0.0;
    } else if (p.kind == 0) syrk_gram_inplace<SHAPE, 0>(tt, p, D, acc, blk, S);
    else if (p.kind == 1) syrk_gram_inplace<SHAPE, 1>(tt, p, D, acc, blk, S);
    else syrk_gram_inplace<SHAPE, 2>(tt, p, D, acc, blk, S);
    __syncthreads();                                        // the coordinates are no longer read: panel and rhs take their place
    diag_reg_body<W>(d, acc, S);
}

__global__ __launch_bounds__(256, DSMGP_DIAGR_WGS) void diag_fused_reg_kernel(const DiagFusedTask* __restrict__ tasks,
                                                                              const KParam* __restrict__ kp, int D) {
    extern __shared__ __attribute__((aligned(16))) double S[];   // DIAGR_LDS_BYTES: ring / coordinates / panel share the space
    static_assert(GRAM_FUSE_MAX_D * TB * (int)sizeof(double) <= DIAGR_LDS_BYTES, "coordinates must fit");
    const DiagFusedTask ft = tasks[blockIdx.x];
    TileTask tt{};
    tt.A = ft.A;
    tt.lda = ft.d.ld;
    tt.k0 = 0;
    tt.k1 = ft.k1;
    tt.kid = ft.kid;
    tt.gxa = ft.gx;
    tt.glda = ft.glda;
    tt.gna = tt.gnb = ft.d.nvalid;
    tt.C = ft.d.T;
    tt.ldc = ft.d.ld;
    // (Giving the role that owns six of the eight diagonal 16x16 blocks -- 3: their pivot chains -- to a different wave in every
    // workgroup, so that the chains of co-resident tasks sit on different SIMDs, measured nothing: depth 4 0.0516-0.0522 s with
    // the role shifted by blockIdx, blockIdx / 8 or blockIdx / 256 as without.)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w == 0) diag_fused_reg<0>(tt, ft.d, kp, D, S);
    else if (w == 1) diag_fused_reg<1>(tt, ft.d, kp, D, S);
    else if (w == 2) diag_fused_reg<2>(tt, ft.d, kp, D, S);
    else diag_fused_reg<3>(tt, ft.d, kp, D, S);
}

// ---------------------------------------------------------------------------------------------
// The diagonal block INSIDE the update launch (round 4).  In a classic block step the chain update -> (reduce) -> diagonal
// block -> panel solve is four dependent launches, and the diagonal-block launch -- one workgroup per leaf, 35 us of a
// latency-bound pivot chain -- keeps the whole chip waiting: 104 steps x 38 us on the headline model and on every shard of a
// multi-GPU job, a third of the step time of a single GP (config 2).  Two streams do not hide it (rounds 2 and 3: concurrent
// kernels slow the update launch by more than the chain they hide), and neither does a task that WAITS inside the update
// launch for the pieces of its tile (built first in round 4: arrival counter, agent-scope release / acquire -- correct, -6 %
// on config 2, but the pieces at the front of a many-round launch break the lockstep in which its tiles share their B panels
// through L2: headline +8 ms of update time for 4 ms of chain).  What does is moving the diagonal tile's update ONE STEP
// AHEAD, where nothing waits for it:
//   step k-1  the update launch also carries, per leaf, the tile (k,k) over the columns [0, 128 (k-1)) -- an ordinary tile of
//             that launch's depth sharing its A panel with the tile (k, k-1) next to it;
//   step k    the update launch carries one DiagFinishTask per leaf: S = tile - F[k,k-1] F[k,k-1]^T (the one block column the
//             panel solves of step k-1 have just finished), factorisation, inverse, z_k -- every input final before the launch
//             starts, so the task is an ordinary member of the grid, running beside the updates of the tiles below it, which
//             do not need it; the panel-solve launch that follows finds L_kk, Dinv_k and z_k ready.
// (src/AdvancedCholeskey.jl:161-171 per step, with the potrf of the NEXT diagonal block overlapped with the trailing update:
// the classic one-step lookahead of a right-looking factorisation, expressed as tasks of one launch.)
struct DiagFinishTask {
    DiagTask d;
    const double* A;         // F[k, K - 128 .. K): the last finished block of the block row, ld = d.ld
};

__device__ __forceinline__ void diag_finish_body(const DiagFinishTask& ft, double* S) {
    TileTask tt{};
    tt.A = ft.A;
    tt.lda = ft.d.ld;
    tt.k0 = 0;
    tt.k1 = TB;
    tt.C = ft.d.T;
    tt.ldc = ft.d.ld;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w == 3) {
        const int blk[6] = {0, 1, 3, 4, 6, 7};
        diag_finish_front<1>(tt, S, blk);
    } else {
        const int rbase = (w == 2) ? 2 : 5, cbase = (w == 1) ? 3 : 0;
        const int blk[6] = {rbase, rbase + 1, rbase + 2, cbase, cbase + 1, cbase + 2};
        diag_finish_front<0>(tt, S, blk);
    }
    chol_diag_packed_body(ft.d, S, true);                    // its first barrier publishes the image
}

// ROLE only names the instantiation (same code): 0 = update launches (whole tiles and split-K pieces), 1 = panel
// solves (K = 128), so that profilers report the two populations as two kernels
// (tile_gemm_kernel_v2<false, 0> is the dominant kernel of bench.py's roofline).
// PAD: the launch carries enough short tiles (TileTask.mrows <= 96: padding rows below) for the column-split form
// (tile_rows_body) to pay; without it the check is compiled out (it costs the other launches ~0.3 % through the register
// allocation of the main path).
// Grid of an update launch that carries diagonal-block tasks: [0, dpos) tile tasks (tasks[0..dpos)), [dpos, dpos + ndfin) the
// DiagFinishTasks, then the remaining tile tasks (tasks[dpos..]).  ndfin = 0: tile tasks only.
constexpr int UPD_LDS_DOUBLES = (2 * NRING * KC2 * LDP > PIMG + 2 * TB + 64) ? 2 * NRING * KC2 * LDP : PIMG + 2 * TB + 64;
template <bool STAMP, int ROLE = 0, bool PAD = false>
__global__ __launch_bounds__(256, 2) void tile_gemm_kernel_v2(const TileTask* __restrict__ tasks,
                                                              unsigned long long* __restrict__ stamps,
                                                              const KParam* __restrict__ kp, int D, int dpos,
                                                              const DiagFinishTask* __restrict__ dfin, int ndfin) {
    __shared__ __attribute__((aligned(16))) double smem[UPD_LDS_DOUBLES];
    double (*sA)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(smem);
    double (*sB)[KC2 * LDP] = reinterpret_cast<double (*)[KC2 * LDP]>(smem + NRING * KC2 * LDP);
    static_assert(GRAM_FUSE_MAX_D * TB <= NRING * KC2 * LDP, "coordinate image of a fused-Gram tile must fit the ring");
    static_assert(UPD_LDS_DOUBLES * sizeof(double) >= (size_t)DIAGP_LDS_BYTES, "a diagonal-block task must fit the launch's LDS");
    unsigned long long r0 = 0;
    if (STAMP) r0 = __builtin_amdgcn_s_memrealtime();
    int b = blockIdx.x;
    if (!STAMP && ndfin > 0 && b >= dpos) {      // workgroup-uniform
        if (b < dpos + ndfin) {
            diag_finish_body(dfin[b - dpos], smem);
            return;
        }
        b -= ndfin;
    }
    const TileTask tk = tasks[b];
    if (tk.sym) {       // diagonal tile of the factorisation: lower blocks only (workgroup-uniform branch)
        tile_syrk_body(tk, sA, kp, D);
    } else if (PAD && !STAMP && tk.update == 1 && tk.mrows != 0 && tk.mrows <= 96 && tk.wi == nullptr && tk.rev == 0) {
        // short tile (rows mrows.. are padding): the waves split the columns, every SIMD does mrows/128 of a tile's work
        if (tk.mrows <= 32) tile_rows_body<2>(tk, sA, sB, kp, D);
        else if (tk.mrows <= 64) tile_rows_body<4>(tk, sA, sB, kp, D);
        else tile_rows_body<6>(tk, sA, sB, kp, D);
    } else {
        d4 acc[4][4];
        gemm_mainloop_v2<STAMP>(tk, acc, sA, sB, stamps);
        tile_epilogue(tk, acc, &sA[0][0], kp, D, &sB[0][0]);
        if (STAMP && (threadIdx.x & 63) == 0) {
            unsigned long long* s = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
            s[6] = r0;                                    // kernel entry / exit of this wave (100 MHz wall ticks)
            s[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

}  // namespace dsmgp
