// C ABI (include/dsmgp_hip.h) and host orchestration of the GP-expert hot path on one MI355X.
// The host builds, once per leaf table, flat task lists for every block step of the batched
// left-looking Cholesky; fit!/predict then replay them as a fixed launch sequence on one stream.
#include "../../include/dsmgp_hip.h"
#ifdef DSMGP_DIAG
#include "../../include/dsmgp_hip_diag.h"
#endif
#include "kernels.hpp"
#include "kernels_fused.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <array>
#include <chrono>
#include <dlfcn.h>
#include <limits>
#include <vector>

using namespace dsmgp;

namespace {

std::string g_create_error;

struct HyperHost {
    int kind = -1;
    std::vector<double> loghyp;   // [logl..., logs, logNoise]
};

struct LeafHost {
    int n = 0, npad = 0, nb = 0, kid = 0;
    double mean = 0.0;
    int64_t obs_off = 0;
    int op = DSMGP_SHARE_FULL, src = -1;
    int64_t prefix = 0;
    int owner = -1;          // leaf whose factor buffer this leaf uses (itself unless COPY)
    int kb = 0;              // PREFIX: number of leading 128-blocks copied from src
    size_t f_off = 0, dinv_off = 0, vec_off = 0, xg_off = 0;
    // prediction
    int nt = 0, ntpad = 0;
    size_t vt_off = 0, xt_off = 0, pv_off = 0;
    int64_t route_off = 0;
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t count = 0;       // entries in use (what a launch may cover)
    size_t cap = 0;         // entries allocated: a list that is replaced keeps its allocation while the new one fits (dev_upload)
};

// DSMGP_HOSTLOG=1: wall time of the host-side phases of plan building to stderr (diagnostic)
struct HostLog {
    const char* what;
    std::chrono::steady_clock::time_point t0;
    explicit HostLog(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {}
    void lap(const char* next) {      // close the running phase, open the next
        done();
        what = next;
        t0 = std::chrono::steady_clock::now();
    }
    void done() {
        static const bool on = std::getenv("DSMGP_HOSTLOG") != nullptr;
        if (on && what)
            std::fprintf(stderr, "hostlog %-28s %.4f s\n", what,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        what = nullptr;
    }
    ~HostLog() { done(); }
};

enum { STEP_CLASSIC = 0, STEP_FUSED = 1 };
constexpr int MAX_LANES = 4;
#ifndef DSMGP_FUSED_SHALLOW
#define DSMGP_FUSED_SHALLOW 4
#endif
constexpr int FUSED_SHALLOW_STEPS = DSMGP_FUSED_SHALLOW;   // block steps 0..4 (K <= 512) run fused where at least ...
constexpr int FUSED_SHALLOW_MIN_LEAVES = 32;               // ... this many leaves take part in the step
#ifndef DSMGP_STREAM_FLAGS
#define DSMGP_STREAM_FLAGS hipStreamDefault
#endif
#ifndef DSMGP_LANES_AUTO_MIN
#define DSMGP_LANES_AUTO_MIN 8
#endif
// two leaf lanes (dsmgp_ctx::nlanes) from this many sharing groups on.  Same-box A/B, one lane -> two (profiles/r05_lanes_ab.log): headline
// (144 leaves) 0.4084 / 0.4079 -> 0.3926 / 0.3912 s; depth 4 (18k leaves) 0.0512 / 0.0514 -> 0.0498 / 0.0499; the shards of 2-, 4- and
// 8-rank jobs (72 / 36 / 18 leaves) 0.2110 -> 0.1994, 0.1082 -> 0.1044, 0.0584 -> 0.0572; PoE of 128 experts 4.72 -> 4.67 ms
constexpr int LANES_AUTO_MIN_LEAVES = DSMGP_LANES_AUTO_MIN;
// The tail of an update launch (UpdateSplitter::add_step): K pieces per tile, and full rounds of tiles that join the tail in
// launches of under two rounds -- with ONE leaf lane, and with two or more (where the other lane's launch fills the slots a tail
// leaves idle and cutting it only costs: see add_step)
#ifndef DSMGP_TAIL_SPLIT_DEFAULT
#define DSMGP_TAIL_SPLIT_DEFAULT 4
#endif
#ifndef DSMGP_TAIL_ROUNDS_DEFAULT
#define DSMGP_TAIL_ROUNDS_DEFAULT 1
#endif
#ifndef DSMGP_TAIL_SPLIT_LANES
#define DSMGP_TAIL_SPLIT_LANES 1
#endif
#ifndef DSMGP_TAIL_ROUNDS_LANES
#define DSMGP_TAIL_ROUNDS_LANES 0
#endif
#ifndef DSMGP_SYM_SHARE
#define DSMGP_SYM_SHARE 0                // > 0: diagonal tiles run as FULL tiles where they are under 1 / this of a launch's tiles (A/B builds)
#endif
#ifndef DSMGP_PAD_SHARE
#define DSMGP_PAD_SHARE 0                // > 0: short tiles take the column-split form only from 1 / this of a launch's tiles on (A/B builds)
#endif
#ifndef DSMGP_SOLO_FACTOR
#define DSMGP_SOLO_FACTOR 1.56             // time of one workgroup alone on a CU relative to its share of a co-resident pair
#endif
#ifndef DSMGP_DFIN_BACK
#define DSMGP_DFIN_BACK (-1)             // < 0: the diagonal-block tasks of a full update launch go last; >= 0: that many rounds of
                                         // whole tiles before its tail pieces (diagnostic builds: A/B of the placement)
#endif

struct StepLists {
    // one phase (wave) of factorisation: per block step k the tasks for update / split-K reduce / diag / trsm
    std::vector<int> upd_off, red_off, diag_off, trsm_off;   // size nsteps+1
    std::vector<int> step_tiles;                             // whole update tiles per step (before split-K)
    std::vector<char> pad;                                   // per step: >= 10 % of the tiles have rows 64.. all padding
    DevBuf<TileTask> upd, trsm;
    DevBuf<ReduceTask> red;
    DevBuf<DiagTask> diag;
    // steps with more diagonal blocks than CUs run fused (kernels_fused.hpp): diag_fused_reg_kernel, then tile_fused8_kernel
    std::vector<int> fdiag_off, ftile8_off;                  // size nsteps+1; a fused step has no classic tasks and vice versa
    std::vector<char> mode;                                  // STEP_* per step
    DevBuf<DiagFusedTask> fdiag;
    DevBuf<FusedTask8> ftile8;                               // eight 16-row blocks below a diagonal block each
    // classic steps with the diagonal block INSIDE the update launch (DiagFinishTask, kernels_fused.hpp): the update launch of
    // step k carries dfin[dfin_off[k] .. dfin_off[k+1]) behind its first dpos[k] tile tasks; a leaf whose diagonal block rides
    // there has no `diag` task in that step
    std::vector<int> dpos, dfin_off;                         // size nsteps / nsteps+1
    DevBuf<DiagFinishTask> dfin;
    int nsteps = 0;
};

// Collects the update tiles of one block step and splits their K range over several workgroups when
// the step has too few tiles to fill the chip (tail of the factorisation, prediction sweeps).
// Cost model in units of one K column on one CU: a workgroup costs (K/S + C0), rounds = ceil(T*S / CUs).
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its L2), so a run
// of tasks that read the same B panel should sit at positions x, x+8, x+16, ...  Reorder [begin,end) so
// that XCD slot x receives a contiguous chunk of the natural (leaf-major) order.  Speed only.
template <class T, class U>
void xcd_permute(std::vector<T>& a, std::vector<U>& b, size_t begin, size_t end, bool enable) {
    const size_t n = end - begin;
    if (!enable || n < 16) return;
    const size_t q = n / 8, r = n % 8;
    std::vector<T> ta(a.begin() + begin, a.begin() + end);
    std::vector<U> tb(b.begin() + begin, b.begin() + end);
    size_t src = 0;
    for (size_t x = 0; x < 8; ++x) {
        const size_t len = q + (x < r ? 1 : 0);
        for (size_t j = 0; j < len; ++j, ++src) {
            a[begin + x + 8 * j] = ta[src];
            b[begin + x + 8 * j] = tb[src];
        }
    }
}

// The same for a launch whose tasks differ in depth (the contraction tiles of the gradient pass: one launch, depths from
// 1 to nb blocks).  An XCD slot works through its share of the list on its own, so equal COUNTS per slot leave the slots
// with unequal work and the launch waits for the slowest eighth of the chip.  Blocks of tasks that share operands
// (`block_start`, natural order = deepest first) are dealt to the slot with the least work so far; the counts are then
// evened out (the positions of a slot are x, x+8, ...: every slot needs n/8 tasks) by moving tasks from the ends of the
// fuller slots, which are the shallowest.
template <class T, class U>
void xcd_deal_by_work(std::vector<T>& a, std::vector<U>& b, size_t begin, size_t end, const std::vector<size_t>& block_start,
                      const std::vector<double>& work, bool enable) {   // block_start / work: relative to `begin`
    const size_t n = end - begin;
    if (!enable || n < 16) return;
    std::vector<std::vector<size_t>> q(8);
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t bi = 0; bi + 1 < block_start.size(); ++bi) {
        int x = 0;
        for (int y = 1; y < 8; ++y)
            if (load[y] < load[x]) x = y;
        for (size_t i = block_start[bi]; i < block_start[bi + 1]; ++i) {
            q[x].push_back(i);
            load[x] += work[i];
        }
    }
    std::vector<size_t> pool;
    for (size_t x = 0; x < 8; ++x) {
        const size_t target = n / 8 + (x < n % 8 ? 1 : 0);
        while (q[x].size() > target) {
            pool.push_back(q[x].back());
            q[x].pop_back();
        }
    }
    for (size_t x = 0; x < 8; ++x) {
        const size_t target = n / 8 + (x < n % 8 ? 1 : 0);
        while (q[x].size() < target) {
            q[x].push_back(pool.back());
            pool.pop_back();
        }
    }
    const std::vector<T> ta(a.begin() + begin, a.begin() + end);
    const std::vector<U> tb(b.begin() + begin, b.begin() + end);
    for (size_t x = 0; x < 8; ++x)
        for (size_t j = 0; j < q[x].size(); ++j) {
            a[begin + x + 8 * j] = ta[q[x][j]];
            b[begin + x + 8 * j] = tb[q[x][j]];
        }
}

struct UpdateSplitter {
    int ncu = 256;
    bool xcd = true;
    int tail_split = DSMGP_TAIL_SPLIT_DEFAULT;     // K pieces per tile in the tail of a launch (1 = the tail is not cut)
    int tail_rounds = DSMGP_TAIL_ROUNDS_DEFAULT;   // full rounds (of ncu tiles) that belong to the tail
    std::vector<TileTask> upd;
    std::vector<ReduceTask> red;
    std::vector<int64_t> upd_slab, red_slab;   // slab index of a task's output / first slab, -1 = none
    size_t tail_begin = 0;                     // add_step: index in `upd` where the tail pieces of the last step begin (its end if none)
    size_t max_slabs = 0;

    // Cost model in units of one K column at the matrix pipe's full rate on one CU.  A CU holds two workgroups: two co-resident
    // pieces of depth d share the pipe and are both done after 2 d; ONE piece alone on a CU cannot saturate it (a wave issues an
    // f64 MFMA every ~100 cycles where the pipe takes one every 64: kernels.hpp, chol_diag_packed_body) and takes 1.56 d.
    // (Until round 4 the model counted rounds over `ncu` slots at d per round: right for pairs, but it left a step of 129..255
    // tiles unsplit -- one workgroup per CU at 64 % of the pipe -- where two half-depth pieces per CU take 36 % less time.)
    static int choose_split(int T, int K, int ncu) {
        if (T <= 0 || K < 512) return 1;
        const double C0 = 96.0, CRED = 64.0;
        const int chunks = K / KC;
        auto cost = [&](int S) {
            const double depth = (double)((chunks + S - 1) / S) * KC;
            const long P = (long)T * S, slots = 2L * ncu;
            const long full = P / slots, rem = P % slots;
            double t = (double)full * 2.0 * (depth + C0);
            if (rem > ncu) t += 2.0 * (depth + C0);
            else if (rem > 0) t += DSMGP_SOLO_FACTOR * depth + C0;
            return t + (S > 1 ? CRED : 0.0);
        };
        int best = 1;
        double bc = cost(1);
        for (int S = 2; S <= 64 && S * 256 <= K && (long)T * S <= 4096; ++S) {
            const double c = cost(S);
            if (c < 0.93 * bc) {     // (0.65 / 0.80 / 1.0 under two lanes: no difference, profiles/r05_tail_ab.log)
                bc = c;
                best = S;
            }
        }
        return best;
    }
    // Emit `tiles[from, to)` split S ways (S = 1: as they are), split-major, XCD-ordered.
    void emit(const std::vector<TileTask>& tiles, size_t from, size_t to, int S, size_t& slab) {
        const size_t begin = upd.size();
        if (S <= 1) {
            for (size_t i = from; i < to; ++i) {
                upd.push_back(tiles[i]);
                upd_slab.push_back(-1);
            }
            xcd_permute(upd, upd_slab, begin, upd.size(), xcd);
            return;
        }
        const size_t first = slab;
        for (size_t i = from; i < to; ++i) {
            ReduceTask r{};
            r.C = tiles[i].C;
            r.ldc = tiles[i].ldc;
            r.nsplit = S;
            // piece 0 stores product - Gram value, or the tile is defined as -product: nothing to read from it
            r.fresh = (tiles[i].gram != 0 || tiles[i].update == 2) ? 1 : 0;
            red.push_back(r);
            red_slab.push_back((int64_t)(first + (i - from) * S));
        }
        // split-major order: tasks with the same K range (and, per leaf, the same B panel) stay adjacent
        for (int s = 0; s < S; ++s)
            for (size_t i = from; i < to; ++i) {
                TileTask p = tiles[i];
                const long chunks = (tiles[i].k1 - tiles[i].k0) / KC;   // this tile's own K range
                p.k0 = tiles[i].k0 + (int)(chunks * s / S) * KC;
                p.k1 = tiles[i].k0 + (int)(chunks * (s + 1) / S) * KC;
                p.update = 0;
                p.C = nullptr;
                p.ldc = TB;
                if (s != 0) p.gram = 0;
                upd.push_back(p);
                upd_slab.push_back((int64_t)(first + (i - from) * S + s));
            }
        xcd_permute(upd, upd_slab, begin, upd.size(), xcd);
        slab += (to - from) * (size_t)S;
    }
    // tiles: tasks of one block step with update = 1 and (nearly) equal depth K.
    // Fewer tiles than CUs: split all of them (cost model).  Otherwise the step runs in rounds of one tile
    // per CU; the last, partial round would hold the whole launch for a full tile time, so only the
    // remainder tiles are split, finely enough to fit one short extra round, and issued last.
    void add_step(const std::vector<TileTask>& tiles, int K) {
        const size_t T = tiles.size();
        size_t slab = 0;
        if ((int)T < ncu) {
            tail_begin = upd.size();       // every tile is cut along K: one short round
            emit(tiles, 0, T, choose_split((int)T, K, ncu), slab);
        } else {
            // Whole tiles first, in rounds of one per CU; the tail of the launch -- the partial last round plus
            // `tail_rounds` full ones -- is cut into `tail_split` K pieces per tile and issued last, so the
            // launch ends on short tasks instead of idling CUs for up to a whole tile time.
            // The extra full round(s) in the tail pay off only in launches of under two rounds (measured, round 2: a round
            // of 256 extra tiles cut in four costs the headline step 3 ms of reduce traffic for nothing, while the 8-rank
            // shard, whose launches are that short, gains 1.7 % from it; with this rule: headline 0.4048 / 0.4019 ->
            // 0.3989 / 0.3993 s, 4-rank shard 0.1080 / 0.1070 -> 0.1051 / 0.1054 s, 8-rank shard and depth 4 unchanged).
            // Round 5: with two leaf lanes the other lane's launch runs in the slots a tail leaves idle, and the cut tail only
            // costs (its slabs, the reduce launch on the chain): tail_split 4 -> 1 and tail_rounds 1 -> 0 WHERE THE PLAN HAS LANES,
            // same box, alternating: headline 0.3871 / 0.3862 -> 0.3825 / 0.3838 s, 8-rank shards 0.0548 / 0.0559 -> 0.0531 / 0.0534
            // and 0.0551 / 0.0557 -> 0.0531 / 0.0531, 4-rank shard 0.1018 / 0.1024 -> 0.1000 / 0.1009, depth 4 unchanged; with one
            // lane the same setting costs 4 % (0.4090 / 0.4116 -> 0.4261 / 0.4266; shards +2-3 %): profiles/r05_tail_ab.log.  What
            // stays with lanes: a remainder of under a quarter round is cut finely enough to fill one round (below).
            const size_t r = T % (size_t)ncu;
            size_t ntail = r + (T < (size_t)(2 * ncu) ? (size_t)tail_rounds * ncu : 0);
            if (ntail > T) ntail = T;
            int S = std::min(tail_split, std::max(1, K / 256));
            if (r > 0 && r * 4 < (size_t)ncu) S = std::max(S, (int)std::min<size_t>((size_t)ncu / r, 16));   // tiny remainder: finer
            S = std::min(S, std::max(1, K / 256));
            if (S <= 1 || r == 0) ntail = 0;   // an exact multiple of the CU count already ends evenly
            // (Round 4 tried cutting the last ncu whole tiles in two along K where the whole tiles make an odd number of ncu-rounds
            // -- the reasoning: two workgroups per CU, so an odd round would run one workgroup per CU at 64 % of the pipe.  No
            // effect: headline 0.3847 / 0.3848 / 0.3854 without, 0.3847 / 0.3860 / 0.3853 with; slots free up one by one, there
            // is no such round.)
            emit(tiles, 0, T - ntail, 1, slab);
            tail_begin = upd.size();
            if (ntail) emit(tiles, T - ntail, T, S, slab);
        }
        max_slabs = std::max(max_slabs, slab);
    }
    // A step whose tiles have every depth from 1 to `kblocks` blocks (the blocks of L^-T), listed leaf by leaf, deepest
    // first.  Tasks are dispatched in list order, so the deepest tiles of the last leaves start when the launch is nearly
    // over and the chip waits for them (15 % of a launch at k = 41).  Reordering by depth is not an option -- the tiles
    // of a leaf must run together to share their B panel through L2 (passes over the leaves by depth class: 35 % slower)
    // -- so the last two rounds' worth of tiles are cut along K into pieces of at most a quarter of the step's depth.
    int ragged_rounds = 2, ragged_div = 4;
    void add_step_ragged(const std::vector<TileTask>& tiles, int Kavg, int kblocks) {
        const size_t T = tiles.size();
        if (kblocks < 8) {
            add_step(tiles, Kavg);
            return;
        }
        size_t slab = 0;
        size_t cut = 0;
        int maxb;
        if (T < (size_t)((ragged_rounds + 2) * ncu)) {
            // few tiles (the late steps: a handful of leaves, every depth): all of them in pieces of equal depth, about
            // three pieces per workgroup slot of the chip
            long W = 0;
            for (const TileTask& u : tiles) W += (u.k1 - u.k0) / TB;
            maxb = (int)std::min<long>(kblocks, std::max<long>(2, W / (long)(2 * ncu * 3)));
        } else {
            cut = T - (size_t)(ragged_rounds * ncu);
            emit(tiles, 0, cut, 1, slab);
            maxb = std::max(1, kblocks / ragged_div);
        }
        auto pieces = [&](const TileTask& u) { return std::max(1, ((u.k1 - u.k0) / TB + maxb - 1) / maxb); };
        for (size_t i = cut; i < T;) {
            const int S = pieces(tiles[i]);
            size_t j = i + 1;
            while (j < T && pieces(tiles[j]) == S) ++j;
            emit(tiles, i, j, S, slab);
            i = j;
        }
        max_slabs = std::max(max_slabs, slab);
    }
    void bind(double* workspace) {
        for (size_t i = 0; i < upd.size(); ++i)
            if (upd_slab[i] >= 0) upd[i].C = workspace + (size_t)upd_slab[i] * TB * TB;
        for (size_t i = 0; i < red.size(); ++i) red[i].slabs = workspace + (size_t)red_slab[i] * TB * TB;
    }
};

}  // namespace

struct dsmgp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int profile = 0;                // 0: totals only, 1: events around the update launches (dominant kernel), 2: every category
    bool alt_names = false;         // launches under the instantiation names of profile 0 whatever the level (dsmgp_set_profile 3)

    int64_t N = 0;
    int D = 0;
    double* dX = nullptr;
    double* dy = nullptr;

    int L = 0;
    std::vector<LeafHost> leaves;
    std::vector<int64_t> obs_ptr, obs_idx;
    int64_t* d_obs_ptr = nullptr;
    int64_t* d_obs_idx = nullptr;
    bool plan_ready = false;

    std::vector<HyperHost> hyper;
    KParam* d_kp = nullptr;
    double* d_l2 = nullptr;
    size_t kp_cap = 0, l2_cap = 0;   // capacities of d_kp / d_l2 (elements)

    double* arenaF = nullptr;
    double* arenaDinv = nullptr;
    double* arenaVec = nullptr;     // per leaf: yc, w, z, alpha (4 x npad)
    double* arenaXg = nullptr;
    int* d_info = nullptr;
    int* d_owner = nullptr;         // per leaf: the leaf whose factor (and info slot) it uses (dsmgp_fit_exchange packs info per owner)
    double* d_mll = nullptr;
    LeafDev* d_leaves = nullptr;
    std::vector<LeafDev> h_leaves;
    size_t bytes_needed = 0;

    DevBuf<GramTask> gram;          // Gram launch of fit!: every lower tile, or (fused) the tiles no update task writes
    bool fuse_gram = true;          // update tasks of fit! evaluate the Gram values of their tile themselves (TileTask.gram)
    bool fuse_steps = true;         // block steps with more diagonal blocks than CUs run as two fused launches (kernels_fused.hpp)
    bool diag_in_update = true;     // classic steps: the diagonal tile's update runs one step ahead and its factorisation rides in
                                    // the update launch (DiagFinishTask, kernels_fused.hpp)
    // per phase and block step (decided by build_plan):
    //   STEP_CLASSIC    update (all tiles, split-K) / reduce / diagonal block / panel solve launches, one after the other
    //   STEP_FUSED      many leaves: diag_fused_reg_kernel, then tile_fused8_kernel, from the kernel function (kernels_fused.hpp)
    // (A third, lookahead schedule -- the update of step k cut at its last block column, the bulk on this stream, the rank-128
    // finish + diagonal block + solve on a second, high-priority stream beside the bulk of step k + 1 -- was built in round 3
    // and measured in every regime it was meant for: headline 0.4037 / 0.4055 s against 0.3906 / 0.3992, config 2 3.51 against
    // 3.35 ms, 8-rank shards 0.0571 / 0.0569 / 0.0568 / 0.0572 against 0.0577 / 0.0575 / 0.0574 / 0.0575, a 4-rank shard 0.1089 /
    // 0.1082 against 0.1054 / 0.1053.  Removed in round 4; the numbers are in DESIGN.md section 8d.)
    // (Round 4 also built the whole classic step as ONE launch -- diagonal blocks ahead and first in the grid, tile tasks that
    // update from the kernel function, wait inside the launch for their leaf's flag, solve from their registers and write once,
    // K-pieces of split tiles meeting at a ticket -- commit 9fc7be2, parity-green, and measured slower everywhere: headline
    // 0.3947 -> 0.4096 s, 8-rank shard 0.0562 -> 0.0645, config 2 2.51 -> 4.40 ms (profiles/r04_one_launch_ab.log): the last
    // piece to arrive sums up to 46 slabs alone where the reduce launch spreads a tile over 8 workgroups, and the row-split
    // main loop the in-register solve needs runs 5-10 % behind the 2 x 2 one at K >= 2000.  Removed again.)
    std::vector<char> fused_step[2];
    // Leaf lanes (round 5, DSMGP_OPT_LANES): the leaves of a table are dealt to one or two LANES (longest-processing-time on n^3,
    // sharing groups together); every lane has step lists and a split-K workspace of its own and runs them on a stream of its
    // own, joined at the end of fit!.  One lane's latency-bound launches (diagonal blocks, panel solves, split-K reduces) then
    // run under the other's update launches -- what several contexts per GPU bought (hipabi.MultiContext: headline -1.4 %, depth
    // 4 -3..-4 %) without a second copy of X, a second plan or host threads.  Per-leaf results do not depend on the lane.
    int lanes_opt = 0;              // 0 = automatic, 1 .. MAX_LANES
    int nlanes = 1;                 // lanes of the current plan
    hipStream_t lane_stream[MAX_LANES] = {};           // [0] = stream
    hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {};
    std::vector<char> leaf_lane;    // per leaf (COPY / PREFIX leaves: their source's)
    StepLists phase[MAX_LANES][2];  // [lane][0: FULL leaves, 1: PREFIX leaves (need their source first)]
    std::vector<char> leaf_group;   // per leaf: the phase it belongs to (COPY leaves ride with their source)
    // Optional device pool (dsmgp_reserve): the large arenas are carved out of one allocation made once, in stack
    // order plan < test < gradients, instead of hipMalloc/hipFree per leaf table -- the driver clears memory on
    // allocation (5 s per 230 GB group measured), which dominated the streaming mode's wall time.
    char* pool_base = nullptr;
    size_t pool_cap = 0, pool_top = 0, pool_mark_plan = 0;
    double* slabF[MAX_LANES] = {};  // split-K workspace of the factorisation, per lane
    StepLists phaseJ[MAX_LANES][2]; // the same with the resident test rows riding along (built by set_test)
    double* slabJ[MAX_LANES] = {};
    double alg_flops_joint = 0.0;
    bool joint = true;              // fit advances the resident test rows too
    bool joint_ready = false;
    bool vt_valid = false;          // Vt holds K_tn L^-T for the current factor
    bool last_fit_joint = false;
    double* slabP = nullptr;        // ... of the prediction sweep
    int ncu = 256;
    bool xcd_order = true;          // XCD-aware task order (speed only)
    int tail_split = DSMGP_TAIL_SPLIT_DEFAULT, tail_rounds = DSMGP_TAIL_ROUNDS_DEFAULT;          // plans with one lane
    int tail_split_lanes = DSMGP_TAIL_SPLIT_LANES, tail_rounds_lanes = DSMGP_TAIL_ROUNDS_LANES;  // plans with two or more
    int ragged_rounds = 2, ragged_div = 4;   // UpdateSplitter::add_step_ragged (launches of the gradient pass)
    std::vector<int> fwd_off, bwd_off;
    DevBuf<SolveTask> fwd, bwd;
    int solve_steps = 0;
    bool fitted = false;
    // The launch sequence of fit! as a captured hipGraph (one per variant: fit alone / with the resident test rows), replayed
    // while per-launch timing is off (dsmgp_set_profile 0): every kernel argument is a pointer into the plan's task lists or
    // arenas, which live as long as the plan, and the hyper-parameters go through d_kp's CONTENTS -- so a graph stays valid
    // until the plan, the test set or the KParam table's allocation changes (drop_graphs).  What it buys is the host-side
    // launch cost between dependent kernels of a latency-bound chain (config 2: 32 steps x 3 launches).
    bool use_graph = false;         // opt-in (DSMGP_OPT_FIT_GRAPH): config 2 2.50 -> 2.47 ms, nothing elsewhere; capture does not mix with
                                    // several contexts driven from concurrent host threads (hipabi.MultiContext)
    hipGraphExec_t fit_graph[2] = {nullptr, nullptr};
    int graph_launches[2][2] = {{0, 0}, {0, 0}};
    bool alpha_valid = false;       // alpha = L^-T z has been computed for the current factors (ensure_alpha)
    bool dinv_complete = false;     // every Dinv_k holds the whole inverse for the current factors (ensure_dinv); a fit leaves the
                                    // blocks of its fused steps with their 16x16 diagonal inverses only
    DevBuf<DiagTask> dinvc_prefix;  // copied blocks of PREFIX leaves whose source factorised them in a fused step: completed right
                                    // after the copy (the classic steps of the PREFIX phase solve against Dinv_k)
    DevBuf<DiagTask> dinvc_fwd;     // blocks of the leaves whose z comes from the forward sweep (COPY, PREFIX): completed before it
    DevBuf<DiagTask> dinvc_all;     // every diagonal block a fused step factorises WITHOUT its inverse (only L_kk and the 16x16 diagonal
                                    // inverses exist after a fit): dinv_complete_kernel over this list produces the rest of Dinv_k for
                                    // whoever needs it (ensure_dinv).  Owned by the PLAN -- the list is the same with and without test
                                    // rows riding along, and must outlive a test set that is replaced between a fit and its first use

    // prediction
    double* dXt = nullptr;
    int64_t n_t = 0;
    int64_t* d_route_ptr = nullptr;
    int64_t* d_route_idx = nullptr;
    std::vector<int64_t> route_ptr;
    double* arenaVt = nullptr;
    size_t arenaVt_count = 0;       // doubles allocated for the K_tn arena: a test set that replaces another takes it over while it
                                    // fits (releasing ~10 GB and asking the driver for them again took up to 0.6 s of a 0.06 s predict)
    // capacities of the test set's other buffers (dev_grow: kept across registrations, freed with the context / the plan)
    size_t cap_dXt = 0, cap_route_ptr = 0, cap_route_idx = 0, cap_row_ptr = 0, cap_row_ent = 0, cap_ent_leaf = 0, cap_agg_coef = 0,
           cap_agg_group = 0, cap_agg_out = 0, cap_Xt = 0, cap_PV = 0, cap_slabP = 0;
    // pinned host staging for the uploads of a registration (stage_upload): the lists of a test set are ~10 MB at the headline
    // model; through hipMemcpy from pageable vectors they took 9 ms of a 25 ms registration and left the runtime busy behind them
    char* stage = nullptr;
    size_t stage_cap = 0, stage_top = 0;
    DevBuf<SweepSeg> psegs;         // (leaf, block step) pairs of the sweep's fused steps: build_sweep8_kernel makes the tasks
    double* arenaXt = nullptr;
    double* arenaPV = nullptr;      // mu | var (route order, unpadded) | macc | sacc (padded accumulators of the sweep)
    size_t acc_off = 0, acc_count = 0;
    DevBuf<PredTask> ptasks_slow;   // test tiles of leaves whose z is not produced during the factorisation
    DevBuf<GramTask> pgram;         // K_tn tiles the standalone sweep reads from memory (all of them only when D > 32)
    DevBuf<GramTask> pgram0;        // ... of the joint fit when the Gram is fused: block column 0 only
    DevBuf<PredTask> ptasks;
    std::vector<int> pupd_off, pred_off, ptrsm_off;
    DevBuf<FusedTask8> psweep8;     // the sweep's tasks in the block steps that run fused: update + solve of eight 16-row blocks each
    std::vector<int> psweep8_off;
    DevBuf<TileTask> pupd, ptrsm;
    DevBuf<ReduceTask> pred;
    int psteps = 0;
    int plan_lanes_test = 1;        // lanes the registered test set's sweep lists were built for
    bool test_ready = false;
    bool predicted = false;
    int64_t route_total = 0;

    // aggregation of the leaf moments per test row + scores (dsmgp_aggregate*, dsmgp_scores)
    int64_t* d_row_ptr = nullptr;   // n_t + 1: entries of every test row (built by set_test)
    int32_t* d_row_ent = nullptr;   // entry positions, ascending per row
    int32_t* d_ent_leaf = nullptr;  // leaf of every entry position
    double* d_agg_part = nullptr;   // W x n_t partial sums
    size_t agg_part_cap = 0;
    double* d_agg_coef = nullptr;   // L
    int32_t* d_agg_group = nullptr; // L
    double* d_agg_out = nullptr;    // mu | var (n_t each) | y_test (n_t) | score block sums
    int agg_family = -1, agg_G = 0, agg_W = 0;
    bool agg_partial_ready = false, agg_done = false;
    bool agg_total = false;         // d_agg_part holds the sum over ranks (dsmgp_aggregate_exchange ran on these partial sums)

    // gradients (built on first use)
    bool grad_ready = false;
    std::vector<char> grad_active;  // dsmgp_set_gradient_leaves: leaves whose gradients are wanted (empty = all)
    double* arenaX = nullptr;       // Xt = L^-T per factor owner, npad x npad
    size_t arenaX_count = 0;
    double* slabG = nullptr;
    size_t slabG_count = 0;
    DevBuf<TransTask> gtrans;
    std::vector<int> gupd_off, gred_off, gtrsm_off;   // entry q * gsteps + k: block step k of lane q
    int glanes = 1;                                    // lanes of the gradient plan (= nlanes of the plan it was built on)
    DevBuf<TileTask> gupd, gtrsm;
    DevBuf<ReduceTask> gred;
    int gsteps = 0;
    DevBuf<FrobTask> gfrob;
    std::vector<int> gfrob_leaf;    // owner leaf of each frob task
    DevBuf<GradTask> gdot;
    std::vector<int> gdot_leaf;     // leaf of each graddot task
    bool ard_true_gradient = false; // DSMGP_OPT_ARD_LENGTHSCALE_GRADIENT
    int gstride = 2;                // doubles per contraction task in d_gpart
    std::vector<int> grad_src;      // per leaf: the leaf whose contraction it shares (COPY leaf with the same mean), or -1
    double* d_gpart = nullptr;      // partial results: frob | graddot pairs | per-leaf dots
    size_t gpart_count = 0, gpart_cap = 0;

    // multi-GPU exchange over RCCL (dsmgp_comm_*, dsmgp_allgather): librccl.so is loaded on first use
    void* comm = nullptr;           // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    double* d_xchg = nullptr;       // send | recv staging
    size_t xchg_cap = 0;
    // the model's tree as flat arrays in HBM (dsmgp_set_tree): the routing of predict runs on the device (dsmgp_set_test_routed)
    RouteTree rtree{};
    int64_t rtree_nodes = 0;
    int rtree_max_leaf = -1;        // largest local leaf index a region names (checked against the leaf table at routing time)
    bool rtree_ready = false;
    // routing workspace, kept across registrations: row counts | leaf counts | outside flag, bitmap, word prefixes
    int32_t* rws_counts = nullptr;
    size_t rws_counts_cap = 0;
    uint32_t* rws_bits = nullptr;
    size_t rws_bits_cap = 0;
    hipStream_t side = nullptr;     // clock sampler (dsmgp_clock_sample_*): a one-wave kernel beside the context's own launches
    unsigned long long* d_clock = nullptr;
    bool clock_pending = false;
    double timings[DSMGP_N_TIMINGS] = {0};
    std::vector<hipEvent_t> event_pool;   // PhaseTimer's events, reused across calls
    double alg_flops_update = 0.0;  // algorithmic flops of the Cholesky update launches
    double alg_flops_fused = 0.0, alg_flops_fused_joint = 0.0;   // ... of the fused tile launches (update + solve), fit alone / joint
    int n_fused_launches = 0;
    bool phase_ready = false;       // `phase` (fit! without resident test rows) has been built for the current plan
    int n_update_launches = 0;
};

namespace {

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return (e_ == hipErrorOutOfMemory) ? DSMGP_E_NOMEM : DSMGP_E_HIP;                    \
        }                                                                                        \
    } while (0)

int fail(dsmgp_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}

// `count` entries for a task list: the allocation is kept while the new list fits (a test set that replaces another, a
// gradient mask that changes: hipFree + hipMalloc per list and registration were 3-4 ms of a 16 ms predict at depth 4)
template <class T>
int dev_reserve(dsmgp_ctx* ctx, DevBuf<T>& buf, size_t count) {
    buf.count = 0;
    if (count > buf.cap || (count && !buf.p)) {
        if (buf.p) {
            HIPCHK(ctx, hipFree(buf.p));
            buf.p = nullptr;
        }
        buf.cap = 0;
        const size_t want = count + count / 8;
        HIPCHK(ctx, hipMalloc(&buf.p, want * sizeof(T)));
        buf.cap = want;
    }
    buf.count = count;
    return 0;
}
template <class T>
int dev_upload(dsmgp_ctx* ctx, DevBuf<T>& buf, const std::vector<T>& host) {
    if (int rc = dev_reserve(ctx, buf, host.size())) return rc;
    if (host.empty()) return 0;
    HIPCHK(ctx, hipMemcpy(buf.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

template <class T>
void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}
template <class T>
void dev_free(DevBuf<T>& b) {      // a freed list is an EMPTY list: nobody may launch over its old count
    dev_free(b.p);
    b.count = 0;
    b.cap = 0;
}
template <class T>
void dev_drop(DevBuf<T>& b, bool keep) {    // keep: the list is emptied, its allocation stays for the list that replaces it
    if (keep) b.count = 0;
    else dev_free(b);
}
// a plain device array that grows on demand and otherwise stays (the buffers of a registered test set)
template <class T>
int dev_grow(dsmgp_ctx* ctx, T*& p, size_t& cap, size_t need) {
    if (p && need <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(1, need + need / 8);
    HIPCHK(ctx, hipMalloc(&p, want * sizeof(T)));
    cap = want;
    return 0;
}
template <class T>
void dev_drop(T*& p, size_t& cap, bool keep) {
    if (keep) return;
    dev_free(p);
    cap = 0;
}

bool in_pool(const dsmgp_ctx* c, const void* p) {
    return c->pool_base && (const char*)p >= c->pool_base && (const char*)p < c->pool_base + c->pool_cap;
}
// `bytes` from host memory to the device through the context's PINNED staging buffer, asynchronously on the context's stream: the
// caller's memory is free again on return, the copy is done once the stream is synchronised (stage_done, which every user calls
// before it returns).  Grows on demand; growing waits for the copies in flight.
int stage_upload(dsmgp_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return 0;
    const size_t at = (c->stage_top + 63) & ~size_t(63);
    if (at + bytes > c->stage_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->stage) (void)hipHostFree(c->stage);
        c->stage = nullptr;
        c->stage_cap = 0;
        c->stage_top = 0;
        const size_t want = std::max<size_t>(size_t(8) << 20, 2 * (at + bytes));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->stage), want, hipHostMallocDefault));
        c->stage_cap = want;
        return stage_upload(c, dst, src, bytes);
    }
    std::memcpy(c->stage + at, src, bytes);
    c->stage_top = at + bytes;
    HIPCHK(c, hipMemcpyAsync(dst, c->stage + at, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int stage_done(dsmgp_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stage_top = 0;
    return 0;
}
template <class T>
int stage_upload_list(dsmgp_ctx* c, DevBuf<T>& buf, const std::vector<T>& host) {     // dev_upload through the staging buffer
    if (int rc = dev_reserve(c, buf, host.size())) return rc;
    return stage_upload(c, buf.p, host.data(), host.size() * sizeof(T));
}

// `count` doubles for one of the large arenas: from the pool when there is one, else its own allocation
int arena_get(dsmgp_ctx* c, double*& p, size_t count) {
    const size_t bytes = std::max<size_t>(1, count) * sizeof(double);
    if (c->pool_base) {
        const size_t off = (c->pool_top + 255) & ~size_t(255);
        if (off + bytes > c->pool_cap)
            return fail(c, DSMGP_E_NOMEM, "reserved device pool too small: need " + std::to_string((off + bytes) >> 20) +
                                              " MiB, pool has " + std::to_string(c->pool_cap >> 20) + " MiB");
        p = reinterpret_cast<double*>(c->pool_base + off);
        c->pool_top = off + bytes;
        return 0;
    }
    HIPCHK(c, hipMalloc(&p, bytes));
    return 0;
}
void arena_put(dsmgp_ctx* c, double*& p) {
    if (p && !in_pool(c, p)) (void)hipFree(p);
    p = nullptr;
}

void drop_graphs(dsmgp_ctx* c) {
    for (auto& g : c->fit_graph)
        if (g) {
            (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
}

// the task lists of the gradient pass (they depend on the set of active leaves): emptied, their allocations and the L^-T arena
// stay -- finetune! changes the mask L times per iteration (src/finetuning.jl:34-57), and the lists that replace these fit into
// the same buffers (dev_upload) instead of a hipFree + hipMalloc per list and pass
void free_grad_lists(dsmgp_ctx* c) {
    dev_drop(c->gtrans, true);
    dev_drop(c->gupd, true);
    dev_drop(c->gtrsm, true);
    dev_drop(c->gred, true);
    dev_drop(c->gfrob, true);
    dev_drop(c->gdot, true);
    c->grad_ready = false;
}

void free_grad(dsmgp_ctx* c) {
    arena_put(c, c->slabG);
    c->slabG_count = 0;
    arena_put(c, c->arenaX);
    dev_free(c->gtrans);
    dev_free(c->gupd);
    dev_free(c->gtrsm);
    dev_free(c->gred);
    dev_free(c->gfrob);
    dev_free(c->gdot);
    dev_free(c->d_gpart);
    c->gpart_cap = 0;
    c->grad_ready = false;
}

void free_tree(dsmgp_ctx* c) {
    dev_free(const_cast<int8_t*&>(c->rtree.kind));
    dev_free(const_cast<int32_t*&>(c->rtree.first));
    dev_free(const_cast<int32_t*&>(c->rtree.nchild));
    dev_free(const_cast<int32_t*&>(c->rtree.sdim));
    dev_free(const_cast<int32_t*&>(c->rtree.leaf));
    dev_free(const_cast<double*&>(c->rtree.thr));
    c->rtree_ready = false;
    c->rtree_nodes = 0;
    c->rtree_max_leaf = -1;
}

void free_test(dsmgp_ctx* c, bool keep = false);
void free_plan(dsmgp_ctx* c) {
    drop_graphs(c);
    if (c->pool_base) {      // stack order: everything above the plan goes with it
        free_test(c);
        c->pool_top = 0;
        c->pool_mark_plan = 0;
    }
    arena_put(c, c->arenaF);
    arena_put(c, c->arenaDinv);
    arena_put(c, c->arenaVec);
    arena_put(c, c->arenaXg);
    dev_free(c->d_info);
    dev_free(c->d_owner);
    dev_free(c->d_mll);
    dev_free(c->d_leaves);
    dev_free(c->gram);
    for (auto& lane : c->phase)
        for (auto& ph : lane) {
            dev_free(ph.upd);
            dev_free(ph.trsm);
            dev_free(ph.red);
            dev_free(ph.diag);
            dev_free(ph.fdiag);
            dev_free(ph.ftile8);
            dev_free(ph.dfin);
        }
    for (auto& sl : c->slabF) arena_put(c, sl);
    dev_free(c->fwd);
    dev_free(c->bwd);
    dev_free(c->dinvc_prefix);
    dev_free(c->dinvc_fwd);
    dev_free(c->dinvc_all);
    free_grad(c);
    c->plan_ready = false;
    c->phase_ready = false;
    c->fitted = false;
}

// keep: a registration that replaces another -- every buffer of the old test set stays allocated for the new one (they are
// emptied, not freed).  With a device pool the arenas are carved out of the pool's stack and go with its top.
void free_test(dsmgp_ctx* c, bool keep) {
    drop_graphs(c);
    if (c->pool_base) {      // the gradient arenas sit above (or would be clobbered below) the test arenas
        free_grad(c);
        c->pool_top = c->pool_mark_plan;
        keep = false;
    }
    dev_drop(c->dXt, c->cap_dXt, keep);
    dev_drop(c->d_route_ptr, c->cap_route_ptr, keep);
    dev_drop(c->d_route_idx, c->cap_route_idx, keep);
    dev_drop(c->d_row_ptr, c->cap_row_ptr, keep);
    dev_drop(c->d_row_ent, c->cap_row_ent, keep);
    dev_drop(c->d_ent_leaf, c->cap_ent_leaf, keep);
    dev_drop(c->d_agg_part, c->agg_part_cap, keep);
    dev_drop(c->d_agg_coef, c->cap_agg_coef, keep);
    dev_drop(c->d_agg_group, c->cap_agg_group, keep);
    dev_drop(c->d_agg_out, c->cap_agg_out, keep);
    c->agg_partial_ready = c->agg_done = c->agg_total = false;
    if (!keep) {
        arena_put(c, c->arenaVt);
        c->arenaVt_count = 0;
        arena_put(c, c->arenaXt);
        arena_put(c, c->arenaPV);
        arena_put(c, c->slabP);
        c->cap_Xt = c->cap_PV = c->cap_slabP = 0;
    }
    dev_drop(c->pgram, keep);
    dev_drop(c->pgram0, keep);
    dev_drop(c->ptasks, keep);
    dev_drop(c->ptasks_slow, keep);
    dev_drop(c->pupd, keep);
    dev_drop(c->ptrsm, keep);
    dev_drop(c->pred, keep);
    dev_drop(c->psweep8, keep);
    dev_drop(c->psegs, keep);
    for (auto& lane : c->phaseJ)
        for (auto& ph : lane) {
            dev_drop(ph.upd, keep);
            dev_drop(ph.trsm, keep);
            dev_drop(ph.red, keep);
            dev_drop(ph.diag, keep);
            dev_drop(ph.fdiag, keep);
            dev_drop(ph.ftile8, keep);
            dev_drop(ph.dfin, keep);
        }
    for (auto& sl : c->slabJ) arena_put(c, sl);
    c->joint_ready = false;
    c->vt_valid = false;
    c->test_ready = false;
    c->predicted = false;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// TileTask.mrows of a row tile with `valid` data rows: rounded up to the MFMA tile height; 0 = the whole tile
inline int tile_mrows(int valid) {
    const int m = round_up(std::max(0, std::min(TB, valid)), 16);
    return m >= TB ? 0 : std::max(16, m);
}

// Upload the KParam table from the host hyper-parameters.
int upload_hyper(dsmgp_ctx* c) {
    const int nk = (int)c->hyper.size();
    std::vector<double> l2pool;
    std::vector<KParam> kp(nk);
    std::vector<size_t> off(nk);
    for (int k = 0; k < nk; ++k) {
        const HyperHost& h = c->hyper[k];
        off[k] = l2pool.size();
        if (h.kind < 0) {   // id never set: no leaf may use it (check_hyper)
            kp[k] = KParam{0, 0, 1.0, 1.0, 1.0, nullptr, nullptr, 0.0, 1.0};
            continue;
        }
        const int nl = (int)h.loghyp.size() - 2;
        off[k] = l2pool.size();
        for (int i = 0; i < nl; ++i) {
            const double l = std::exp(h.loghyp[i]);
            l2pool.push_back(l * l);
        }
        kp[k].kind = h.kind;
        kp[k].nl = nl;
        const double logs = h.loghyp[nl];
        const double logn = h.loghyp[nl + 1];
        kp[k].sigma2 = (h.kind == DSMGP_KIND_ISO_LINEAR) ? 1.0 : std::exp(2.0 * logs);
        kp[k].sigma = (h.kind == DSMGP_KIND_ISO_LINEAR) ? 1.0 : std::exp(logs);
        kp[k].noise = std::exp(2.0 * logn);
    }
    const size_t nslots = l2pool.size();
    for (size_t i = 0; i < nslots; ++i) l2pool.push_back(-0.5 / l2pool[i]);   // second half: the exponent's factor
    if (l2pool.size() > c->l2_cap || !c->d_l2) {   // (re)allocate only when the table grows: fit is called in loops
        drop_graphs(c);
        dev_free(c->d_l2);
        c->l2_cap = std::max<size_t>(16, 2 * l2pool.size());
        HIPCHK(c, hipMalloc(&c->d_l2, c->l2_cap * sizeof(double)));
    }
    if ((size_t)nk > c->kp_cap || !c->d_kp) {
        drop_graphs(c);
        dev_free(c->d_kp);
        c->kp_cap = std::max<size_t>(4, 2 * (size_t)nk);
        HIPCHK(c, hipMalloc(&c->d_kp, c->kp_cap * sizeof(KParam)));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_l2, l2pool.data(), l2pool.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (int k = 0; k < nk; ++k) {
        kp[k].l2 = c->d_l2 + off[k];
        kp[k].nh = c->d_l2 + nslots + off[k];
        if (c->hyper[k].kind >= 0) {
            kp[k].nh0 = l2pool[nslots + off[k]];
            kp[k].il2 = 1.0 / l2pool[off[k]];
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->d_kp, kp.data(), nk * sizeof(KParam), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // l2pool / kp are stack storage
    return 0;
}

int check_hyper(dsmgp_ctx* c) {
    for (int l = 0; l < c->L; ++l) {
        const int kid = c->leaves[l].kid;
        if (kid < 0 || kid >= (int)c->hyper.size() || c->hyper[kid].kind < 0)
            return fail(c, DSMGP_E_STATE, "leaf " + std::to_string(l) + " uses kernel id " + std::to_string(kid) +
                                              " without hyper-parameters");
        const HyperHost& h = c->hyper[kid];
        const int nl = (int)h.loghyp.size() - 2;
        if (h.kind == DSMGP_KIND_ARD_SE && nl != c->D)
            return fail(c, DSMGP_E_ARG, "ArdSE needs one lengthscale per input dimension");
        if (h.kind != DSMGP_KIND_ARD_SE && nl != 1) return fail(c, DSMGP_E_ARG, "Iso kernels take one lengthscale");
    }
    return 0;
}

// Algorithmic flops of the left-looking update of a leaf of (unpadded) size n: for every block column k,
// 2*K flops (K = 128k) per lower-triangle element of that block column.
// A PREFIX leaf keeps the leading kb x kb blocks of its source: in block columns k < kb only the rows >= 128 kb are
// updated (they are the launches of build_factor_steps), the copied part costs nothing.
template <class Depth>
double update_flops(int n, int kb, Depth depth /* block step -> blocks of K the timed launch covers */) {
    double f = 0.0;
    for (int k = 1; k * TB < n; ++k) {
        const int c0 = k * TB, c1 = std::min(n, c0 + TB);
        double elems = 0.0;
        if (k < kb) elems = (double)(c1 - c0) * (double)std::max(0, n - kb * TB);
        else
            for (int c = c0; c < c1; ++c) elems += (double)(n - c);
        f += 2.0 * (double)(depth(k) * TB) * elems;
    }
    return f;
}

// Per-leaf algorithmic flops of the test rows riding through the sweep: 2*K per (test row, column) element.
template <class Depth>
double predict_update_flops(int n, int nt, Depth depth) {
    double f = 0.0;
    for (int k = 1; k * TB < n; ++k) f += 2.0 * (double)(depth(k) * TB) * (double)nt * (double)std::min(TB, n - k * TB);
    return f;
}

// Gram values evaluated by the update tasks themselves (TileTask.gram) instead of written by the Gram launch and read
// back: needs the coordinate image of a tile to fit the kernel's LDS ring.
bool gram_fused(const dsmgp_ctx* c) { return c->fuse_gram && c->D <= GRAM_FUSE_MAX_D; }

// The 16-row blocks of one leaf below the diagonal block of step k -> fused tile tasks of eight (tile_fused8_kernel): all of
// them read the leaf's B panel F[k, 0:K], L_kk behind it and the coordinates of the block's columns.  Empties `blocks`.
void push_fused8_tasks(std::vector<FusedTask8>& out, std::vector<RowBlock>& blocks, const LeafDev& d, const LeafHost& lf, int k) {
    for (size_t b0 = 0; b0 < blocks.size(); b0 += 8) {
        FusedTask8 f{};
        f.B = d.F + (size_t)k * TB;
        f.Dinv = d.Dinv + (size_t)k * TB * TB;
        f.gxb = d.Xg + (size_t)k * TB;
        f.ldb = f.gldb = lf.npad;
        f.gnb = std::max(0, std::min(TB, lf.n - k * TB));
        f.k1 = k * TB;
        f.kid = lf.kid;
        f.nblk = (int)std::min<size_t>(8, blocks.size() - b0);
        for (int q = 0; q < f.nblk; ++q) {
            f.rb[q] = blocks[b0 + q];
            if (f.rb[q].wi != nullptr) f.zk = d.z + (size_t)k * TB;     // riders read z_k
        }
        out.push_back(f);
    }
    blocks.clear();
}

// Step lists of the batched left-looking factorisation.  phase[0]: leaves factorised in full (and, with
// `with_test`, the test rows of those leaves and of the COPY leaves that alias them); phase[1]: PREFIX leaves.
// with_test: the rows of K_tn (Vt) of every leaf are appended below its factor and advance through the same
// update / panel-solve launches -- prediction's triangular solves (src/gaussianprocess.jl:120) cost no launches
// of their own when the test set is resident at fit time.
int build_factor_steps(dsmgp_ctx* c, int lane, bool with_test, StepLists (&phase)[2], double*& slab_ws, double& alg_flops,
                       double& alg_flops_fused, bool slab_outside_pool = false) {
    const int L = c->L;
    alg_flops = 0.0;
    const bool fused = gram_fused(c);
    UpdateSplitter split[2];
    for (int ph = 0; ph < 2; ++ph) {
        StepLists& S = phase[ph];
        UpdateSplitter& U = split[ph];
        U.ncu = c->ncu;
        U.xcd = c->xcd_order;
        U.tail_split = c->nlanes > 1 ? c->tail_split_lanes : c->tail_split;
        U.tail_rounds = c->nlanes > 1 ? c->tail_rounds_lanes : c->tail_rounds;
        auto in_phase = [&](const LeafHost& lf) {
            const size_t l = (size_t)(&lf - c->leaves.data());
            return c->leaf_group[l] == ph && c->leaf_lane[l] == lane;
        };
        int nsteps = 0;
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (!in_phase(lf)) continue;
            if (lf.owner == l || (with_test && lf.nt > 0)) nsteps = std::max(nsteps, lf.nb);
        }
        S.nsteps = nsteps;
        std::vector<TileTask> trsm;
        std::vector<DiagTask> diag;
        std::vector<FusedTask8> ftile8;
        std::vector<RowBlock> blocks8;      // 16-row blocks of one leaf below the step's diagonal block: factor rows, then test rows
        std::vector<DiagFusedTask> fdiag;
        std::vector<DiagFinishTask> dfin;
        S.dpos.assign(nsteps, 0);
        S.dfin_off.assign(nsteps + 1, 0);
        S.upd_off.assign(nsteps + 1, 0);
        S.red_off.assign(nsteps + 1, 0);
        S.trsm_off.assign(nsteps + 1, 0);
        S.diag_off.assign(nsteps + 1, 0);
        S.fdiag_off.assign(nsteps + 1, 0);
        S.ftile8_off.assign(nsteps + 1, 0);
        S.step_tiles.assign(nsteps, 0);
        S.pad.assign(nsteps, 0);
        S.mode.assign(nsteps, STEP_CLASSIC);
        for (int k = 0; k < nsteps; ++k) {
            S.upd_off[k] = (int)U.upd.size();
            S.red_off[k] = (int)U.red.size();
            S.trsm_off[k] = (int)trsm.size();
            S.diag_off[k] = (int)diag.size();
            S.fdiag_off[k] = (int)fdiag.size();
            S.ftile8_off[k] = (int)ftile8.size();
            S.dfin_off[k] = (int)dfin.size();
            const int mode = k < (int)c->fused_step[ph].size() ? c->fused_step[ph][k] : STEP_CLASSIC;
            S.mode[k] = (char)mode;
            const bool fstep = mode != STEP_CLASSIC;               // diagonal blocks and tiles below go through the fused kernels
            std::vector<TileTask> tiles;
            // One-step lookahead of the diagonal blocks (DiagFinishTask): where steps k-1 and k are both classic, the update
            // launch of step k-1 has left tile (k,k) = K(k,k) - F[k,0:K-128] F[k,0:K-128]^T (`ahead` below, one step on), and
            // the diagonal-block task of step k -- rank-128 finish + factorisation -- rides in the update launch of step k.
            auto classic_at = [&](int q) {
                return q >= 0 && q < nsteps && (q >= (int)c->fused_step[ph].size() || c->fused_step[ph][q] == STEP_CLASSIC);
            };
            const bool finish_here = c->diag_in_update && gram_fused(c) && !fstep && k >= 2 && classic_at(k - 1);
            const bool ahead = c->diag_in_update && gram_fused(c) && !fstep && k >= 1 && classic_at(k + 1);
            size_t nsym = 0;
            for (int l = 0; l < L; ++l) {
                const LeafHost& lf = c->leaves[l];
                if (lf.nb <= k || !in_phase(lf)) continue;
                const LeafDev& d = c->h_leaves[l];
                const int ld = lf.npad;
                if (lf.owner == l) {
                    // a PREFIX leaf keeps the copied leading kb x kb blocks: for k < kb only rows >= kb are new
                    const int i_first = (k < lf.kb) ? lf.kb : k;
                    const bool own_diag = (k >= lf.kb);
                    const bool fin = finish_here && own_diag;    // tile (k,k) got all but its last block column one step ago
                    for (int i = i_first; i < lf.nb; ++i) {
                        if (fstep) {      // fused step: the diagonal tile belongs to the diagonal-block task; of the tiles below
                                          // the 16-row blocks that hold data are updated and solved, eight to a task (the rows
                                          // beyond stay zero: zero_pad_rows_kernel)
                            if (i == k) continue;
                            const int rows = std::max(0, std::min(TB, lf.n - i * TB));
                            for (int r = 0; r < rows; r += 16) {
                                RowBlock b{};
                                b.A = d.F + (size_t)i * TB + r;
                                b.C = d.F + (size_t)i * TB + r + (size_t)k * TB * ld;
                                b.gx = d.Xg + (size_t)i * TB + r;
                                b.lda = b.ldc = b.glda = ld;
                                b.nvalid = std::min(16, rows - r);
                                if (ph != 1) b.wi = d.w + (size_t)i * TB + r;     // fused forward substitution
                                blocks8.push_back(b);
                            }
                            continue;
                        }
                        if (k > 0 && !(i == k && fin)) {
                            TileTask u{};
                            u.A = d.F + (size_t)i * TB;
                            u.B = d.F + (size_t)k * TB;
                            u.C = d.F + (size_t)i * TB + (size_t)k * TB * ld;
                            u.lda = u.ldb = u.ldc = ld;
                            u.k0 = 0;
                            u.k1 = k * TB;
                            u.update = 1;
                            u.mrows = tile_mrows(lf.n - i * TB);
                            u.sym = (i == k) ? 1 : 0;   // diagonal tile: A == B, lower blocks only
                            nsym += (size_t)u.sym;
                            if (fused) {
                                u.gram = 1 | (i == k ? 2 : 0) | 4;   // bit 2: a tile of the factor is first written here, padding rows too
                                u.kid = lf.kid;
                                u.gxa = d.Xg + (size_t)i * TB;
                                u.gxb = d.Xg + (size_t)k * TB;
                                u.glda = u.gldb = ld;
                                u.gna = std::max(0, std::min(TB, lf.n - i * TB));
                                u.gnb = std::max(0, std::min(TB, lf.n - k * TB));
                            }
                            tiles.push_back(u);
                        }
                        if (i == k + 1 && ahead && k + 1 >= lf.kb) {
                            // the NEXT step's diagonal tile over the columns this step's tiles cover: same depth, the A panel
                            // of the tile (k+1, k) beside it; its last block column and its factorisation follow in the
                            // DiagFinishTask of the next update launch
                            TileTask u{};
                            u.A = u.B = d.F + (size_t)i * TB;
                            u.C = d.F + (size_t)i * TB + (size_t)i * TB * ld;
                            u.lda = u.ldb = u.ldc = ld;
                            u.k0 = 0;
                            u.k1 = k * TB;
                            u.update = 1;
                            u.mrows = tile_mrows(lf.n - i * TB);
                            u.sym = 1;
                            nsym += 1;
                            u.gram = 1 | 2 | 4;
                            u.kid = lf.kid;
                            u.gxa = u.gxb = d.Xg + (size_t)i * TB;
                            u.glda = u.gldb = ld;
                            u.gna = u.gnb = std::max(0, std::min(TB, lf.n - i * TB));
                            tiles.push_back(u);
                        }
                        if (i > k) {
                            TileTask s{};
                            s.A = d.F + (size_t)i * TB + (size_t)k * TB * ld;
                            s.B = d.Dinv + (size_t)k * TB * TB;
                            s.C = const_cast<double*>(s.A);
                            s.lda = ld;
                            s.ldb = TB;
                            s.ldc = ld;
                            s.k0 = 0;
                            s.k1 = TB;
                            s.update = 0;
                            s.mrows = tile_mrows(lf.n - i * TB);
                            if (ph != 1) {   // fused forward substitution for leaves factorised in full
                                s.zk = d.z + (size_t)k * TB;
                                s.wi = d.w + (size_t)i * TB;
                            }
                            trsm.push_back(s);
                        }
                    }
                    if (own_diag) {
                        DiagTask g{};
                        g.T = d.F + (size_t)k * TB + (size_t)k * TB * ld;
                        g.Dinv = d.Dinv + (size_t)k * TB * TB;
                        if (ph != 1) {
                            g.wk = d.w + (size_t)k * TB;
                            g.zk = d.z + (size_t)k * TB;
                        }
                        g.info = d.info;
                        g.ld = ld;
                        g.nvalid = std::max(0, std::min(TB, lf.n - k * TB));
                        g.row0 = k * TB;
                        if (fstep) {
                            DiagFusedTask fg{};
                            fg.d = g;
                            fg.A = d.F + (size_t)k * TB;
                            fg.gx = d.Xg + (size_t)k * TB;
                            fg.k1 = k * TB;
                            fg.glda = ld;
                            fg.kid = lf.kid;
                            fdiag.push_back(fg);
                        } else if (fin) {
                            DiagFinishTask ft{};
                            ft.d = g;
                            ft.A = d.F + (size_t)k * TB + (size_t)(k - 1) * TB * ld;
                            dfin.push_back(ft);
                        } else {
                            diag.push_back(g);
                        }
                    }
                }
                if (with_test && lf.nt > 0) {
                    for (int ti = 0; ti < lf.ntpad / TB; ++ti) {
                        double* tile = d.Vt + (size_t)ti * TB + (size_t)k * TB * lf.ntpad;
                        if (fstep) {
                            const int rows = std::max(0, std::min(TB, lf.nt - ti * TB));
                            for (int r = 0; r < rows; r += 16) {
                                RowBlock b{};
                                b.A = d.Vt + (size_t)ti * TB + r;
                                b.C = tile + r;
                                b.gx = d.Xtg + (size_t)ti * TB + r;
                                b.lda = b.ldc = b.glda = lf.ntpad;
                                b.nvalid = std::min(16, rows - r);
                                if (d.zfused) {
                                    b.wi = d.macc + (size_t)ti * TB + r;
                                    b.sq = d.sacc + (size_t)ti * TB + r;
                                }
                                blocks8.push_back(b);
                            }
                            continue;
                        }
                        if (k > 0) {
                            TileTask u{};
                            u.A = d.Vt + (size_t)ti * TB;
                            u.B = d.F + (size_t)k * TB;
                            u.C = tile;
                            u.lda = lf.ntpad;
                            u.ldb = ld;
                            u.ldc = lf.ntpad;
                            u.k0 = 0;
                            u.k1 = k * TB;
                            u.update = 1;
                            u.mrows = tile_mrows(lf.nt - ti * TB);
                            if (fused) {
                                u.gram = 1;
                                u.kid = lf.kid;
                                u.gxa = d.Xtg + (size_t)ti * TB;
                                u.gxb = d.Xg + (size_t)k * TB;
                                u.glda = lf.ntpad;
                                u.gldb = ld;
                                u.gna = std::max(0, std::min(TB, lf.nt - ti * TB));
                                u.gnb = std::max(0, std::min(TB, lf.n - k * TB));
                            }
                            tiles.push_back(u);
                        }
                        TileTask s{};
                        s.A = tile;
                        s.B = d.Dinv + (size_t)k * TB * TB;
                        s.C = tile;
                        s.lda = lf.ntpad;
                        s.ldb = TB;
                        s.ldc = lf.ntpad;
                        s.k0 = 0;
                        s.k1 = TB;
                        s.update = 0;
                        s.mrows = tile_mrows(lf.nt - ti * TB);
                        if (d.zfused) {   // z_k exists by the time this launch runs: accumulate mu and the variance term
                            s.zk = d.z + (size_t)k * TB;
                            s.wi = d.macc + (size_t)ti * TB;
                            s.sq = d.sacc + (size_t)ti * TB;
                        }
                        trsm.push_back(s);
                    }
                }
                // fused tile tasks: the leaf's 16-row blocks of this step, eight to a task (the last factor rows and the test
                // rows share tasks)
                push_fused8_tasks(ftile8, blocks8, d, lf, k);
            }
            // The lower-blocks-only form of a diagonal tile takes ~0.6 of a full tile (tools/bench_tile_sym.py), the column-split
            // form of a short tile (padding rows below: tile_rows_body) its share of rows: together 3.4 % of the headline
            // model's executed flops (full diagonal tiles 1.5 %, the last row tile of every leaf and of its test rows 1.9 %).
            // Until round 5 both forms were kept for launches in which such tiles are a large share (1/5, 1/10: depth 4, -6 %
            // there) -- in launches dominated by full tiles a few shorter tasks measured +1.2 % (round 3, +0.4 % when
            // issued last).  With the schedule as it is now they pay everywhere: headline, same box, four alternating runs,
            // 0.3819 -> 0.3792 s with both (two lanes), 0.3981 -> 0.3951 with one lane; depth 4, configs 2 and 3, the shards of 4- and 8-rank
            // jobs unchanged (profiles/r05_sym_pad_ab.log).
            if (DSMGP_SYM_SHARE > 0 && nsym * DSMGP_SYM_SHARE < tiles.size())
                for (auto& u : tiles) u.sym = 0;
            {   // short tiles: the PAD instantiations of the kernels (step 0 has no update launch: its panel solves decide)
                size_t npad = 0, ntot = tiles.size();
                for (const auto& u : tiles) npad += (u.mrows != 0 && u.mrows <= 96) ? 1 : 0;
                if (tiles.empty()) {
                    ntot = trsm.size() - (size_t)S.trsm_off[k];
                    for (size_t q = (size_t)S.trsm_off[k]; q < trsm.size(); ++q) npad += (trsm[q].mrows != 0 && trsm[q].mrows <= 64) ? 1 : 0;
                }
                const bool share = DSMGP_PAD_SHARE == 0 || npad * (size_t)DSMGP_PAD_SHARE >= ntot;
                // big launches only: the PAD instantiation in EVERY launch leaves the headline, its shards and depth 4 where
                // they are, costs a single GP of 32 blocks 2 % (2.28 -> 2.33 ms) and gives the PoE of 128 experts 2.4 % (4.50 ->
                // 4.39 ms): profiles/r05_sym_pad_ab.log
                S.pad[k] = (npad > 0 && share && ntot >= (size_t)(2 * c->ncu)) ? 1 : 0;
            }
            // (The shorter tasks stay with the tiles of their leaf, whose B panel they share through L2: moved behind the whole
            // tiles of the launch the headline step measures 0.3853 -> 0.3883 s, four alternating runs: profiles/r05_sym_pad_ab.log.)
            U.add_step(tiles, k * TB);
            S.step_tiles[k] = (int)tiles.size();
            // Where the step's diagonal-block tasks sit in its update launch.  They are ~45 us each, and a slot that runs one
            // finishes its share of the launch that much later: in a launch that keeps every workgroup slot busy from start
            // to end (rounds of equal tiles) the launch ends that much later wherever they sit (measured: at the front the
            // update launches of the headline model and of an 8-rank shard grew by 37 and 30 us a step, the diagonal-block
            // launch they replace was 38).  So: a launch that leaves slots idle anyway (fewer tasks than the chip's 2 x CUs
            // workgroup slots) takes them first; a fuller one takes them LAST, behind its tail pieces, where they land in the
            // slots that drain first while the last pieces are still running.
            {
                const int nu_k = (int)U.upd.size() - S.upd_off[k], nd_k = (int)dfin.size() - S.dfin_off[k];
                if (nu_k + nd_k <= 2 * c->ncu) S.dpos[k] = 0;
                else if (DSMGP_DFIN_BACK < 0) S.dpos[k] = nu_k;
                else S.dpos[k] = std::max(0, (int)U.tail_begin - S.upd_off[k] - DSMGP_DFIN_BACK * c->ncu);
            }
            {   // panel solves of a leaf read the same Dinv_k: keep them on one XCD (small leaves have 3-4 of them per
                // step; dealt round-robin every one of them fetched the 128 KB block from HBM on its own: the depth-4
                // model's panel-solve launches fetched 2.6x their tile bytes)
                std::vector<int> unused(trsm.size());
                xcd_permute(trsm, unused, (size_t)S.trsm_off[k], trsm.size(), c->xcd_order);
                std::vector<int> unused3(ftile8.size());     // the fused tile tasks of a leaf share its B panel and L_kk
                xcd_permute(ftile8, unused3, (size_t)S.ftile8_off[k], ftile8.size(), c->xcd_order);
            }
        }
        S.upd_off[nsteps] = (int)U.upd.size();
        S.red_off[nsteps] = (int)U.red.size();
        S.trsm_off[nsteps] = (int)trsm.size();
        S.diag_off[nsteps] = (int)diag.size();
        S.fdiag_off[nsteps] = (int)fdiag.size();
        S.ftile8_off[nsteps] = (int)ftile8.size();
        S.dfin_off[nsteps] = (int)dfin.size();
        if (int rc = dev_upload(c, S.dfin, dfin)) return rc;
        if (int rc = dev_upload(c, S.trsm, trsm)) return rc;
        if (int rc = dev_upload(c, S.diag, diag)) return rc;
        if (int rc = dev_upload(c, S.fdiag, fdiag)) return rc;
        if (int rc = dev_upload(c, S.ftile8, ftile8)) return rc;
    }
    {
        const size_t slabs = std::max(split[0].max_slabs, split[1].max_slabs);
        arena_put(c, slab_ws);
        if (slabs) {
            if (slab_outside_pool && c->pool_base) HIPCHK(c, hipMalloc(&slab_ws, slabs * TB * TB * sizeof(double)));
            else if (int rc = arena_get(c, slab_ws, slabs * TB * TB)) return rc;
        }
        for (int ph = 0; ph < 2; ++ph) {
            split[ph].bind(slab_ws);
            if (int rc = dev_upload(c, phase[ph].upd, split[ph].upd)) return rc;
            if (int rc = dev_upload(c, phase[ph].red, split[ph].red)) return rc;
        }
        if (std::getenv("DSMGP_HOSTLOG")) {      // how many tiles of the factorisation pass through a split-K reduce before their panel solve
            size_t tiles = 0, panel = 0;
            for (int ph = 0; ph < 2; ++ph) {
                for (int v : phase[ph].step_tiles) tiles += (size_t)v;
                panel += (size_t)phase[ph].trsm_off[phase[ph].nsteps];
            }
            std::fprintf(stderr, "hostlog lane %d: %zu update tiles, %zu of them cut along K (reduce tasks), %zu panel-solve tiles\n", lane, tiles,
                         split[0].red.size() + split[1].red.size(), panel);
        }
    }
    // algorithmic flops of the launches timed as "update" (slot 1: tile_gemm_kernel_v2): 2 K per element of block column k with
    // K = 128 k, nothing where it runs fused.  Fused tile launches (slot 18: tile_fused8_kernel) are counted apart: their update flops
    // plus the triangular solve of the tiles below the diagonal block (c_k^2 per row, c_k = columns of block k).
    alg_flops_fused = 0.0;
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        if (c->leaf_lane[l] != lane) continue;
        const std::vector<char>& md = c->fused_step[(int)c->leaf_group[l]];
        auto mode = [&](int k) { return k < (int)md.size() ? (int)md[k] : (int)STEP_CLASSIC; };
        auto depth = [&](int k) { return mode(k) == STEP_FUSED ? 0 : k; };
        auto depth_fused = [&](int k) { return mode(k) == STEP_FUSED ? k : 0; };
        if (lf.owner == l) {
            alg_flops += update_flops(lf.n, lf.kb, depth);
            alg_flops_fused += update_flops(lf.n, lf.kb, depth_fused);
        }
        if (with_test && lf.nt > 0) {
            alg_flops += predict_update_flops(lf.n, lf.nt, depth);
            alg_flops_fused += predict_update_flops(lf.n, lf.nt, depth_fused);
        }
        for (int k = 0; k * TB < lf.n; ++k) {
            if (mode(k) != STEP_FUSED) continue;
            const double ck = (double)std::min(TB, lf.n - k * TB);
            if (lf.owner == l) alg_flops_fused += (double)std::max(0, lf.n - std::max(k + 1, lf.kb) * TB) * ck * ck;
            if (with_test && lf.nt > 0) alg_flops_fused += (double)lf.nt * ck * ck;
        }
    }
    return 0;
}

// Build arenas, the LeafDev table and every task list for the current leaf table + sharing schedule.
int build_plan(dsmgp_ctx* c) {
    HostLog hl_total("build_plan");
    {
        HostLog hl("build_plan: free_plan");
        free_plan(c);
    }
    const int L = c->L;
    if (L == 0) return fail(c, DSMGP_E_STATE, "no leaves set");
    size_t fTot = 0, dTot = 0, vTot = 0, xTot = 0;
    for (int l = 0; l < L; ++l) {
        LeafHost& lf = c->leaves[l];
        lf.owner = (lf.op == DSMGP_SHARE_COPY) ? lf.src : l;
        lf.kb = (lf.op == DSMGP_SHARE_PREFIX) ? (int)(lf.prefix / TB) : 0;
        if (lf.op == DSMGP_SHARE_PREFIX && lf.kb == 0) {
            lf.op = DSMGP_SHARE_FULL;   // nothing worth copying
            lf.src = -1;
        }
    }
    for (int l = 0; l < L; ++l) {
        LeafHost& lf = c->leaves[l];
        if (lf.owner == l) {
            lf.f_off = fTot;
            fTot += (size_t)lf.npad * lf.npad;
            lf.dinv_off = dTot;
            dTot += (size_t)lf.nb * TB * TB;
        }
        lf.vec_off = vTot;
        vTot += (size_t)4 * lf.npad;
        lf.xg_off = xTot;
        xTot += (size_t)lf.npad * c->D;
    }
    for (int l = 0; l < L; ++l) {
        LeafHost& lf = c->leaves[l];
        if (lf.owner != l) {
            lf.f_off = c->leaves[lf.owner].f_off;
            lf.dinv_off = c->leaves[lf.owner].dinv_off;
        }
    }
    c->bytes_needed = (fTot + dTot + vTot + xTot) * sizeof(double);
    size_t freeB = 0, totalB = 0;
    HIPCHK(c, hipMemGetInfo(&freeB, &totalB));
    if (!c->pool_base && c->bytes_needed + (size_t(1) << 30) > freeB)
        return fail(c, DSMGP_E_NOMEM, "leaf table needs " + std::to_string(c->bytes_needed >> 20) + " MiB, device has " +
                                          std::to_string(freeB >> 20) + " MiB free");
    {
        HostLog hl("build_plan: arenas");
        if (int rc = arena_get(c, c->arenaF, fTot)) return rc;
        if (int rc = arena_get(c, c->arenaDinv, dTot)) return rc;
        // the diagonal-block kernel writes the lower blocks of Dinv_k only: the blocks above the diagonal are zero from here on
        HIPCHK(c, hipMemsetAsync(c->arenaDinv, 0, std::max<size_t>(1, dTot) * sizeof(double), c->stream));
        if (int rc = arena_get(c, c->arenaVec, vTot)) return rc;
        if (int rc = arena_get(c, c->arenaXg, xTot)) return rc;
    }
    HIPCHK(c, hipMalloc(&c->d_info, L * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->d_owner, L * sizeof(int)));
    {   // the owner table changes with the leaf table / sharing schedule only: uploaded here, once
        std::vector<int> owner(L);
        for (int l = 0; l < L; ++l) owner[l] = c->leaves[l].owner;
        HIPCHK(c, hipMemcpy(c->d_owner, owner.data(), (size_t)L * sizeof(int), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc(&c->d_mll, L * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_leaves, L * sizeof(LeafDev)));

    c->h_leaves.assign(L, LeafDev{});
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        LeafDev& d = c->h_leaves[l];
        d.F = c->arenaF + lf.f_off;
        d.Dinv = c->arenaDinv + lf.dinv_off;
        d.Xg = c->arenaXg + lf.xg_off;
        d.yc = c->arenaVec + lf.vec_off;
        d.w = d.yc + lf.npad;
        d.z = d.w + lf.npad;
        d.alpha = d.z + lf.npad;
        d.info = c->d_info + lf.owner;
        d.mean = lf.mean;
        d.n = lf.n;
        d.npad = lf.npad;
        d.nb = lf.nb;
        d.kid = lf.kid;
    }
    HIPCHK(c, hipMemcpy(c->d_leaves, c->h_leaves.data(), L * sizeof(LeafDev), hipMemcpyHostToDevice));

    // gather X rows and centred y of every leaf (replaces the per-leaf views + apply_subtract!,
    // src/gaussianprocess.jl:72-74, src/treeStructure.jl:271-273)
    {
        int maxpad = 0;
        for (auto& lf : c->leaves) maxpad = std::max(maxpad, lf.npad);
        for (int l0 = 0; l0 < L; l0 += 32768) {
            const int cnt = std::min(32768, L - l0);
            dim3 grid((maxpad + 255) / 256, cnt);
            gather_leaf_kernel<<<grid, 256, 0, c->stream>>>(c->d_leaves, c->d_obs_ptr, c->d_obs_idx, c->dX, c->dy,
                                                            c->N, c->D, l0);
        }
        HIPCHK(c, hipGetLastError());
    }
    // The eight-wave fused tile tasks write the 16-row blocks of a factor that hold data and nothing else: the rows below them
    // in a leaf's last row tile -- padding, which the classic steps and every sweep over the factor expect to be zero -- are
    // zeroed here, once per plan (nothing writes anything else there afterwards)
    DevBuf<ZeroRowsTask> zrows;
    if (gram_fused(c)) {
        std::vector<ZeroRowsTask> zr;
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            const int last = lf.n - (lf.nb - 1) * TB;
            if (lf.owner != l || lf.nb < 2 || last >= TB) continue;
            ZeroRowsTask z{};
            z.p = c->h_leaves[l].F + (size_t)(lf.nb - 1) * TB;
            z.ld = lf.npad;
            z.r0 = (std::max(0, last) + 15) / 16 * 16;
            z.ncols = (lf.nb - 1) * TB;
            if (z.r0 < TB) zr.push_back(z);
        }
        if (int rc = dev_upload(c, zrows, zr)) return rc;
        if (!zr.empty()) zero_pad_rows_kernel<<<(int)zr.size(), 256, 0, c->stream>>>(zrows.p);
    }
    // the upper 16x16 blocks of every diagonal tile: zero once per plan, written by nobody afterwards (zero_upper_blocks_kernel)
    DevBuf<ZeroUpperTask> zupper;
    {
        std::vector<ZeroUpperTask> zu;
        int maxnb = 1;
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (lf.owner != l) continue;
            zu.push_back(ZeroUpperTask{c->h_leaves[l].F, lf.npad, lf.nb});
            maxnb = std::max(maxnb, lf.nb);
        }
        if (int rc = dev_upload(c, zupper, zu)) return rc;
        for (size_t b0 = 0; b0 < zu.size(); b0 += 32768) {
            const size_t cnt = std::min<size_t>(32768, zu.size() - b0);
            zero_upper_blocks_kernel<<<dim3((unsigned)cnt, (unsigned)std::min(maxnb, 64)), 256, 0, c->stream>>>(zupper.p + b0);
        }
        const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(c->stream);
        dev_free(zrows);          // before anything below can return
        dev_free(zupper);
        HIPCHK(c, e1);
        HIPCHK(c, e2);
    }

    // Phases: PREFIX leaves run after their sources (phase 1), everything else in phase 0 (COPY leaves ride with their source).
    // (Round 3 also built a split of the phase-0 leaves into two groups whose fused steps ran merged -- one launch carrying the
    // diagonal blocks of one group next to the tiles of the other, so that a CU mostly holds one latency-bound and one
    // pipe-bound workgroup: correct, and no faster -- depth 4 0.0601 s against 0.0586-0.0604 s -- so it is not in the tree.)
    c->leaf_group.assign(L, 0);
    for (int l = 0; l < L; ++l)
        if (c->leaves[l].op == DSMGP_SHARE_PREFIX) c->leaf_group[l] = 1;
    // Lanes: sharing groups (a source with its COPY and PREFIX leaves) dealt longest-processing-time first on n^3.
    {
        std::vector<int> unit(L);
        std::vector<double> cost(L, 0.0);
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            unit[l] = lf.op == DSMGP_SHARE_FULL ? l : lf.src;
            const double n = (double)lf.n;
            cost[unit[l]] += lf.op == DSMGP_SHARE_COPY ? n * n : n * n * n;
        }
        int nunits = 0;
        for (int l = 0; l < L; ++l) nunits += unit[l] == l ? 1 : 0;
        // (also under a reserved pool -- the streaming context -- since round 6: every lane's split-K workspace is one more
        // arena_get on the pool's stack, lane after lane; config 5 at full size 111.4 -> 110.3 s with two lanes in every group)
        c->nlanes = c->lanes_opt > 0 ? c->lanes_opt : (nunits >= LANES_AUTO_MIN_LEAVES ? 2 : 1);
        c->nlanes = std::max(1, std::min(c->nlanes, nunits));
        c->leaf_lane.assign(L, 0);
        if (c->nlanes > 1) {
            std::vector<int> order;
            for (int l = 0; l < L; ++l)
                if (unit[l] == l) order.push_back(l);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
            // (Weighing in the chain of a lane's largest leaf -- in the shard of an 8-rank job the lane that holds it ends at 52.3 ms,
            // the other, of equal flops and 77 steps instead of 103, at 47.5 -- by a cost per block step moves no end time that
            // matters: shards 0.0520 / 0.0513 / 0.0975 s without, 0.0519 / 0.0514 / 0.0971 with 0.13 ms a step, slower beyond;
            // tools/lane_timeline.py, profiles/r05_tail_ab.log.  What the early lane leaves behind runs on the whole chip.)
            double load[MAX_LANES] = {};
            std::vector<char> of_unit(L, 0);
            for (int u : order) {
                int lane = 0;
                for (int q = 1; q < c->nlanes; ++q)
                    if (load[q] < load[lane]) lane = q;
                of_unit[u] = (char)lane;
                load[lane] += cost[u];
            }
            for (int l = 0; l < L; ++l) c->leaf_lane[l] = of_unit[unit[l]];
        }
    }
    // How every block step runs (STEP_*).  Fused: the steps whose diagonal blocks alone fill the chip (or shallow ones).
    for (int ph = 0; ph < 2; ++ph) {
        int ns = 0;
        for (int l = 0; l < L; ++l)
            if (c->leaf_group[l] == ph) ns = std::max(ns, c->leaves[l].nb);
        c->fused_step[ph].assign(ns, STEP_CLASSIC);
        if (!gram_fused(c)) continue;
        for (int k = 0; k < ns; ++k) {
            int nd = 0;
            for (int l = 0; l < L; ++l) {
                const LeafHost& lf = c->leaves[l];
                if (c->leaf_group[l] == ph && lf.owner == l && lf.nb > k && k >= lf.kb) ++nd;
            }
            // (a depth limit on the first rule -- classic steps from block step 6 / 10 / 14 on even where the diagonal blocks fill
            // the chip -- gives the depth-3 model, whose fused steps reach K = 2304, 1.2 % at every limit and leaves depth 4
            // where it is: profiles/r05_tail_ab.log)
            // fused: the diagonal blocks alone fill the chip -- or the step is shallow (K <= 512: the one workgroup that
            // updates a diagonal tile before factorising it is done in a few microseconds; deeper, that update belongs in
            // the many-workgroup update launch, split along K)
            // (round 4: the shallow rule only where the step has leaves enough to give the two fused launches something to
            // do -- with the diagonal blocks one step ahead the classic steps of a handful of leaves are the shorter chain:
            // single GP 2.49 -> 2.25-2.31 ms, 8-rank shards 0.0560-0.0568 -> 0.0553-0.0560 s with no shallow step fused; the
            // headline model's 144 leaves and a PoE's 128 experts keep it)
            if (c->fuse_steps && (nd > c->ncu || (k <= FUSED_SHALLOW_STEPS && nd >= FUSED_SHALLOW_MIN_LEAVES))) c->fused_step[ph][k] = STEP_FUSED;
        }
    }
    // Gram tasks: lower tiles of every owner; with the Gram fused into the update tasks only the tiles that have none --
    // block column 0 (a PREFIX leaf: below the blocks it copies from its source; the copied blocks need no Gram at all),
    // and none at all where step 0 runs fused (its tasks start from the kernel function)
    std::vector<GramTask> gram;
    const bool fused = gram_fused(c);
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        if (lf.owner != l) continue;
        {
            const std::vector<char>& fs = c->fused_step[(int)c->leaf_group[l]];
            if (fused && !fs.empty() && fs[0] != STEP_CLASSIC) continue;
        }
        const LeafDev& d = c->h_leaves[l];
        for (int j = 0; j < (fused ? 1 : lf.nb); ++j)
            for (int i = (fused ? std::max(j, lf.kb) : j); i < lf.nb; ++i) {
                GramTask g{};
                g.xa = d.Xg + (size_t)i * TB;
                g.xb = d.Xg + (size_t)j * TB;
                g.out = d.F + (size_t)i * TB + (size_t)j * TB * lf.npad;
                g.lda = g.ldb = g.ldo = lf.npad;
                g.na = std::max(0, std::min(TB, lf.n - i * TB));
                g.nb = std::max(0, std::min(TB, lf.n - j * TB));
                g.sym = 1;
                g.diag = (i == j);
                g.kid = lf.kid;
                gram.push_back(g);
            }
    }
    if (int rc = dev_upload(c, c->gram, gram)) return rc;

    // Selective completions of Dinv_k inside fit! (the fused steps leave the 16x16 diagonal inverses only, kernels.hpp):
    {
        auto fused_in = [&](int ph, int k) { return k < (int)c->fused_step[ph].size() && c->fused_step[ph][k] == STEP_FUSED; };
        auto block_task = [&](int l, int k) {
            const LeafHost& lf = c->leaves[l];
            const LeafDev& d = c->h_leaves[l];
            DiagTask g{};
            g.T = d.F + (size_t)k * TB + (size_t)k * TB * lf.npad;
            g.Dinv = d.Dinv + (size_t)k * TB * TB;
            g.info = d.info;
            g.ld = lf.npad;
            g.nvalid = std::max(0, std::min(TB, lf.n - k * TB));
            g.row0 = k * TB;
            return g;
        };
        std::vector<DiagTask> pre, fw;
        std::vector<char> done(L, 0);
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (lf.op == DSMGP_SHARE_PREFIX)      // its copied blocks: produced by the source's phase-0 steps
                for (int k = 0; k < lf.kb; ++k)
                    if (fused_in(0, k)) pre.push_back(block_task(l, k));
            if (lf.op == DSMGP_SHARE_FULL) continue;
            const int o = lf.owner;               // COPY: the source's buffers; PREFIX: its own
            if (done[o]) continue;
            done[o] = 1;
            const LeafHost& lo = c->leaves[o];
            const int pho = (int)c->leaf_group[o];
            for (int k = lo.kb; k < lo.nb; ++k)
                if (fused_in(pho, k)) fw.push_back(block_task(o, k));
        }
        if (int rc = dev_upload(c, c->dinvc_prefix, pre)) return rc;
        if (int rc = dev_upload(c, c->dinvc_fwd, fw)) return rc;
        // ... and on first use after it (ensure_dinv): every block an owner's fused step factorises (the copied blocks of a
        // PREFIX leaf are in `pre`: completed inside every fit)
        std::vector<DiagTask> all;
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (lf.owner != l) continue;
            for (int k = lf.kb; k < lf.nb; ++k)
                if (fused_in((int)c->leaf_group[l], k)) all.push_back(block_task(l, k));
        }
        if (int rc = dev_upload(c, c->dinvc_all, all)) return rc;
    }
    // The step lists of the factorisation are built on first use: those for the train rows alone by the first fit! without a
    // resident test set (ensure_phase), those with the test rows riding along by dsmgp_set_test -- a context that only ever
    // fits with its test set resident (the streaming mode: 20 million tile tasks over the groups of config 5) builds one set.
    c->phase_ready = false;

    // solve sweeps.  Forward: only leaves whose factor came from elsewhere (COPY, PREFIX) -- leaves factorised
    // in full get z = L^-1 y from the factorisation itself (chol_diag_packed_kernel + the panel-solve epilogue).
    // Backward: every leaf.
    {
        int nsteps = 0;
        for (auto& lf : c->leaves) nsteps = std::max(nsteps, lf.nb);
        c->solve_steps = nsteps;
        std::vector<SolveTask> fwd, bwd;
        c->fwd_off.assign(nsteps + 1, 0);
        c->bwd_off.assign(nsteps + 1, 0);
        for (int k = 0; k < nsteps; ++k) {
            c->fwd_off[k] = (int)fwd.size();
            for (int l = 0; l < L; ++l) {
                const LeafHost& lf = c->leaves[l];
                if (lf.nb <= k) continue;
                if (lf.owner == l && lf.op == DSMGP_SHARE_FULL) continue;   // fused
                const LeafDev& d = c->h_leaves[l];
                for (int i = k; i < lf.nb; ++i) {
                    SolveTask s{};
                    s.Dk = d.Dinv + (size_t)k * TB * TB;
                    s.vk = d.w + (size_t)k * TB;
                    s.ldt = lf.npad;
                    if (i == k) {
                        s.self = 1;
                        s.out_k = d.z + (size_t)k * TB;
                    } else {
                        s.T = d.F + (size_t)i * TB + (size_t)k * TB * lf.npad;
                        s.vi = d.w + (size_t)i * TB;
                    }
                    fwd.push_back(s);
                }
            }
        }
        c->fwd_off[nsteps] = (int)fwd.size();
        // backward sweep on w = copy of z (z itself is kept: the predictive mean is m + (K_tn L^-T) z).
        // launch 0: alpha of every leaf's last block; launch s >= 1: block kb = nb-s updates the blocks left of
        // it with alpha_kb, and the task of block kb-1 finishes alpha_{kb-1} in the same workgroup.
        for (int s_ = 0; s_ < nsteps; ++s_) {
            c->bwd_off[s_] = (int)bwd.size();
            for (int l = 0; l < L; ++l) {
                const LeafHost& lf = c->leaves[l];
                const LeafDev& d = c->h_leaves[l];
                if (s_ == 0) {
                    SolveTask s{};
                    s.self = 1;
                    s.Dk = d.Dinv + (size_t)(lf.nb - 1) * TB * TB;
                    s.vk = d.w + (size_t)(lf.nb - 1) * TB;
                    s.out_k = d.alpha + (size_t)(lf.nb - 1) * TB;
                    s.ldt = lf.npad;
                    bwd.push_back(s);
                    continue;
                }
                const int kb = lf.nb - s_;
                if (kb < 1) continue;
                for (int j = 0; j < kb; ++j) {
                    SolveTask s{};
                    s.T = d.F + (size_t)kb * TB + (size_t)j * TB * lf.npad;
                    s.ldt = lf.npad;
                    s.vk = d.alpha + (size_t)kb * TB;
                    s.vi = d.w + (size_t)j * TB;
                    if (j == kb - 1) {
                        s.Dk = d.Dinv + (size_t)j * TB * TB;
                        s.out_k = d.alpha + (size_t)j * TB;
                    }
                    bwd.push_back(s);
                }
            }
        }
        c->bwd_off[nsteps] = (int)bwd.size();
        if (int rc = dev_upload(c, c->fwd, fwd)) return rc;
        if (int rc = dev_upload(c, c->bwd, bwd)) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->plan_ready = true;
    c->pool_mark_plan = c->pool_top;
    return 0;
}

// Step lists of a fit! without resident test rows.  Built after the plan, possibly after the test arenas: with a device pool
// its split-K workspace is allocated outside the pool (the pool is a stack: plan < test < gradients).
int ensure_phase(dsmgp_ctx* c) {
    if (c->phase_ready) return 0;
    HostLog hl("ensure_phase: factor steps");
    c->alg_flops_update = c->alg_flops_fused = 0.0;
    for (int lane = 0; lane < c->nlanes; ++lane) {
        double fu = 0.0, ff = 0.0;
        if (int rc = build_factor_steps(c, lane, false, c->phase[lane], c->slabF[lane], fu, ff, true)) return rc;
        c->alg_flops_update += fu;
        c->alg_flops_fused += ff;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->phase_ready = true;
    return 0;
}

// Step lists of a fit! with the registered test rows riding along: built on first use (see dsmgp_set_test).
int ensure_joint(dsmgp_ctx* c) {
    if (c->joint_ready) return 0;
    HostLog hl("ensure_joint: joint factor steps");
    c->alg_flops_joint = c->alg_flops_fused_joint = 0.0;
    for (int lane = 0; lane < c->nlanes; ++lane) {
        double fu = 0.0, ff = 0.0;
        if (int rc = build_factor_steps(c, lane, true, c->phaseJ[lane], c->slabJ[lane], fu, ff)) return rc;
        c->alg_flops_joint += fu;
        c->alg_flops_fused_joint += ff;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->joint_ready = true;
    return 0;
}

// dpos / dfin / ndfin: an update launch that carries its step's diagonal-block tasks (tile_gemm_kernel_v2's grid layout)
void launch_tiles(dsmgp_ctx* c, const TileTask* tasks, int n, int role = 0 /* 0 update, 1 panel solve */, bool pad = false,
                  int dpos = 0, const DiagFinishTask* dfin = nullptr, int ndfin = 0, hipStream_t st = nullptr) {
    const int g = n + ndfin;
    if (!st) st = c->stream;
    if (pad) {   // launches with many padding-row tiles (small leaves): waves without data rows stay off the matrix pipe
        if (role == 1) tile_trsm_kernel<true><<<n, 256, 0, st>>>(tasks);
        else if (c->profile == 0 || c->alt_names) tile_gemm_kernel_v2<false, 2, true><<<g, 256, 0, st>>>(tasks, nullptr, c->d_kp, c->D, dpos, dfin, ndfin);
        else tile_gemm_kernel_v2<false, 0, true><<<g, 256, 0, st>>>(tasks, nullptr, c->d_kp, c->D, dpos, dfin, ndfin);
        return;
    }
    // ROLE only names the instantiation: with per-launch timing switched off (dsmgp_set_profile(ctx, 0)) the same code
    // runs as <false, 2>, so that a profiler's per-kernel average of <false, 0> covers exactly the launches bench.py times
    if (role == 1) tile_trsm_kernel<false><<<n, 256, 0, st>>>(tasks);   // B = inverse of a diagonal block, K = 128
    else if (c->profile == 0 || c->alt_names) tile_gemm_kernel_v2<false, 2><<<g, 256, 0, st>>>(tasks, nullptr, c->d_kp, c->D, dpos, dfin, ndfin);
    else tile_gemm_kernel_v2<false, 0><<<g, 256, 0, st>>>(tasks, nullptr, c->d_kp, c->D, dpos, dfin, ndfin);
}

// Two events bracketing a call on the context's stream; destroyed on every exit path.
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t init() {
        hipError_t e = hipEventCreate(&a);
        return e != hipSuccess ? e : hipEventCreate(&b);
    }
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// hipEvent spans per kernel category.  The events come from a pool owned by the context (created once, reused by
// every fit/predict), so the timed region neither creates nor destroys events.
struct PhaseTimer {
    dsmgp_ctx* c;
    struct Span { int slot; int e0, e1; int step, tasks, red; };
    std::vector<Span> spans;
    size_t used = 0;
    explicit PhaseTimer(dsmgp_ctx* c_) : c(c_) {}
    bool on = false;
    int take() {
        if (used == c->event_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            c->event_pool.push_back(e);
        }
        return (int)used++;
    }
    void begin(int slot, hipStream_t st = nullptr) {
        on = c->profile >= 2 || (c->profile == 1 && (slot == 1 || slot == 18));
        if (!on) return;
        const int a = take(), b = take();
        if (a < 0 || b < 0) {
            on = false;
            return;
        }
        (void)hipEventRecord(c->event_pool[a], st ? st : c->stream);
        spans.push_back({slot, a, b, -1, 0, 0});
    }
    void note(int step, int ntasks, int nred) {
        if (on) {
            spans.back().step = step;
            spans.back().tasks = ntasks;
            spans.back().red = nred;
        }
    }
    void end(hipStream_t st = nullptr) {
        if (!on) return;
        (void)hipEventRecord(c->event_pool[spans.back().e1], st ? st : c->stream);
    }
    // ref: an event recorded before every span (the start of the call).  With two lanes the launches of a slot overlap in time:
    // next to the SUM of their durations (timings[slot]: what a profiler's per-kernel total is) the time during which ANY of
    // them ran -- the union of the intervals -- goes to timings[19] (update launches) and [20] (fused tile launches).
    void collect(hipEvent_t ref = nullptr) {
        const bool log = std::getenv("DSMGP_STEPLOG") != nullptr;
        std::vector<std::pair<float, float>> iv[2];
        for (const Span& s : spans) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c->event_pool[s.e0], c->event_pool[s.e1]);
            c->timings[s.slot] += ms * 1e-3;
            float a = -1.f;
            if (ref && (log || s.slot == 1 || s.slot == 18)) (void)hipEventElapsedTime(&a, ref, c->event_pool[s.e0]);
            if (ref && (s.slot == 1 || s.slot == 18)) iv[s.slot == 1 ? 0 : 1].push_back({a, a + ms});
            if (log && s.step >= 0)
                std::fprintf(stderr, "steplog slot %d step %d tasks %d tiles %d ms %.4f at %.4f\n", s.slot, s.step, s.tasks, s.red, ms, a);
        }
        for (int q = 0; q < 2; ++q) {
            std::sort(iv[q].begin(), iv[q].end());
            double tot = 0.0;
            float lo = 0.f, hi = -1.f;
            for (const auto& p : iv[q]) {
                if (hi < 0.f || p.first > hi) {
                    if (hi >= 0.f) tot += hi - lo;
                    lo = p.first;
                    hi = p.second;
                } else if (p.second > hi) {
                    hi = p.second;
                }
            }
            if (hi >= 0.f) tot += hi - lo;
            c->timings[19 + q] += tot * 1e-3;
        }
        spans.clear();
        used = 0;
    }
};

// One block step of one lane's factorisation phase, on that lane's stream.  Classic step: update (-> split-K reduce) -> diagonal
// block -> panel solve.  Fused step (many leaves, or shallow): diag_fused_reg_kernel -> tile_fused8_kernel.
void run_step(dsmgp_ctx* c, StepLists& S, int k, PhaseTimer& pt, hipStream_t st, bool count_launches) {
    if (k >= S.nsteps) return;
    const int nfd = S.fdiag_off[k + 1] - S.fdiag_off[k], nft8 = S.ftile8_off[k + 1] - S.ftile8_off[k];
    const int nu = S.upd_off[k + 1] - S.upd_off[k];
    if (nfd > 0 || nft8 > 0) {     // fused step: diagonal blocks (their tile's update included), then the tiles below them
        if (nfd > 0) {
            pt.begin(2, st);
            diag_fused_reg_kernel<<<nfd, 256, DIAGR_LDS_BYTES, st>>>(S.fdiag.p + S.fdiag_off[k], c->d_kp, c->D);
            pt.note(k, nfd, 0);
            pt.end(st);
        }
        if (nft8 > 0) {
            pt.begin(18, st);
            if (c->profile == 0 || c->alt_names) tile_fused8_kernel<2><<<nft8, 512, 0, st>>>(S.ftile8.p + S.ftile8_off[k], c->d_kp, c->D);
            else tile_fused8_kernel<0><<<nft8, 512, 0, st>>>(S.ftile8.p + S.ftile8_off[k], c->d_kp, c->D);
            pt.note(k, nft8, nft8);
            pt.end(st);
            if (count_launches) c->n_fused_launches++;
        }
        return;
    }
    const int ndf = S.dfin_off[k + 1] - S.dfin_off[k];
    if (nu > 0 || ndf > 0) {
        pt.begin(1, st);
        launch_tiles(c, S.upd.p + S.upd_off[k], nu, 0, S.pad[k] != 0, S.dpos[k], S.dfin.p + S.dfin_off[k], ndf, st);
        pt.note(k, nu, S.step_tiles[k]);
        pt.end(st);
        const int nr = S.red_off[k + 1] - S.red_off[k];
        if (nr > 0) {
            pt.begin(13, st);
            tile_reduce_kernel<<<nr * REDUCE_WGS, 256, 0, st>>>(S.red.p + S.red_off[k]);
            pt.end(st);
        }
        if (count_launches) c->n_update_launches++;
    }
    const int nd = S.diag_off[k + 1] - S.diag_off[k];
    if (nd > 0) {
        pt.begin(2, st);
        chol_diag_packed_kernel<<<nd, 256, DIAGP_LDS_BYTES, st>>>(S.diag.p + S.diag_off[k]);
        pt.note(k, nd, 0);
        pt.end(st);
    }
    const int ns = S.trsm_off[k + 1] - S.trsm_off[k];
    if (ns > 0) {
        pt.begin(3, st);
        launch_tiles(c, S.trsm.p + S.trsm_off[k], ns, 1, S.pad[k] != 0, 0, nullptr, 0, st);
        pt.note(k, ns, 0);
        pt.end(st);
    }
}

// Phase `ph` of every lane: the lanes' streams wait for what the context's stream has queued so far (fork), take their block
// steps -- enqueued step by step, lane after lane, so that both queues stay fed -- and the context's stream waits for them (join).
int run_lanes(dsmgp_ctx* c, StepLists (*lists)[2], int ph, PhaseTimer& pt, bool count_launches) {
    int nsteps = 0;
    for (int lane = 0; lane < c->nlanes; ++lane) nsteps = std::max(nsteps, lists[lane][ph].nsteps);
    if (nsteps == 0) return 0;
    if (c->nlanes > 1) {
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int lane = 1; lane < c->nlanes; ++lane) HIPCHK(c, hipStreamWaitEvent(c->lane_stream[lane], c->ev_fork, 0));
    }
    // (Starting the second lane half a step late -- after the first launch of the first lane's step 0, so that the latency-bound
    // launches of one lane meet the pipe-bound ones of the other from the start -- measured nothing: headline 0.3870 / 0.3865 /
    // 0.3928 plain, 0.3870 / 0.3878 / 0.3922 staggered; depth 4 0.0495 / 0.0500 / 0.0495 against 0.0501 / 0.0498 / 0.0500
    // (profiles/r05_lane_stagger_ab.log): the lanes drift apart by themselves within a few steps.)
    for (int k = 0; k < nsteps; ++k)
        for (int lane = 0; lane < c->nlanes; ++lane) run_step(c, lists[lane][ph], k, pt, c->lane_stream[lane], count_launches);
    HIPCHK(c, hipGetLastError());
    for (int lane = 1; lane < c->nlanes; ++lane) {
        HIPCHK(c, hipEventRecord(c->ev_join[lane], c->lane_stream[lane]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[lane], 0));
    }
    return 0;
}

// The whole inverse of every diagonal block (the fused steps of a fit leave L_kk and its 16x16 diagonal inverses only): what the
// standalone prediction sweep, the gradients and the sweeps for alpha multiply with.  Queued on the context's stream.
int ensure_dinv(dsmgp_ctx* c) {
    if (c->dinv_complete) return 0;
    if (c->dinvc_all.count)
        dinv_complete_kernel<<<(int)c->dinvc_all.count, 256, DIAGP_LDS_BYTES, c->stream>>>(c->dinvc_all.p);
    HIPCHK(c, hipGetLastError());
    c->dinv_complete = true;
    return 0;
}

// alpha = L^-T z by the backward block sweep on w = copy of z (z stays: the predictive mean is m + V^T z).
int ensure_alpha(dsmgp_ctx* c) {
    if (c->alpha_valid) return 0;
    if (int rc = ensure_dinv(c)) return rc;
    EventPair ev;
    HIPCHK(c, ev.init());
    HIPCHK(c, hipEventRecord(ev.a, c->stream));
    int maxpad = 0;
    for (auto& lf : c->leaves) maxpad = std::max(maxpad, lf.npad);
    copy_z_kernel<<<dim3((maxpad + 255) / 256, c->L), 256, 0, c->stream>>>(c->d_leaves);
    for (int s = 0; s < c->solve_steps; ++s) {
        const int n = c->bwd_off[s + 1] - c->bwd_off[s];
        if (n > 0) solve_bwd_kernel<<<n, 256, 0, c->stream>>>(c->bwd.p + c->bwd_off[s]);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(ev.b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev.a, ev.b));
    c->timings[14] = ms * 1e-3;
    c->alpha_valid = true;
    return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int dsmgp_create(int32_t device_id, dsmgp_ctx** out) {
    if (!out) return fail(nullptr, DSMGP_E_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, DSMGP_E_NODEVICE, "no HIP device visible: the GP-expert path has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, DSMGP_E_ARG, "device id out of range");
    dsmgp_ctx* c = new dsmgp_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, DSMGP_STREAM_FLAGS) != hipSuccess) {
        delete c;
        return fail(nullptr, DSMGP_E_HIP, "cannot initialise device");
    }
    c->lane_stream[0] = c->stream;
    bool lanes_ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int lane = 1; lane < MAX_LANES && lanes_ok; ++lane)
        lanes_ok = hipStreamCreateWithFlags(&c->lane_stream[lane], DSMGP_STREAM_FLAGS) == hipSuccess &&
                   hipEventCreateWithFlags(&c->ev_join[lane], hipEventDisableTiming) == hipSuccess;
    if (!lanes_ok) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, DSMGP_E_HIP, "cannot create the lanes' streams");
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_packed_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, DIAGP_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dinv_complete_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, DIAGP_LDS_BYTES);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
            c->ncu = prop.multiProcessorCount;
    }
#ifdef DSMGP_DIAG   // scheduling knobs of the diagnostic build (DSMGP_XCD, DSMGP_TAIL_SPLIT, DSMGP_TAIL_ROUNDS); the product build reads no tuning variables
    if (const char* s = std::getenv("DSMGP_XCD")) c->xcd_order = std::atoi(s) != 0;
    if (const char* s = std::getenv("DSMGP_TAIL_SPLIT")) c->tail_split = c->tail_split_lanes = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("DSMGP_TAIL_ROUNDS")) c->tail_rounds = c->tail_rounds_lanes = std::max(0, std::atoi(s));
    if (const char* s = std::getenv("DSMGP_RAGGED_ROUNDS")) c->ragged_rounds = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("DSMGP_RAGGED_DIV")) c->ragged_div = std::max(1, std::atoi(s));
#endif
    const char* p = std::getenv("DSMGP_PROFILE");
    c->profile = p ? std::atoi(p) * 2 : 0;   // DSMGP_PROFILE=1 -> every category
    *out = c;
    return 0;
}

int dsmgp_destroy(dsmgp_ctx* c) {
    if (!c) return DSMGP_E_ARG;
    (void)hipSetDevice(c->device);
    free_plan(c);
    free_test(c);
    if (c->pool_base) (void)hipFree(c->pool_base);
    c->pool_base = nullptr;
    dev_free(c->dX);
    dev_free(c->dy);
    dev_free(c->d_obs_ptr);
    dev_free(c->d_obs_idx);
    dev_free(c->d_kp);
    dev_free(c->d_l2);
    (void)dsmgp_comm_destroy(c);
    drop_graphs(c);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    dev_free(c->d_clock);
    free_tree(c);
    dev_free(c->rws_counts);
    dev_free(c->rws_bits);
    if (c->stage) (void)hipHostFree(c->stage);
    c->stage = nullptr;
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int lane = 1; lane < MAX_LANES; ++lane) {
        if (c->ev_join[lane]) (void)hipEventDestroy(c->ev_join[lane]);
        if (c->lane_stream[lane]) (void)hipStreamDestroy(c->lane_stream[lane]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

const char* dsmgp_last_error(dsmgp_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int dsmgp_device_name(dsmgp_ctx* c, char* buf, int32_t len) {
    if (!c || !buf || len <= 0) return DSMGP_E_ARG;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    std::snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int dsmgp_set_joint(dsmgp_ctx* c, int32_t on) {
    if (!c) return DSMGP_E_ARG;
    c->joint = on != 0;
    return 0;
}

int dsmgp_set_option(dsmgp_ctx* c, int32_t option, int32_t value) {
    if (!c) return DSMGP_E_ARG;
    if (option == DSMGP_OPT_ARD_LENGTHSCALE_GRADIENT) {
        if ((value != 0) != c->ard_true_gradient) {
            HIPCHK(c, hipSetDevice(c->device));
            free_grad(c);     // the contraction tile list depends on it
        }
        c->ard_true_gradient = value != 0;
        return 0;
    }
    if (option == DSMGP_OPT_FUSED_GRAM) {
        if ((value != 0) != c->fuse_gram) {
            HIPCHK(c, hipSetDevice(c->device));
            free_plan(c);     // task lists of fit! (and, through the plan, of the resident test set) depend on it
            free_test(c);
        }
        c->fuse_gram = value != 0;
        return 0;
    }
    if (option == DSMGP_OPT_FUSED_STEPS) {
        if ((value != 0) != c->fuse_steps) {
            HIPCHK(c, hipSetDevice(c->device));
            free_plan(c);
            free_test(c);
        }
        c->fuse_steps = value != 0;
        return 0;
    }
    if (option == DSMGP_OPT_FIT_GRAPH) {
        if (value == 0) drop_graphs(c);
        c->use_graph = value != 0;
        return 0;
    }
    if (option == DSMGP_OPT_LANES) {
        if (value < 0 || value > MAX_LANES) return fail(c, DSMGP_E_ARG, "set_option: lanes must be 0 (automatic) or 1 .. 4");
        if (value != c->lanes_opt) {
            HIPCHK(c, hipSetDevice(c->device));
            free_plan(c);
            free_test(c);
        }
        c->lanes_opt = value;
        return 0;
    }
    if (option == DSMGP_OPT_DIAG_IN_UPDATE) {
        if ((value != 0) != c->diag_in_update) {
            HIPCHK(c, hipSetDevice(c->device));
            free_plan(c);
            free_test(c);
        }
        c->diag_in_update = value != 0;
        return 0;
    }
    return fail(c, DSMGP_E_ARG, "set_option: unknown option");
}

int dsmgp_set_profile(dsmgp_ctx* c, int32_t on) {
    if (!c) return DSMGP_E_ARG;
    c->alt_names = on == 3;         // level 3 = level 1 under the kernel names of level 0 (bench.py's single-lane extra)
    c->profile = on < 0 ? 0 : (on == 3 ? 1 : (on > 2 ? 2 : on));
    return 0;
}

int dsmgp_set_train(dsmgp_ctx* c, const double* X, const double* y, int64_t N, int32_t D) {
    if (!c) return DSMGP_E_ARG;
    if (!X || !y || N <= 0 || D <= 0) return fail(c, DSMGP_E_ARG, "set_train: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    free_plan(c);
    free_test(c);
    free_tree(c);
    dev_free(c->dX);
    dev_free(c->dy);
    c->N = N;
    c->D = D;
    HIPCHK(c, hipMalloc(&c->dX, (size_t)N * D * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->dy, (size_t)N * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->dX, X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dy, y, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    c->L = 0;
    c->leaves.clear();
    return 0;
}

int dsmgp_set_leaves(dsmgp_ctx* c, int32_t L, const int64_t* obs_ptr, const int64_t* obs_idx,
                     const int32_t* kernel_id, const double* mean) {
    if (!c) return DSMGP_E_ARG;
    if (!c->dX) return fail(c, DSMGP_E_STATE, "set_leaves before set_train");
    if (L <= 0 || !obs_ptr || !obs_idx || !kernel_id || !mean) return fail(c, DSMGP_E_ARG, "set_leaves: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    free_plan(c);
    free_test(c);
    free_tree(c);           // its regions name indices of the OLD leaf table
    if (obs_ptr[0] != 0) return fail(c, DSMGP_E_ARG, "obs_ptr[0] must be 0");
    c->leaves.assign(L, LeafHost{});
    c->grad_active.clear();         // a new leaf table: gradients of every leaf again
    for (int l = 0; l < L; ++l) {
        const int64_t a = obs_ptr[l], b = obs_ptr[l + 1];
        if (b <= a) return fail(c, DSMGP_E_ARG, "leaf " + std::to_string(l) + " has no observations");
        if (b - a > (int64_t)1 << 20) return fail(c, DSMGP_E_ARG, "leaf too large");
        for (int64_t i = a; i < b; ++i) {
            if (obs_idx[i] < 0 || obs_idx[i] >= c->N) return fail(c, DSMGP_E_ARG, "observation index out of range");
            if (i > a && obs_idx[i] <= obs_idx[i - 1]) return fail(c, DSMGP_E_ARG, "obs lists must be strictly ascending");
        }
        if (kernel_id[l] < 0 || kernel_id[l] >= DSMGP_MAX_KERNEL_IDS) return fail(c, DSMGP_E_ARG, "kernel id out of range");
        LeafHost& lf = c->leaves[l];
        lf.n = (int)(b - a);
        lf.npad = round_up(lf.n, TB);
        lf.nb = lf.npad / TB;
        lf.kid = kernel_id[l];
        lf.mean = mean[l];
        lf.obs_off = a;
    }
    c->L = L;
    c->obs_ptr.assign(obs_ptr, obs_ptr + L + 1);
    c->obs_idx.assign(obs_idx, obs_idx + obs_ptr[L]);
    dev_free(c->d_obs_ptr);
    dev_free(c->d_obs_idx);
    HIPCHK(c, hipMalloc(&c->d_obs_ptr, (L + 1) * sizeof(int64_t)));
    HIPCHK(c, hipMalloc(&c->d_obs_idx, std::max<size_t>(1, c->obs_idx.size()) * sizeof(int64_t)));
    HIPCHK(c, hipMemcpy(c->d_obs_ptr, obs_ptr, (L + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_obs_idx, obs_idx, c->obs_idx.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    return 0;
}

int dsmgp_set_sharing(dsmgp_ctx* c, const int32_t* op, const int32_t* src, const int64_t* prefix_len) {
    if (!c) return DSMGP_E_ARG;
    if (c->L == 0) return fail(c, DSMGP_E_STATE, "set_sharing before set_leaves");
    const int L = c->L;
    // the whole schedule is validated into temporaries and committed at the end: a rejected schedule leaves the
    // context (leaf table, task lists of the previous schedule) exactly as it was
    struct Share { int op; int src; int64_t prefix; };
    std::vector<Share> sh(L, Share{DSMGP_SHARE_FULL, -1, 0});
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        const int o = op ? op[l] : DSMGP_SHARE_FULL;
        if (o == DSMGP_SHARE_FULL) continue;
        if (!src || src[l] < 0 || src[l] >= L || src[l] == l) return fail(c, DSMGP_E_ARG, "sharing: bad source leaf");
        const LeafHost& s = c->leaves[src[l]];
        const int so = op[src[l]];
        if (so != DSMGP_SHARE_FULL) return fail(c, DSMGP_E_ARG, "sharing: source leaf must be factorised in full");
        if (s.kid != lf.kid) return fail(c, DSMGP_E_ARG, "sharing: kernel ids differ");
        const int64_t* a = c->obs_idx.data() + lf.obs_off;
        const int64_t* b = c->obs_idx.data() + s.obs_off;
        if (o == DSMGP_SHARE_COPY) {
            if (s.n != lf.n || std::memcmp(a, b, sizeof(int64_t) * lf.n) != 0)
                return fail(c, DSMGP_E_ARG, "sharing: COPY needs identical observation lists");
            sh[l] = Share{o, src[l], 0};
        } else if (o == DSMGP_SHARE_PREFIX) {
            if (!prefix_len || prefix_len[l] != s.n || s.n >= lf.n || std::memcmp(a, b, sizeof(int64_t) * s.n) != 0)
                return fail(c, DSMGP_E_ARG, "sharing: PREFIX needs the source's list as a strict prefix");
            sh[l] = Share{o, src[l], prefix_len[l]};
        } else {
            return fail(c, DSMGP_E_ARG, "sharing: unknown op");
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    free_plan(c);
    free_test(c);     // the task lists of a resident test set point into the plan's arenas: they go with it
    for (int l = 0; l < L; ++l) {
        c->leaves[l].op = sh[l].op;
        c->leaves[l].src = sh[l].src;
        c->leaves[l].prefix = sh[l].prefix;
    }
    return 0;
}

int dsmgp_set_hyper(dsmgp_ctx* c, int32_t kernel_id, int32_t kind, const double* loghyp, int32_t n) {
    if (!c) return DSMGP_E_ARG;
    if (kernel_id < 0 || kernel_id >= DSMGP_MAX_KERNEL_IDS || !loghyp || n < 3) return fail(c, DSMGP_E_ARG, "set_hyper: bad arguments");
    if (kind < 0 || kind > 2) return fail(c, DSMGP_E_ARG, "set_hyper: unknown kernel kind");
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(loghyp[i])) return fail(c, DSMGP_E_ARG, "set_hyper: non-finite hyper-parameter");
    if ((int)c->hyper.size() <= kernel_id) c->hyper.resize(kernel_id + 1);
    if (c->hyper[kernel_id].kind != kind) free_grad(c);   // the contraction tiles depend on the kernel kind
    c->hyper[kernel_id].kind = kind;
    c->hyper[kernel_id].loghyp.assign(loghyp, loghyp + n);
    c->fitted = false;
    c->predicted = false;
    c->vt_valid = false;
    return 0;
}

int dsmgp_fit(dsmgp_ctx* c, double* mll_out, int32_t* info_out, double* seconds) {
    if (!c) return DSMGP_E_ARG;
    if (c->L == 0) return fail(c, DSMGP_E_STATE, "fit before set_leaves");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_hyper(c)) return rc;
    if (!c->plan_ready)
        if (int rc = build_plan(c)) return rc;
    if (int rc = upload_hyper(c)) return rc;
    // With a resident test set the rows of K_tn ride through the same launches (build_factor_steps).
    const bool joint = c->joint && c->test_ready;
    if (joint) {
        if (int rc = ensure_joint(c)) return rc;
    } else if (int rc = ensure_phase(c)) {
        return rc;
    }
    const int L = c->L;
    for (int i = 0; i < 6; ++i) c->timings[i] = 0.0;
    c->timings[11] = 0.0;
    c->timings[13] = 0.0;
    c->timings[14] = 0.0;
    c->timings[18] = c->timings[19] = c->timings[20] = 0.0;
    c->n_update_launches = 0;
    c->n_fused_launches = 0;
    PhaseTimer pt(c);
    EventPair ev;
    HIPCHK(c, ev.init());
    const hipEvent_t t0 = ev.a, t1 = ev.b;
    HIPCHK(c, hipEventRecord(t0, c->stream));

    StepLists (*phases)[2] = joint ? c->phaseJ : c->phase;
    auto enqueue = [&]() -> int {
        HIPCHK(c, hipMemsetAsync(c->d_info, 0, L * sizeof(int), c->stream));
        if (joint && c->acc_count) HIPCHK(c, hipMemsetAsync(c->arenaPV + c->acc_off, 0, c->acc_count * sizeof(double), c->stream));
        // 1. kernel matrices K + (noise + eps) I, lower tiles   (src/gaussianprocess.jl:83-98) [+ K_tn tiles]
        //    (fused into the update tasks: only the tiles of block column 0 are written here)
        {
            const DevBuf<GramTask>& tg = (joint && gram_fused(c)) ? c->pgram0 : c->pgram;
            pt.begin(0);
            if (c->gram.count) gram_tile_kernel<<<2 * (int)c->gram.count, 256, 0, c->stream>>>(c->gram.p, c->d_kp, c->D);
            if (joint && tg.count) gram_tile_kernel<<<2 * (int)tg.count, 256, 0, c->stream>>>(tg.p, c->d_kp, c->D);
            pt.end();
        }
        // 2. factorisation, full leaves first                    (src/gaussianprocess.jl:101); w = y - m rides along
        {
            int maxpad = 0;
            for (auto& lf : c->leaves) maxpad = std::max(maxpad, lf.npad);
            copy_vec_kernel<<<dim3((maxpad + 255) / 256, L), 256, 0, c->stream>>>(c->d_leaves);
        }
        if (int rc = run_lanes(c, phases, 0, pt, true)) return rc;
        // 3. prefix leaves: copy the leading blocks of the source factor, continue (src/fit.jl:276-278)
        bool any_prefix = false;
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (lf.op != DSMGP_SHARE_PREFIX) continue;
            any_prefix = true;
            const LeafDev& d = c->h_leaves[l];
            const LeafDev& s = c->h_leaves[lf.src];
            const size_t rows = (size_t)lf.kb * TB;
            HIPCHK(c, hipMemcpy2DAsync(d.F, (size_t)d.npad * sizeof(double), s.F, (size_t)s.npad * sizeof(double),
                                       rows * sizeof(double), rows, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d.Dinv, s.Dinv, rows * TB * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
        if (any_prefix) {
            if (c->dinvc_prefix.count)      // copied blocks that a fused step factorised: their whole inverse, for the classic steps below
                dinv_complete_kernel<<<(int)c->dinvc_prefix.count, 256, DIAGP_LDS_BYTES, c->stream>>>(c->dinvc_prefix.p);
            if (int rc = run_lanes(c, phases, 1, pt, true)) return rc;
        }
        // 4. z = L^-1 (y - m) for the leaves whose factor came from another leaf (COPY, PREFIX); leaves factorised in
        //    full produced z during the factorisation.  alpha = L^-T z (src/gaussianprocess.jl:105) is NOT computed here:
        //    neither the log-marginal (z.z) nor the prediction (V^T z) needs it -- ensure_alpha() runs the backward sweep
        //    on first use (gradients, dsmgp_download_factor).
        {
            pt.begin(4);
            if (c->dinvc_fwd.count)
                dinv_complete_kernel<<<(int)c->dinvc_fwd.count, 256, DIAGP_LDS_BYTES, c->stream>>>(c->dinvc_fwd.p);
            for (int k = 0; k < c->solve_steps; ++k) {
                const int n = c->fwd_off[k + 1] - c->fwd_off[k];
                if (n > 0) solve_fwd_kernel<<<n, 256, 0, c->stream>>>(c->fwd.p + c->fwd_off[k]);
            }
            pt.end();
        }
        // 5. log marginal likelihood                              (src/gaussianprocess.jl:163)
        pt.begin(5);
        mll_kernel<<<L, 256, 0, c->stream>>>(c->d_leaves, c->d_mll);
        pt.end();
        return 0;
    };
    c->vt_valid = false;
    c->alpha_valid = false;
    // Replay the sequence as a graph while nothing inside it records events (profile 0); capture it on first use
    const int gk = joint ? 1 : 0;
    if (c->use_graph && c->profile == 0) {
        if (!c->fit_graph[gk]) {
            hipGraph_t g = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue();
            const hipError_t ce = hipStreamEndCapture(c->stream, &g);
            if (rc != 0 || ce != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                c->use_graph = false;               // this runtime cannot capture the sequence: plain launches from here on
                if (rc != 0) return rc;
            } else {
                const hipError_t ie = hipGraphInstantiate(&c->fit_graph[gk], g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (ie != hipSuccess) {
                    c->fit_graph[gk] = nullptr;
                    (void)hipGetLastError();
                    c->use_graph = false;
                }
                c->graph_launches[gk][0] = c->n_update_launches;      // counted by the capture pass: the same on every replay
                c->graph_launches[gk][1] = c->n_fused_launches;
            }
        }
        if (c->fit_graph[gk]) {
            HIPCHK(c, hipGraphLaunch(c->fit_graph[gk], c->stream));
            c->n_update_launches = c->graph_launches[gk][0];
            c->n_fused_launches = c->graph_launches[gk][1];
        } else if (int rc = enqueue()) {
            return rc;
        }
    } else {
        if (int rc = enqueue()) return rc;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(t1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
    pt.collect(t0);
    c->timings[11] = ms * 1e-3;
    if (seconds) *seconds = ms * 1e-3;
    if (mll_out) HIPCHK(c, hipMemcpy(mll_out, c->d_mll, L * sizeof(double), hipMemcpyDeviceToHost));
    if (info_out) {
        // info lives per factor owner
        std::vector<int> owner_info(L);
        HIPCHK(c, hipMemcpy(owner_info.data(), c->d_info, L * sizeof(int), hipMemcpyDeviceToHost));
        for (int l = 0; l < L; ++l) info_out[l] = owner_info[c->leaves[l].owner];
    }
    c->fitted = true;
    c->predicted = false;
    c->vt_valid = joint;
    c->last_fit_joint = joint;
    c->dinv_complete = c->dinvc_all.count == 0;
    return 0;
}

// -------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
// Second half of a test-set registration, common to dsmgp_set_test (routes from the host) and dsmgp_set_test_routed (routes
// made on the device): per-leaf sizes from c->route_ptr, the K_tn arena, the gathered test rows, the task lists of the sweep.
// The first half has put the rows (dXt), the CSR (d_route_ptr / d_route_idx) and the per-row entry index (d_row_ptr / d_row_ent /
// d_ent_leaf) into HBM and the per-leaf row offsets into c->route_ptr.
int register_test(dsmgp_ctx* c, HostLog& hl) {
    const int L = c->L;
    const int64_t n_t = c->n_t, total = c->route_total;
    const int64_t* rp = c->route_ptr.data();
    hl.lap("set_test: sizes");
    size_t vTot = 0, xTot = 0;
    size_t accTot = 0;
    for (int l = 0; l < L; ++l) accTot += (size_t)round_up((int)(rp[l + 1] - rp[l]), TB);
    const size_t pTot = 2 * (size_t)total + 2 * accTot + 2 * TB;
    for (int l = 0; l < L; ++l) {
        LeafHost& lf = c->leaves[l];
        lf.nt = (int)(rp[l + 1] - rp[l]);
        lf.ntpad = round_up(lf.nt, TB);
        lf.route_off = rp[l];
        lf.vt_off = vTot;
        vTot += (size_t)lf.ntpad * lf.npad;
        lf.xt_off = xTot;
        xTot += (size_t)lf.ntpad * c->D;
        lf.pv_off = (size_t)rp[l];   // mu / var of all leaves are contiguous in route order (unpadded)
    }
    size_t freeB = 0, totalB = 0;
    HIPCHK(c, hipMemGetInfo(&freeB, &totalB));
    const size_t need = (vTot + xTot + pTot) * sizeof(double) + (size_t)n_t * c->D * sizeof(double);
    if (!c->pool_base) freeB += (c->arenaVt_count + c->cap_Xt + c->cap_PV) * sizeof(double);   // taken over or released first (arena_fit)
    if (!c->pool_base && need + (size_t(1) << 30) > freeB)
        return fail(c, DSMGP_E_NOMEM, "test set needs " + std::to_string(need >> 20) + " MiB, device has " +
                                          std::to_string(freeB >> 20) + " MiB free");
    hl.lap("set_test: arenas");
    // every arena of the test set this one replaces is taken over while it fits (and is not wastefully large: at most twice what
    // is needed); with a device pool they are carved out of the pool's stack
    auto arena_fit = [&](double*& p, size_t& cap, size_t need_) -> int {
        if (c->pool_base) return arena_get(c, p, need_);
        if (p && cap >= need_ && cap <= 2 * need_ + (size_t(1) << 20)) return 0;
        arena_put(c, p);
        cap = 0;
        const size_t want = need_ + need_ / 8;
        if (int rc = arena_get(c, p, want)) return rc;
        cap = want;
        return 0;
    };
    if (int rc = arena_fit(c->arenaVt, c->arenaVt_count, vTot)) return rc;
    if (int rc = arena_fit(c->arenaXt, c->cap_Xt, xTot)) return rc;
    if (int rc = arena_fit(c->arenaPV, c->cap_PV, pTot)) return rc;
    int maxpad = 0;
    size_t accOff = 0;
    c->acc_off = 2 * (size_t)total;
    c->acc_count = 2 * accTot;
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        LeafDev& d = c->h_leaves[l];
        d.Vt = c->arenaVt + lf.vt_off;
        d.Xtg = c->arenaXt + lf.xt_off;
        d.mu = c->arenaPV + lf.pv_off;
        d.var = c->arenaPV + (size_t)total + lf.pv_off;
        d.macc = c->arenaPV + 2 * (size_t)total + accOff;
        d.sacc = d.macc + accTot;
        d.zfused = (lf.owner == l && lf.op == DSMGP_SHARE_FULL) ? 1 : 0;
        accOff += (size_t)lf.ntpad;
        d.nt = lf.nt;
        d.ntpad = lf.ntpad;
        maxpad = std::max(maxpad, lf.ntpad);
    }
    if (int rc = stage_upload(c, c->d_leaves, c->h_leaves.data(), (size_t)L * sizeof(LeafDev))) return rc;
    if (maxpad > 0) {
        for (int l0 = 0; l0 < L; l0 += 32768) {
            const int cnt = std::min(32768, L - l0);
            gather_test_kernel<<<dim3((maxpad + 255) / 256, cnt), 256, 0, c->stream>>>(
                c->d_leaves, c->d_route_ptr, c->d_route_idx, c->dXt, n_t, c->D, l0);
        }
        HIPCHK(c, hipGetLastError());
    }
    hl.lap("set_test: task lists (host)");
    // task lists
    // In the block steps that a fit runs fused (many leaves, or shallow: build_plan) the sweep does too: one tile_fused8_kernel
    // launch whose tasks evaluate K_tn themselves, update and solve eight 16-row blocks of test rows each and write them once
    // -- the routed rows of a small leaf are a fraction of a 128-row tile (44 of 128 at depth 4), which the update / panel
    // solve launches of the other steps execute in full.  A COPY leaf goes with its source's group, as in the factorisation.
    auto fused_at = [&](int l, int k) {
        const std::vector<char>& fs = c->fused_step[(int)c->leaf_group[l]];
        return gram_fused(c) && k < (int)fs.size() && fs[(size_t)k] == STEP_FUSED;
    };
    std::vector<GramTask> pg, pg0;  // K_tn tiles the sweep reads from memory (block column 0 of the classic steps; every column where the
                                    // update tasks cannot evaluate them: D > 32); those of block column 0 for the joint fit
    std::vector<PredTask> ptk, ptk_slow;
    int nsteps = 0;
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        if (lf.nt == 0) continue;
        nsteps = std::max(nsteps, lf.nb);
        const LeafDev& d = c->h_leaves[l];
        for (int ti = 0; ti < lf.ntpad / TB; ++ti) {
            ptk.push_back(PredTask{l, ti * TB});
            if (!d.zfused) ptk_slow.push_back(PredTask{l, ti * TB});
            for (int j = 0; j < lf.nb; ++j) {
                if (fused_at(l, j)) continue;       // evaluated inside the fused tasks (sweep and joint fit alike)
                if (j > 0 && gram_fused(c)) continue;   // ... and inside the update tasks of the classic steps (TileTask.gram)
                GramTask g{};
                g.xa = d.Xtg + (size_t)ti * TB;
                g.xb = d.Xg + (size_t)j * TB;
                g.out = d.Vt + (size_t)ti * TB + (size_t)j * TB * lf.ntpad;
                g.lda = lf.ntpad;
                g.ldb = lf.npad;
                g.ldo = lf.ntpad;
                g.na = std::max(0, std::min(TB, lf.nt - ti * TB));
                g.nb = std::max(0, std::min(TB, lf.n - j * TB));
                g.sym = 0;
                g.diag = 0;
                g.kid = lf.kid;
                pg.push_back(g);
                if (j == 0) pg0.push_back(g);
            }
        }
    }
    c->psteps = nsteps;
    // The sweep runs lane by lane like the factorisation (dsmgp_ctx::nlanes): the lists of lane q's block step k are entry
    // q * nsteps + k of the offset tables, one lane's panel solves and reduces run under the other's update launches.
    const int nl = c->nlanes, nv = nl * nsteps;
    c->plan_lanes_test = nl;
    std::vector<UpdateSplitter> U((size_t)nl);
    for (UpdateSplitter& u : U) {
        u.ncu = c->ncu;
        u.xcd = c->xcd_order;
        u.tail_split = c->nlanes > 1 ? c->tail_split_lanes : c->tail_split;
        u.tail_rounds = c->nlanes > 1 ? c->tail_rounds_lanes : c->tail_rounds;
    }
    std::vector<std::vector<TileTask>> trsm_l((size_t)nl);
    std::vector<int> upd_loc((size_t)nv + 1, 0), red_loc((size_t)nv + 1, 0), trsm_loc((size_t)nv + 1, 0);   // offsets inside the lane's own lists
    c->pupd_off.assign(nv + 1, 0);
    c->pred_off.assign(nv + 1, 0);
    c->ptrsm_off.assign(nv + 1, 0);
    c->psweep8_off.assign(nv + 1, 0);
    // Fused steps: the tasks are made on the device (build_sweep8_kernel) from one SweepSeg per (leaf, step); the host counts them
    // per step -- a leaf of nt routed rows has ceil(ceil(nt / 16) / 8) tasks in each of its fused steps -- and says where each
    // pair's tasks start.  Classic steps: the leaves that have them, per step (few: the largest leaves of a many-leaf model,
    // all leaves beyond the shallow steps of a model with few).
    std::vector<SweepSeg> segs;
    std::vector<std::vector<int>> classic((size_t)nv);
    {
        std::vector<int> cnt8((size_t)nv, 0);
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (lf.nt == 0) continue;
            const int nt8 = ((lf.nt + 15) / 16 + 7) / 8;
            const size_t v0 = (size_t)c->leaf_lane[l] * (size_t)nsteps;
            for (int k = 0; k < lf.nb; ++k) {
                if (fused_at(l, k)) cnt8[v0 + (size_t)k] += nt8;
                else classic[v0 + (size_t)k].push_back(l);
            }
        }
        // (The K_tn arena is NOT cleared.  Rounds 1-4 zeroed it at every registration because the update / panel-solve tiles of
        // the classic steps read and write whole 128-row tiles while the Gram kernel writes data rows only -- but a row of a tile
        // product depends on its own row of the first operand alone, every rider sums per row, and nothing reads a row beyond a
        // leaf's routed ones (pred_finish / pred_mu / pred_var stop at nt): what sits in the padding rows never reaches a result.
        // Clearing cost more than its own time: 9.6 GB per registration at depth 4.)
        for (int v = 0; v < nv; ++v) c->psweep8_off[(size_t)v + 1] = c->psweep8_off[(size_t)v] + cnt8[(size_t)v];
        std::vector<int> cursor((size_t)nv, 0);
        segs.reserve((size_t)L * 4);
        for (int l = 0; l < L; ++l) {
            const LeafHost& lf = c->leaves[l];
            if (lf.nt == 0) continue;
            const int nt8 = ((lf.nt + 15) / 16 + 7) / 8;
            const size_t v0 = (size_t)c->leaf_lane[l] * (size_t)nsteps;
            for (int k = 0; k < lf.nb; ++k)
                if (fused_at(l, k)) {
                    SweepSeg sg{};
                    sg.leaf = l;
                    sg.k = k;
                    sg.src0 = cursor[v0 + (size_t)k];
                    sg.begin = c->psweep8_off[v0 + (size_t)k];
                    sg.n = cnt8[v0 + (size_t)k];
                    segs.push_back(sg);
                    cursor[v0 + (size_t)k] += nt8;
                }
        }
    }
    // (the lanes' lists are independent: one host thread per lane -- 90k tile tasks at the headline model, 10 ms on one thread)
    auto build_lane = [&](int lane) {
      for (int k = 0; k < nsteps; ++k) {
        const int v = lane * nsteps + k;
        UpdateSplitter& Ul = U[(size_t)lane];
        std::vector<TileTask>& trsm = trsm_l[(size_t)lane];
        upd_loc[(size_t)v] = (int)Ul.upd.size();
        red_loc[(size_t)v] = (int)Ul.red.size();
        trsm_loc[(size_t)v] = (int)trsm.size();
        std::vector<TileTask> tiles;
        for (int l : classic[(size_t)v]) {
            const LeafHost& lf = c->leaves[l];
            const LeafDev& d = c->h_leaves[l];
            for (int ti = 0; ti < lf.ntpad / TB; ++ti) {
                double* tile = d.Vt + (size_t)ti * TB + (size_t)k * TB * lf.ntpad;
                if (k > 0) {
                    TileTask u{};
                    u.A = d.Vt + (size_t)ti * TB;
                    u.B = d.F + (size_t)k * TB;
                    u.C = tile;
                    u.lda = lf.ntpad;
                    u.ldb = lf.npad;
                    u.ldc = lf.ntpad;
                    u.k0 = 0;
                    u.k1 = k * TB;
                    u.update = 1;
                    u.mrows = tile_mrows(lf.nt - ti * TB);
                    if (gram_fused(c)) {    // the task evaluates its K_tn tile itself, as in the joint fit: only block column 0 of
                        u.gram = 1;         // K_tn ever exists in memory
                        u.kid = lf.kid;
                        u.gxa = d.Xtg + (size_t)ti * TB;
                        u.gxb = d.Xg + (size_t)k * TB;
                        u.glda = lf.ntpad;
                        u.gldb = lf.npad;
                        u.gna = std::max(0, std::min(TB, lf.nt - ti * TB));
                        u.gnb = std::max(0, std::min(TB, lf.n - k * TB));
                    }
                    tiles.push_back(u);
                }
                TileTask s{};
                s.A = tile;
                s.B = d.Dinv + (size_t)k * TB * TB;
                s.C = tile;
                s.lda = lf.ntpad;
                s.ldb = TB;
                s.ldc = lf.ntpad;
                s.k0 = 0;
                s.k1 = TB;
                s.update = 0;
                s.mrows = tile_mrows(lf.nt - ti * TB);
                s.zk = d.z + (size_t)k * TB;           // predictive mean and variance ride along
                s.wi = d.macc + (size_t)ti * TB;
                s.sq = d.sacc + (size_t)ti * TB;
                trsm.push_back(s);
            }
        }
        Ul.add_step(tiles, k * TB);
      }
    };
    {
        // nothing may throw across the ABI, and nothing may leave a worker thread (std::terminate) or unwind past a joinable one:
        // every lane's build runs inside a catch that records the failure (bad_alloc from the 90k-task vectors of a large
        // registration), the workers are joined first, then the call returns DSMGP_E_NOMEM
        std::atomic<bool> lane_failed{false};
        auto build_lane_safe = [&](int q) {
            try {
                build_lane(q);
            } catch (...) {
                lane_failed.store(true);
            }
        };
        std::vector<std::thread> th;
        int started = 1;                    // lanes [1, started) have a thread of their own
        try {
            for (int q = 1; q < nl; ++q) {
                th.emplace_back(build_lane_safe, q);
                started = q + 1;
            }
        } catch (...) {                     // no thread to be had: the remaining lanes are built here
        }
        build_lane_safe(0);
        for (std::thread& t : th) t.join();
        for (int q = started; q < nl; ++q) build_lane_safe(q);
        if (lane_failed.load()) return fail(c, DSMGP_E_NOMEM, "set_test: out of host memory while building the sweep's task lists");
    }
    // one split-K workspace per lane, one list of each kind for all lanes (lane after lane)
    size_t slab_tot = 0;
    std::vector<size_t> slab_base((size_t)nl, 0);
    for (int q = 0; q < nl; ++q) {
        slab_base[(size_t)q] = slab_tot;
        slab_tot += U[(size_t)q].max_slabs;
    }
    hl.lap("set_test: task uploads");
    if (int rc = stage_upload_list(c, c->psegs, segs)) return rc;
    if (int rc = dev_reserve(c, c->psweep8, (size_t)c->psweep8_off[(size_t)nv])) return rc;
    if (!segs.empty())
        build_sweep8_kernel<<<(unsigned)((segs.size() + 127) / 128), 128, 0, c->stream>>>(c->psegs.p, (int)segs.size(), c->d_leaves, c->psweep8.p,
                                                                                        c->xcd_order ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    if (slab_tot) {
        if (c->pool_base) {
            if (int rc = arena_get(c, c->slabP, slab_tot * TB * TB)) return rc;
        } else if (!c->slabP || c->cap_slabP < slab_tot * TB * TB) {
            arena_put(c, c->slabP);
            if (int rc = arena_get(c, c->slabP, slab_tot * TB * TB)) return rc;
            c->cap_slabP = slab_tot * TB * TB;
        }
    }
    {   // the lanes' lists one behind the other in one device list of each kind, uploaded lane by lane
        size_t nu = 0, nr = 0, nt_ = 0;
        for (int q = 0; q < nl; ++q) {
            nu += U[(size_t)q].upd.size();
            nr += U[(size_t)q].red.size();
            nt_ += trsm_l[(size_t)q].size();
        }
        if (int rc = dev_reserve(c, c->pupd, nu)) return rc;
        if (int rc = dev_reserve(c, c->pred, nr)) return rc;
        if (int rc = dev_reserve(c, c->ptrsm, nt_)) return rc;
        int bu = 0, br = 0, bt = 0;
        for (int q = 0; q < nl; ++q) {
            UpdateSplitter& Uq = U[(size_t)q];
            Uq.bind(c->slabP + slab_base[(size_t)q] * TB * TB);
            for (int k = 0; k < nsteps; ++k) {
                const size_t v = (size_t)q * (size_t)nsteps + (size_t)k;
                c->pupd_off[v] = bu + upd_loc[v];
                c->pred_off[v] = br + red_loc[v];
                c->ptrsm_off[v] = bt + trsm_loc[v];
            }
            if (int rc = stage_upload(c, c->pupd.p + bu, Uq.upd.data(), Uq.upd.size() * sizeof(TileTask))) return rc;
            if (int rc = stage_upload(c, c->pred.p + br, Uq.red.data(), Uq.red.size() * sizeof(ReduceTask))) return rc;
            if (int rc = stage_upload(c, c->ptrsm.p + bt, trsm_l[(size_t)q].data(), trsm_l[(size_t)q].size() * sizeof(TileTask))) return rc;
            bu += (int)Uq.upd.size();
            br += (int)Uq.red.size();
            bt += (int)trsm_l[(size_t)q].size();
        }
        c->pupd_off[(size_t)nv] = bu;
        c->pred_off[(size_t)nv] = br;
        c->ptrsm_off[(size_t)nv] = bt;
    }
    if (int rc = stage_upload_list(c, c->pgram, pg)) return rc;
    if (int rc = stage_upload_list(c, c->pgram0, pg0)) return rc;
    if (int rc = stage_upload_list(c, c->ptasks, ptk)) return rc;
    if (int rc = stage_upload_list(c, c->ptasks_slow, ptk_slow)) return rc;
    // The same test rows as riders of the factorisation launches (used by fit while this test set is resident).  With a device
    // pool (the streaming context: the pool is a stack, plan < test < gradients, and a fit follows at once) the lists are built
    // here; otherwise by the first fit that wants them (ensure_joint) -- predict(model, x) on rows the model has not seen
    // registers them and runs its own sweep, and should not wait for task lists only a later fit! would use (0.067 s of the
    // 0.088 s this call took at the headline model).
    c->joint_ready = false;
    if (c->pool_base)
        if (int rc = ensure_joint(c)) return rc;
    hl.lap("set_test: final sync");
    if (int rc = stage_done(c)) return rc;
    c->test_ready = true;
    c->vt_valid = false;
    return 0;
}
}  // namespace

extern "C" {

int dsmgp_set_test(dsmgp_ctx* c, const double* Xt, int64_t n_t, int32_t D, const int64_t* route_ptr, const int64_t* route_idx) {
    if (!c) return DSMGP_E_ARG;
    if (c->L == 0) return fail(c, DSMGP_E_STATE, "set_test before set_leaves");
    if (!Xt || n_t <= 0 || !route_ptr) return fail(c, DSMGP_E_ARG, "set_test: bad arguments");
    if (D != c->D) return fail(c, DSMGP_E_ARG, "set_test: the test matrix has " + std::to_string(D) + " columns, the training data " + std::to_string(c->D));
    HIPCHK(c, hipSetDevice(c->device));
    HostLog hl_total("set_test");
    if (!c->plan_ready)
        if (int rc = build_plan(c)) return rc;
    HostLog hl("set_test: free old");
    free_test(c, true);     // every buffer of the set this one replaces is kept for it
    c->stage_top = 0;       // (the stream is idle: every entry point synchronises before it returns)
    hl.lap("set_test: validate");
    const int L = c->L;
    if (route_ptr[0] != 0) return fail(c, DSMGP_E_ARG, "route_ptr[0] must be 0");
    const int64_t total = route_ptr[L];
    if (total > 0 && !route_idx) return fail(c, DSMGP_E_ARG, "route_idx is NULL");
    for (int l = 0; l < L; ++l)
        if (route_ptr[l + 1] < route_ptr[l]) return fail(c, DSMGP_E_ARG, "route_ptr must be non-decreasing");
    for (int64_t i = 0; i < total; ++i)
        if (route_idx[i] < 0 || route_idx[i] >= n_t) return fail(c, DSMGP_E_ARG, "route index out of range");
    c->n_t = n_t;
    c->route_total = total;
    c->route_ptr.assign(route_ptr, route_ptr + L + 1);
    hl.lap("set_test: uploads + row index");
    if (int rc = dev_grow(c, c->dXt, c->cap_dXt, (size_t)n_t * c->D)) return rc;
    HIPCHK(c, hipMemcpy(c->dXt, Xt, (size_t)n_t * c->D * sizeof(double), hipMemcpyHostToDevice));
    if (int rc = dev_grow(c, c->d_route_ptr, c->cap_route_ptr, (size_t)L + 1)) return rc;
    HIPCHK(c, hipMemcpy(c->d_route_ptr, route_ptr, (L + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (int rc = dev_grow(c, c->d_route_idx, c->cap_route_idx, (size_t)total)) return rc;
    if (total) HIPCHK(c, hipMemcpy(c->d_route_idx, route_idx, total * sizeof(int64_t), hipMemcpyHostToDevice));
    {
        // per test row, the (leaf, row) entries that carry its moments, in ascending entry order (= leaf order):
        // the index agg_partial_kernel walks
        if (total > (int64_t)INT32_MAX) return fail(c, DSMGP_E_ARG, "set_test: more than 2^31 routed rows");
        std::vector<int64_t> rptr((size_t)n_t + 1, 0);
        for (int64_t i = 0; i < total; ++i) rptr[route_idx[i] + 1]++;
        for (int64_t r = 0; r < n_t; ++r) rptr[r + 1] += rptr[r];
        std::vector<int32_t> rent((size_t)std::max<int64_t>(1, total)), eleaf((size_t)std::max<int64_t>(1, total));
        std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
        for (int l = 0; l < L; ++l)
            for (int64_t i = route_ptr[l]; i < route_ptr[l + 1]; ++i) {
                rent[fill[route_idx[i]]++] = (int32_t)i;
                eleaf[i] = l;
            }
        if (int rc = dev_grow(c, c->d_row_ptr, c->cap_row_ptr, (size_t)n_t + 1)) return rc;
        if (int rc = dev_grow(c, c->d_row_ent, c->cap_row_ent, rent.size())) return rc;
        if (int rc = dev_grow(c, c->d_ent_leaf, c->cap_ent_leaf, eleaf.size())) return rc;
        HIPCHK(c, hipMemcpy(c->d_row_ptr, rptr.data(), rptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_row_ent, rent.data(), rent.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_ent_leaf, eleaf.data(), eleaf.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return register_test(c, hl);
}


// ---- routing on the device --------------------------------------------------------------------------------------------------
int dsmgp_set_tree(dsmgp_ctx* c, int64_t n_nodes, const int8_t* kind, const int64_t* first_child, const int64_t* n_child,
                   const int64_t* split_dim, const double* thr, int64_t thr_ld, const int64_t* leaf_id) {
    if (!c) return DSMGP_E_ARG;
    if (c->L == 0) return fail(c, DSMGP_E_STATE, "set_tree before set_leaves");
    if (n_nodes <= 0 || n_nodes > (int64_t)INT32_MAX || !kind || !first_child || !n_child || !split_dim || !thr || !leaf_id || thr_ld <= 0)
        return fail(c, DSMGP_E_ARG, "set_tree: bad arguments");
    std::vector<int32_t> first((size_t)n_nodes), nch((size_t)n_nodes), sdim((size_t)n_nodes), leaf((size_t)n_nodes), need((size_t)n_nodes, 0);
    int max_leaf = -1;
    for (int64_t i = 0; i < n_nodes; ++i) {
        if (kind[i] < 0 || kind[i] > 2) return fail(c, DSMGP_E_ARG, "set_tree: unknown node kind");
        if (kind[i] == 0) {
            if (leaf_id[i] < -1 || leaf_id[i] >= c->L) return fail(c, DSMGP_E_ARG, "set_tree: a region names a leaf outside the leaf table");
            max_leaf = std::max(max_leaf, (int)leaf_id[i]);
        } else if (n_child[i] <= 0 || first_child[i] <= i || first_child[i] + n_child[i] > n_nodes) {
            return fail(c, DSMGP_E_ARG, "set_tree: children must follow their parent, consecutively");
        }
        if (kind[i] == 1 && (n_child[i] > thr_ld || split_dim[i] < 0 || split_dim[i] >= c->D))
            return fail(c, DSMGP_E_ARG, "set_tree: a split node needs a threshold per child and a split dimension below D");
        first[(size_t)i] = (int32_t)first_child[i];
        nch[(size_t)i] = (int32_t)n_child[i];
        sdim[(size_t)i] = (int32_t)split_dim[i];
        leaf[(size_t)i] = kind[i] == 0 ? (int32_t)leaf_id[i] : -1;
    }
    // pending nodes of a row's depth-first walk (route_walk_row's stack)
    const int stack_need = route_stack_need(n_nodes, kind, first.data(), nch.data(), need.data());
    if (stack_need > ROUTE_STACK)
        return fail(c, DSMGP_E_ARG, "set_tree: the tree needs " + std::to_string(stack_need) + " pending nodes per row, the walk holds " +
                                        std::to_string(ROUTE_STACK));
    HIPCHK(c, hipSetDevice(c->device));
    free_tree(c);
    auto up = [&](auto*& dst, const auto* src, size_t n) -> int {
        using T = std::remove_cv_t<std::remove_pointer_t<std::remove_reference_t<decltype(dst)>>>;
        T* p = nullptr;
        HIPCHK(c, hipMalloc(&p, std::max<size_t>(1, n) * sizeof(T)));
        dst = p;
        HIPCHK(c, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    };
    if (int rc = up(c->rtree.kind, kind, (size_t)n_nodes)) return rc;
    if (int rc = up(c->rtree.first, first.data(), (size_t)n_nodes)) return rc;
    if (int rc = up(c->rtree.nchild, nch.data(), (size_t)n_nodes)) return rc;
    if (int rc = up(c->rtree.sdim, sdim.data(), (size_t)n_nodes)) return rc;
    if (int rc = up(c->rtree.leaf, leaf.data(), (size_t)n_nodes)) return rc;
    if (int rc = up(c->rtree.thr, thr, (size_t)n_nodes * (size_t)thr_ld)) return rc;
    c->rtree.thr_ld = (int)thr_ld;
    c->rtree_nodes = n_nodes;
    c->rtree_max_leaf = max_leaf;
    c->rtree_ready = true;
    return 0;
}

int dsmgp_set_test_routed(dsmgp_ctx* c, const double* Xt, int64_t n_t, int32_t D) {
    if (!c) return DSMGP_E_ARG;
    if (c->L == 0) return fail(c, DSMGP_E_STATE, "set_test before set_leaves");
    if (!c->rtree_ready) return fail(c, DSMGP_E_STATE, "set_test_routed before set_tree");
    if (!Xt || n_t <= 0) return fail(c, DSMGP_E_ARG, "set_test_routed: bad arguments");
    if (D != c->D) return fail(c, DSMGP_E_ARG, "set_test_routed: the test matrix has " + std::to_string(D) + " columns, the training data " + std::to_string(c->D));
    HIPCHK(c, hipSetDevice(c->device));
    HostLog hl_total("set_test_routed");
    if (!c->plan_ready)
        if (int rc = build_plan(c)) return rc;
    HostLog hl("set_test: free old");
    free_test(c, true);
    c->stage_top = 0;
    hl.lap("set_test: route on the device");
    const int L = c->L;
    const int64_t wpl = (n_t + 31) / 32;
    const size_t nbits = (size_t)L * (size_t)wpl;
    if (nbits * 8 > (size_t(8) << 30)) return fail(c, DSMGP_E_NOMEM, "set_test_routed: routing bitmap above 8 GiB; route on the host (dsmgp_set_test)");
    const size_t ncounts = (size_t)n_t + (size_t)L + 1;
    if (ncounts > c->rws_counts_cap) {
        dev_free(c->rws_counts);
        c->rws_counts_cap = ncounts + ncounts / 4;
        HIPCHK(c, hipMalloc(&c->rws_counts, c->rws_counts_cap * sizeof(int32_t)));
    }
    if (2 * nbits > c->rws_bits_cap) {
        dev_free(c->rws_bits);
        c->rws_bits_cap = 2 * nbits + nbits / 2;
        HIPCHK(c, hipMalloc(&c->rws_bits, c->rws_bits_cap * sizeof(uint32_t)));
    }
    int32_t* row_cnt = c->rws_counts;
    int32_t* leaf_cnt = row_cnt + n_t;
    int* outside = reinterpret_cast<int*>(leaf_cnt + L);
    uint32_t* bitmap = c->rws_bits;
    uint32_t* wprefix = bitmap + nbits;
    if (int rc = dev_grow(c, c->dXt, c->cap_dXt, (size_t)n_t * c->D)) return rc;
    if (int rc = stage_upload(c, c->dXt, Xt, (size_t)n_t * c->D * sizeof(double))) return rc;
    if (int rc = dev_grow(c, c->d_route_ptr, c->cap_route_ptr, (size_t)L + 1)) return rc;
    if (int rc = dev_grow(c, c->d_row_ptr, c->cap_row_ptr, (size_t)n_t + 1)) return rc;
    HIPCHK(c, hipMemsetAsync(bitmap, 0, nbits * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(outside, 0, sizeof(int), c->stream));
    const unsigned gr = (unsigned)((n_t + 255) / 256);
    route_walk_kernel<false><<<gr, 256, 0, c->stream>>>(c->rtree, c->dXt, n_t, row_cnt, bitmap, wpl, outside, nullptr, nullptr, nullptr, nullptr);
    scan_counts_kernel<<<1, 1024, 0, c->stream>>>(row_cnt, n_t, c->d_row_ptr);
    route_rank_kernel<<<L, 256, 0, c->stream>>>(bitmap, wpl, wprefix, leaf_cnt);
    scan_counts_kernel<<<1, 1024, 0, c->stream>>>(leaf_cnt, (int64_t)L, c->d_route_ptr);
    HIPCHK(c, hipGetLastError());
    c->route_ptr.assign((size_t)L + 1, 0);
    int flag = 0;
    HIPCHK(c, hipMemcpyAsync(c->route_ptr.data(), c->d_route_ptr, ((size_t)L + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&flag, outside, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (flag != 0) {
        free_test(c);
        return fail(c, DSMGP_E_DOMAIN, "set_test_routed: a test row lies outside the region of a split node (the reference loops forever there)");
    }
    const int64_t total = c->route_ptr[(size_t)L];
    if (total > (int64_t)INT32_MAX) {
        free_test(c);
        return fail(c, DSMGP_E_ARG, "set_test: more than 2^31 routed rows");
    }
    c->n_t = n_t;
    c->route_total = total;
    if (int rc = dev_grow(c, c->d_route_idx, c->cap_route_idx, (size_t)total)) return rc;
    if (int rc = dev_grow(c, c->d_row_ent, c->cap_row_ent, (size_t)total)) return rc;
    if (int rc = dev_grow(c, c->d_ent_leaf, c->cap_ent_leaf, (size_t)total)) return rc;
    route_fill_kernel<<<L, 256, 0, c->stream>>>(bitmap, wpl, wprefix, c->d_route_ptr, c->d_route_idx, c->d_ent_leaf);
    route_walk_kernel<true><<<gr, 256, 0, c->stream>>>(c->rtree, c->dXt, n_t, nullptr, bitmap, wpl, nullptr, c->d_row_ptr, c->d_route_ptr, wprefix,
                                                       c->d_row_ent);
    HIPCHK(c, hipGetLastError());
    return register_test(c, hl);
}

int dsmgp_routes(dsmgp_ctx* c, int64_t* route_ptr, int64_t* route_idx) {
    if (!c) return DSMGP_E_ARG;
    if (!c->test_ready) return fail(c, DSMGP_E_STATE, "routes before set_test");
    HIPCHK(c, hipSetDevice(c->device));
    if (route_ptr) std::memcpy(route_ptr, c->route_ptr.data(), ((size_t)c->L + 1) * sizeof(int64_t));
    if (route_idx && c->route_total)
        HIPCHK(c, hipMemcpy(route_idx, c->d_route_idx, (size_t)c->route_total * sizeof(int64_t), hipMemcpyDeviceToHost));
    return 0;
}

int dsmgp_predict_run(dsmgp_ctx* c, double* seconds) {
    if (!c) return DSMGP_E_ARG;
    if (!c->fitted) return fail(c, DSMGP_E_STATE, "predict before fit");
    if (!c->test_ready) return fail(c, DSMGP_E_STATE, "predict before set_test");
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 6; i < 10; ++i) c->timings[i] = 0.0;
    c->timings[12] = 0.0;
    HostLog hl("predict_run: events");
    PhaseTimer pt(c);
    EventPair ev;
    HIPCHK(c, ev.init());
    const hipEvent_t t0 = ev.a, t1 = ev.b;
    HIPCHK(c, hipEventRecord(t0, c->stream));
    hl.lap("predict_run: enqueue");
    if (c->ptasks.count) {
        const bool standalone = !c->vt_valid;
        if (standalone) {
            if (int rc = ensure_dinv(c)) return rc;       // the panel solves of the sweep multiply with Dinv_k
            HIPCHK(c, hipMemsetAsync(c->arenaPV + c->acc_off, 0, c->acc_count * sizeof(double), c->stream));
            // K_tn tiles of the classic steps                (src/gaussianprocess.jl:133)
            if (c->pgram.count) {
                pt.begin(6);
                gram_tile_kernel<<<2 * (int)c->pgram.count, 256, 0, c->stream>>>(c->pgram.p, c->d_kp, c->D);
                pt.end();
            }
            // V^T = K_tn L^-T, block column by block column (src/gaussianprocess.jl:120), lane by lane on the lanes' streams
            const int nl = c->plan_lanes_test;
            if (nl > 1) {
                HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
                for (int lane = 1; lane < nl; ++lane) HIPCHK(c, hipStreamWaitEvent(c->lane_stream[lane], c->ev_fork, 0));
            }
            for (int k = 0; k < c->psteps; ++k)
                for (int lane = 0; lane < nl; ++lane) {
                    const int v = lane * c->psteps + k;
                    hipStream_t st = c->lane_stream[lane];
                    const int n8 = c->psweep8_off[v + 1] - c->psweep8_off[v];
                    if (n8 > 0) {
                        pt.begin(7, st);
                        tile_fused8_kernel<1><<<n8, 512, 0, st>>>(c->psweep8.p + c->psweep8_off[v], c->d_kp, c->D);
                        pt.end(st);
                    }
                    const int nu = c->pupd_off[v + 1] - c->pupd_off[v];
                    if (nu > 0) {
                        pt.begin(7, st);
                        launch_tiles(c, c->pupd.p + c->pupd_off[v], nu, 0, false, 0, nullptr, 0, st);
                        const int nr = c->pred_off[v + 1] - c->pred_off[v];
                        if (nr > 0) tile_reduce_kernel<<<nr * REDUCE_WGS, 256, 0, st>>>(c->pred.p + c->pred_off[v]);
                        pt.end(st);
                    }
                    const int ns = c->ptrsm_off[v + 1] - c->ptrsm_off[v];
                    if (ns > 0) {
                        pt.begin(8, st);
                        launch_tiles(c, c->ptrsm.p + c->ptrsm_off[v], ns, 1, false, 0, nullptr, 0, st);
                        pt.end(st);
                    }
                }
            for (int lane = 1; lane < nl; ++lane) {
                HIPCHK(c, hipEventRecord(c->ev_join[lane], c->lane_stream[lane]));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[lane], 0));
            }
            c->vt_valid = true;
        }
        // mu = m + V^T z (= m + K_tn alpha), var = diag(Ktt - V'V) + noise   (src/gaussianprocess.jl:117-126):
        // both sums were accumulated by the panel-solve epilogues of the sweep; leaves whose z did not exist yet
        // while their rows rode through the factorisation (COPY / PREFIX) are finished from the stored rows
        pt.begin(9);
        pred_finish_kernel<<<(int)c->ptasks.count, 128, 0, c->stream>>>(c->d_leaves, c->ptasks.p, c->d_kp, c->D);
        if (!standalone && c->ptasks_slow.count) {
            pred_mu_kernel<<<(int)c->ptasks_slow.count, 256, 0, c->stream>>>(c->d_leaves, c->ptasks_slow.p);
            pred_var_kernel<<<(int)c->ptasks_slow.count, 256, 0, c->stream>>>(c->d_leaves, c->ptasks_slow.p, c->d_kp, c->D);
        }
        pt.end();
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(t1, c->stream));
    hl.lap("predict_run: wait");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hl.done();
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
    pt.collect();
    c->timings[12] = ms * 1e-3;
    if (seconds) *seconds = ms * 1e-3;
    c->predicted = true;
    c->agg_partial_ready = c->agg_done = c->agg_total = false;
    return 0;
}

int dsmgp_predict_fetch(dsmgp_ctx* c, double* mu_out, double* var_out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->predicted) return fail(c, DSMGP_E_STATE, "predict_fetch before predict_run");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->route_total;
    if (n && mu_out) HIPCHK(c, hipMemcpyAsync(mu_out, c->arenaPV, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (n && var_out) HIPCHK(c, hipMemcpyAsync(var_out, c->arenaPV + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int dsmgp_predict_leaves(dsmgp_ctx* c, const double* Xt, int64_t n_t, int32_t D, const int64_t* route_ptr,
                         const int64_t* route_idx, double* mu_out, double* var_out) {
    if (int rc = dsmgp_set_test(c, Xt, n_t, D, route_ptr, route_idx)) return rc;
    if (int rc = dsmgp_predict_run(c, nullptr)) return rc;
    return dsmgp_predict_fetch(c, mu_out, var_out);
}

// -------------------------------------------------------------------------------------------------
// predict(model, x): aggregation over the leaves of every test row, and the score functions, on the device
namespace {
int agg_width(int family, int G) { return family == AGG_MIXTURE ? 3 : (family == AGG_RBCM ? 2 * G : 2); }
}  // namespace

int dsmgp_aggregate_partial(dsmgp_ctx* c, int32_t family, const double* leaf_coef, const int32_t* leaf_group,
                            int32_t n_groups, double* partial_out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->predicted) return fail(c, DSMGP_E_STATE, "aggregate before predict_run");
    if (family < AGG_MIXTURE || family > AGG_RBCM) return fail(c, DSMGP_E_ARG, "aggregate: unknown family");
    if (family == AGG_RBCM) {
        if (!leaf_group || n_groups <= 0 || n_groups > 4096) return fail(c, DSMGP_E_ARG, "aggregate: rBCM needs leaf groups");
        for (int l = 0; l < c->L; ++l)
            if (leaf_group[l] < 0 || leaf_group[l] >= n_groups) return fail(c, DSMGP_E_ARG, "aggregate: leaf group out of range");
    } else if (!leaf_coef) {
        return fail(c, DSMGP_E_ARG, "aggregate: leaf_coef is NULL");
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int L = c->L;
    const int G = family == AGG_RBCM ? n_groups : 0;
    const int W = agg_width(family, G);
    const size_t need = (size_t)W * (size_t)c->n_t;
    if (int rc = dev_grow(c, c->d_agg_part, c->agg_part_cap, need)) return rc;
    if (int rc = dev_grow(c, c->d_agg_coef, c->cap_agg_coef, (size_t)L)) return rc;
    if (int rc = dev_grow(c, c->d_agg_group, c->cap_agg_group, (size_t)L)) return rc;
    if (leaf_coef) HIPCHK(c, hipMemcpyAsync(c->d_agg_coef, leaf_coef, (size_t)L * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (family == AGG_RBCM)
        HIPCHK(c, hipMemcpyAsync(c->d_agg_group, leaf_group, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    AggArgs a{};
    a.row_ptr = c->d_row_ptr;
    a.row_ent = c->d_row_ent;
    a.ent_leaf = c->d_ent_leaf;
    a.mu = c->arenaPV;
    a.var = c->arenaPV + (size_t)c->route_total;
    a.coef = c->d_agg_coef;
    a.group = c->d_agg_group;
    a.part = c->d_agg_part;
    a.n_t = c->n_t;
    a.family = family;
    a.G = G;
    agg_partial_kernel<<<(unsigned)((c->n_t + 255) / 256), 256, 0, c->stream>>>(a);
    HIPCHK(c, hipGetLastError());
    if (partial_out) HIPCHK(c, hipMemcpyAsync(partial_out, c->d_agg_part, need * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // leaf_coef / leaf_group are the caller's
    c->agg_family = family;
    c->agg_G = G;
    c->agg_W = W;
    c->agg_partial_ready = true;
    c->agg_total = false;
    c->agg_done = false;
    return 0;
}

int dsmgp_aggregate_finish(dsmgp_ctx* c, const double* partial_in, int32_t plain, int32_t prior_kernel_id,
                           double* mu_out, double* var_out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->agg_partial_ready) return fail(c, DSMGP_E_STATE, "aggregate_finish before aggregate_partial");
    if (c->agg_family == AGG_RBCM &&
        (prior_kernel_id < 0 || prior_kernel_id >= (int)c->hyper.size() || c->hyper[prior_kernel_id].kind < 0))
        return fail(c, DSMGP_E_ARG, "aggregate_finish: rBCM needs the kernel id of the model's first leaf");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nt = (size_t)c->n_t;
    const size_t nblk = (nt + 255) / 256;
    if (int rc = dev_grow(c, c->d_agg_out, c->cap_agg_out, 3 * nt + 3 * nblk + 8)) return rc;
    if (partial_in)   // sums over all ranks / contexts, added by the caller
        HIPCHK(c, hipMemcpyAsync(c->d_agg_part, partial_in, (size_t)c->agg_W * nt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    agg_finish_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(c->d_agg_part, c->n_t, c->agg_family, c->agg_G, plain ? 1 : 0,
                                                           c->d_kp, c->agg_family == AGG_RBCM ? prior_kernel_id : 0, c->dXt,
                                                           c->D, c->d_agg_out, c->d_agg_out + nt);
    HIPCHK(c, hipGetLastError());
    if (mu_out) HIPCHK(c, hipMemcpyAsync(mu_out, c->d_agg_out, nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (var_out) HIPCHK(c, hipMemcpyAsync(var_out, c->d_agg_out + nt, nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->agg_done = true;
    return 0;
}

int dsmgp_aggregate(dsmgp_ctx* c, int32_t family, const double* leaf_coef, const int32_t* leaf_group, int32_t n_groups,
                    int32_t plain, int32_t prior_kernel_id, double* mu_out, double* var_out) {
    if (int rc = dsmgp_aggregate_partial(c, family, leaf_coef, leaf_group, n_groups, nullptr)) return rc;
    return dsmgp_aggregate_finish(c, nullptr, plain, prior_kernel_id, mu_out, var_out);
}

int dsmgp_scores(dsmgp_ctx* c, const double* y_test, double* out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->agg_done) return fail(c, DSMGP_E_STATE, "scores before aggregate");
    if (!y_test || !out) return fail(c, DSMGP_E_ARG, "scores: NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nt = (size_t)c->n_t;
    const size_t nblk = (nt + 255) / 256;
    double* dy = c->d_agg_out + 2 * nt;
    double* dsum = dy + nt;
    HIPCHK(c, hipMemcpyAsync(dy, y_test, nt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    std::vector<double> blk(3 * nblk);
    auto pass = [&](int which, double mse, double mae, double (&tot)[3]) -> int {
        agg_scores_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(dy, c->d_agg_out, c->d_agg_out + nt, c->n_t, which, mse, mae, dsum);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(blk.data(), dsum, blk.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        tot[0] = tot[1] = tot[2] = 0.0;
        for (size_t b = 0; b < nblk; ++b)
            for (int k = 0; k < 3; ++k) tot[k] += blk[3 * b + k];
        return 0;
    };
    double t0[3], t1[3];
    if (int rc = pass(0, 0.0, 0.0, t0)) return rc;
    const double n = (double)nt;
    const double mse = t0[0] / n, mae = t0[1] / n;
    if (int rc = pass(1, mse, mae, t1)) return rc;
    out[0] = mse;                                                        // mse  src/scorefunctions.jl:8
    out[1] = nt > 1 ? std::sqrt(t1[0] / (n - 1.0)) / std::sqrt(n) : NAN; // sse  :9  (std = unbiased)
    out[2] = mae;                                                        // mae  :13
    out[3] = nt > 1 ? std::sqrt(t1[1] / (n - 1.0)) / std::sqrt(n) : NAN; // sae  :14
    out[4] = t0[2] / n;                                                  // nlpd :16
    return 0;
}

namespace {

// Task lists of the gradient pass: Xt = L^-T by a blocked triangular inversion on the same tile kernels
// (row tile t of Xt is e_t^T L^-T: zero left of block t, Dinv_t^T on the diagonal, then a left-looking sweep
// whose K range starts at column 128 t), then the contraction tiles of tile_graddot_kernel.
int build_grad_plan(dsmgp_ctx* c) {
    const int L = c->L;
    size_t xTot = 0;
    std::vector<size_t> xoff(L, 0);
    for (int l = 0; l < L; ++l)
        if (c->leaves[l].owner == l) {
            xoff[l] = xTot;
            xTot += (size_t)c->leaves[l].npad * c->leaves[l].npad;
        }
    size_t freeB = 0, totalB = 0;
    HIPCHK(c, hipMemGetInfo(&freeB, &totalB));
    if (!c->pool_base && xTot * sizeof(double) + (size_t(2) << 30) > freeB)
        return fail(c, DSMGP_E_NOMEM, "gradients need " + std::to_string((xTot * 8) >> 20) + " MiB for L^-1, device has " +
                                          std::to_string(freeB >> 20) + " MiB free");
    if (!c->arenaX || c->arenaX_count != xTot) {      // kept across changes of the active set (dsmgp_set_gradient_leaves)
        arena_put(c, c->arenaX);
        if (int rc = arena_get(c, c->arenaX, xTot)) return rc;
        c->arenaX_count = xTot;
    }
    auto Xt = [&](int l) { return c->arenaX + xoff[c->leaves[l].owner]; };
    // Active leaves (dsmgp_set_gradient_leaves; default all).  A leaf's gradient needs tr K_y^-1 = |L^-1|_F^2 of its factor
    // owner and, for the kernels with a length-scale term, the contraction tiles -- its own, or its source's where a COPY leaf
    // shares them (grad_src): needC = leaves whose contraction is computed, needX = owners whose L^-T is built.
    auto active = [&](int l) { return c->grad_active.empty() || c->grad_active[l] != 0; };
    c->grad_src.assign(L, -1);
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        if (lf.op == DSMGP_SHARE_COPY && lf.mean == c->leaves[lf.src].mean) c->grad_src[l] = lf.src;
    }
    std::vector<char> needC(L, 0), needX(L, 0);
    for (int l = 0; l < L; ++l)
        if (active(l)) {
            needC[c->grad_src[l] >= 0 ? c->grad_src[l] : l] = 1;
            needX[c->leaves[l].owner] = 1;
        }
    for (int l = 0; l < L; ++l)
        if (needC[l]) needX[c->leaves[l].owner] = 1;

    std::vector<TransTask> trans;
    std::vector<FrobTask> frob;
    c->gfrob_leaf.clear();
    int nsteps = 0;
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        if (lf.owner != l || !needX[l]) continue;
        nsteps = std::max(nsteps, lf.nb);
        const LeafDev& d = c->h_leaves[l];
        for (int t = 0; t < lf.nb; ++t) {
            TransTask tt{};
            tt.src = d.Dinv + (size_t)t * TB * TB;
            tt.dst = Xt(l) + (size_t)t * TB + (size_t)t * TB * lf.npad;
            tt.ldd = lf.npad;
            trans.push_back(tt);
            FrobTask f{};
            f.X = Xt(l) + (size_t)t * TB;
            f.ld = lf.npad;
            f.col0 = t * TB;
            f.col1 = lf.npad;
            f.nrows = std::max(0, std::min(TB, lf.n - t * TB));
            f.n = lf.n;
            frob.push_back(f);
            c->gfrob_leaf.push_back(l);
        }
    }
    c->gsteps = nsteps;
    // The inversion runs lane by lane like the factorisation (dsmgp_ctx::nlanes; a leaf inverts in the lane that factorised it):
    // the lists of lane q's block step k are entry q * nsteps + k of the offset tables, one lane's panel solves and reduces run
    // under the other's update launches.
    const int nl = std::max(1, c->nlanes);
    c->glanes = nl;
    const int nv = nl * nsteps;
    std::vector<UpdateSplitter> U((size_t)nl);
    for (UpdateSplitter& u : U) {
        u.ncu = c->ncu;
        u.xcd = c->xcd_order;
        u.tail_split = nl > 1 ? c->tail_split_lanes : c->tail_split;
        u.tail_rounds = nl > 1 ? c->tail_rounds_lanes : c->tail_rounds;
        u.ragged_rounds = c->ragged_rounds;      // (0 / 1 / 2 rounds cut under lanes: no difference, profiles/r05_grad_lanes_ab.log)
        u.ragged_div = c->ragged_div;
    }
    std::vector<std::vector<TileTask>> trsm_l((size_t)nl);
    std::vector<int> upd_loc((size_t)nv + 1, 0), red_loc((size_t)nv + 1, 0), trsm_loc((size_t)nv + 1, 0);   // inside the lane's own lists
    for (int q = 0; q < nl; ++q) {
        std::vector<TileTask>& trsm = trsm_l[q];
        for (int k = 1; k < nsteps; ++k) {
            const int v = q * nsteps + k;
            upd_loc[v] = (int)U[q].upd.size();
            red_loc[v] = (int)U[q].red.size();
            trsm_loc[v] = (int)trsm.size();
            std::vector<TileTask> tiles;
            double depth = 0.0;
            for (int l = 0; l < L; ++l) {
                const LeafHost& lf = c->leaves[l];
                if (lf.owner != l || lf.nb <= k || !needX[l] || (nl > 1 && c->leaf_lane[l] != q)) continue;
                const LeafDev& d = c->h_leaves[l];
                for (int t = 0; t < k; ++t) {
                    double* tile = Xt(l) + (size_t)t * TB + (size_t)k * TB * lf.npad;
                    TileTask u{};
                    u.A = Xt(l) + (size_t)t * TB;
                    u.B = d.F + (size_t)k * TB;
                    u.C = tile;
                    u.lda = u.ldb = u.ldc = lf.npad;
                    u.k0 = t * TB;
                    u.k1 = k * TB;
                    u.update = 2;            // the block is defined here: -product, nothing to read (Xt needs no zero fill:
                    u.rev = 1;               //   every later task reads row tile t from column 128 t on only)
                    tiles.push_back(u);
                    depth += u.k1 - u.k0;
                    TileTask s{};
                    s.A = tile;
                    s.B = d.Dinv + (size_t)k * TB * TB;
                    s.C = tile;
                    s.lda = lf.npad;
                    s.ldb = TB;
                    s.ldc = lf.npad;
                    s.k0 = 0;
                    s.k1 = TB;
                    s.update = 0;
                    trsm.push_back(s);
                }
            }
            const int Kavg = tiles.empty() ? 0 : (int)(depth / tiles.size()) / TB * TB;
            U[q].add_step_ragged(tiles, std::max(TB, Kavg), k);
        }
    }
    // one list of each kind for all lanes: lane q's tasks behind those of the lanes before it, its slabs behind theirs
    c->gupd_off.assign((size_t)nv + 1, 0);
    c->gred_off.assign((size_t)nv + 1, 0);
    c->gtrsm_off.assign((size_t)nv + 1, 0);
    std::vector<TileTask> upd_all, trsm_all;
    std::vector<ReduceTask> red_all;
    size_t slabs_total = 0;
    std::vector<size_t> slab_base((size_t)nl, 0);
    for (int q = 0; q < nl; ++q) {
        slab_base[q] = slabs_total;
        slabs_total += U[q].max_slabs;
    }
    if (slabs_total * TB * TB > c->slabG_count) {
        arena_put(c, c->slabG);
        if (int rc = arena_get(c, c->slabG, slabs_total * TB * TB)) return rc;
        c->slabG_count = slabs_total * TB * TB;
    }
    for (int q = 0; q < nl; ++q) {
        U[q].bind(c->slabG + slab_base[q] * TB * TB);
        const int ub = (int)upd_all.size(), rb = (int)red_all.size(), tb = (int)trsm_all.size();
        for (int k = 0; k < nsteps; ++k) {
            const int v = q * nsteps + k;
            c->gupd_off[v] = ub + (k >= 1 ? upd_loc[v] : 0);
            c->gred_off[v] = rb + (k >= 1 ? red_loc[v] : 0);
            c->gtrsm_off[v] = tb + (k >= 1 ? trsm_loc[v] : 0);
        }
        upd_all.insert(upd_all.end(), U[q].upd.begin(), U[q].upd.end());
        red_all.insert(red_all.end(), U[q].red.begin(), U[q].red.end());
        trsm_all.insert(trsm_all.end(), trsm_l[q].begin(), trsm_l[q].end());
    }
    c->gupd_off[nv] = (int)upd_all.size();
    c->gred_off[nv] = (int)red_all.size();
    c->gtrsm_off[nv] = (int)trsm_all.size();

    // contraction tiles: every IsoSE leaf (COPY leaves too: their alpha is their own)
    // Order.  A 128x128 tile task moves 2 x 128 x K operand doubles for 2 x 128^2 x K flops: 8 flop/B, below the
    // ridge of the chip unless operands are shared through L2.  Tasks that are adjacent in this list run at the same
    // time on one XCD (xcd_permute below), so the lower tiles of a leaf are listed in super-tiles of GS x GS tiles:
    // the GS^2 tasks of a super-tile read GS row panels and GS column panels of L^-T between them (measured: tiles
    // listed row by row 55 TFLOP/s; sorted by depth across leaves, i.e. no sharing at all, 26).
    constexpr int GS = 4;
    std::vector<GradTask> gd;
    std::vector<size_t> gblock;
    c->gdot_leaf.clear();
    bool any_ard = false;
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        const int kind_l = c->hyper[lf.kid].kind;
        const bool ard = kind_l == DSMGP_KIND_ARD_SE && c->ard_true_gradient;
        if (kind_l != DSMGP_KIND_ISO_SE && !ard) continue;
        any_ard = any_ard || ard;
        // Shared gradients (the idea of src/fit.jl:313-395: a leaf whose observation set equals its main leaf's takes
        // that leaf's gradients, `copygradients`): a COPY leaf has its source's factor and kernel id; with the same
        // ConstMean its alpha is the source's too, so its contraction is the source's and is not computed again.
        if (c->grad_src[l] >= 0 || !needC[l]) continue;
        const LeafDev& d = c->h_leaves[l];
        for (int ib = 0; ib < lf.nb; ib += GS) {
            gblock.push_back(gd.size());           // one block per (leaf, GS tile rows): these tasks share their A panels
            for (int jb = 0; jb <= ib; jb += GS)
                for (int i = ib; i < std::min(ib + GS, lf.nb); ++i)
                    for (int j = jb; j < std::min(jb + GS, i + 1); ++j) {
                        GradTask g{};
                        g.gemm.A = Xt(l) + (size_t)i * TB;
                        g.gemm.B = Xt(l) + (size_t)j * TB;
                        g.gemm.C = nullptr;
                        g.gemm.lda = g.gemm.ldb = lf.npad;
                        g.gemm.ldc = TB;
                        g.gemm.k0 = i * TB;
                        g.gemm.k1 = lf.npad;
                        g.gemm.update = 0;
                        g.xa = d.Xg + (size_t)i * TB;
                        g.xb = d.Xg + (size_t)j * TB;
                        g.alpha_a = d.alpha + (size_t)i * TB;
                        g.alpha_b = d.alpha + (size_t)j * TB;
                        g.ldx = lf.npad;
                        g.na = std::max(0, std::min(TB, lf.n - i * TB));
                        g.nb = std::max(0, std::min(TB, lf.n - j * TB));
                        g.diag = (i == j);
                        g.kid = lf.kid;
                        gd.push_back(g);
                        c->gdot_leaf.push_back(l);
                    }
        }
    }
    // neighbours in the list sit 8 apart in the launch: they run on one XCD and share its L2; the XCD slots get equal work
    {
        gblock.push_back(gd.size());
        std::vector<double> work(gd.size());
        for (size_t i = 0; i < gd.size(); ++i) work[i] = (double)(gd[i].gemm.k1 - gd[i].gemm.k0) + 64.0;   // + per-task overhead
        xcd_deal_by_work(gd, c->gdot_leaf, 0, gd.size(), gblock, work, c->xcd_order);
    }
    if (int rc = dev_upload(c, c->gtrans, trans)) return rc;
    if (int rc = dev_upload(c, c->gfrob, frob)) return rc;
    if (int rc = dev_upload(c, c->gupd, upd_all)) return rc;
    if (int rc = dev_upload(c, c->gred, red_all)) return rc;
    if (int rc = dev_upload(c, c->gtrsm, trsm_all)) return rc;
    if (int rc = dev_upload(c, c->gdot, gd)) return rc;
    if (any_ard && c->D > GRADDOT_STAGE_D)
        return fail(c, DSMGP_E_ARG, "ArdSE length-scale gradients need D <= " + std::to_string(GRADDOT_STAGE_D));
    c->gstride = any_ard ? 2 + c->D : 2;
    c->gpart_count = frob.size() + (size_t)c->gstride * gd.size() + 2 * (size_t)L;
    if (int rc = dev_grow(c, c->d_gpart, c->gpart_cap, c->gpart_count)) return rc;
    c->grad_ready = true;
    return 0;
}

}  // namespace

int dsmgp_set_gradient_leaves(dsmgp_ctx* c, const int32_t* active) {
    if (!c) return DSMGP_E_ARG;
    if (c->L == 0) return fail(c, DSMGP_E_STATE, "set_gradient_leaves before set_leaves");
    std::vector<char> m;
    if (active) {
        m.resize(c->L);
        bool all = true;
        for (int l = 0; l < c->L; ++l) {
            m[l] = active[l] != 0;
            all = all && m[l];
        }
        if (all) m.clear();
    }
    if (m != c->grad_active) {
        HIPCHK(c, hipSetDevice(c->device));
        free_grad_lists(c);
        c->grad_active = std::move(m);
    }
    return 0;
}

int dsmgp_gradients(dsmgp_ctx* c, double* grad_out, int32_t stride) {
    if (!c) return DSMGP_E_ARG;
    if (!c->fitted) return fail(c, DSMGP_E_STATE, "gradients before fit");
    if (!grad_out) return fail(c, DSMGP_E_ARG, "grad_out is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const int L = c->L;
    for (int l = 0; l < L; ++l) {
        const HyperHost& h = c->hyper[c->leaves[l].kid];
        if ((int)h.loghyp.size() > stride) return fail(c, DSMGP_E_ARG, "gradients: stride smaller than the hyper-vector");
    }
    if (!c->grad_ready)
        if (int rc = build_grad_plan(c)) return rc;
    if (int rc = ensure_dinv(c)) return rc;
    if (int rc = ensure_alpha(c)) return rc;
    c->timings[10] = 0.0;
    EventPair ev;
    HIPCHK(c, ev.init());
    const hipEvent_t t0 = ev.a, t1 = ev.b;
    HIPCHK(c, hipEventRecord(t0, c->stream));
    for (int i = 15; i < 18; ++i) c->timings[i] = 0.0;
    EventPair e_inv, e_dot;     // three spans: L^-T | contraction | traces and dots
    HIPCHK(c, e_inv.init());
    HIPCHK(c, e_dot.init());
    // Xt = L^-T (blocks left of the diagonal are never written and never read)
    if (c->gtrans.count) transpose_tile_kernel<<<(int)c->gtrans.count * 16, 256, 0, c->stream>>>(c->gtrans.p);
    const int gl = c->glanes;
    if (gl > 1) {       // fork: the lanes' streams wait for the transposes
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int q = 1; q < gl; ++q) HIPCHK(c, hipStreamWaitEvent(c->lane_stream[q], c->ev_fork, 0));
    }
    for (int k = 1; k < c->gsteps; ++k)
        for (int q = 0; q < gl; ++q) {
            const int v = q * c->gsteps + k;
            const hipStream_t st = gl > 1 ? c->lane_stream[q] : c->stream;
            const int nu = c->gupd_off[v + 1] - c->gupd_off[v];
            if (nu > 0) {
                launch_tiles(c, c->gupd.p + c->gupd_off[v], nu, 0, false, 0, nullptr, 0, st);
                const int nr = c->gred_off[v + 1] - c->gred_off[v];
                if (nr > 0) tile_reduce_kernel<<<nr * REDUCE_WGS, 256, 0, st>>>(c->gred.p + c->gred_off[v]);
            }
            const int ns = c->gtrsm_off[v + 1] - c->gtrsm_off[v];
            if (ns > 0) launch_tiles(c, c->gtrsm.p + c->gtrsm_off[v], ns, 1, false, 0, nullptr, 0, st);
        }
    for (int q = 1; q < gl; ++q) {      // join
        HIPCHK(c, hipEventRecord(c->ev_join[q], c->lane_stream[q]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[q], 0));
    }
    HIPCHK(c, hipEventRecord(e_inv.a, c->stream));
    double* pfrob = c->d_gpart;
    double* pdot = pfrob + c->gfrob.count;
    double* pleaf = pdot + (size_t)c->gstride * c->gdot.count;
    if (c->gdot.count)
        tile_graddot_kernel<<<(int)c->gdot.count, 256, 0, c->stream>>>(c->gdot.p, c->d_kp, c->D, pdot, c->gstride);
    HIPCHK(c, hipEventRecord(e_dot.a, c->stream));
    if (c->gfrob.count) frob_kernel<<<(int)c->gfrob.count, 256, 0, c->stream>>>(c->gfrob.p, pfrob);
    dots_kernel<<<L, 256, 0, c->stream>>>(c->d_leaves, pleaf);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(t1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
    c->timings[10] = ms * 1e-3;
    HIPCHK(c, hipEventElapsedTime(&ms, t0, e_inv.a));
    c->timings[15] = ms * 1e-3;
    HIPCHK(c, hipEventElapsedTime(&ms, e_inv.a, e_dot.a));
    c->timings[16] = ms * 1e-3;
    HIPCHK(c, hipEventElapsedTime(&ms, e_dot.a, t1));
    c->timings[17] = ms * 1e-3;
    std::vector<double> part(c->gpart_count);
    HIPCHK(c, hipMemcpy(part.data(), c->d_gpart, c->gpart_count * sizeof(double), hipMemcpyDeviceToHost));
    // host assembly (fixed summation order -> reproducible)
    std::vector<double> trK(L, 0.0), S1(L, 0.0);
    for (size_t i = 0; i < c->gfrob.count; ++i) trK[c->gfrob_leaf[i]] += part[i];
    for (int l = 0; l < L; ++l)
        if (c->leaves[l].owner != l) trK[l] = trK[c->leaves[l].owner];
    const double* pd = part.data() + c->gfrob.count;
    const size_t gs = (size_t)c->gstride;
    std::vector<double> Sd;           // per leaf and dimension: contraction with dK / dlog l_d (ArdSE option)
    if (gs > 2) Sd.assign((size_t)L * c->D, 0.0);
    for (size_t i = 0; i < c->gdot.count; ++i) {
        const int l = c->gdot_leaf[i];
        if (c->hyper[c->leaves[l].kid].kind == DSMGP_KIND_ARD_SE) {
            for (int d = 0; d < c->D; ++d) Sd[(size_t)l * c->D + d] += pd[gs * i + 2 + d];
        } else {
            S1[l] += pd[gs * i];
        }
    }
    for (int l = 0; l < L; ++l)
        if (c->grad_src[l] >= 0) {   // copygradients (src/fit.jl:352-356)
            S1[l] = S1[c->grad_src[l]];
            if (gs > 2)
                for (int d = 0; d < c->D; ++d) Sd[(size_t)l * c->D + d] = Sd[(size_t)c->grad_src[l] * c->D + d];
        }
    const double* pl = pd + gs * c->gdot.count;
    for (int l = 0; l < L; ++l) {
        const LeafHost& lf = c->leaves[l];
        const HyperHost& h = c->hyper[lf.kid];
        const int nl = (int)h.loghyp.size() - 2;
        const double noise = std::exp(2.0 * h.loghyp[nl + 1]);
        const double cc = noise + 1e-8;
        const double ya = pl[2 * l], aa = pl[2 * l + 1];
        const double n = (double)lf.n;
        // tr(precomp K) with K = K_y - c I:  (y.alpha - c alpha.alpha) - (n - c tr K_y^-1)
        const double trPK = (ya - cc * aa) - (n - cc * trK[l]);
        double* g = grad_out + (size_t)l * stride;
        for (int j = 0; j < stride; ++j) g[j] = 0.0;
        if (!c->grad_active.empty() && !c->grad_active[l]) continue;      // not asked for (dsmgp_set_gradient_leaves): zeros
        if (h.kind == DSMGP_KIND_ISO_SE) {
            const double sigma = std::exp(h.loghyp[1]);
            const double ell2 = std::exp(2.0 * h.loghyp[0]);
            g[0] = 0.5 * sigma * S1[l] / ell2;                // src/kernels.jl:95-97
            g[1] = sigma * trPK;                              // src/kernels.jl:90-93
        } else if (h.kind == DSMGP_KIND_ARD_SE) {
            const double sigma = std::exp(h.loghyp[nl]);
            for (int d = 0; d < nl; ++d)                      // src/kernels.jl:161: identically zero unless the
                g[d] = (c->ard_true_gradient && gs > 2) ? 0.5 * Sd[(size_t)l * c->D + d] : 0.0;   // true gradient is asked for
            g[nl] = sigma * trPK;                             // src/kernels.jl:157
        } else {
            g[0] = -trPK;                                     // src/kernels.jl:198
            g[1] = 0.0;                                       // src/kernels.jl:201
        }
        g[nl + 1] = noise * (aa - trK[l]);                    // src/gaussianprocess.jl:176
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
int dsmgp_kernel_matrix(dsmgp_ctx* c, int32_t kernel_id, const double* x1, int64_t n1, const double* x2,
                        int64_t n2, double* K_out) {
    if (!c) return DSMGP_E_ARG;
    if (!x1 || !x2 || !K_out || n1 <= 0 || n2 <= 0 || c->D <= 0) return fail(c, DSMGP_E_ARG, "kernel_matrix: bad arguments");
    if (kernel_id < 0 || kernel_id >= (int)c->hyper.size() || c->hyper[kernel_id].kind < 0)
        return fail(c, DSMGP_E_STATE, "kernel_matrix: kernel id without hyper-parameters");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_hyper(c)) return rc;
    const int D = c->D;
    const int p1 = round_up((int)n1, TB), p2 = round_up((int)n2, TB);
    double *dx1 = nullptr, *dx2 = nullptr, *dK = nullptr;
    HIPCHK(c, hipMalloc(&dx1, (size_t)n1 * D * sizeof(double)));
    HIPCHK(c, hipMalloc(&dx2, (size_t)n2 * D * sizeof(double)));
    HIPCHK(c, hipMalloc(&dK, (size_t)p1 * p2 * sizeof(double)));
    HIPCHK(c, hipMemcpy(dx1, x1, (size_t)n1 * D * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dx2, x2, (size_t)n2 * D * sizeof(double), hipMemcpyHostToDevice));
    std::vector<GramTask> g;
    for (int j = 0; j < p2 / TB; ++j)
        for (int i = 0; i < p1 / TB; ++i) {
            GramTask t{};
            t.xa = dx1 + (size_t)i * TB;
            t.xb = dx2 + (size_t)j * TB;
            t.out = dK + (size_t)i * TB + (size_t)j * TB * p1;
            t.lda = (int)n1;
            t.ldb = (int)n2;
            t.ldo = p1;
            t.na = (int)std::min<int64_t>(TB, n1 - (int64_t)i * TB);
            t.nb = (int)std::min<int64_t>(TB, n2 - (int64_t)j * TB);
            t.kid = kernel_id;
            g.push_back(t);
        }
    DevBuf<GramTask> dg;
    if (int rc = dev_upload(c, dg, g)) return rc;
    gram_tile_kernel<<<2 * (int)g.size(), 256, 0, c->stream>>>(dg.p, c->d_kp, D);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(K_out, (size_t)n1 * sizeof(double), dK, (size_t)p1 * sizeof(double), (size_t)n1 * sizeof(double),
                          (size_t)n2, hipMemcpyDeviceToHost));
    dev_free(dg);
    dev_free(dx1);
    dev_free(dx2);
    dev_free(dK);
    return 0;
}

int dsmgp_download_factor(dsmgp_ctx* c, int32_t leaf, double* F, double* alpha) {
    if (!c) return DSMGP_E_ARG;
    if (!c->fitted) return fail(c, DSMGP_E_STATE, "download_factor before fit");
    if (leaf < 0 || leaf >= c->L) return fail(c, DSMGP_E_ARG, "leaf out of range");
    HIPCHK(c, hipSetDevice(c->device));
    const LeafHost& lf = c->leaves[leaf];
    const LeafDev& d = c->h_leaves[leaf];
    if (F) {
        HIPCHK(c, hipMemcpy2D(F, (size_t)lf.n * sizeof(double), d.F, (size_t)lf.npad * sizeof(double),
                              (size_t)lf.n * sizeof(double), (size_t)lf.n, hipMemcpyDeviceToHost));
        for (int col = 1; col < lf.n; ++col)
            for (int r = 0; r < col; ++r) F[r + (size_t)col * lf.n] = 0.0;
    }
    if (alpha) {
        if (int rc = ensure_alpha(c)) return rc;
        HIPCHK(c, hipMemcpy(alpha, d.alpha, lf.n * sizeof(double), hipMemcpyDeviceToHost));
    }
    return 0;
}

int dsmgp_timings(dsmgp_ctx* c, double* out) {
    if (!c || !out) return DSMGP_E_ARG;
    for (int i = 0; i < DSMGP_N_TIMINGS; ++i) out[i] = c->timings[i];
    return 0;
}

int dsmgp_work_gradients(dsmgp_ctx* c, double* alg_flops_inverse, double* alg_flops_contraction, int32_t* n_contraction_tiles) {
    if (!c) return DSMGP_E_ARG;
    // L^-T of every factor owner: n^3/3; contraction (alpha alpha^T - K_y^-1) o K o P of every IsoSE leaf: the lower
    // tiles of L^-T L^-1, n^3/3 again (2 x 128 x 128 x K per tile with K = n - 128 i)
    double fi = 0.0, fc = 0.0;
    for (int l = 0; l < c->L; ++l) {
        const LeafHost& lf = c->leaves[l];
        const double n = (double)lf.n;
        if (lf.owner == l) fi += n * n * n / 3.0;
        if (lf.kid < (int)c->hyper.size() && c->hyper[lf.kid].kind == DSMGP_KIND_ISO_SE) fc += n * n * n / 3.0;
    }
    if (alg_flops_inverse) *alg_flops_inverse = fi;
    if (alg_flops_contraction) *alg_flops_contraction = fc;
    if (n_contraction_tiles) *n_contraction_tiles = (int32_t)c->gdot.count;
    return 0;
}

int dsmgp_work(dsmgp_ctx* c, double* alg_flops_update, int32_t* n_update_launches) {
    if (!c) return DSMGP_E_ARG;
    if (alg_flops_update) *alg_flops_update = c->last_fit_joint ? c->alg_flops_joint : c->alg_flops_update;
    if (n_update_launches) *n_update_launches = c->n_update_launches;
    return 0;
}

int dsmgp_lanes(dsmgp_ctx* c, int32_t* lanes) {
    if (!c || !lanes) return DSMGP_E_ARG;
    *lanes = c->plan_ready ? c->nlanes : 0;
    return 0;
}

int dsmgp_work_fused(dsmgp_ctx* c, double* alg_flops_fused, int32_t* n_fused_launches) {
    if (!c) return DSMGP_E_ARG;
    if (alg_flops_fused) *alg_flops_fused = c->last_fit_joint ? c->alg_flops_fused_joint : c->alg_flops_fused;
    if (n_fused_launches) *n_fused_launches = c->n_fused_launches;
    return 0;
}

// Free every device buffer that scales with the leaf sizes (factors, inverse blocks, K_tn rows, L^-1, task lists),
// keeping the training data, the leaf table, the sharing schedule and the hyper-parameters: the next fit rebuilds
// them.  This is the "discard" half of the factor-and-discard streaming mode (hipabi.StreamingContext).
int dsmgp_reserve(dsmgp_ctx* c, int64_t bytes) {
    if (!c || bytes < 0) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_plan(c);
    free_test(c);
    if (c->pool_base) (void)hipFree(c->pool_base);
    c->pool_base = nullptr;
    c->pool_cap = c->pool_top = c->pool_mark_plan = 0;
    if (bytes == 0) return 0;
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, DSMGP_E_NOMEM, "cannot reserve " + std::to_string(bytes >> 20) + " MiB of device memory");
    }
    c->pool_base = static_cast<char*>(p);
    c->pool_cap = (size_t)bytes;
    return 0;
}

int dsmgp_release(dsmgp_ctx* c) {
    if (!c) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HostLog hl("release");
    free_plan(c);
    free_test(c);
    return 0;
}

// Device bytes a leaf table of these sizes needs resident (factor + inverse blocks + gathered inputs + vectors,
// K_tn rows for n_test routed rows per leaf, and L^-1 when gradients are wanted); no context state involved.
int64_t dsmgp_estimate_bytes(int32_t L, const int64_t* n, const int64_t* n_test, int32_t D, int32_t with_gradients) {
    int64_t tot = 0;
    for (int l = 0; l < L; ++l) {
        const int64_t np_ = (n[l] + TB - 1) / TB * TB;
        const int64_t nt_ = n_test ? (n_test[l] + TB - 1) / TB * TB : 0;
        tot += np_ * np_ * (with_gradients ? 2 : 1) + np_ * TB + np_ * (D + 4) + nt_ * np_ + nt_ * (D + 4);
    }
    return tot * (int64_t)sizeof(double);
}

int dsmgp_memory(dsmgp_ctx* c, int64_t* needed, int64_t* free_bytes) {
    if (!c) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    if (needed) *needed = (int64_t)c->bytes_needed;
    if (free_bytes) *free_bytes = (int64_t)f;
    return 0;
}

int dsmgp_probe_f64_mfma(dsmgp_ctx* c, double* tflops) {
    double d[4];
    if (!tflops) return DSMGP_E_ARG;
    if (int rc = dsmgp_probe_f64_mfma_detail(c, 8, d)) return rc;
    *tflops = d[0];
    return 0;
}

// out[0] = TFLOP/s over the chip, out[1] = shader cycles per MFMA issued by one wave (median wave),
// out[2] = shader clock in GHz held during the loop, out[3] = waves per SIMD used
int dsmgp_probe_f64_mfma_detail(dsmgp_ctx* c, int32_t blocks_per_cu, double* out) {
    if (!c || !out || blocks_per_cu < 1 || blocks_per_cu > 8) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int blocks = c->ncu * blocks_per_cu;
    const int iters = 6000 / blocks_per_cu;
    const int nwaves = blocks * 4;
    double* sink = nullptr;
    unsigned long long* stamps = nullptr;
    HIPCHK(c, hipMalloc(&sink, (size_t)blocks * 256 * sizeof(double)));
    HIPCHK(c, hipMalloc(&stamps, (size_t)nwaves * 2 * sizeof(unsigned long long)));
    EventPair ev;
    HIPCHK(c, ev.init());
    const hipEvent_t t0 = ev.a, t1 = ev.b;
    for (int rep = 0; rep < 3; ++rep) mfma_probe_kernel<<<blocks, 256, 0, c->stream>>>(sink, stamps, iters);
    HIPCHK(c, hipEventRecord(t0, c->stream));
    mfma_probe_kernel<<<blocks, 256, 0, c->stream>>>(sink, stamps, iters);
    HIPCHK(c, hipEventRecord(t1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
    std::vector<unsigned long long> hs((size_t)nwaves * 2);
    HIPCHK(c, hipMemcpy(hs.data(), stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    dev_free(sink);
    dev_free(stamps);
    std::vector<double> cyc(nwaves), clk(nwaves);
    for (int i = 0; i < nwaves; ++i) {
        cyc[i] = (double)hs[2 * i] / ((double)iters * 16.0);
        clk[i] = hs[2 * i + 1] ? (double)hs[2 * i] / ((double)hs[2 * i + 1] * 10.0) : 0.0;   // cycles per ns
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double flops = (double)blocks * 4.0 * (double)iters * 16.0 * 2048.0;
    out[0] = flops / (ms * 1e-3) / 1e12;
    out[1] = cyc[nwaves / 2];
    out[2] = clk[nwaves / 2];   // cycles per 10 ns tick / 10 = GHz
    out[3] = blocks_per_cu;
    return 0;
}

// Shader clock held under load: a one-wave kernel on a stream of its own that sleeps for `milliseconds` of wall time beside
// whatever the context launches meanwhile and reports shader cycles per wall tick.  start returns at once; read waits for it.
int dsmgp_clock_sample_start(dsmgp_ctx* c, double milliseconds) {
    if (!c) return DSMGP_E_ARG;
    if (!(milliseconds > 0.0) || milliseconds > 5000.0) return fail(c, DSMGP_E_ARG, "clock_sample_start: 0 < milliseconds <= 5000");
    if (c->clock_pending) return fail(c, DSMGP_E_STATE, "clock_sample_start: a sample is already running (read it first)");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->side) HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    if (!c->d_clock) HIPCHK(c, hipMalloc(&c->d_clock, 2 * sizeof(unsigned long long)));
    clock_sample_kernel<<<1, 64, 0, c->side>>>(c->d_clock, (unsigned long long)(milliseconds * 1e5));   // 100 MHz ticks
    HIPCHK(c, hipGetLastError());
    c->clock_pending = true;
    return 0;
}

int dsmgp_clock_sample_read(dsmgp_ctx* c, double* ghz, double* milliseconds) {
    if (!c) return DSMGP_E_ARG;
    if (!c->clock_pending) return fail(c, DSMGP_E_STATE, "clock_sample_read before clock_sample_start");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->side));
    c->clock_pending = false;
    unsigned long long h[2] = {0, 0};
    HIPCHK(c, hipMemcpy(h, c->d_clock, sizeof(h), hipMemcpyDeviceToHost));
    if (ghz) *ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;      // cycles per 10 ns tick / 10 = GHz
    if (milliseconds) *milliseconds = (double)h[1] * 1e-5;
    return 0;
}

#ifdef DSMGP_DIAG
// ---- diagnostic build only (libdsmgp_hip_diag.so, include/dsmgp_hip_diag.h): never part of the product library ----
// Diagnostic: f64 MFMA and f64 VALU FMA alone and co-issued (two waves per SIMD).  out[3*mode + {0,1,2}] =
// {MFMA TFLOP/s, VALU TFLOP/s, wall ms} for mode 0 (MFMA only), 1 (VALU only), 2 (one wave of each per SIMD).
int dsmgp_probe_coissue(dsmgp_ctx* c, double* out) {
    if (!c || !out) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int blocks = c->ncu, iters = 2000;
    double* sink = nullptr;
    unsigned long long* st = nullptr;
    HIPCHK(c, hipMalloc(&sink, (size_t)blocks * 512 * sizeof(double)));
    HIPCHK(c, hipMalloc(&st, (size_t)blocks * 8 * 2 * sizeof(unsigned long long)));
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t t0, t1;
        HIPCHK(c, hipEventCreate(&t0));
        HIPCHK(c, hipEventCreate(&t1));
        coissue_probe_kernel<<<blocks, 512, 0, c->stream>>>(sink, st, iters, mode);
        HIPCHK(c, hipEventRecord(t0, c->stream));
        coissue_probe_kernel<<<blocks, 512, 0, c->stream>>>(sink, st, iters, mode);
        HIPCHK(c, hipEventRecord(t1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        const double waves_mfma = (mode == 0) ? 8.0 : (mode == 2 ? 4.0 : 0.0);
        const double waves_valu = (mode == 1) ? 8.0 : (mode == 2 ? 4.0 : 0.0);
        const double fl_m = (double)blocks * waves_mfma * iters * 16.0 * 2048.0;
        const double fl_v = (double)blocks * waves_valu * iters * 8.0 * 32.0 * 128.0;   // 64 lanes x 2 flops per v_fma_f64
        out[3 * mode] = fl_m / (ms * 1e-3) / 1e12;
        out[3 * mode + 1] = fl_v / (ms * 1e-3) / 1e12;
        out[3 * mode + 2] = ms;
    }
    dev_free(sink);
    dev_free(st);
    return 0;
}

// Diagnostic: steady-state rate of tile_gemm_kernel on a uniform batch (no factorisation around it).
// mode 0: every tile has its own A panel, groups of `group` tiles share a B panel (the access pattern of
// one block step); mode 1: all tiles read the same A and B panels (operands stay in L2).
int dsmgp_bench_tile(dsmgp_ctx* c, int32_t ntiles, int32_t K, int32_t mode, int32_t group, int32_t reps,
                     double* seconds_per_launch) {
    if (!c || ntiles <= 0 || K <= 0 || K % KC || !seconds_per_launch || group <= 0) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t panel = (size_t)TB * K;
    const int nA = mode == 1 ? 1 : ((mode == 3 || mode == 5) ? (ntiles + group - 1) / group * group : ntiles);
    const int nB = mode == 1 ? 1 : (ntiles + group - 1) / group;
    if (mode == 2 && std::getenv("DSMGP_STAMPS")) return fail(c, DSMGP_E_ARG, "no stamps in mode 2");
    double *A = nullptr, *B = nullptr, *C = nullptr;
    HIPCHK(c, hipMalloc(&A, nA * panel * sizeof(double)));
    HIPCHK(c, hipMalloc(&B, nB * panel * sizeof(double)));
    HIPCHK(c, hipMalloc(&C, (size_t)ntiles * TB * TB * sizeof(double)));
    HIPCHK(c, hipMemset(A, 0, nA * panel * sizeof(double)));
    HIPCHK(c, hipMemset(B, 0, nB * panel * sizeof(double)));
    HIPCHK(c, hipMemset(C, 0, (size_t)ntiles * TB * TB * sizeof(double)));
    {   // non-trivial operand values (random-ish, bounded)
        std::vector<double> hv(panel);
        for (size_t i = 0; i < panel; ++i) hv[i] = 1e-3 * (double)((i * 2654435761u) % 2001) - 1.0;
        for (int i = 0; i < nA; ++i) HIPCHK(c, hipMemcpy(A + i * panel, hv.data(), panel * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < nB; ++i) HIPCHK(c, hipMemcpy(B + i * panel, hv.data(), panel * sizeof(double), hipMemcpyHostToDevice));
    }
    std::vector<TileTask> tasks(ntiles);
    for (int i = 0; i < ntiles; ++i) {
        TileTask t{};
        t.A = A + (mode == 1 ? 0 : (size_t)i * panel);
        t.B = B + (mode == 1 ? 0 : (size_t)(i / group) * panel);
        t.C = C + (size_t)i * TB * TB;
        t.lda = t.ldb = TB;
        if (mode == 3 || mode == 5) {   // A tiles are row tiles of a (group*128) x K column-major matrix, like the rows of Vt
            t.A = A + (size_t)(i / group) * group * panel + (size_t)(i % group) * TB;
            t.lda = group * TB;
        }
        t.ldc = TB;
        t.k0 = 0;
        t.k1 = K;
        t.update = 1;
        if (mode == 4 || mode == 5) {   // diagonal tiles of the factorisation: C -= A A^T, lower blocks only
            t.B = t.A;
            t.ldb = t.lda;
            t.sym = 1;
        }
        tasks[i] = t;
    }
    // mode 0/1: the raw batch; mode 2: the batch as UpdateSplitter would schedule it (split-K + reduce)
    UpdateSplitter U;
    U.ncu = c->ncu;
    U.xcd = c->xcd_order;
    U.tail_split = c->tail_split;
    U.tail_rounds = c->tail_rounds;
    double* slabs = nullptr;
    if (mode == 2) {
        U.add_step(tasks, K);
        if (U.max_slabs) HIPCHK(c, hipMalloc(&slabs, U.max_slabs * TB * TB * sizeof(double)));
        U.bind(slabs);
        tasks = U.upd;
        std::fprintf(stderr, "  splitter: %zu tiles -> %zu tasks, %zu reduces\n", (size_t)ntiles, tasks.size(), U.red.size());
    } else {
        std::vector<int> dummy(tasks.size());
        xcd_permute(tasks, dummy, 0, tasks.size(), c->xcd_order && (mode == 0 || mode == 6));
        // mode 6: mode 0 with the first workgroup of every CU at half depth -- the two workgroups of a CU then run half a task
        // apart for the rest of the launch instead of reaching their epilogues (and the next tasks' first loads) together
        if (mode == 6)
            for (int i = 0; i < std::min(ntiles, c->ncu); ++i) tasks[i].k1 = K / 2 / KC * KC;
    }
    DevBuf<TileTask> dt;
    DevBuf<ReduceTask> dr;
    if (int rc = dev_upload(c, dt, tasks)) return rc;
    if (int rc = dev_upload(c, dr, U.red)) return rc;
    const int nt_ = (int)tasks.size();
    auto run = [&]() {
        launch_tiles(c, dt.p, nt_);
        if (dr.count) tile_reduce_kernel<<<(int)dr.count * REDUCE_WGS, 256, 0, c->stream>>>(dr.p);
    };
    hipEvent_t t0, t1;
    HIPCHK(c, hipEventCreate(&t0));
    HIPCHK(c, hipEventCreate(&t1));
    run();
    HIPCHK(c, hipEventRecord(t0, c->stream));
    for (int r = 0; r < reps; ++r) run();
    HIPCHK(c, hipEventRecord(t1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    dev_free(dr);
    dev_free(slabs);
    *seconds_per_launch = ms * 1e-3 / reps;
    if (std::getenv("DSMGP_STAMPS")) {
        unsigned long long* st = nullptr;
        HIPCHK(c, hipMalloc(&st, (size_t)ntiles * 32 * sizeof(unsigned long long)));
        tile_gemm_kernel_v2<true><<<ntiles, 256, 0, c->stream>>>(dt.p, st, nullptr, 0, 0, nullptr, 0);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<unsigned long long> hs((size_t)ntiles * 32);
        HIPCHK(c, hipMemcpy(hs.data(), st, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        dev_free(st);
        double tot = 0, mf = 0, bd = 0, nchs = 0, pro = 0, loop = 0, epi = 0;
        unsigned long long first = ~0ull, last = 0;
        for (size_t i = 0; i < (size_t)ntiles * 4; ++i) {
            const unsigned long long* h = hs.data() + 8 * i;
            tot += (double)h[0];
            mf += (double)h[1];
            bd += (double)h[2];
            nchs += (double)h[3];
            pro += (double)(h[4] - h[6]);
            loop += (double)(h[5] - h[4]);
            epi += (double)(h[7] - h[5]);
            first = std::min(first, h[6]);
            last = std::max(last, h[7]);
        }
        const double nw = (double)ntiles * 4.0;
        std::fprintf(stderr,
                     "  stamps: cycles/chunk-pair total %.0f  mfma-span %.0f  boundary %.0f (per wave, mean); per wave us: "
                     "prologue %.2f  loop %.2f  epilogue %.2f  -> loop clock %.3f GHz; launch span %.1f us, sum of wave "
                     "times / (span x slots) = %.3f\n",
                     tot / nchs, mf / nchs, bd / nchs, pro / nw * 0.01, loop / nw * 0.01, epi / nw * 0.01,
                     tot / loop / 10.0, (double)(last - first) * 0.01,
                     (pro + loop + epi) / ((double)(last - first) * std::min<double>(2.0 * c->ncu * 4.0, nw)));
    }
    dev_free(dt);
    dev_free(A);
    dev_free(B);
    dev_free(C);
    return 0;
}
// Diagnostic: the eight-wave fused tile task (tile_fused8_kernel) on a uniform batch of `ntasks` tasks of depth K -- eight full
// 16-row blocks each with their own A rows, groups of `group` tasks sharing a B panel, the access pattern of bench_tile's mode 0.
// A task of depth K is its kernel function, K / 128 block columns of product, one substitution and a store: the SLOPE of the
// launch time over K is the steady-state rate of the eight-wave product loop (four waves per SIMD), to set against
// tile_gemm_kernel_v2's (two waves per SIMD) on the same shape.
int dsmgp_bench_fused8(dsmgp_ctx* c, int32_t ntasks, int32_t K, int32_t group, int32_t reps, double* seconds_per_launch) {
    if (!c || ntasks <= 0 || K < 0 || K % TB || !seconds_per_launch || group <= 0 || reps <= 0) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int D = 8;
    const size_t apanel = (size_t)TB * K, bpanel = (size_t)TB * (K + TB);
    const int nB = (ntasks + group - 1) / group;
    double *A = nullptr, *B = nullptr, *Cc = nullptr, *Dv = nullptr, *gx = nullptr, *l2 = nullptr;
    KParam* kp = nullptr;
    HIPCHK(c, hipMalloc(&A, std::max<size_t>(1, (size_t)ntasks * apanel) * sizeof(double)));
    HIPCHK(c, hipMalloc(&B, (size_t)nB * bpanel * sizeof(double)));
    HIPCHK(c, hipMalloc(&Cc, (size_t)ntasks * TB * TB * sizeof(double)));
    HIPCHK(c, hipMalloc(&Dv, (size_t)TB * TB * sizeof(double)));
    HIPCHK(c, hipMalloc(&gx, (size_t)TB * D * sizeof(double)));
    HIPCHK(c, hipMalloc(&l2, 2 * sizeof(double)));
    HIPCHK(c, hipMalloc(&kp, sizeof(KParam)));
    {
        std::vector<double> hv(std::max(apanel, bpanel));
        for (size_t i = 0; i < hv.size(); ++i) hv[i] = 1e-3 * (double)((i * 2654435761u) % 2001) - 1.0;
        for (int i = 0; i < ntasks && apanel; ++i) HIPCHK(c, hipMemcpy(A + i * apanel, hv.data(), apanel * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < nB; ++i) HIPCHK(c, hipMemcpy(B + i * bpanel, hv.data(), bpanel * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double> id((size_t)TB * TB, 0.0), xs((size_t)TB * D);
        for (int i = 0; i < TB; ++i) id[i + (size_t)i * TB] = 1.0;
        for (size_t i = 0; i < xs.size(); ++i) xs[i] = 1e-3 * (double)((i * 40503u) % 1000);
        HIPCHK(c, hipMemcpy(Dv, id.data(), id.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(gx, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
        const double hl2[2] = {1.0, -0.5};
        HIPCHK(c, hipMemcpy(l2, hl2, sizeof(hl2), hipMemcpyHostToDevice));
        KParam h{};
        h.kind = 0;
        h.nl = 1;
        h.sigma2 = h.sigma = 1.0;
        h.noise = 0.01;
        h.l2 = l2;
        h.nh = l2 + 1;
        h.nh0 = -0.5;
        h.il2 = 1.0;
        HIPCHK(c, hipMemcpy(kp, &h, sizeof(h), hipMemcpyHostToDevice));
    }
    std::vector<FusedTask8> tasks(ntasks);
    for (int i = 0; i < ntasks; ++i) {
        FusedTask8 t{};
        t.B = B + (size_t)(i / group) * bpanel;
        t.Dinv = Dv;
        t.zk = nullptr;
        t.gxb = gx;
        t.ldb = TB;
        t.gldb = TB;
        t.gnb = TB;
        t.k1 = K;
        t.kid = 0;
        t.nblk = 8;
        for (int w = 0; w < 8; ++w) {
            RowBlock& r = t.rb[w];
            r.A = (K ? A + (size_t)i * apanel : B) + 16 * w;
            r.C = Cc + (size_t)i * TB * TB + 16 * w;
            r.gx = gx + 16 * w;
            r.wi = nullptr;
            r.sq = nullptr;
            r.lda = TB;
            r.ldc = TB;
            r.glda = TB;
            r.nvalid = 16;
        }
        tasks[i] = t;
    }
    {
        std::vector<int> dummy(tasks.size());
        xcd_permute(tasks, dummy, 0, tasks.size(), c->xcd_order);
    }
    DevBuf<FusedTask8> dt;
    if (int rc = dev_upload(c, dt, tasks)) return rc;
    auto run = [&]() { tile_fused8_kernel<2><<<ntasks, 512, 0, c->stream>>>(dt.p, kp, D); };
    EventPair ev;
    HIPCHK(c, ev.init());
    run();
    HIPCHK(c, hipEventRecord(ev.a, c->stream));
    for (int r = 0; r < reps; ++r) run();
    HIPCHK(c, hipEventRecord(ev.b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev.a, ev.b));
    *seconds_per_launch = ms * 1e-3 / reps;
    dev_free(dt);
    dev_free(A);
    dev_free(B);
    dev_free(Cc);
    dev_free(Dv);
    dev_free(gx);
    dev_free(l2);
    dev_free(kp);
    return 0;
}
// Diagnostic: the diagonal-block kernel alone on `ntiles` well-conditioned blocks (microseconds per launch over `reps`
// launches) and the wall-clock phases of block 0's wave 0 from one stamped launch: phases_us[0..19] = load, first
// 16x16 block, then (P1, P2) of the 8 block steps, write-back, inverse phase; [20], [21] = inside step 3's P2 on wave 0:
// the trailing product of the next diagonal block, its potrf + inverse (potrf_inv16 with its LDS reads and writes);
// [22] = shader cycles of [21].
int dsmgp_probe_diag(dsmgp_ctx* c, int32_t ntiles, int32_t ld, int32_t reps, double* kernel_us, double* phases_us) {
    if (!c || ntiles <= 0 || ld < TB || reps <= 0 || !kernel_us || !phases_us) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t tile = (size_t)ld * TB;
    std::vector<double> h(tile, 0.0);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        return (double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    };
    for (int cidx = 0; cidx < TB; ++cidx)
        for (int r = cidx; r < TB; ++r) h[r + (size_t)cidx * ld] = (r == cidx) ? 64.0 + rnd() : rnd();
    double *T0 = nullptr, *T = nullptr, *Dinv = nullptr, *wz = nullptr;
    int* info = nullptr;
    unsigned long long* stamps = nullptr;
    DiagTask* dt = nullptr;
    HIPCHK(c, hipMalloc(&T0, tile * sizeof(double)));
    HIPCHK(c, hipMalloc(&T, ntiles * tile * sizeof(double)));
    HIPCHK(c, hipMalloc(&Dinv, (size_t)ntiles * TB * TB * sizeof(double)));
    HIPCHK(c, hipMalloc(&wz, (size_t)ntiles * 2 * TB * sizeof(double)));
    HIPCHK(c, hipMalloc(&info, ntiles * sizeof(int)));
    HIPCHK(c, hipMalloc(&stamps, (size_t)ntiles * 24 * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&dt, ntiles * sizeof(DiagTask)));
    HIPCHK(c, hipMemcpy(T0, h.data(), tile * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(wz, 0, (size_t)ntiles * 2 * TB * sizeof(double)));
    HIPCHK(c, hipMemset(Dinv, 0, (size_t)ntiles * TB * TB * sizeof(double)));
    HIPCHK(c, hipMemset(info, 0, ntiles * sizeof(int)));
    std::vector<DiagTask> tasks(ntiles);
    for (int i = 0; i < ntiles; ++i) {
        DiagTask g{};
        g.T = T + i * tile;
        g.Dinv = Dinv + (size_t)i * TB * TB;
        g.wk = wz + (size_t)i * 2 * TB;
        g.zk = wz + (size_t)i * 2 * TB + TB;
        g.info = info + i;
        g.ld = ld;
        g.nvalid = TB;
        g.row0 = 0;
        tasks[i] = g;
    }
    HIPCHK(c, hipMemcpy(dt, tasks.data(), ntiles * sizeof(DiagTask), hipMemcpyHostToDevice));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_packed_stamp_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, DIAGP_LDS_BYTES);
    const size_t lds = DIAGP_LDS_BYTES;
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    auto refill = [&]() {
        for (int i = 0; i < ntiles; ++i)
            (void)hipMemcpyAsync(T + i * tile, T0, tile * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    };
    double total = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        refill();
        HIPCHK(c, hipEventRecord(e0, c->stream));
        chol_diag_packed_kernel<<<ntiles, 256, lds, c->stream>>>(dt);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
    }
    *kernel_us = total / reps * 1e3;
    refill();
    chol_diag_packed_stamp_kernel<<<ntiles, 256, lds, c->stream>>>(dt, stamps);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long hs[24];
    HIPCHK(c, hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
    for (int i = 0; i < 20; ++i) phases_us[i] = (double)(hs[i + 1] - hs[i]) * 0.01;   // 100 MHz
    phases_us[20] = (double)(hs[21] - hs[9]) * 0.01;
    phases_us[21] = (double)(hs[22] - hs[21]) * 0.01;
    phases_us[22] = (double)hs[23];                       // shader cycles of the interval of [21]
    int bad = 0;
    HIPCHK(c, hipMemcpy(&bad, info, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(T0); (void)hipFree(T); (void)hipFree(Dinv); (void)hipFree(wz);
    (void)hipFree(info); (void)hipFree(stamps); (void)hipFree(dt);
    if (bad != 0) return fail(c, DSMGP_E_STATE, "probe block was not positive definite");
    return 0;
}

// diagnostic: the diagonal-block task of a FUSED step (diag_fused_reg_kernel) alone on ntiles synthetic blocks of depth K: IsoSE
// values of random points in [0, 1)^8 (l = 0.3, noise 0.01) minus a small product -- what a depth-4 fit launches 18k of per step.
// ntiles = 256 / 512 / 768 puts one / two / three tasks on every CU: the latency of a task alone and what co-residents cost it.
int dsmgp_probe_diag_fused(dsmgp_ctx* c, int32_t ntiles, int32_t K, int32_t reps, double* kernel_us) {
    if (!c || ntiles <= 0 || K < 0 || K % TB != 0 || reps <= 0 || !kernel_us) return DSMGP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int D = 8;
    const size_t leaf = (size_t)TB * (size_t)(K + TB);          // block row: 128 rows x (K + 128) columns, ld = 128
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        return (double)(st >> 11) * (1.0 / 9007199254740992.0);
    };
    std::vector<double> hx((size_t)TB * D), ha((size_t)TB * std::max(K, 1));
    for (double& v : hx) v = rnd();
    for (double& v : ha) v = (rnd() - 0.5) * 2e-3;
    double *F = nullptr, *X = nullptr, *Dinv = nullptr, *wz = nullptr, *nh = nullptr;
    int* info = nullptr;
    DiagFusedTask* dt = nullptr;
    KParam* kp = nullptr;
    HIPCHK(c, hipMalloc(&F, (size_t)ntiles * leaf * sizeof(double)));
    HIPCHK(c, hipMalloc(&X, (size_t)TB * D * sizeof(double)));
    HIPCHK(c, hipMalloc(&Dinv, (size_t)ntiles * TB * TB * sizeof(double)));
    HIPCHK(c, hipMalloc(&wz, (size_t)ntiles * 2 * TB * sizeof(double)));
    HIPCHK(c, hipMalloc(&info, ntiles * sizeof(int)));
    HIPCHK(c, hipMalloc(&dt, ntiles * sizeof(DiagFusedTask)));
    HIPCHK(c, hipMalloc(&kp, sizeof(KParam)));
    HIPCHK(c, hipMalloc(&nh, 2 * sizeof(double)));
    HIPCHK(c, hipMemset(F, 0, (size_t)ntiles * leaf * sizeof(double)));
    HIPCHK(c, hipMemset(Dinv, 0, (size_t)ntiles * TB * TB * sizeof(double)));
    HIPCHK(c, hipMemset(wz, 0, (size_t)ntiles * 2 * TB * sizeof(double)));
    HIPCHK(c, hipMemset(info, 0, ntiles * sizeof(int)));
    HIPCHK(c, hipMemcpy(X, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice));
    if (K > 0)
        for (int i = 0; i < ntiles; ++i)
            HIPCHK(c, hipMemcpyAsync(F + (size_t)i * leaf, ha.data(), (size_t)TB * K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const double l2 = 0.09, hnh[2] = {-0.5 / l2, l2};
    HIPCHK(c, hipMemcpy(nh, hnh, sizeof(hnh), hipMemcpyHostToDevice));
    KParam p{};
    p.kind = DSMGP_KIND_ISO_SE;
    p.nl = 1;
    p.sigma2 = p.sigma = 1.0;
    p.noise = 0.01;
    p.l2 = nh + 1;
    p.nh = nh;
    p.nh0 = hnh[0];
    p.il2 = 1.0 / l2;
    HIPCHK(c, hipMemcpy(kp, &p, sizeof(p), hipMemcpyHostToDevice));
    std::vector<DiagFusedTask> tasks(ntiles);
    for (int i = 0; i < ntiles; ++i) {
        DiagFusedTask f{};
        f.d.T = F + (size_t)i * leaf + (size_t)K * TB;
        f.d.Dinv = Dinv + (size_t)i * TB * TB;
        f.d.wk = wz + (size_t)i * 2 * TB;
        f.d.zk = wz + (size_t)i * 2 * TB + TB;
        f.d.info = info + i;
        f.d.ld = TB;
        f.d.nvalid = TB;
        f.d.row0 = 0;
        f.A = F + (size_t)i * leaf;
        f.gx = X;
        f.k1 = K;
        f.glda = TB;
        f.kid = 0;
        tasks[i] = f;
    }
    HIPCHK(c, hipMemcpy(dt, tasks.data(), ntiles * sizeof(DiagFusedTask), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    double total = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        HIPCHK(c, hipEventRecord(e0, c->stream));
        diag_fused_reg_kernel<<<ntiles, 256, DIAGR_LDS_BYTES, c->stream>>>(dt, kp, D);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
    }
    *kernel_us = total / reps * 1e3;
    int bad = 0;
    HIPCHK(c, hipMemcpy(&bad, info, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(F); (void)hipFree(X); (void)hipFree(Dinv); (void)hipFree(wz); (void)hipFree(nh);
    (void)hipFree(info); (void)hipFree(dt); (void)hipFree(kp);
    if (bad != 0) return fail(c, DSMGP_E_STATE, "probe block was not positive definite");
    return 0;
}

#endif  // DSMGP_DIAG

// -------------------------------------------------------------------------------------------------
// The one exchange step of the path when leaves are sharded over the GPUs of a node: an all-gather of
// per-leaf log-marginals after fit! and of the aggregation's partial sums after predict, over RCCL (xGMI inside a node)
// on the context's stream.  librccl.so is opened with dlopen on first use: a single-GPU process never loads it, and
// the library itself has no link-time dependency on it.
namespace {
struct RcclApi {
    void* lib = nullptr;
    struct UniqueId { char internal[128]; };                         // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // First choice: the librccl that sits NEXT TO the HIP runtime this library is bound to.  A process can hold two
        // ROCm runtimes -- the system's, and the one a PyTorch wheel bundles and maps under the plain names "librccl.so" /
        // "libamdhip64.so" -- and only the first of them to initialise owns the GPU.  dlopen("librccl.so") returns whichever
        // copy is already mapped under that name: with this library loaded first and torch imported later that was torch's
        // librccl on top of torch's never-initialised runtime, and ncclCommInitRank failed with 'no ROCm-capable device is
        // detected'.  (Loaded after torch, this library binds to torch's runtime by SONAME and the first choice IS torch's copy.)
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &di) && di.dli_fname) {
            const std::string f(di.dli_fname);
            const size_t p = f.rfind('/');
            if (p != std::string::npos) {
                names.push_back(f.substr(0, p + 1) + "librccl.so.1");
                names.push_back(f.substr(0, p + 1) + "librccl.so");
            }
        }
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) names.push_back(n);
        for (const std::string& name : names) {
            lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            err = std::string("cannot load librccl.so: ") + dlerror();
            return false;
        }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy) {
            err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
            lib = nullptr;
            return false;
        }
        return true;
    }
    std::string what(int rc) { return GetErrorString ? GetErrorString(rc) : ("nccl error " + std::to_string(rc)); }
};
RcclApi g_rccl;
constexpr int NCCL_FLOAT64 = 8;   // ncclDataType_t ncclFloat64 (rccl.h)
}  // namespace

int dsmgp_comm_unique_id(char* id_out) {
    if (!id_out) return DSMGP_E_ARG;
    if (!g_rccl.load()) return fail(nullptr, DSMGP_E_STATE, g_rccl.err);
    RcclApi::UniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, DSMGP_E_HIP, "ncclGetUniqueId: " + g_rccl.what(rc));
    std::memcpy(id_out, id.internal, sizeof(id.internal));
    return 0;
}

int dsmgp_comm_init(dsmgp_ctx* c, int32_t rank, int32_t world, const char* id) {
    if (!c) return DSMGP_E_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, DSMGP_E_ARG, "comm_init: bad arguments");
    if (c->comm) return fail(c, DSMGP_E_STATE, "comm_init: communicator already initialised");
    if (!g_rccl.load()) return fail(c, DSMGP_E_STATE, g_rccl.err);
    HIPCHK(c, hipSetDevice(c->device));
    RcclApi::UniqueId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    const int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (rc != 0) {
        c->comm = nullptr;
        return fail(c, DSMGP_E_HIP, "ncclCommInitRank: " + g_rccl.what(rc));
    }
    c->comm_rank = rank;
    c->comm_world = world;
    return 0;
}

int dsmgp_allgather(dsmgp_ctx* c, const double* send, int64_t count, double* recv) {
    if (!c) return DSMGP_E_ARG;
    if (!c->comm) return fail(c, DSMGP_E_STATE, "allgather before comm_init");
    if (!send || !recv || count <= 0) return fail(c, DSMGP_E_ARG, "allgather: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)count, need = n * (size_t)(c->comm_world + 1);
    if (need > c->xchg_cap) {
        dev_free(c->d_xchg);
        HIPCHK(c, hipMalloc(&c->d_xchg, need * sizeof(double)));
        c->xchg_cap = need;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_xchg, send, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int rc = g_rccl.AllGather(c->d_xchg, c->d_xchg + n, n, NCCL_FLOAT64, c->comm, c->stream);
    if (rc != 0) return fail(c, DSMGP_E_HIP, "ncclAllGather: " + g_rccl.what(rc));
    HIPCHK(c, hipMemcpyAsync(recv, c->d_xchg + n, n * (size_t)c->comm_world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

namespace {
// all-gather of `n` doubles per rank, device to device on the context's stream: d_xchg = [send n | recv world * n]
int xchg_reserve(dsmgp_ctx* c, size_t n) {
    const size_t need = n * (size_t)(c->comm_world + 1);
    if (need > c->xchg_cap) {
        dev_free(c->d_xchg);
        HIPCHK(c, hipMalloc(&c->d_xchg, need * sizeof(double)));
        c->xchg_cap = need;
    }
    return 0;
}
int xchg_gather(dsmgp_ctx* c, size_t n) {
    const int rc = g_rccl.AllGather(c->d_xchg, c->d_xchg + n, n, NCCL_FLOAT64, c->comm, c->stream);
    if (rc != 0) return fail(c, DSMGP_E_HIP, "ncclAllGather: " + g_rccl.what(rc));
    return 0;
}
}  // namespace

int dsmgp_fit_exchange(dsmgp_ctx* c, int64_t count, double* out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->comm) return fail(c, DSMGP_E_STATE, "fit_exchange before comm_init");
    if (!out || count <= 0 || count < c->L) return fail(c, DSMGP_E_ARG, "fit_exchange: count must cover this rank's leaves");
    if (c->L > 0 && !c->fitted) return fail(c, DSMGP_E_STATE, "fit_exchange before fit");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = 2 * (size_t)count;
    if (int rc = xchg_reserve(c, n)) return rc;
    if (c->L > 0) {
        // info lives per factor owner (the table is the plan's: build_plan)
        pack_mll_info_kernel<<<(unsigned)((count + 255) / 256), 256, 0, c->stream>>>(c->d_mll, c->d_info, c->d_owner, c->L, count, c->d_xchg);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemsetAsync(c->d_xchg, 0, n * sizeof(double), c->stream));
    }
    if (int rc = xchg_gather(c, n)) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->d_xchg + n, n * (size_t)c->comm_world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int dsmgp_aggregate_exchange(dsmgp_ctx* c, double* total_out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->comm) return fail(c, DSMGP_E_STATE, "aggregate_exchange before comm_init");
    if (!c->agg_partial_ready) return fail(c, DSMGP_E_STATE, "aggregate_exchange before aggregate_partial");
    if (c->agg_total) return fail(c, DSMGP_E_STATE, "aggregate_exchange: these partial sums already hold the total over ranks");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->agg_W * (size_t)c->n_t;
    if (int rc = xchg_reserve(c, n)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_xchg, c->d_agg_part, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (int rc = xchg_gather(c, n)) return rc;
    sum_ranks_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(c->d_xchg + n, c->comm_world, (int64_t)n, c->d_agg_part);
    HIPCHK(c, hipGetLastError());
    if (total_out) HIPCHK(c, hipMemcpyAsync(total_out, c->d_agg_part, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->agg_total = true;
    return 0;
}

int dsmgp_aggregate_exchange_empty(dsmgp_ctx* c, int32_t W, int64_t n_t, double* total_out) {
    if (!c) return DSMGP_E_ARG;
    if (!c->comm) return fail(c, DSMGP_E_STATE, "aggregate_exchange before comm_init");
    if (W <= 0 || n_t <= 0) return fail(c, DSMGP_E_ARG, "aggregate_exchange_empty: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)W * (size_t)n_t;
    if (int rc = xchg_reserve(c, n)) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_xchg, 0, n * sizeof(double), c->stream));
    if (int rc = xchg_gather(c, n)) return rc;
    if (total_out) {
        sum_ranks_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(c->d_xchg + n, c->comm_world, (int64_t)n, c->d_xchg);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(total_out, c->d_xchg, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int dsmgp_comm_destroy(dsmgp_ctx* c) {
    if (!c) return DSMGP_E_ARG;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    dev_free(c->d_xchg);
    c->xchg_cap = 0;
    c->comm_world = 1;
    c->comm_rank = 0;
    return 0;
}

}  // extern "C"
