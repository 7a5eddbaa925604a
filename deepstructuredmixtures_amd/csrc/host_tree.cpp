// Host-only entry points of the C ABI (include/dsmgp_hip.h): the random partition tree and the "main leaf" search of
// the sharing schedule.  No device is touched and nothing here needs the HIP headers: the file is compiled as plain
// C++ and linked into libdsmgp_hip.so beside dsmgp_hip.cpp (build.sh).
#include "../../include/dsmgp_hip.h"
#include "route_walk.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <unistd.h>

namespace {
// Worker threads of the host routines, one per allowed CPU at most, each pinned to its own CPU: left alone, the
// scheduler starts new threads on the creator's CPU and takes about a second to spread them (measured: the first
// threaded call of a process ran 8 threads at 1.1 CPUs), which is the whole run time of these routines.
int allowed_cpus(std::vector<int>* list = nullptr) {
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &set)) {
                ++n;
                if (list) list->push_back(c);
            }
    if (n == 0) n = (int)std::max(1u, std::thread::hardware_concurrency());
    return n;
}
template <class F>
void run_spread(int nthr, F&& work) {
    if (nthr <= 1) {
        work(0);
        return;
    }
    std::vector<int> cpus;
    allowed_cpus(&cpus);
    // several processes of one node (one per GPU) share an affinity mask: each takes its own group of CPUs
    const size_t groups = std::max<size_t>(1, cpus.size() / (size_t)nthr);
    const size_t first = ((size_t)getpid() % groups) * (size_t)nthr;
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t) {
        th.emplace_back(work, t);
        if ((int)cpus.size() >= nthr) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[first + (size_t)t], &one);
            (void)pthread_setaffinity_np(th.back().native_handle(), sizeof(one), &one);
        }
    }
    for (auto& x : th) x.join();
}
}  // namespace

// Host-only helper of the sharing schedule (src/fit.jl:12-39,78-86) for leaf tables too large for the dense L x L
// overlap matrix: for every leaf j, main[j] = argmax_i D[i,j] D[j,i] with D[a,b] = 1 - (|a| - |a n b|) / |a| for
// overlapping a != b and 0 otherwise (first maximum; 0 when the leaf overlaps nothing, like argmax of a zero column),
// and c_main[j] = |j n main[j]|.  Intersection counts come from an inverted index (point -> leaves, ascending).  The
// product is symmetric in (i, j), so every pair is counted once, from its lower leaf: leaf j walks the leaves l > j of
// its points and offers the pair to both ends.  O(sum_p deg(p)^2 / 2) increments, O(L + sum n) memory, no L x L array;
// threads take chunks of leaves from a shared counter and keep their own candidate tables, merged by the same rule
// (larger product, then lower leaf index), so the result does not depend on the interleaving.  No device is touched.
namespace {
struct MainCand {
    double prod;
    int32_t leaf, count;
};
inline void offer(MainCand& m, double prod, int32_t leaf, int32_t count) {
    if (prod > m.prod || (prod == m.prod && m.prod > 0.0 && leaf < m.leaf)) m = MainCand{prod, leaf, count};
}
template <class CT>
void overlap_main_impl(int32_t L, const int64_t* obs_ptr, const int64_t* obs_idx, const std::vector<int64_t>& pptr,
                       const std::vector<int32_t>& pleaf, int nthr, int64_t* main_out, int64_t* c_main_out) {
    std::vector<std::vector<MainCand>> tabs((size_t)nthr);
    std::atomic<int32_t> next{0};
    const int32_t chunk = 16;
    auto work = [&](int t) {
        std::vector<MainCand>& tab = tabs[(size_t)t];
        tab.assign((size_t)L, MainCand{0.0, 0, 0});
        std::vector<CT> cnt((size_t)L, 0);
        std::vector<int32_t> touched;
        for (;;) {
            const int32_t j0 = next.fetch_add(chunk);
            if (j0 >= L) break;
            const int32_t j1 = std::min(L, j0 + chunk);
            for (int32_t j = j0; j < j1; ++j) {
                touched.clear();
                for (int64_t e = obs_ptr[j]; e < obs_ptr[j + 1]; ++e) {
                    const int64_t p = obs_idx[e];
                    for (int64_t q = pptr[p + 1] - 1; q >= pptr[p]; --q) {
                        const int32_t l = pleaf[q];
                        if (l <= j) break;
                        if (cnt[l]++ == 0) touched.push_back(l);
                    }
                }
                const double nj = (double)(obs_ptr[j + 1] - obs_ptr[j]);
                for (int32_t l : touched) {
                    const int32_t c = (int32_t)cnt[l];
                    cnt[l] = 0;
                    const double nl = (double)(obs_ptr[l + 1] - obs_ptr[l]);
                    const double d_jl = 1.0 - (nj - (double)c) / nj;      // D[j, l]
                    const double d_lj = 1.0 - (nl - (double)c) / nl;      // D[l, j]
                    const double prod = d_lj * d_jl;
                    offer(tab[(size_t)j], prod, l, c);
                    offer(tab[(size_t)l], prod, j, c);
                }
            }
        }
    };
    run_spread(nthr, work);
    for (int32_t j = 0; j < L; ++j) {
        MainCand m{0.0, 0, 0};
        for (int t = 0; t < nthr; ++t) {
            const MainCand& c = tabs[(size_t)t][(size_t)j];
            if (c.prod > 0.0) offer(m, c.prod, c.leaf, c.count);
        }
        main_out[j] = m.prod > 0.0 ? m.leaf : 0;
        c_main_out[j] = m.prod > 0.0 ? m.count : 0;
    }
}
}  // namespace

extern "C" int dsmgp_overlap_main(int32_t L, const int64_t* obs_ptr, const int64_t* obs_idx, int64_t N, int64_t* main_out,
                       int64_t* c_main_out) {
    if (L < 0 || !obs_ptr || (!obs_idx && L > 0 && obs_ptr[L] > 0) || N <= 0 || !main_out || !c_main_out) return DSMGP_E_ARG;
    const int64_t total = L ? obs_ptr[L] : 0;
    int64_t nmax = 0;
    for (int32_t l = 0; l < L; ++l) {
        if (obs_ptr[l + 1] < obs_ptr[l]) return DSMGP_E_ARG;
        nmax = std::max(nmax, obs_ptr[l + 1] - obs_ptr[l]);
    }
    for (int64_t e = 0; e < total; ++e)
        if (obs_idx[e] < 0 || obs_idx[e] >= N) return DSMGP_E_ARG;
    // inverted index: leaves of every point, ascending leaf order
    std::vector<int64_t> pptr(N + 1, 0);
    for (int64_t e = 0; e < total; ++e) pptr[obs_idx[e] + 1]++;
    for (int64_t p = 0; p < N; ++p) pptr[p + 1] += pptr[p];
    std::vector<int32_t> pleaf(total);
    {
        std::vector<int64_t> fill(pptr.begin(), pptr.end() - 1);
        for (int32_t l = 0; l < L; ++l)
            for (int64_t e = obs_ptr[l]; e < obs_ptr[l + 1]; ++e) pleaf[fill[obs_idx[e]]++] = l;
    }
    // one candidate table per thread: bounded at 256 MB in all
    const int64_t by_mem = std::max<int64_t>(1, (int64_t)(256u << 20) / std::max<int64_t>(1, (int64_t)L * (int64_t)sizeof(MainCand)));
    const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>({16, (int64_t)allowed_cpus(), (int64_t)L / 512 + 1, by_mem}));
    // a pair's count is at most the smaller leaf: 16-bit counters (half the cache footprint) whenever that fits
    if (nmax < 65536) overlap_main_impl<uint16_t>(L, obs_ptr, obs_idx, pptr, pleaf, nthr, main_out, c_main_out);
    else overlap_main_impl<int32_t>(L, obs_ptr, obs_idx, pptr, pleaf, nthr, main_out, c_main_out);
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Host-only: the random partition tree of buildTree (src/treeStructure.jl:4-307) -- getSplits (:23-129), _buildSplit
// (:131-210), _buildSum (:212-243), the regions of _buildGP (:245-307) -- as one native recursion over index lists
// instead of one interpreted call per node (26k nodes at depth 4).  Draws come from the portable
// counter stream (deepstructuredmixtures_amd/datagen.py: SplitMix64, draw i = mix(seed + (i+1) GAMMA)) in exactly the
// order of the Python builder tree.py, every floating-point expression is evaluated as NumPy evaluates it (pairwise
// sum of the ranges, median of an even count = (a+b)/2, no fused multiply-add), so both builders return the same tree
// bit for bit (tests/test_host_cpu.py).  No device is touched.
namespace {
#pragma clang fp contract(off)

// NumPy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src): plain loop below 8 elements, eight running sums up to
// 128, halves (the first a multiple of 8 long) above.
static double np_pairwise(const double* a, int64_t n) {
    if (n < 8) {
        double r = -0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

// add.reduce of a contiguous float64 vector as NumPy evaluates it: the reduction walks the vector in chunks of the ufunc buffer
// size (8192 elements, np.getbufsize()), each chunk pairwise, the chunk sums added in order.  Until round 6 this was ONE pairwise
// sum over the whole vector -- equal to NumPy's up to 8192 elements, an ulp off above (found by oracle/tree.py on the headline
// model: the ConstMean of 15 of its 144 leaves, n > 8192).
static double np_sum(const double* a, int64_t n) {
    constexpr int64_t NP_BUFSIZE = 8192;
    double out = 0.0;
    for (int64_t i = 0; i < n; i += NP_BUFSIZE) out += np_pairwise(a + i, std::min(NP_BUFSIZE, n - i));
    return out;
}

struct TreeBuild {
    const double* X;                                     // column-major N x D
    std::vector<double> Xr;                              // row-major copy: the ranges of a region in one pass over its rows
    int64_t N;
    int D, minData, K, V, maxDepth, nKernels;
    double bnoise;
    bool sumRoot;
    uint64_t seed, pos = 0;
    // node table (creation order = pre-order)
    std::vector<int32_t> kind, split_dim, parent;        // kind: 0 region (GP or sum of GPs), 1 split, 2 sum
    std::vector<double> lb, ub;                          // D per node
    std::vector<int64_t> thr_ptr, obs_ptr;               // per node (+1)
    std::vector<double> thr, dir_u;
    std::vector<int32_t> obs;                            // N < 2^31 (checked at the entry point)
    // Work space of build_split, allocated once: the coordinates of a region (vals), their selection inside the bounds
    // (sel) and the child slot of every row (key) are dead before the recursion goes down, so every call uses the same
    // N-sized buffers (uninitialised: a std::vector per call zero-fills what the pass then overwrites and, above the
    // allocator's mmap threshold, faults its pages in again at every node); the children's row lists live across the
    // recursion and come from a stack of blocks that is popped on the way up.
    std::unique_ptr<double[]> vals, sel, sel2;
    std::unique_ptr<int32_t[]> key;
    std::unique_ptr<uint16_t[]> bucket;                  // median_of: bucket of every value (HIST_B <= 65536)
    struct RowStack {
        std::vector<std::unique_ptr<int32_t[]>> block;
        std::vector<size_t> cap;
        size_t cur = 0, top = 0;                         // block in use, entries used in it
        int32_t* push(size_t n) {
            while (cur < block.size() && top + n > cap[cur]) {
                ++cur;
                top = 0;
            }
            if (cur == block.size()) {
                const size_t c = std::max<size_t>(n, (size_t)1 << 20);
                block.emplace_back(new int32_t[c]);
                cap.push_back(c);
                top = 0;
            }
            int32_t* p = block[cur].get() + top;
            top += n;
            return p;
        }
    } rows;

    double rand() {
        uint64_t z = seed + (pos + 1) * 0x9E3779B97F4A7C15ull;
        ++pos;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
    double beta22() {
        double a = rand(), b = rand(), c = rand();
        if (a > b) std::swap(a, b);
        if (b > c) std::swap(b, c);
        if (a > b) std::swap(a, b);
        return b;
    }
    int new_node(int k, int par, const std::vector<double>& l, const std::vector<double>& u) {
        kind.push_back(k);
        split_dim.push_back(-1);
        parent.push_back(par);
        lb.insert(lb.end(), l.begin(), l.end());
        ub.insert(ub.end(), u.begin(), u.end());
        thr_ptr.push_back((int64_t)thr.size());
        obs_ptr.push_back((int64_t)obs.size());
        return (int)kind.size() - 1;
    }
    // Median of the n > 0 values a[0..n) (all inside (l, u]) as NumPy's median returns it: the middle element, or the mean
    // of the two middle ones.  Only the SET matters, so the selection is free to be any exact one: a histogram over
    // (l, u] finds the bucket that holds the middle rank (x -> (x - l) * scale is monotone in floating point: a smaller
    // bucket never holds a larger value), its members are collected into w and the rank is selected among them; the
    // largest value of the buckets below serves the lower middle of an even count when the bucket starts at the rank.
    // Two sequential passes with no data-dependent branch (the bucket of every value is kept from the first for the
    // second) instead of a quickselect's mispredicted ones, which were half of the builder's time at depth 4.  Small or
    // degenerate inputs go to std::nth_element on a copy.
    static constexpr int HIST_B = 1024;
    int32_t hist[HIST_B];
    double median_of(const double* a, int64_t n, double l, double u, double* w) {
        const int64_t h = n / 2;
        int B = 32;                                       // about 8 values per bucket: clearing and scanning the
        while (B < HIST_B && (int64_t)B * 8 < n) B *= 2;  // counters must stay small beside the two passes
        const double scale = (double)B / (u - l);
        double hi, lomax;
        if (n < 192 || !(scale > 0.0) || !std::isfinite(scale)) {
            std::copy(a, a + n, w);
            std::nth_element(w, w + h, w + n);
            hi = w[h];
            lomax = (n % 2) ? hi : *std::max_element(w, w + h);
        } else {
            std::memset(hist, 0, (size_t)B * sizeof(hist[0]));
            uint16_t* const bk = bucket.get();
            for (int64_t i = 0; i < n; ++i) {
                const int k = std::min(B - 1, (int)((a[i] - l) * scale));
                bk[i] = (uint16_t)k;
                hist[k]++;
            }
            int64_t below = 0;
            int kb = 0;
            while (below + hist[kb] <= h) below += hist[kb++];
            int64_t c = 0;
            for (int64_t i = 0; i < n; ++i) {
                w[c] = a[i];
                c += (bk[i] == (uint16_t)kb) ? 1 : 0;
            }
            const int64_t r = h - below;                  // rank inside the bucket
            std::nth_element(w, w + r, w + c);
            hi = w[r];
            lomax = hi;
            if (n % 2 == 0) {
                if (r > 0) {
                    lomax = *std::max_element(w, w + r);
                } else {                                  // the bucket starts at the rank: the largest value below it
                    lomax = -std::numeric_limits<double>::infinity();
                    for (int64_t i = 0; i < n; ++i) lomax = (bk[i] < (uint16_t)kb && a[i] > lomax) ? a[i] : lomax;
                }
            }
        }
        // + 0.0 turns -0.0 into +0.0: which of two signed zeros a selection returns is not defined (NumPy's median and
        // a selection differ), and the sign would reach the stored thresholds through 0 * a + m
        return ((n % 2) ? hi : (lomax + hi) / 2.0) + 0.0;
    }
    // getSplits (src/treeStructure.jl:23-129) on a[0..n): the coordinates of the region inside the caller's (l, u].  The
    // reference filters the whole region again at every recursion level, with l = max(lower, region min) and
    // u = min(upper, region max); the bounds only shrink on the way down and a child's bound is the cut s_new itself, so
    // the two children's selections are exactly the two sides of the parent's selection around s_new: the values are
    // dealt to the two ends of the second buffer w (same length) and each child works on its side, with the roles of
    // the two buffers exchanged -- only when a child is large enough to cut again.  Only the SET matters to every
    // quantity computed here (median, counts).  rmin / rmax: extrema of the WHOLE region on this dimension, as the
    // reference uses them.
    void get_splits(double* a, int64_t n, double* w, double rmin, double rmax, double lower, double upper, int depth,
                    std::vector<double>& s) {
        int K_ = depth * depth;
        const double l = std::max(lower, rmin), u = std::min(upper, rmax);
        const double v = u - l;
        if (n <= 2 * (int64_t)minData) return;
        const double m = median_of(a, n, l, u, w);
        int64_t z1 = 0, z2 = 0;
        int cnt = 0;
        double s_new = m;
        while (z1 == 0 || z2 == 0) {
            const double x = beta22() * v + l;
            const double t1 = bnoise * x, t2 = (1.0 - bnoise) * m;
            s_new = t1 + t2;
            const double cut = s_new;
            int64_t le = 0;
            for (int64_t i = 0; i < n; ++i) le += (a[i] <= cut) ? 1 : 0;
            z1 = le;
            z2 = n - z1;
            if (++cnt > 100) return;
        }
        const bool first_low = (1 + (int)(rand() * 2.0)) == 1;
        bool dealt = false;
        for (int posn = 0; posn < 2; ++posn) {
            const bool low = (posn == 0) == first_low;
            const int64_t z = low ? z1 : z2;
            if (z > minData && K_ < K) {
                if (z > 2 * (int64_t)minData) {           // (a smaller side returns at once and draws nothing)
                    if (!dealt) {
                        const double cut = s_new;
                        int64_t lo = 0, hi = 0;
                        for (int64_t i = 0; i < n; ++i) {
                            const double x = a[i];
                            const bool le = x <= cut;
                            w[lo] = x;
                            w[n - 1 - hi] = x;            // one of the two stays: lo + hi <= i, the cursors never cross
                            lo += le ? 1 : 0;
                            hi += le ? 0 : 1;
                        }
                        dealt = true;
                    }
                    if (low) get_splits(w, z1, a, rmin, rmax, lower, s_new, depth + 1, s);
                    else get_splits(w + z1, z2, a + z1, rmin, rmax, s_new, upper, depth + 1, s);
                }
                if (posn == 0) K_ += 1;
            }
        }
        s.push_back(s_new);
    }
    void build_gp(int par, const int32_t* idx, int64_t n, const std::vector<double>& l, const std::vector<double>& u) {
        new_node(0, par, l, u);
        obs.insert(obs.end(), idx, idx + n);
        for (int k = 0; k < nKernels; ++k) dir_u.push_back(rand());
    }
    void build_split(int par, const int32_t* idx, int64_t n, const std::vector<double>& lowerBound,
                     const std::vector<double>& upperBound, int depth, int d) {
        const double* xd = X + (size_t)d * N;
        double* const vals = this->vals.get();
        double rmin = xd[idx[0]], rmax = rmin;
        for (int64_t q = 0; q < n; ++q) {
            if (q + 24 < n) __builtin_prefetch(xd + idx[q + 24]);      // rows of a deep region lie a page apart
            const double x = xd[idx[q]];
            vals[q] = x;
            rmin = std::min(rmin, x);
            rmax = std::max(rmax, x);
        }
        std::vector<double> s;
        {
            const double l = std::max(lowerBound[d], rmin), u = std::min(upperBound[d], rmax);
            double* const sel = this->sel.get();
            int64_t ns = 0;
            for (int64_t q = 0; q < n; ++q) {
                const double x = vals[q];
                sel[ns] = x;
                ns += (x > l && x <= u) ? 1 : 0;
            }
            get_splits(sel, ns, sel2.get(), rmin, rmax, lowerBound[d], upperBound[d], 1, s);
        }
        std::sort(s.begin(), s.end());
        const double lo = lowerBound[d], up = upperBound[d];
        if (s.empty()) {
            int32_t* const sub = key.get();
            int64_t m = 0;
            for (int64_t q = 0; q < n; ++q)
                if (vals[q] > lo && vals[q] <= up) sub[m++] = idx[q];
            build_gp(par, sub, m, lowerBound, upperBound);
            return;
        }
        const int id = new_node(1, par, lowerBound, upperBound);
        split_dim[id] = d;
        s.push_back(up);
        thr.insert(thr.end(), s.begin(), s.end());
        // child k holds the points with s[k-1] < x <= s[k] (s[-1] = lowerBound[d]): k = number of cuts below x; a
        // counting pass and a scatter into one buffer, the children are spans of it
        const size_t nc = s.size();
        int32_t* const key = this->key.get();
        std::vector<int64_t> off(nc + 2, 0);
        for (int64_t q = 0; q < n; ++q) {
            const double x = vals[q];
            size_t k = 0;
            if (nc <= 16)
                for (size_t j = 0; j < nc; ++j) k += (s[j] < x) ? 1 : 0;
            else
                k = (size_t)(std::lower_bound(s.begin(), s.end(), x) - s.begin());
            if (!(x > lo)) k = nc;                       // slot nc: outside (lowerBound, upperBound]
            key[q] = (int32_t)k;
            off[k + 1]++;
        }
        for (size_t k = 0; k <= nc; ++k) off[k + 1] += off[k];
        const size_t mark_cur = rows.cur, mark_top = rows.top;
        int32_t* const buf = rows.push((size_t)off[nc] + 1);
        {
            std::vector<int64_t> fill(off.begin(), off.begin() + nc + 1);
            fill[nc] = off[nc];                          // rows outside the bounds land in one spare entry
            for (int64_t q = 0; q < n; ++q) {
                const size_t k = (size_t)key[q];
                buf[fill[k]] = idx[q];
                fill[k] += k < nc ? 1 : 0;
            }
        }
        std::vector<double> lb_(lowerBound), ub_(upperBound);
        for (size_t k = 0; k < nc; ++k) {
            const double si = s[k];
            ub_[d] = si;
            const int32_t* sub = buf + off[k];
            const int64_t m = off[k + 1] - off[k];
            if (depth < maxDepth && m > (int64_t)minData) {
                if (sumRoot) build_sum(id, sub, m, lb_, ub_, depth);
                else build_split(id, sub, m, lb_, ub_, depth, 0);
            } else {
                build_gp(id, sub, m, lb_, ub_);
            }
            lb_[d] = si;
        }
        rows.cur = mark_cur;
        rows.top = mark_top;
    }
    void build_sum(int par, const int32_t* idx, int64_t n, const std::vector<double>& lowerBound,
                   const std::vector<double>& upperBound, int depth) {
        const int id = new_node(2, par, lowerBound, upperBound);
        std::vector<double> mn(Xr.begin() + (size_t)idx[0] * D, Xr.begin() + (size_t)(idx[0] + 1) * D), mx(mn), phi(D);
        for (int64_t q = 0; q < n; ++q) {
            if (q + 16 < n) __builtin_prefetch(Xr.data() + (size_t)idx[q + 16] * D);
            const double* row = Xr.data() + (size_t)idx[q] * D;
            for (int d = 0; d < D; ++d) {
                mn[d] = std::min(mn[d], row[d]);
                mx[d] = std::max(mx[d], row[d]);
            }
        }
        for (int d = 0; d < D; ++d) phi[d] = mx[d] - mn[d];
        const double tot = np_sum(phi.data(), (int64_t)D);
        if (tot > 0.0)
            for (int d = 0; d < D; ++d) phi[d] = phi[d] / tot;
        else
            for (int d = 0; d < D; ++d) phi[d] = 1.0 / (double)D;
        for (int v = 0; v < V; ++v) {
            // Categorical(phi): cumulative sums, first index with c[i] > u c[-1]
            std::vector<double> cs(D);
            double acc = 0.0;
            for (int d = 0; d < D; ++d) {
                acc += phi[d];
                cs[d] = acc;
            }
            const double x = rand() * cs[D - 1];
            int dsel = (int)(std::upper_bound(cs.begin(), cs.end(), x) - cs.begin());
            dsel = std::min(dsel, D - 1);
            build_split(id, idx, n, lowerBound, upperBound, depth + 1, dsel);
        }
    }
};
}  // namespace

struct dsmgp_tree {
    TreeBuild b;
};

extern "C" {

int dsmgp_tree_build(const double* X, int64_t N, int32_t D, int32_t min_data, int32_t n_splits, int32_t n_sum_children,
                     int32_t depth, double bnoise, int32_t sum_root, int32_t n_kernels, uint64_t seed, dsmgp_tree** out) {
    if (!X || N <= 0 || N > (int64_t)std::numeric_limits<int32_t>::max() || D <= 0 || !out || min_data < 0 || n_splits < 1 ||
        n_sum_children < 1 || depth < 0 || n_kernels < 0)
        return DSMGP_E_ARG;
    for (int64_t i = 0; i < N * (int64_t)D; ++i)
        if (!std::isfinite(X[i])) return DSMGP_E_ARG;
    dsmgp_tree* t = new dsmgp_tree();
    TreeBuild& b = t->b;
    b.X = X;
    b.N = N;
    b.D = D;
    b.minData = min_data;
    b.K = n_splits;
    b.V = n_sum_children;
    b.maxDepth = depth;
    b.bnoise = bnoise;
    b.sumRoot = sum_root != 0;
    b.nKernels = n_kernels;
    b.seed = seed;
    if (b.sumRoot) {
        b.Xr.resize((size_t)N * D);
        for (int d = 0; d < D; ++d)
            for (int64_t i = 0; i < N; ++i) b.Xr[(size_t)i * D + d] = X[(size_t)d * N + i];
        // every level of sum nodes multiplies the rows held by regions by V (at most: regions can only lose rows)
        double rows = (double)N;
        for (int l = 0; l < depth && rows < 1e9; ++l) rows *= (double)n_sum_children;
        if (rows < 1e9) b.obs.reserve((size_t)rows);
    }
    b.vals.reset(new double[(size_t)N]);
    b.sel.reset(new double[(size_t)N]);
    b.sel2.reset(new double[(size_t)N]);
    b.key.reset(new int32_t[(size_t)N]);
    b.bucket.reset(new uint16_t[(size_t)N]);
    std::vector<int32_t> all((size_t)N);
    for (int64_t i = 0; i < N; ++i) all[i] = (int32_t)i;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> lbv(D, -inf), ubv(D, inf);
    if (b.sumRoot) b.build_sum(-1, all.data(), N, lbv, ubv, 0);
    else b.build_split(-1, all.data(), N, lbv, ubv, 0, 0);
    b.thr_ptr.push_back((int64_t)b.thr.size());
    b.obs_ptr.push_back((int64_t)b.obs.size());
    b.X = nullptr;
    std::vector<double>().swap(b.Xr);
    b.vals.reset();
    b.sel.reset();
    b.sel2.reset();
    b.key.reset();
    b.bucket.reset();
    b.rows = TreeBuild::RowStack();
    *out = t;
    return 0;
}

int dsmgp_tree_sizes(const dsmgp_tree* t, int64_t* n_nodes, int64_t* n_thr, int64_t* n_obs, int64_t* n_dir) {
    if (!t) return DSMGP_E_ARG;
    if (n_nodes) *n_nodes = (int64_t)t->b.kind.size();
    if (n_thr) *n_thr = (int64_t)t->b.thr.size();
    if (n_obs) *n_obs = (int64_t)t->b.obs.size();
    if (n_dir) *n_dir = (int64_t)t->b.dir_u.size();
    return 0;
}

int dsmgp_tree_export(const dsmgp_tree* t, int32_t* kind, int32_t* parent, int32_t* split_dim, double* lb, double* ub,
                      int64_t* thr_ptr, double* thr, int64_t* obs_ptr, int64_t* obs, double* dir_u) {
    if (!t || !kind || !parent || !split_dim || !lb || !ub || !thr_ptr || !obs_ptr) return DSMGP_E_ARG;
    const TreeBuild& b = t->b;
    const size_t n = b.kind.size();
    std::memcpy(kind, b.kind.data(), n * sizeof(int32_t));
    std::memcpy(parent, b.parent.data(), n * sizeof(int32_t));
    std::memcpy(split_dim, b.split_dim.data(), n * sizeof(int32_t));
    std::memcpy(lb, b.lb.data(), b.lb.size() * sizeof(double));
    std::memcpy(ub, b.ub.data(), b.ub.size() * sizeof(double));
    std::memcpy(thr_ptr, b.thr_ptr.data(), (n + 1) * sizeof(int64_t));
    std::memcpy(obs_ptr, b.obs_ptr.data(), (n + 1) * sizeof(int64_t));
    if (thr && !b.thr.empty()) std::memcpy(thr, b.thr.data(), b.thr.size() * sizeof(double));
    if (obs)
        for (size_t e = 0; e < b.obs.size(); ++e) obs[e] = (int64_t)b.obs[e];
    if (dir_u && !b.dir_u.empty()) std::memcpy(dir_u, b.dir_u.data(), b.dir_u.size() * sizeof(double));
    return 0;
}

int dsmgp_tree_means(const dsmgp_tree* t, const double* y, int64_t N, double* mean_out) {
    if (!t || !y || !mean_out || N <= 0) return DSMGP_E_ARG;
    const TreeBuild& b = t->b;
    if (N != b.N) return DSMGP_E_ARG;
    std::vector<int64_t> reg;                       // node index of every region, creation order
    for (size_t i = 0; i < b.kind.size(); ++i)
        if (b.kind[i] == 0) reg.push_back((int64_t)i);
    const int64_t R = (int64_t)reg.size();
    const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>({16, (int64_t)allowed_cpus(), R / 256 + 1}));
    auto work = [&](int t) {
        std::vector<double> g;
        for (int64_t r = R * t / nthr; r < R * (t + 1) / nthr; ++r) {
            const int64_t e0 = b.obs_ptr[reg[r]], e1 = b.obs_ptr[reg[r] + 1];
            g.resize((size_t)(e1 - e0));
            for (int64_t e = e0; e < e1; ++e) g[e - e0] = y[b.obs[e]];
            mean_out[r] = e1 > e0 ? np_sum(g.data(), e1 - e0) / (double)(e1 - e0) : 0.0;
        }
    };
    run_spread(nthr, work);
    return 0;
}

int dsmgp_tree_free(dsmgp_tree* t) {
    delete t;
    return 0;
}

}  // extern "C"

// Routing of test rows (src/common.jl:181-196,275-292) over the flat tree: a sum node forwards its rows to every child, a split
// node to the first child whose threshold is not below x[d].  Row by row with the walk the device kernels run
// (route_walk.hpp: route_walk_row), rows ascending -- so every leaf's list comes out ascending: one pass counts, one fills.
// (Until round 5 this was a partition of the row set level by level; the per-row walk takes the same 14 ms for 10k rows at
// depth 4 and is the code the device path shares.)
extern "C" int dsmgp_tree_route(int64_t n_nodes, const int8_t* kind, const int64_t* first_child, const int64_t* n_child,
                                const int64_t* split_dim, const double* thr, int64_t thr_ld, const int64_t* leaf_id,
                                int64_t n_leaves, const double* x, int64_t n_t, int64_t D, int64_t row_stride,
                                int64_t col_stride, int64_t* route_ptr, int64_t* route_idx, int64_t capacity,
                                int64_t* n_routes_out) {
    if (n_nodes <= 0 || n_nodes > (int64_t)INT32_MAX || !kind || !first_child || !n_child || !split_dim || !thr || !leaf_id ||
        n_leaves < 0 || n_leaves > (int64_t)INT32_MAX || n_t < 0 || D <= 0 || thr_ld <= 0 || (n_t > 0 && !x) || !route_ptr ||
        capacity < 0 || (capacity > 0 && !route_idx))
        return -1;      // DSMGP_E_ARG
    std::vector<int32_t> first((size_t)n_nodes), nch((size_t)n_nodes), sdim((size_t)n_nodes), leaf((size_t)n_nodes), need((size_t)n_nodes);
    for (int64_t i = 0; i < n_nodes; ++i) {
        if (kind[i] < 0 || kind[i] > 2) return -1;
        if (kind[i] == 0 ? (leaf_id[i] < 0 || leaf_id[i] >= n_leaves)
                         : (n_child[i] <= 0 || first_child[i] <= i || first_child[i] + n_child[i] > n_nodes))
            return -1;
        if (kind[i] == 1 && (n_child[i] > thr_ld || split_dim[i] < 0 || split_dim[i] >= D)) return -1;   // x has D columns
        first[(size_t)i] = (int32_t)first_child[i];
        nch[(size_t)i] = (int32_t)n_child[i];
        sdim[(size_t)i] = (int32_t)split_dim[i];
        leaf[(size_t)i] = kind[i] == 0 ? (int32_t)leaf_id[i] : -1;
    }
    // the walk's pending-node stack on the heap, sized by THIS tree: the host routine takes any tree (the device walk has the
    // fixed ROUTE_STACK and dsmgp_set_tree refuses a tree beyond it -- the caller then routes here)
    std::vector<int32_t> stack((size_t)dsmgp::route_stack_need(n_nodes, kind, first.data(), nch.data(), need.data()));
    const dsmgp::RouteTree t{kind, first.data(), nch.data(), sdim.data(), leaf.data(), thr, (int)thr_ld};
    std::vector<int64_t> cnt((size_t)n_leaves, 0);
    for (int64_t r = 0; r < n_t; ++r)
        if (dsmgp::route_walk_row_on(stack.data(), t, x, row_stride, col_stride, r, [&](int l, int) { ++cnt[(size_t)l]; }) < 0)
            return -6;  // DSMGP_E_DOMAIN: a row outside the region of a split node (NaN included), not a malformed call
    route_ptr[0] = 0;
    for (int64_t l = 0; l < n_leaves; ++l) route_ptr[l + 1] = route_ptr[l] + cnt[(size_t)l];
    if (n_routes_out) *n_routes_out = route_ptr[n_leaves];
    if (capacity < route_ptr[n_leaves]) return -4;  // DSMGP_E_NOMEM: *n_routes_out says how much
    std::vector<int64_t> fill(route_ptr, route_ptr + n_leaves);
    for (int64_t r = 0; r < n_t; ++r)
        (void)dsmgp::route_walk_row_on(stack.data(), t, x, row_stride, col_stride, r, [&](int l, int) { route_idx[fill[(size_t)l]++] = r; });
    return 0;
}
