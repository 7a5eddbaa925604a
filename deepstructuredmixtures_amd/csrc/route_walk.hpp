// The walk of ONE test row through the model's tree (src/common.jl:101-122,181-196,275-292: a sum node forwards the row to every
// child, a split node to the first child k with x[d] <= s_k), shared by the device routing kernels (kernels.hpp,
// route_walk_kernel) and the library's host routine dsmgp_tree_route (host_tree.cpp): one source, so the CPU tests of the host
// routine -- rows on a threshold, +-Inf bounds, NaN, ragged trees, kernel vectors -- are tests of what the device walks.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define DSMGP_HD __host__ __device__
#else
#define DSMGP_HD
#endif

namespace dsmgp {

struct RouteTree {
    const int8_t* kind;       // 0 region (leaf), 1 split, 2 sum; node 0 = root, children of node i at first[i] .. + nchild[i] - 1
    const int32_t* first;
    const int32_t* nchild;
    const int32_t* sdim;
    const int32_t* leaf;      // regions: index in the caller's leaf table, -1 = not held here (another rank): skipped
    const double* thr;        // split nodes: ascending thresholds thr[i * thr_ld + 0 .. nchild[i] - 1], the last = upper bound
    int thr_ld;
};
constexpr int ROUTE_STACK = 96;       // pending nodes of one row's walk (route_stack_need checks a tree against it)

// Depth-first, children in order: leaves are numbered depth-first, so the row meets its leaves in ascending leaf order.
// visit(leaf, i) for the i-th leaf reached; returns their number, or -1 when the row lies outside the region of a split node
// (beyond its last threshold -- the reference loops forever there; a NaN coordinate is beyond every threshold).
// `stack` holds at least route_stack_need(tree) entries: the device walk's fixed ROUTE_STACK (route_walk_row below), the host
// routine's heap buffer sized by the tree (dsmgp_tree_route: any tree).
template <class Visit>
DSMGP_HD inline int route_walk_row_on(int32_t* stack, const RouteTree& t, const double* x, int64_t row_stride, int64_t col_stride, int64_t r,
                                      Visit&& visit) {
    int sp = 0, cnt = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const int node = stack[--sp];
        const int kd = t.kind[node];
        if (kd == 0) {
            const int l = t.leaf[node];
            if (l >= 0) {
                visit(l, cnt);
                ++cnt;
            }
        } else if (kd == 2) {
            const int c0 = t.first[node];
            for (int j = t.nchild[node] - 1; j >= 0; --j) stack[sp++] = c0 + j;       // popped in child order
        } else {
            const double v = x[r * row_stride + (int64_t)t.sdim[node] * col_stride];
            const double* th = t.thr + (int64_t)node * t.thr_ld;
            const int nc = t.nchild[node];
            int k = 0;
            while (k < nc && !(v <= th[k])) ++k;          // first k with v <= s_k (np.searchsorted(th, v, side = "left"))
            if (k >= nc) return -1;
            stack[sp++] = t.first[node] + k;
        }
    }
    return cnt;
}

template <class Visit>
DSMGP_HD inline int route_walk_row(const RouteTree& t, const double* x, int64_t row_stride, int64_t col_stride, int64_t r, Visit&& visit) {
    int32_t stack[ROUTE_STACK];
    return route_walk_row_on(stack, t, x, row_stride, col_stride, r, visit);
}

// Most pending nodes any row's walk can hold: need[i] over the subtree of node i, children after their parents (one backward
// pass).  The caller compares need[0] (at least 1: the root) with ROUTE_STACK.
inline int route_stack_need(int64_t n_nodes, const int8_t* kind, const int32_t* first, const int32_t* nchild, int32_t* need) {
    for (int64_t i = n_nodes - 1; i >= 0; --i) {
        need[i] = 0;
        if (kind[i] == 0) continue;
        const int32_t c0 = first[i], nc = nchild[i];
        int32_t m = kind[i] == 2 ? nc : 1;
        for (int32_t j = 0; j < nc; ++j) {
            const int32_t v = (kind[i] == 2 ? nc - 1 - j : 0) + need[c0 + j];
            m = v > m ? v : m;
        }
        need[i] = m;
    }
    return need[0] > 1 ? need[0] : 1;
}

}  // namespace dsmgp
