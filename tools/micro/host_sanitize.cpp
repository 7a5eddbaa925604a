#include "../../deepstructuredmixtures_amd/csrc/host_tree.cpp"
#include <cstdio>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0, 1);
    struct Cfg { int64_t N; int D, M, K, V, depth, nk, sum; };
    Cfg cfgs[] = {{3000, 3, 20, 4, 3, 3, 0, 1}, {2500, 16, 15, 4, 3, 2, 2, 1}, {4000, 2, 50, 8, 1, 2, 0, 0}, {100, 1, 10, 4, 3, 2, 0, 1},
                  {6000, 8, 60, 4, 3, 4, 0, 1}, {5000, 9, 30, 5, 2, 3, 0, 1}, {70000, 2, 30000, 2, 2, 1, 0, 1}, {5, 2, 1, 2, 2, 2, 1, 1}};
    for (auto& c : cfgs) {
        std::vector<double> X(c.N * c.D), y(c.N);
        for (auto& v : X) v = U(rng);
        for (auto& v : y) v = U(rng);
        dsmgp_tree* t = nullptr;
        int rc = dsmgp_tree_build(X.data(), c.N, c.D, c.M, c.K, c.V, c.depth, 0.5, c.sum, c.nk, 11, &t);
        if (rc) { printf("rc %d\n", rc); return 1; }
        int64_t nn, nt, no, nd; dsmgp_tree_sizes(t, &nn, &nt, &no, &nd);
        std::vector<int32_t> kind(nn), par(nn), sd(nn);
        std::vector<double> lb(nn * c.D), ub(nn * c.D), thr(nt + 1), du(nd + 1);
        std::vector<int64_t> tp(nn + 1), op(nn + 1), obs(no + 1);
        dsmgp_tree_export(t, kind.data(), par.data(), sd.data(), lb.data(), ub.data(), tp.data(), thr.data(), op.data(), obs.data(), du.data());
        int64_t R = 0; for (auto k : kind) R += k == 0;
        std::vector<double> mean(R + 1);
        dsmgp_tree_means(t, y.data(), c.N, mean.data());
        std::vector<int64_t> lptr{0}, lidx;
        for (int64_t i = 0; i < nn; ++i) if (kind[i] == 0 && op[i + 1] > op[i]) { lidx.insert(lidx.end(), obs.begin() + op[i], obs.begin() + op[i + 1]); lptr.push_back((int64_t)lidx.size()); }
        int L = (int)lptr.size() - 1;
        std::vector<int64_t> mn(L + 1), cm(L + 1);
        rc = dsmgp_overlap_main(L, lptr.data(), lidx.data(), c.N, mn.data(), cm.data());
        printf("N %ld D %d: nodes %ld regions %ld leaves %d overlap rc %d mean0 %.6f main0 %ld\n", (long)c.N, c.D, (long)nn, (long)R, L, rc, mean[0], (long)mn[0]);
        {   // dsmgp_tree_route on the same tree: breadth-first renumbering (children consecutive), leaves = regions in order
            std::vector<std::vector<int64_t>> ch(nn);
            for (int64_t i = 1; i < nn; ++i) ch[par[i]].push_back(i);
            std::vector<int64_t> order{0}, newid(nn, -1), region(nn, -1);
            int64_t rcount = 0;
            for (int64_t i = 0; i < nn; ++i) if (kind[i] == 0) region[i] = rcount++;
            for (size_t q = 0; q < order.size(); ++q) { newid[order[q]] = (int64_t)q; for (auto cidx : ch[order[q]]) order.push_back(cidx); }
            int64_t width = 1;
            for (int64_t i = 0; i < nn; ++i) width = std::max<int64_t>(width, tp[i + 1] - tp[i]);
            std::vector<int8_t> k2(nn); std::vector<int64_t> first(nn), nch(nn), sd2(nn), leaf(nn, -1);
            std::vector<double> th2(nn * width, std::numeric_limits<double>::infinity());
            size_t next = 1;
            for (size_t q = 0; q < order.size(); ++q) {
                const int64_t i = order[q];
                k2[q] = (int8_t)kind[i]; nch[q] = (int64_t)ch[i].size(); first[q] = (int64_t)next; next += ch[i].size();
                sd2[q] = kind[i] == 1 ? sd[i] : 0; leaf[q] = region[i];
                for (int64_t e = tp[i]; e < tp[i + 1] && kind[i] == 1; ++e) th2[q * width + (e - tp[i])] = thr[e];
            }
            const int64_t nq = 257;
            std::vector<double> xq(nq * c.D);
            for (auto& v : xq) v = U(rng);
            std::vector<int64_t> rp(rcount + 1), ri((size_t)nq * (size_t)rcount + 1);
            int64_t nr = 0;
            rc = dsmgp_tree_route(nn, k2.data(), first.data(), nch.data(), sd2.data(), th2.data(), width, leaf.data(), rcount,
                                  xq.data(), nq, c.D, c.D, 1, rp.data(), ri.data(), (int64_t)ri.size(), &nr);
            printf("   route rc %d routes %ld (%.1f per row)\n", rc, (long)nr, (double)nr / nq);
        }
        dsmgp_tree_free(t);
    }
    return 0;
}
