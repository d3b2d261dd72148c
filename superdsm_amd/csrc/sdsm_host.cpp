// Host-side combinatorial steps of the global-energy-minimisation stage as native code (no device access): the approximate
// min-weight set cover (reference: superdsm/minsetcover.py:4-88, Algorithm 2 of Kostrykin & Rohr, TPAMI 2023) and the greedy
// max-weight set packing (superdsm/maxsetpack.py:8-24).  They stay on the host by design (BASELINE.json north_star); once the
// candidate solves run on the GPU the Python loops of the set cover are the largest part of the stage's wall clock per image.
// Same decisions as the Python restatement in superdsm_amd/minsetcover.py / maxsetpack.py, which follows the reference: same
// arithmetic (IEEE double, same order of the sums), same tie-breaking (first minimum / maximum in list order, stable sort).
// Footprints are bit sets over the atoms of one cluster: `words` uint64 per object.
#include <algorithm>
#include <cstdint>
#include <unordered_set>
#include <vector>

#include "../../include/sdsm.h"

namespace {

struct Family {
    int n, words;
    const uint64_t *foot;
    const double *energy;
    const uint64_t *fp(int i) const { return foot + (size_t)i * words; }
};

inline int popcount_and(const uint64_t *a, const uint64_t *b, int words)
{
    int c = 0;
    for (int w = 0; w < words; w++) c += __builtin_popcountll(a[w] & b[w]);
    return c;
}
inline bool disjoint(const uint64_t *a, const uint64_t *b, int words)
{
    for (int w = 0; w < words; w++) if (a[w] & b[w]) return false;
    return true;
}
inline bool subset(const uint64_t *a, const uint64_t *b, int words)   // a <= b
{
    for (int w = 0; w < words; w++) if (a[w] & ~b[w]) return false;
    return true;
}

// greedy phase + merge phase for one beta (minsetcover.py:24-50)
std::vector<int> solve_once(const Family &F, double beta, bool merge)
{
    const int W = F.words;
    std::vector<uint64_t> uncovered(W, 0);
    for (int i = 0; i < F.n; i++) for (int w = 0; w < W; w++) uncovered[w] |= F.fp(i)[w];
    std::vector<int> cand(F.n), accepted;
    for (int i = 0; i < F.n; i++) cand[i] = i;
    while (!cand.empty()) {
        int best = -1;
        double best_price = 0;
        for (int c : cand) {
            const double price = (F.energy[c] + beta) / popcount_and(F.fp(c), uncovered.data(), W);
            if (best < 0 || price < best_price) { best = c; best_price = price; }     // first minimum in list order
        }
        accepted.push_back(best);
        for (int w = 0; w < W; w++) uncovered[w] &= ~F.fp(best)[w];
        std::vector<int> rest;
        for (int c : cand) if (!disjoint(F.fp(c), uncovered.data(), W)) rest.push_back(c);
        cand.swap(rest);
    }
    if (!merge) return accepted;
    std::vector<char> taken(F.n, 0);
    for (int a : accepted) taken[a] = 1;
    std::vector<int> others;
    for (int i = 0; i < F.n; i++) if (!taken[i]) others.push_back(i);
    std::stable_sort(others.begin(), others.end(), [&](int a, int b) { return F.energy[a] + beta < F.energy[b] + beta; });
    for (int nw : others) {
        std::vector<int> inside;
        bool valid = true;
        for (int c : accepted) {
            if (disjoint(F.fp(c), F.fp(nw), W)) continue;
            if (!subset(F.fp(c), F.fp(nw), W)) { valid = false; break; }           // partial overlap: not a valid replacement
            inside.push_back(c);
        }
        if (!valid) continue;
        double sum = 0;                                                             // Python's sum(): left to right from 0
        for (int c : inside) sum += F.energy[c] + beta;
        if (F.energy[nw] + beta < sum) {
            std::vector<int> next;
            for (int c : accepted) if (std::find(inside.begin(), inside.end(), c) == inside.end()) next.push_back(c);
            next.push_back(nw);
            accepted.swap(next);
        }
    }
    return accepted;
}

double price_of(const Family &F, const std::vector<int> &sol, double beta)
{
    double s = 0;
    for (int c : sol) s += F.energy[c];
    return s + beta * (double)sol.size();
}

std::vector<int> solve_cover(const Family &F, double beta, bool merge, int max_iter, double gamma)
{
    std::vector<int> solution = solve_once(F, beta, merge);
    if (max_iter > 1 && beta > 0) {
        std::vector<int> retry = solve_cover(F, beta * gamma, merge, max_iter - 1, gamma);
        if (price_of(F, retry, beta) < price_of(F, solution, beta)) return retry;   // priced with the beta of THIS level
    }
    return solution;
}

}  // namespace

extern "C" int sdsm_minsetcover(int n, int words, const uint64_t *footprints, const double *energies, double beta, int merge, int max_iter,
                                double gamma, int32_t *selected, int32_t *n_selected)
{
    if (n < 0 || words < 1 || !selected || !n_selected || (n > 0 && (!footprints || !energies)) || !(beta >= 0) || !(gamma > 0 && gamma < 1)) return SDSM_ERR_ARGUMENT;
    Family F{n, words, footprints, energies};
    std::vector<int> sol = n > 0 ? solve_cover(F, beta, merge != 0, max_iter, gamma) : std::vector<int>();
    *n_selected = (int32_t)sol.size();
    for (size_t i = 0; i < sol.size(); i++) selected[i] = sol[i];
    return SDSM_OK;
}

// The same for SEVERAL independent families in one call (MinSetCover.update: one cover per touched cluster and generation -- a hundred
// calls per image, each paying the trip through the foreign-function interface): family f has n[f] objects of words[f] uint64 each, its
// footprints / energies / selected indices start at the running sums of n[f] * words[f] / n[f] / n[f].
extern "C" int sdsm_minsetcover_multi(int n_families, const int32_t *n, const int32_t *words, const uint64_t *footprints, const double *energies, double beta,
                                      int merge, int max_iter, double gamma, int32_t *selected, int32_t *n_selected)
{
    if (n_families < 0 || (n_families > 0 && (!n || !words || !footprints || !energies || !selected || !n_selected)) || !(beta >= 0) || !(gamma > 0 && gamma < 1)) return SDSM_ERR_ARGUMENT;
    size_t fo = 0, eo = 0;
    for (int f = 0; f < n_families; f++) {
        if (n[f] < 0 || words[f] < 1) return SDSM_ERR_ARGUMENT;
        Family F{n[f], words[f], footprints + fo, energies + eo};
        std::vector<int> sol = n[f] > 0 ? solve_cover(F, beta, merge != 0, max_iter, gamma) : std::vector<int>();
        n_selected[f] = (int32_t)sol.size();
        for (size_t i = 0; i < sol.size(); i++) selected[eo + i] = sol[i];
        fo += (size_t)n[f] * words[f];
        eo += (size_t)n[f];
    }
    return SDSM_OK;
}

extern "C" int sdsm_maxsetpack(int n, int words, const uint64_t *footprints, const double *energies, int32_t *selected, int32_t *n_selected)
{
    if (n < 0 || words < 1 || !selected || !n_selected || (n > 0 && (!footprints || !energies))) return SDSM_ERR_ARGUMENT;
    Family F{n, words, footprints, energies};
    std::vector<int> pool(n), packed;
    for (int i = 0; i < n; i++) pool[i] = i;
    while (!pool.empty()) {
        int top = pool[0];
        for (int c : pool) if (energies[c] > energies[top]) top = c;                 // first maximum in list order
        packed.push_back(top);
        std::vector<int> rest;
        for (int c : pool) if (disjoint(F.fp(c), F.fp(top), words)) rest.push_back(c);
        pool.swap(rest);
    }
    *n_selected = (int32_t)packed.size();
    for (size_t i = 0; i < packed.size(); i++) selected[i] = packed[i];
    return SDSM_OK;
}

// Size of the search space of the iterations (globalenergymin.py:292-323, _estimate_progress from the atoms on): for every cluster
// the number of footprints that growing the single atoms by one adjacent atom at a time produces, generation after generation
// (de-duplicated within a generation) -- the connected atom subsets of 2 .. n atoms (n - 1 with skip_last: the universe is computed
// separately) whose atoms are pairwise compatible (seed distance).  Atoms of cluster k are offsets[k] .. offsets[k + 1] - 1; adj /
// compat hold, per atom, the bit set of its neighbours / of the atoms it may share a footprint with, as LOCAL bit indices of its
// cluster (compat == NULL: all).  Clusters of more than 64 atoms get -1 (the caller enumerates them itself).  Stops counting once the
// total exceeds max_amount (the caller raises, as the reference does).
extern "C" int sdsm_count_growth(int n_clusters, const int32_t *offsets, const uint64_t *adj, const uint64_t *compat, int skip_last,
                                 int64_t max_amount, int64_t *counts)
{
    if (n_clusters < 0 || !offsets || !counts || (n_clusters > 0 && !adj)) return SDSM_ERR_ARGUMENT;
    int64_t total = 0;
    for (int k = 0; k < n_clusters; k++) {
        const int lo = offsets[k], n = offsets[k + 1] - lo;
        if (n > 64) { counts[k] = -1; continue; }
        counts[k] = 0;
        std::vector<uint64_t> current(n);
        for (int a = 0; a < n; a++) current[a] = 1ull << a;
        int size = 1;
        while (!current.empty() && total <= max_amount) {
            if (skip_last && size + 1 == n) break;                               // no footprint of this generation is grown
            std::unordered_set<uint64_t> next;
            for (uint64_t fp : current) {
                uint64_t nb = 0;
                for (uint64_t m = fp; m; m &= m - 1) nb |= adj[lo + __builtin_ctzll(m)];
                nb &= ~fp;
                for (uint64_t m = nb; m; m &= m - 1) {
                    const int a = __builtin_ctzll(m);
                    if (compat && (compat[lo + a] & fp) != fp) continue;
                    next.insert(fp | (1ull << a));
                }
            }
            counts[k] += (int64_t)next.size();
            total += (int64_t)next.size();
            current.assign(next.begin(), next.end());
            size++;
        }
    }
    return SDSM_OK;
}
