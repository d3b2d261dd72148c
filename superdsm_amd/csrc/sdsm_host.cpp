// Host-side combinatorial steps of the global-energy-minimisation stage as native code (no device access): the approximate
// min-weight set cover (reference: superdsm/minsetcover.py:4-88, Algorithm 2 of Kostrykin & Rohr, TPAMI 2023) and the greedy
// max-weight set packing (superdsm/maxsetpack.py:8-24).  They stay on the host by design (BASELINE.json north_star); once the
// candidate solves run on the GPU the Python loops of the set cover are the largest part of the stage's wall clock per image.
// Same decisions as the Python restatement in superdsm_amd/minsetcover.py / maxsetpack.py, which follows the reference: same
// arithmetic (IEEE double, same order of the sums), same tie-breaking (first minimum / maximum in list order, stable sort).
// Footprints are bit sets over the atoms of one cluster: `words` uint64 per object.
#include <algorithm>
#include <cstdint>
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
