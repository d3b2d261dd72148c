// Candidate setup: region crop packed by RUNS (the region pixels of one image row inside one aligned 4-column cell), compressed
// coordinates, greedy sub-sample grid, G~ of the runs (one entry = a grid point with the weights of the run's four pixels; run
// positions sorted by their number of entries).
// One 256- or 1024-thread workgroup per candidate.
//
// Reference behaviour restated here (never its code):
//   region        superdsm/objects.py:93,126-127  in1d(atoms, footprint) & y_mask & (EDT(y<=0) <= margin)
//                 (the candidate-independent part `valid` comes from sdsm_image_prepare)
//   trivial rule  superdsm/objects.py:184-191
//   mask compression, "too small" rule       superdsm/dsm.py:185-187
//   greedy grid   superdsm/dsm.py:164-181
//   PSF gather + float32 row normalisation   superdsm/dsm.py:145-161,192-193,232
#include "sdsm_common.h"

namespace {

// The PSF table: in LDS when it fits, else in global memory.  The two are read through pointers of their OWN address space (one uniform
// branch): a select between a generic LDS and a generic global pointer compiles to flat loads, which wait for every global store the
// wavefront has in flight (vmcnt(0)) -- the entries the loop has just written: 1.2 us per grid point in the rows of G~.
typedef const float __attribute__((address_space(3))) *lds_cfloat_p;
// (which of the two is a TEMPLATE argument of the loops over the grid points, not a branch inside them: where the two paths of a branch
// meet the compiler waits for vmcnt(0) -- all global stores of the wavefront included -- whichever path was taken: the loop that writes
// the entries then waited for its own stores once per grid point, 2-3 us each: 150-260 of the 250-390 us of sdsm_k_setup_rows)
template <bool PSF_LDS>
__device__ __forceinline__ float psf_at(const float *psf_lds, const float *psf, int pidx)
{
    if constexpr (PSF_LDS) return ((lds_cfloat_p)psf_lds)[pidx];
    else return ((g_cfloat_p)psf)[pidx];
}

struct WeightCtx {
    int cr, cc, R, k;
    const float *psf;       // global table
    const float *psf_lds;   // the same in LDS, or nullptr
    const uint32_t *keys;   // sorted grid keys in LDS
    int jlo, jhi;           // grid points whose (compressed) row lies within R of the pixel's: the only ones that can be in the window
    int nnz;                // entries of the row (grid points inside the PSF window)
    float wmax;             // largest of them
};

template <bool PSF_LDS>
__device__ __forceinline__ float wval(WeightCtx &c, int j)
{
    // (no branch: the eight calls of a group then have their reads of the key and of the PSF table in flight together; a point outside the
    // window reads the table's centre and contributes +0)
    const uint32_t key = c.keys[j];
    const int dr = (int)(key >> 16) - c.cr, dc = (int)(key & 0xffffu) - c.cc;
    const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
    const bool in = adr <= c.R && adc <= c.R;
    const int pidx = in ? (c.R + dr) * c.k + (c.R + dc) : c.R * c.k + c.R;
    const float t = psf_at<PSF_LDS>(c.psf_lds, c.psf, pidx);
    const float v = in ? t : 0.f;
    c.wmax = v > c.wmax ? v : c.wmax;
    c.nnz += in ? 1 : 0;
    return v;
}

// numpy's pairwise float32 sum of one block of <= 128 elements (loops_utils.h.src).  Only the grid points j in
// [c.jlo, c.jhi) can be non-zero; the others are skipped, which changes nothing: an element is added to accumulator
// (j - lo) % 8 in the order of j as numpy does, and adding +0.0f is exact.
template <bool PSF_LDS>
__device__ float pw_block(WeightCtx &c, int lo, int n)
{
    const int a = lo > c.jlo ? lo : c.jlo, b = lo + n < c.jhi ? lo + n : c.jhi;
    if (a >= b) return 0.f;
    if (n < 8) {
        float res = 0.f;
        for (int j = a; j < b; j++) res += wval<PSF_LDS>(c, j);
        return res;
    }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f, r5 = 0.f, r6 = 0.f, r7 = 0.f;
    const int nfull = n - (n % 8);
    const int bend = lo + nfull < b ? lo + nfull : b;
    for (int base = lo + ((a - lo) & ~7); base < bend; base += 8) {
        r0 += wval<PSF_LDS>(c, base); r1 += wval<PSF_LDS>(c, base + 1); r2 += wval<PSF_LDS>(c, base + 2); r3 += wval<PSF_LDS>(c, base + 3);
        r4 += wval<PSF_LDS>(c, base + 4); r5 += wval<PSF_LDS>(c, base + 5); r6 += wval<PSF_LDS>(c, base + 6); r7 += wval<PSF_LDS>(c, base + 7);
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (int j = lo + nfull > a ? lo + nfull : a; j < b; j++) res += wval<PSF_LDS>(c, j);
    return res;
}

// full pairwise recursion (n > 128 splits at n/2 rounded down to a multiple of 8), iteratively
template <bool PSF_LDS>
__device__ float pw_sum(WeightCtx &c, int M)
{
    if (M <= 128) return pw_block<PSF_LDS>(c, 0, M);
    int s_lo[6], s_n[6], s_stage[6];                     // M <= SDSM_MAX_GRID = 2048: at most 5 levels above the 128-element blocks
    float s_left[6];
    int sp = 0;
    s_lo[0] = 0; s_n[0] = M; s_stage[0] = 0; s_left[0] = 0.f;
    float ret = 0.f;
    while (sp >= 0) {
        if (s_n[sp] <= 128) { ret = pw_block<PSF_LDS>(c, s_lo[sp], s_n[sp]); sp--; continue; }
        int n2 = s_n[sp] / 2; n2 -= n2 % 8;
        if (s_stage[sp] == 0) {
            s_stage[sp] = 1;
            s_lo[sp + 1] = s_lo[sp]; s_n[sp + 1] = n2; s_stage[sp + 1] = 0; sp++;
        } else if (s_stage[sp] == 1) {
            s_left[sp] = ret; s_stage[sp] = 2;
            s_lo[sp + 1] = s_lo[sp] + n2; s_n[sp + 1] = s_n[sp] - n2; s_stage[sp + 1] = 0; sp++;
        } else { ret = s_left[sp] + ret; sp--; }
    }
    return ret;
}

// The grid points that can lie inside the PSF windows of a run's pixels, in ascending order of j: those of the rows within R (j in
// [jlo, jhi), growstart) whose column is within [clo, chi] = [the run's first column - R, its last + R].  The keys are sorted by
// (row, column): a point left of the range is skipped to the range's start in its row, one right of it to the start in the next row,
// by binary search.  A region 140 columns wide has ~18 grid points per grid row, 3-4 of them in the range: the loops over the points
// that writes the entries of a run makes a fifth of the iterations it made over all of [jlo, jhi) (its body is ~150 instructions; the cheap
// loops -- entry count, row sums -- are faster over all points: a skip is two dependent chains of LDS reads, divergent across the lanes).
struct GridCursor {
    const uint32_t *keys;
    int j, jhi;
    uint32_t clo, chi;
    __device__ __forceinline__ int lower_bound(uint32_t key, int lo) const
    {
        int hi = jhi;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
        return lo;
    }
    __device__ __forceinline__ void settle()             // forward to the next point inside the column range (or to jhi)
    {
        while (j < jhi) {
            const uint32_t key = keys[j], col = key & 0xffffu, row = key >> 16;
            if (col < clo) j = lower_bound((row << 16) | clo, j + 1);
            else if (col > chi) j = lower_bound(((row + 1) << 16) | clo, j + 1);
            else break;
        }
    }
    __device__ __forceinline__ void next() { j++; settle(); }
};
__device__ __forceinline__ GridCursor grid_cursor(const uint32_t *keys, int jlo, int jhi, const int (&cck)[SDSM_RUN], int R)
{
    int cmin = 1 << 20, cmax = -1;
#pragma unroll
    for (int k = 0; k < SDSM_RUN; k++) if (cck[k] >= 0) { cmin = cck[k] < cmin ? cck[k] : cmin; cmax = cck[k] > cmax ? cck[k] : cmax; }
    GridCursor g;
    g.keys = keys; g.j = jlo; g.jhi = cmax >= 0 ? jhi : jlo;   // (a run without pixels: nothing)
    g.clo = (uint32_t)(cmin - R > 0 ? cmin - R : 0); g.chi = (uint32_t)(cmax + R);
    g.settle();
    return g;
}

__device__ __forceinline__ int cheb(int r0, int c0, int r1, int c1)
{
    int a = r0 - r1, b = c0 - c1;
    a = a < 0 ? -a : a; b = b < 0 ? -b : b;
    return a > b ? a : b;
}

// Multiplier of the low-discrepancy scatter of the runs: scan index of raster run rr = (rr * B) mod NR, B ~ 0.618 NR coprime to NR
// (neighbouring positions are far apart in the image, which decorrelates the LDS atomics of the solve kernel).
__device__ __forceinline__ uint32_t scatter_mult(uint32_t NR)
{
    if (NR <= 2) return 1;
    uint32_t B = (uint32_t)(0.6180339887498949 * (double)NR);
    if (B < 1) B = 1;
    for (;;) {
        uint32_t a = B, b = NR;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        if (a == 1) return B;
        B++;
    }
}

// The pixels of raster run rr: present mask, scan indices (into the per-pixel setup arrays) of the present ones.
struct RunPixels { uint32_t mask; int idx[SDSM_RUN]; };
__device__ __forceinline__ RunPixels run_pixels(const BatchParams &P, const CandDesc &cd, int rr)
{
    RunPixels rp;
    const uint32_t q0m = P.run_q0[cd.run_off + rr];
    rp.mask = q0m >> 28;
    uint32_t q = q0m & 0x0fffffffu;
#pragma unroll
    for (int k = 0; k < SDSM_RUN; k++) {
        rp.idx[k] = -1;
        if ((rp.mask >> k) & 1u) { rp.idx[k] = (int)(((unsigned long long)q * cd.perm_inv) % (unsigned long long)cd.N); q++; }
    }
    return rp;
}

// Runs without G~ (null matrix, or more grid points than the solver holds): the final order is the scan order of the runs.
__device__ __forceinline__ void plain_runs(const BatchParams &P, const CandDesc &cd, int NR, uint32_t B, int tid, int nthreads)
{
    for (int rr = tid; rr < NR; rr += nthreads) {
        const int pos = (int)(((unsigned long long)rr * B) % (unsigned long long)NR);
        const RunPixels rp = run_pixels(P, cd, rr);
        double yk[SDSM_RUN];
        uint32_t rc0 = 0;
        bool first = true;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
            yk[k] = 0;
            if (rp.idx[k] >= 0) {
                yk[k] = P.tmp_y[cd.crop_off + rp.idx[k]];
                if (first) { rc0 = P.tmp_rc[cd.crop_off + rp.idx[k]] - (uint32_t)k; first = false; }
            }
        }
        double *yo = P.crop_y + (cd.run_off + pos) * SDSM_RUN;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) yo[k] = yk[k];
        P.crop_rc[cd.run_off + pos] = rc0;
        P.run_meta[cd.run_off + pos] = rp.mask << 24;
    }
}

#ifdef SDSM_PROFILE
#define ROWS_PROF_PARAM , long long (&rows_t)[5]
#define ROWS_PROF_ARG , rows_t
#define ROWS_T(k) do { long long _n = PROF_NOW(); rows_t[k] += _n - rows_t[4]; rows_t[4] = _n; } while (0)
#else
#define ROWS_PROF_PARAM
#define ROWS_PROF_ARG
#define ROWS_T(k) do { } while (0)
#endif
// G~ of the raster runs [rr0, rr1) of a candidate: per pixel the PSF gather, numpy's pairwise float32 row sum and the float32
// division (dsm.py:192-193), exactly as the reference builds the pixel's row; per run ONE pass over the grid points of the rows
// within R that writes an entry for every point inside the window of at least one of its pixels (weight 0 for the others).
// Runs are taken in RASTER order: the 64 lanes of a wavefront then sit next to each other in the image, see (almost) the same
// grid points and take the same branches in the loops over them.  efirst (LDS, M ints, initialised to j): first coupled column.
template <bool PSF_LDS>
__device__ __forceinline__ void rows_of_runs(const BatchParams &P, const CandDesc &cd, int NR, uint32_t B, int M, int R, int hc, const uint32_t *gridkeys,
                                             const uint16_t *growstart, const float *psf_lds, int *efirst, int rr0, int rr1, int step,
                                             bool &bad, int &hzmax ROWS_PROF_PARAM)
{
    for (int rr = rr0; rr < rr1; rr += step) {
        ROWS_T(3);
        const int si = (int)(((unsigned long long)rr * B) % (unsigned long long)NR);
        const int pos = (int)P.inv[cd.crop_off + si];
        const int nnz = (int)P.dist[cd.crop_off + si];
        const RunPixels rp = run_pixels(P, cd, rr);
        int cr = 0, cck[SDSM_RUN];
        double yk[SDSM_RUN];
        uint32_t rc0 = 0;
        bool first = true;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
            yk[k] = 0; cck[k] = -(1 << 20);              // an absent pixel: no grid point is inside its window
            if (rp.idx[k] >= 0) {
                yk[k] = P.tmp_y[cd.crop_off + rp.idx[k]];
                const uint32_t key = P.crop_cc[cd.crop_off + rp.idx[k]];
                cr = key >> 16; cck[k] = key & 0xffffu;
                if (first) { rc0 = P.tmp_rc[cd.crop_off + rp.idx[k]] - (uint32_t)k; first = false; }
            }
        }
        double *yo = P.crop_y + (cd.run_off + pos) * SDSM_RUN;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) yo[k] = yk[k];
        P.crop_rc[cd.run_off + pos] = rc0;
        ROWS_T(0);
        // pass 1, per pixel: row sum in numpy's order, largest entry, number of entries (nothing is stored)
        const int jlo = growstart[cr - R > 0 ? cr - R : 0], jhi = cr + R + 1 < hc ? growstart[cr + R + 1] : M;
        float sumk[SDSM_RUN], limk[SDSM_RUN];
        bool rowbad = nnz > P.zcap_run;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
            sumk[k] = 1.f; limk[k] = 0.f;
            if (rp.idx[k] >= 0) {
                WeightCtx c;
                c.cr = cr; c.cc = cck[k]; c.R = R; c.k = P.k; c.psf = P.psf; c.psf_lds = psf_lds; c.keys = gridkeys; c.nnz = 0; c.wmax = 0.f;
                c.jlo = jlo; c.jhi = jhi;
                const float sum = pw_sum<PSF_LDS>(c, M);
                if (c.nnz > P.zcap || !(sum > 0.f)) rowbad = true;                      // dsm.py:194
                sumk[k] = sum;
                limk[k] = P.hess_thr * __fdiv_rn(c.wmax, sum);
            }
        }
        if (rowbad) { bad = true; P.run_meta[cd.run_off + pos] = rp.mask << 24; continue; }
        // The float32 quotient psf / sum, correctly rounded (the reference divides in float32, dsm.py:193), as a float64 product with the
        // float64 reciprocal of the sum: the quotient of two 24-bit significands is either a float32 number or at least 2^-49 (relative)
        // away from every rounding boundary of float32, and the product is off by less than 2^-52 -- the same bits as the division,
        // without its dozen instructions and two mode switches per entry.
        double rsumk[SDSM_RUN];
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) rsumk[k] = 1.0 / (double)sumk[k];
        ROWS_T(1);
        // pass 2: one entry per grid point inside the window of at least one pixel: the normalised weights of the four pixels and
        // the pixels for which the entry is a LEADING one (>= hess_thr * the pixel's row maximum: the solver's approximate Hessian
        // uses only those; S and the gradient use all).  Leading entries fill the slots from 0 upwards -- in ascending column
        // order, so the solve kernel knows which of a pair is the row --, the others from slot nnz - 1 downwards.
        const int64_t base = cd.ell_off + pos;
        int hz = 0, others = 0, hzk[SDSM_RUN] = {0, 0, 0, 0}, mn[SDSM_RUN] = {0, 0, 0, 0};
        for (GridCursor gq = grid_cursor(gridkeys, jlo, jhi, cck, R); gq.j < gq.jhi; gq.next()) {   // (the points within the columns of the run's windows only)
            const int j = gq.j;
            const uint32_t gk = gridkeys[j];
            const int dr = (int)(gk >> 16) - cr, gc = (int)(gk & 0xffffu);
            if ((dr < 0 ? -dr : dr) > R) continue;
            // (the four pixels side by side without branches: their reads of the PSF table and their divisions overlap; a pixel whose window does
            // not hold the point reads the table's centre and keeps the weight 0)
            float nw[SDSM_RUN], tv[SDSM_RUN];
            bool ink[SDSM_RUN];
            uint32_t lead = 0;
            bool any = false;
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) {
                const int dc = gc - cck[k];
                ink[k] = (dc < 0 ? -dc : dc) <= R;
                const int pidx = ink[k] ? (R + dr) * P.k + (R + dc) : R * P.k + R;
                tv[k] = psf_at<PSF_LDS>(psf_lds, P.psf, pidx);
                any = any || ink[k];
            }
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) {
                const float q = (float)((double)tv[k] * rsumk[k]);    // == __fdiv_rn(tv[k], sumk[k]), see rsumk
                nw[k] = ink[k] ? q : 0.f;
                if (ink[k] && !(nw[k] < limk[k])) lead |= 1u << k;
            }
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) {
                if ((lead >> k) & 1u) {
                    // grid points coupled by this pixel in the solver's Hessian: every one of them with the smallest of them
                    if (hzk[k] == 0) mn[k] = j; else atomicMin(&efirst[j], mn[k]);
                    hzk[k]++;
                }
            }
            if (!any) continue;
            const int slot = lead ? hz++ : nnz - 1 - others++;
            if (slot < 0 || slot >= nnz) { bad = true; break; }                       // (cannot happen: nnz was counted by the same tests)
            const int64_t e = base + (int64_t)slot * NR;
            f32x4 wv; wv.x = nw[0]; wv.y = nw[1]; wv.z = nw[2]; wv.w = nw[3];
            reinterpret_cast<f32x4 *>(P.ell_w)[e] = wv;
            P.ell_im[e] = (uint32_t)j | (lead << 16);
        }
        // padding (column 0, weights 0) up to the entry count of the first run of this run's 64-position chunk (a wavefront of the
        // solve kernel walks the entries its first lane has for all of its lanes)
        const int kh = (int)P.run_aux[cd.run_off + (pos >> 6)];
        {
            f32x4 zw; zw.x = 0.f; zw.y = 0.f; zw.z = 0.f; zw.w = 0.f;
            for (int sl = nnz; sl < kh; sl++) { const int64_t e = base + (int64_t)sl * NR; reinterpret_cast<f32x4 *>(P.ell_w)[e] = zw; P.ell_im[e] = 0; }
        }
        P.run_meta[cd.run_off + pos] = (uint32_t)nnz | ((uint32_t)hz << 12) | (rp.mask << 24);
        hzmax = hz > hzmax ? hz : hzmax;
        ROWS_T(2);
    }
}

// Envelope storage of the Hessian (BatchParams.env_fst / env_rb) from the first coupled columns: made non-decreasing and
// multiples of SDSM_PANEL (the factorisation works on panels of that many columns), row bases by a running sum.  One thread.
__device__ __forceinline__ int envelope_from_first(const BatchParams &P, const CandDesc &cd, int M, int *efirst)
{
    int run = M;
    for (int a = M - 1; a >= 0; a--) { run = efirst[a] < run ? efirst[a] : run; efirst[a] = run & ~(SDSM_PANEL - 1); }
    int rp = 0;
    for (int a = 0; a < M; a++) {
        P.env_fst[cd.xi_off + a] = efirst[a];
        P.env_rb[cd.xi_off + a] = rp - efirst[a];
        rp += a - efirst[a] + 1;
    }
    return rp + 6 * M + 21;
}

}  // namespace

// Limits of the setup kernel's LDS tables: the host picks the smallest class that holds the plan (sdsm_setup_class).
struct SetupLimS { static constexpr int DIM = 512, GRID = 512, PSFW = 1152, LABELS = 32767, WPE = 4, WG = 256; };      // 13 KB: PSF k <= 33
struct SetupLimM { static constexpr int DIM = 1024, GRID = 1024, PSFW = SDSM_PSF_LDS, LABELS = SDSM_MAX_LABELS, WPE = 4, WG = 1024; };   // 37 KB: PSF k <= 65; plans with regions this large have few candidates: 1024 threads each
struct SetupLimL { static constexpr int DIM = SDSM_MAX_BBOX_DIM, GRID = SDSM_MAX_GRID, PSFW = SDSM_PSF_LDS, LABELS = SDSM_MAX_LABELS, WPE = 4, WG = 1024; };   // 54 KB

// LDS of the setup kernel is sized by the limits T of a plan's class (SetupLimits below, chosen by the host from the plan's
// largest bounding box, bound on M, label and PSF): the common plans -- regions of a few hundred pixels across -- then run four
// workgroups per compute unit instead of the two that the largest tables allow (1.58 -> 1.22 ms on the 8-image launch).
// One candidate by one workgroup (`return` leaves the candidate; every exit is uniform over the workgroup).  A function of its own: with the body written into
// the kernel the register allocator left 2 / 6 spilled registers (12 / 20 B of scratch per lane) in two of the three instantiations, this way none.
template <class T>
__device__ __forceinline__ void setup_candidate(const BatchParams &P)
{
    static_assert((T::LABELS + 1) / 32 <= T::PSFW && T::DIM % 32 == 0, "setup limits");
    // footprint bitset during the region scan, then the PSF table (k * k floats, if it fits) for the rows of G~
    __shared__ uint32_t fp_or_psf[T::PSFW];
    uint32_t *fpbits = fp_or_psf;
    __shared__ uint32_t rowbits[T::DIM / 32], colbits[T::DIM / 32];
    __shared__ uint16_t rowrank[T::DIM], colrank[T::DIM];
    __shared__ uint32_t gridkeys[T::GRID];
    __shared__ unsigned long long mom[4];
    __shared__ unsigned long long scr64[(T::WG / 64)];
    __shared__ int scr32[(T::WG / 64)];
    __shared__ unsigned scr32u[(T::WG / 64)];
    __shared__ int sh_M, sh_npos, sh_nneg, sh_err, sh_yhi;
    __shared__ int cls_cnt[SDSM_MAX_ELL_GROUPS + 1], cls_start[SDSM_MAX_ELL_GROUPS + 1], cls_run[SDSM_MAX_ELL_GROUPS + 1];
    __shared__ int wave_cnt[(T::WG / 64)][SDSM_MAX_ELL_GROUPS + 1];
    __shared__ int efirst[T::GRID < SDSM_MAX_N_GLOBAL ? T::GRID : SDSM_MAX_N_GLOBAL];   // envelope of the solver's Hessian: first coupled column per grid point

#ifdef SDSM_PROFILE
    long long sp_t = PROF_NOW(), sp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SETUP_T(k) do { long long _n = PROF_NOW(); sp_acc[k] += _n - sp_t; sp_t = _n; } while (0)
#else
#define SETUP_T(k) do { } while (0)
#endif
    const int tid = threadIdx.x;
    const int ci = P.order[blockIdx.x];
    const CandDesc cd = P.cand[ci];
    CandState *st = &P.state[ci];
    const ImageRef im = P.img[cd.image];
    const double *__restrict__ y = im.y;
    const int32_t *__restrict__ atoms = im.atoms;
    const uint8_t *__restrict__ valid = im.valid;

    if (blockIdx.x == 0 && tid == 0) *P.wide_ticket = 0;
    if (cd.rows_g > 0 && tid < 2 * SDSM_WIDE_SYNC) reinterpret_cast<int *>(P.wide_pool + cd.wide_off)[tid] = 0;   // counters of sdsm_k_setup_rows and of the workgroup group
    if (cd.N <= 0) {
        if (tid == 0) { CandState s = {}; s.status = ST_ERROR; *st = s; }
        return;
    }
    // A region beyond the tables of this kernel (bounding box; a G~ block of 4 GB and more: the solve kernels address a candidate's entries by
    // 32-bit byte offsets) gets its packed crop and the ELLIPTICAL model only -- flagged unsupported by the solve kernel, a fallback result
    // for the caller as after a failed DSM solve of the reference (objects.py:399-410), never an abort of the batch.
    const bool fits = cd.h <= T::DIM && cd.w <= T::DIM && (long long)cd.NRcap * P.zcap_run < (1ll << 28);
    for (int i = tid; i < (T::LABELS + 1) / 32; i += T::WG) fpbits[i] = 0;
    for (int i = tid; i < T::DIM / 32; i += T::WG) { rowbits[i] = 0; colbits[i] = 0; }
    if (tid < 4) mom[tid] = 0;
    if (tid == 0) { sh_M = 0; sh_npos = 0; sh_nneg = 0; sh_err = 0; sh_yhi = 0; }
    __syncthreads();
    for (int i = tid; i < cd.fp_len; i += T::WG) {
        int l = P.fp_labels[cd.fp_off + i];
        if (l >= 1 && l <= T::LABELS) atomicOr(&fpbits[l >> 5], 1u << (l & 31));
    }
    __syncthreads();

    // ---- 1. region scan in raster order over the bounding box, one aligned 4-column cell per thread and step: the region
    //      pixels of a cell are one RUN; pixels go to the per-pixel setup arrays in scan order (low-discrepancy scatter of the
    //      raster rank, the order the grid steps work in), runs are numbered in raster order ------------------------------------
    const int cb0 = cd.c0 >> 2, ncw = ((cd.c0 + cd.w - 1) >> 2) - cb0 + 1;
    const int ncell = cd.h * ncw;
    int running = 0, nruns = 0;
    unsigned long long m_r = 0, m_c = 0, m_rr = 0, m_cc = 0;
    int npos = 0, nneg = 0, yhi = 0;                      // yhi: high word of the largest |y| (non-negative doubles order like their bits)
    for (int base = 0; base < ncell; base += T::WG) {
        const int e = base + tid;
        bool flag[SDSM_RUN] = {false, false, false, false};
        double yv4[SDSM_RUN] = {0, 0, 0, 0};
        int r = 0, ccell = 0;
        if (e < ncell) {
            r = e / ncw;
            ccell = (cb0 + (e - r * ncw)) << 2;               // image column of the cell's first pixel
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) {
                const int c = ccell + k;
                if (c >= cd.c0 && c < cd.c0 + cd.w) {
                    const size_t p = (size_t)(cd.r0 + r) * im.W + c;
                    const int a = atoms[p];                       // three independent loads (not: label -> validity -> y, a chain of round trips)
                    const uint8_t vl = valid[p];
                    const double yl = y[p];
                    flag[k] = a >= 1 && a <= T::LABELS && ((fpbits[a >> 5] >> (a & 31)) & 1u) && vl;
                    if (flag[k]) yv4[k] = yl;
                }
            }
        }
        const unsigned cnt = (unsigned)flag[0] + (unsigned)flag[1] + (unsigned)flag[2] + (unsigned)flag[3];
        unsigned tot2;
        const unsigned before2 = block_excl_scan_u32<T::WG / 64>(cnt | (cnt ? 1u << 16 : 0u), scr32u, &tot2);     // pixels | runs << 16 (<= 4096 pixels per step)
        if (cnt) {
            int q = running + (int)(before2 & 0xffffu);
            const int rr = nruns + (int)(before2 >> 16);
            uint32_t mask = 0;
            const uint32_t q0 = (uint32_t)q;
            if (fits) atomicOr(&rowbits[r >> 5], 1u << (r & 31));
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) {
                if (!flag[k]) continue;
                mask |= 1u << k;
                const int c = ccell + k - cd.c0;
                const double yv = yv4[k];
                if (q < cd.N) {
                    const int64_t o = cd.crop_off + (int64_t)(((unsigned long long)q * cd.perm_inv) % (unsigned long long)cd.N);
                    P.tmp_y[o] = yv;
                    P.tmp_rc[o] = ((uint32_t)(cd.r0 + r) << 16) | (uint32_t)(cd.c0 + c);
                }
                q++;
                if (fits) atomicOr(&colbits[c >> 5], 1u << (c & 31));
                const int ah = __double2hiint(yv) & 0x7fffffff;
                yhi = ah > yhi ? ah : yhi;
                if (yv < 0) nneg++;
                if (yv > 0) {
                    unsigned long long rq = cd.r0 + r, cq = cd.c0 + c;
                    npos++; m_r += rq; m_c += cq; m_rr += rq * rq; m_cc += cq * cq;
                }
            }
            if (rr < cd.NRcap) P.run_q0[cd.run_off + rr] = q0 | (mask << 28);
        }
        running += (int)(tot2 & 0xffffu);
        nruns += (int)(tot2 >> 16);
    }
    if (nneg) atomicAdd(&sh_nneg, nneg);
    if (yhi) atomicMax(&sh_yhi, yhi);
    if (npos) { atomicAdd(&sh_npos, npos); atomicAdd(&mom[0], m_r); atomicAdd(&mom[1], m_c); atomicAdd(&mom[2], m_rr); atomicAdd(&mom[3], m_cc); }
    __syncthreads();
    const int NR = nruns;
    const uint32_t B = scatter_mult((uint32_t)(NR > 0 ? NR : 1));         // (every thread: a few iterations of Euclid)

    SETUP_T(0);
    if (!fits) {                                          // elliptical model only (the marker M makes the solve kernel flag the result)
        CandState s = {};
        s.npos = sh_npos; s.nneg = sh_nneg; s.sum_r = mom[0]; s.sum_c = mom[1]; s.sum_rr = mom[2]; s.sum_cc = mom[3];
        { int ex = ((sh_yhi >> 20) & 0x7ff) - 1022; s.yexp = ex < -400 ? -400 : (ex > 400 ? 400 : ex); }
        s.NR = NR;
        if (running != cd.N || NR > cd.NRcap || NR < 1) s.status = ST_ERROR;
        else if (s.npos == 1 && !P.no_trivial_rule) s.status = ST_TRIVIAL;
        else { plain_runs(P, cd, NR, B, tid, T::WG); s.M = SDSM_MAX_N_GLOBAL; s.status = ST_OK; }
        if (tid == 0) *st = s;
        return;
    }
    // ---- 2. compressed coordinates: delete empty rows / columns (dsm.py:185-186) ---------------
    for (int r = tid; r < cd.h; r += T::WG) {
        int cnt = 0;
        for (int wd = 0; wd < (r >> 5); wd++) cnt += __popc(rowbits[wd]);
        cnt += __popc(rowbits[r >> 5] & ((1u << (r & 31)) - 1u));
        rowrank[r] = (uint16_t)cnt;
    }
    for (int c = tid; c < cd.w; c += T::WG) {
        int cnt = 0;
        for (int wd = 0; wd < (c >> 5); wd++) cnt += __popc(colbits[wd]);
        cnt += __popc(colbits[c >> 5] & ((1u << (c & 31)) - 1u));
        colrank[c] = (uint16_t)cnt;
    }
    int hc = 0, wc = 0;
    for (int wd = 0; wd < (cd.h + 31) / 32; wd++) hc += __popc(rowbits[wd]);
    for (int wd = 0; wd < (cd.w + 31) / 32; wd++) wc += __popc(colbits[wd]);
    __syncthreads();

    // (the state lives in LDS, every thread stores the same values: 26 registers less across the rest of the kernel)
    __shared__ CandState sh_state;
    CandState &s = sh_state;
    if (tid < (int)(sizeof(CandState) / 4)) reinterpret_cast<int *>(&sh_state)[tid] = 0;
    __syncthreads();
    s.hc = hc; s.wc = wc; s.npos = sh_npos; s.nneg = sh_nneg;
    s.sum_r = mom[0]; s.sum_c = mom[1]; s.sum_rr = mom[2]; s.sum_cc = mom[3];
    { int ex = ((sh_yhi >> 20) & 0x7ff) - 1022; s.yexp = ex < -400 ? -400 : (ex > 400 ? 400 : ex); }
    s.NR = NR;
    if (running != cd.N || NR > cd.NRcap || NR < 1) { s.status = ST_ERROR; if (tid == 0) *st = s; return; }   // plan / image mismatch
    if (s.npos == 1 && !P.no_trivial_rule) { s.status = ST_TRIVIAL; if (tid == 0) *st = s; return; }            // objects.py:184-191 (no solve)

    const int S = P.subsample, R = P.R;
    const bool null_matrix = P.no_deform || hc <= P.k / 2 || wc <= P.k / 2;               // dsm.py:187,225
    for (int i = tid; i < cd.N; i += T::WG) {
        uint32_t rc = P.tmp_rc[cd.crop_off + i];
        int r = (int)(rc >> 16) - cd.r0, c = (int)(rc & 0xffffu) - cd.c0;
        uint32_t key = ((uint32_t)rowrank[r] << 16) | (uint32_t)colrank[c];
        P.crop_cc[cd.crop_off + i] = key;
        if (!null_matrix && rowrank[r] % S == 0 && colrank[c] % S == 0) {                   // dsm.py:165-168
            int j = atomicAdd(&sh_M, 1);
            if (j < T::GRID) gridkeys[j] = key;
        }
    }
    if (null_matrix) {                                  // no G~: the final order of the runs is their scan order
        plain_runs(P, cd, NR, B, tid, T::WG);
        s.M = 0; s.status = ST_OK;
        if (tid == 0) *st = s;
        return;
    }
    __syncthreads();
    const int cap = cd.Mcap < T::GRID ? cd.Mcap : T::GRID;
    int M = sh_M;
    // more grid points than this kernel's table (or the host's bound) holds: the elliptical model only, flagged by the solve kernel
    if (M > cap) { plain_runs(P, cd, NR, B, tid, T::WG); s.M = SDSM_MAX_N_GLOBAL; s.status = ST_OK; if (tid == 0) *st = s; return; }

    SETUP_T(1);
    // ---- 3. greedy completion of the grid (dsm.py:169-181) --------------------------------------
    // Distance of every pixel to the nearest point of the regular grid.  The grid points sit on the lattice (i S, j S) of the
    // compressed coordinates, where the mask has a pixel there: with their presence as a bit per lattice position (in the LDS of the
    // footprint bitset, free by now) a pixel looks at the rings of lattice positions around its own cell instead of at all M points
    // -- a position in ring m is at least (m - 1) S + 1 away, so the search ends after the ring k with best <= k S + 1, normally
    // k = 1: 9 probes instead of M (a 73 k-pixel region with 250 lattice points: most of the 1.2 ms this section took).
    const int LH = (hc - 1) / S + 1, LW = (wc - 1) / S + 1;
    const bool lattice_bits = M > 16 && (long long)LH * LW <= (long long)T::PSFW * 32;
    if (lattice_bits) {
        uint32_t *bm = fp_or_psf;
        for (int w2 = tid; w2 < (LH * LW + 31) / 32; w2 += T::WG) bm[w2] = 0;
        __syncthreads();
        for (int j = tid; j < M; j += T::WG) {
            const int b = (int)(gridkeys[j] >> 16) / S * LW + (int)(gridkeys[j] & 0xffffu) / S;
            atomicOr(&bm[b >> 5], 1u << (b & 31));
        }
        __syncthreads();
        for (int i = tid; i < cd.N; i += T::WG) {
            const uint32_t key = P.crop_cc[cd.crop_off + i];
            const int r = key >> 16, c = key & 0xffffu;
            const int i0 = r / S, j0 = c / S;
            uint32_t d = 0xffffffffu;                   // distance_transform_bf without any grid point
            bool done = false;
            for (int k = 0; k <= 3; k++) {
                const int ilo = i0 - k, ihi = i0 + k, jlo = j0 - k, jhi = j0 + k;
                for (int ii = ilo < 0 ? 0 : ilo; ii <= (ihi < LH - 1 ? ihi : LH - 1); ii++) {
                    const bool edge_row = ii == ilo || ii == ihi;
                    const int step = edge_row || k == 0 ? 1 : 2 * k;            // inner rows of the ring: its two end columns only
                    for (int jj = jlo; jj <= jhi; jj += step) {
                        if (jj < 0 || jj >= LW) continue;
                        const int b = ii * LW + jj;
                        if ((bm[b >> 5] >> (b & 31)) & 1u) {
                            const uint32_t t = (uint32_t)cheb(r, c, ii * S, jj * S);
                            d = t < d ? t : d;
                        }
                    }
                }
                if (d <= (uint32_t)(k * S + 1)) { done = true; break; }
            }
            if (!done) {                                 // no grid point nearby (rare: a pixel far out on a thin part of the mask): all of them
                for (int j = 0; j < M; j++) {
                    const uint32_t t = (uint32_t)cheb(r, c, gridkeys[j] >> 16, gridkeys[j] & 0xffffu);
                    d = t < d ? t : d;
                }
            }
            P.dist[cd.crop_off + i] = d;
        }
        __syncthreads();                                 // (the bitmap's LDS is reused by step 4b)
    } else {
        for (int i = tid; i < cd.N; i += T::WG) {
            uint32_t key = P.crop_cc[cd.crop_off + i];
            int r = key >> 16, c = key & 0xffffu;
            uint32_t d = 0xffffffffu;                       // distance_transform_bf without any grid point
            for (int j = 0; j < M; j++) {
                uint32_t t = (uint32_t)cheb(r, c, gridkeys[j] >> 16, gridkeys[j] & 0xffffu);
                d = t < d ? t : d;
            }
            P.dist[cd.crop_off + i] = d;
        }
    }
    bool unsupported = false;
    // smallest distance >= subsample; ties -> first pixel in raster order of the compressed mask.  ONE pass over the pixels per
    // added grid point: the pass that lowers the distances to the new point also finds the next candidate.
    unsigned long long best = ~0ull;
    for (int i = tid; i < cd.N; i += T::WG) {
        uint32_t d = P.dist[cd.crop_off + i];
        if (d >= (uint32_t)S) {
            unsigned long long key = ((unsigned long long)d << 32) | P.crop_cc[cd.crop_off + i];
            best = key < best ? key : best;
        }
    }
    for (;;) {
        best = block_min_u64<T::WG / 64>(best, scr64);
        if (best == ~0ull) break;
        if (M >= cap) { unsupported = true; break; }
        uint32_t nk = (uint32_t)(best & 0xffffffffull);
        if (tid == 0) gridkeys[M] = nk;
        M++;
        int nr = nk >> 16, nc = nk & 0xffffu;
        best = ~0ull;
        for (int i = tid; i < cd.N; i += T::WG) {
            const uint32_t key = P.crop_cc[cd.crop_off + i];
            const uint32_t t = (uint32_t)cheb(key >> 16, key & 0xffffu, nr, nc);
            uint32_t d = P.dist[cd.crop_off + i];
            if (t < d) { d = t; P.dist[cd.crop_off + i] = t; }
            if (d >= (uint32_t)S) {
                unsigned long long k2 = ((unsigned long long)d << 32) | key;
                best = k2 < best ? k2 : best;
            }
        }
    }
    if (unsupported) { __syncthreads(); plain_runs(P, cd, NR, B, tid, T::WG); s.M = SDSM_MAX_N_GLOBAL; s.status = ST_OK; if (tid == 0) *st = s; return; }
    __syncthreads();
    if (6 + M > SDSM_MAX_N_GLOBAL) {                     // the solve kernel only computes the elliptical model (flagged unsupported)
        plain_runs(P, cd, NR, B, tid, T::WG);
        s.M = M; s.status = ST_OK;
        if (tid == 0) *st = s;
        return;
    }

    SETUP_T(2);
    // ---- 4. columns of G~ = grid points in raster order (np.nonzero(col_mask), dsm.py:159) ------
    for (int j = tid; j < M; j += T::WG) {
        uint32_t key = gridkeys[j];
        int rank = 0;
        for (int q = 0; q < M; q++) rank += gridkeys[q] < key;
        P.grid_rc[cd.xi_off + rank] = key;
    }
    __syncthreads();
    for (int j = tid; j < M; j += T::WG) gridkeys[j] = P.grid_rc[cd.xi_off + j];
    __syncthreads();
    // first grid point of every compressed row (the row / column rank tables are no longer needed: reuse rowrank): a pixel
    // only looks at the grid points of the rows within R of its own
    uint16_t *growstart = rowrank;
    for (int r = tid; r < hc && r < T::DIM; r += T::WG) {
        int lo = 0, hi = M;                               // first j with row(j) >= r
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(gridkeys[mid] >> 16) < r) lo = mid + 1; else hi = mid; }
        growstart[r] = (uint16_t)lo;
    }
    __syncthreads();

    SETUP_T(3);
    // ---- 4b. final order of the runs: stable counting sort of their scan order by the number of G~ entries of the run (grid
    //      points inside the window of at least one of its pixels), largest first.  A wavefront of the solve kernel then walks 64
    //      runs of (nearly) the same length: as many entries as its first lane has. ------
    const int zshift = P.zshift;
    const int ngmax = (P.zcap_run + (1 << zshift) - 1) >> zshift;   // <= SDSM_MAX_ELL_GROUPS
    for (int k = tid; k <= ngmax; k += T::WG) { cls_cnt[k] = 0; cls_run[k] = 0; }
    __syncthreads();
    int cntmax = 0;
    for (int rr = tid; rr < NR; rr += T::WG) {          // raster order (coherent wavefronts), see rows_of_runs
        const int si = (int)(((unsigned long long)rr * B) % (unsigned long long)NR);
        const RunPixels rp = run_pixels(P, cd, rr);
        int cr = 0, cck[SDSM_RUN];
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
            cck[k] = -(1 << 20);
            if (rp.idx[k] >= 0) { const uint32_t key = P.crop_cc[cd.crop_off + rp.idx[k]]; cr = key >> 16; cck[k] = key & 0xffffu; }
        }
        int cnt = 0;
        const int jlo = growstart[cr - R > 0 ? cr - R : 0], jhi = cr + R + 1 < hc ? growstart[cr + R + 1] : M;
        for (int j = jlo; j < jhi; j++) {
            int dr = (int)(gridkeys[j] >> 16) - cr;
            const int gc = (int)(gridkeys[j] & 0xffffu);
            dr = dr < 0 ? -dr : dr;
            bool any = false;
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) { int dc = gc - cck[k]; dc = dc < 0 ? -dc : dc; any = any || dc <= R; }
            cnt += (dr <= R && any) ? 1 : 0;
        }
        cntmax = cnt > cntmax ? cnt : cntmax;
        int k = (cnt + (1 << zshift) - 1) >> zshift;
        k = k > ngmax ? ngmax : k;                       // runs with more entries than zcap_run are reported as errors in step 5
        P.dist[cd.crop_off + si] = (uint32_t)cnt;
        P.inv[cd.crop_off + si] = (uint32_t)k;
        atomicAdd(&cls_cnt[k], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = ngmax; k >= 0; k--) { cls_start[k] = acc; acc += cls_cnt[k]; }
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int base = 0; base < NR; base += T::WG) {
            for (int e = tid; e < (T::WG / 64) * (ngmax + 1); e += T::WG) wave_cnt[e / (ngmax + 1)][e % (ngmax + 1)] = 0;
            __syncthreads();
            const int i = base + tid;
            const int k = i < NR ? (int)P.inv[cd.crop_off + i] : -1;
            int within = 0;
            unsigned long long todo = __ballot(k >= 0);
            while (todo) {                               // one pass per distinct class present in the wavefront
                const int src = __ffsll((long long)todo) - 1;
                const int k0 = __shfl(k, src);
                const unsigned long long same = __ballot(k == k0);
                if (k == k0) {
                    within = __popcll(same & ((1ull << lane) - 1ull));
                    if (lane == src) wave_cnt[wave][k0] = __popcll(same);
                }
                todo &= ~same;
            }
            __syncthreads();
            if (k >= 0) {
                int before = 0;
                for (int w2 = 0; w2 < wave; w2++) before += wave_cnt[w2][k];
                P.inv[cd.crop_off + i] = (uint32_t)(cls_start[k] + cls_run[k] + before + within);
            }
            __syncthreads();
            for (int k2 = tid; k2 <= ngmax; k2 += T::WG) {
                int add = 0;
                for (int w2 = 0; w2 < (T::WG / 64); w2++) add += wave_cnt[w2][k2];
                cls_run[k2] += add;
            }
            __syncthreads();
        }
    }

    for (int j = tid; j < M; j += T::WG) efirst[j] = j;
    __syncthreads();

    SETUP_T(4);
    // entry count the rows of every 64-position chunk are padded to: that of the chunk's first run, rounded up to its sort class
    // (runs of a class are not sorted among themselves), for rows_of_runs
    for (int t = tid; t * 64 < NR; t += T::WG) {
        int kh = 0;
        const int head = t * 64;
        for (int k2 = ngmax; k2 >= 0; k2--) if (cls_cnt[k2] > 0 && head >= cls_start[k2]) kh = k2;
        P.run_aux[cd.run_off + t] = (uint32_t)(kh << zshift);
    }
    const int zmax = -block_min_i32<T::WG / 64>(-cntmax, scr32);
    s.M = M; s.zmax = zmax; s.hzmax = 0;
    __syncthreads();
    if (cd.rows_g > 0) {
        // a large region: its rows are built by several workgroups (sdsm_k_setup_rows), the last of which also finishes the envelope;
        // env_size == -1 marks the state as pending
        for (int j = tid; j < M; j += T::WG) P.env_fst[cd.xi_off + j] = j;
        s.env_size = -1; s.status = ST_OK;
        if (tid == 0) *st = s;
        SETUP_T(5);
#ifdef SDSM_PROFILE
        if (P.prof2 && tid == 0) for (int k = 0; k < 8; k++) P.prof2[(size_t)ci * 8 + k] = sp_acc[k];
#endif
        return;
    }

    // ---- 5. rows of G~ ---------------------------------------------------------------------------
    const float *psf_lds = nullptr;
    if (P.k * P.k <= T::PSFW) {                     // the footprint bitset is no longer needed
        float *pl = reinterpret_cast<float *>(fp_or_psf);
        for (int e = tid; e < P.k * P.k; e += T::WG) pl[e] = P.psf[e];
        psf_lds = pl;
        __syncthreads();
    }
    bool bad = false;
    int hzmax = 0;
#ifdef SDSM_PROFILE
    long long rows_t[5] = {0, 0, 0, 0, PROF_NOW()};
#endif
    if (psf_lds) rows_of_runs<true>(P, cd, NR, B, M, R, hc, gridkeys, growstart, psf_lds, efirst, tid, NR, T::WG, bad, hzmax ROWS_PROF_ARG);
    else rows_of_runs<false>(P, cd, NR, B, M, R, hc, gridkeys, growstart, psf_lds, efirst, tid, NR, T::WG, bad, hzmax ROWS_PROF_ARG);
    if (bad) atomicOr(&sh_err, 1);
    hzmax = -block_min_i32<T::WG / 64>(-hzmax, scr32);
    __syncthreads();
    SETUP_T(5);
    // ---- 6. envelope storage of the Hessian ------------------------------------------------------
    if (tid == 0) s.env_size = envelope_from_first(P, cd, M, efirst);
    s.hzmax = hzmax;
    s.status = sh_err ? ST_ERROR : ST_OK;
    if (tid == 0) *st = s;
    SETUP_T(6);
#ifdef SDSM_PROFILE
    if (P.prof2 && tid == 0) for (int k = 0; k < 8; k++) P.prof2[(size_t)ci * 8 + k] = sp_acc[k];
#endif
}

template <class T>
__global__ __launch_bounds__(T::WG, T::WPE) void sdsm_k_setup(BatchParams P)
{
    setup_candidate<T>(P);
}

// Rows of G~ and envelope of the very large regions: one workgroup per member of the region's workgroup group, each takes
// a slice of the raster ranks.  The members meet through global atomics only (first coupled columns: atomicMin, largest
// number of Hessian entries: atomicMax, error flag: atomicOr); the LAST member to finish (a ticket counter, no waiting)
// builds the envelope and completes the state.
__global__ __launch_bounds__(SDSM_WG) void sdsm_k_setup_rows(BatchParams P)
{
    extern __shared__ __align__(16) unsigned char rows_dyn[];            // gridkeys[P.rows_mcap] | efirst[P.rows_mcap]: sized by the plan (launch)
    uint32_t *gridkeys = reinterpret_cast<uint32_t *>(rows_dyn);
    int *efirst = reinterpret_cast<int *>(rows_dyn) + P.rows_mcap;
    __shared__ uint16_t growstart[SDSM_MAX_BBOX_DIM];
    __shared__ float psf_tab[SDSM_PSF_LDS];
    __shared__ int scr32[SDSM_WAVES];
    __shared__ int sh_last;
    const int tid = threadIdx.x;
#ifdef SDSM_PROFILE
    long long rt0 = PROF_NOW(), rt1 = 0, rt2 = 0;
#endif
    const int entry = P.order[blockIdx.x];
    const int ci = entry & 0xffffff, g = (entry >> 24) & 0xff;
    const CandDesc cd = P.cand[ci];
    CandState *st = &P.state[ci];
    if (st->status != ST_OK || st->env_size != -1) return;       // nothing pending (trivial, no G~, unsupported, ...)
    const int M = st->M, hc = st->hc, R = P.R, G = cd.rows_g, NR = st->NR;
    const uint32_t B = scatter_mult((uint32_t)(NR > 0 ? NR : 1));
    int *sync = reinterpret_cast<int *>(P.wide_pool + cd.wide_off);   // [2] ticket, [3] error flag (zeroed by sdsm_k_setup)
    for (int j = tid; j < M; j += SDSM_WG) { gridkeys[j] = P.grid_rc[cd.xi_off + j]; efirst[j] = j; }
    const float *psf_lds = nullptr;
    if (P.k * P.k <= SDSM_PSF_LDS) {
        for (int e = tid; e < P.k * P.k; e += SDSM_WG) psf_tab[e] = P.psf[e];
        psf_lds = psf_tab;
    }
    __syncthreads();
    for (int r = tid; r < hc && r < SDSM_MAX_BBOX_DIM; r += SDSM_WG) {
        int lo = 0, hi = M;                               // first j with row(j) >= r
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(gridkeys[mid] >> 16) < r) lo = mid + 1; else hi = mid; }
        growstart[r] = (uint16_t)lo;
    }
    __syncthreads();
    const int chunk = (NR + G - 1) / G;
    const int q0 = g * chunk < NR ? g * chunk : NR, q1 = q0 + chunk < NR ? q0 + chunk : NR;
    bool bad = false;
    int hzmax = 0;
#ifdef SDSM_PROFILE
    rt1 = PROF_NOW();
    long long rows_t[5] = {0, 0, 0, 0, rt1};
#endif
    if (psf_lds) rows_of_runs<true>(P, cd, NR, B, M, R, hc, gridkeys, growstart, psf_lds, efirst, q0 + tid, q1, SDSM_WG, bad, hzmax ROWS_PROF_ARG);
    else rows_of_runs<false>(P, cd, NR, B, M, R, hc, gridkeys, growstart, psf_lds, efirst, q0 + tid, q1, SDSM_WG, bad, hzmax ROWS_PROF_ARG);
#ifdef SDSM_PROFILE
    rt2 = PROF_NOW();
#endif
    hzmax = -block_min_i32(-hzmax, scr32);
    __syncthreads();
    for (int j = tid; j < M; j += SDSM_WG) if (efirst[j] < j) atomicMin(&P.env_fst[cd.xi_off + j], efirst[j]);
    if (bad) atomicOr(&sync[3], 1);
    if (tid == 0) atomicMax(&st->hzmax, hzmax);
    __syncthreads();
    if (tid == 0) sh_last = __hip_atomic_fetch_add(&sync[2], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
    __syncthreads();
    if (!sh_last) return;
    // last member: every other member's atomics precede its ticket; read them with agent-scope loads
    for (int j = tid; j < M; j += SDSM_WG) efirst[j] = __hip_atomic_load(&P.env_fst[cd.xi_off + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        st->env_size = envelope_from_first(P, cd, M, efirst);
        if (__hip_atomic_load(&sync[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) st->status = ST_ERROR;
#ifdef SDSM_PROFILE
        if (P.prof2) { P.prof2[(size_t)ci * 8 + 7] = rt1 - rt0; P.prof2[(size_t)ci * 8 + 5] = rt2 - rt1; P.prof2[(size_t)ci * 8 + 6] = PROF_NOW() - rt2; P.prof2[(size_t)ci * 8 + 0] = rows_t[0] + rows_t[3]; P.prof2[(size_t)ci * 8 + 1] = rows_t[1]; P.prof2[(size_t)ci * 8 + 2] = rows_t[2]; }   // (the last member's: prologue, its runs, meeting + envelope)
#endif
    }
}

extern "C" int sdsm_setup_class(int max_dim, int max_mcap, int max_label, int k)
{
    if (max_dim <= SetupLimS::DIM && max_mcap <= SetupLimS::GRID && max_label <= SetupLimS::LABELS && k * k <= SetupLimS::PSFW) return 0;
    if (max_dim <= SetupLimM::DIM && max_mcap <= SetupLimM::GRID) return 1;
    return 2;
}

extern "C" int sdsm_setup_fits_small(int h, int w, int mcap, int max_label, int k)
{
    return h <= SetupLimS::DIM && w <= SetupLimS::DIM && mcap <= SetupLimS::GRID && max_label <= SetupLimS::LABELS && k * k <= SetupLimS::PSFW;
}

extern "C" hipError_t sdsm_launch_setup(const BatchParams &P, hipStream_t stream, const int32_t *order_w, int n_w, int cls, const int32_t *order_small, int n_small,
                                        const int32_t *order_big, int n_big)
{
    BatchParams Pb = P;
    if (n_small > 0) {                                   // the large regions first (longest), the many small ones behind them: the two launches overlap
        Pb.order = order_big; Pb.n = n_big;
        BatchParams Ps = P;
        Ps.order = order_small; Ps.n = n_small;
        if (n_big > 0) {
            if (cls == 1) hipLaunchKernelGGL(sdsm_k_setup<SetupLimM>, dim3(n_big), dim3(SetupLimM::WG), 0, stream, Pb);
            else hipLaunchKernelGGL(sdsm_k_setup<SetupLimL>, dim3(n_big), dim3(SetupLimL::WG), 0, stream, Pb);
        }
        hipLaunchKernelGGL(sdsm_k_setup<SetupLimS>, dim3(n_small), dim3(SetupLimS::WG), 0, stream, Ps);
    }
    else if (cls == 0) hipLaunchKernelGGL(sdsm_k_setup<SetupLimS>, dim3(P.n), dim3(SetupLimS::WG), 0, stream, P);
    else if (cls == 1) hipLaunchKernelGGL(sdsm_k_setup<SetupLimM>, dim3(P.n), dim3(SetupLimM::WG), 0, stream, P);
    else hipLaunchKernelGGL(sdsm_k_setup<SetupLimL>, dim3(P.n), dim3(SetupLimL::WG), 0, stream, P);
    (void)order_w; (void)n_w;                            // the rows of the very large regions: sdsm_launch_setup_rows, on the stream of their workgroup groups
    return hipGetLastError();
}

// Rows of G~ of the very large regions (one workgroup per member of their workgroup groups): queued by sdsm_launch_solve on the side
// stream of the groups, in front of them -- no other solve class waits for it.
extern "C" hipError_t sdsm_launch_setup_rows(const BatchParams &P, hipStream_t stream, const int32_t *order_w, int n_w)
{
    if (n_w <= 0) return hipSuccess;
    BatchParams Pw = P;
    Pw.order = order_w; Pw.n = n_w;                      // (candidate | member << 24) of the workgroup groups
    hipLaunchKernelGGL(sdsm_k_setup_rows, dim3(n_w), dim3(SDSM_WG), (size_t)8 * (Pw.rows_mcap > 0 ? Pw.rows_mcap : 1), stream, Pw);
    return hipGetLastError();
}
