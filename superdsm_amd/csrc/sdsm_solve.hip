// Per-candidate convex solve: elliptical model, deformable shape model, final energy, mask, record.
// One persistent workgroup per candidate (256 or 512 threads by size class); all solver state lives in LDS.
//
// Reference behaviour restated here (never its code):
//   surface S, energy psi, gradient, Hessian   superdsm/dsm.py:86-94, 291-385   
//   solve protocol (elliptical, retry, DSM, fallback)   superdsm/objects.py:321-412   
//   moment initialisation                       superdsm/objects.py:287-296, dsm.py:96-111 (A7)
//   mask tail                                   superdsm/objects.py:198-209, dsm.py:113-128 (A9)
// The reference hands scale*psi to cvxopt.solvers.cp (dsm.py:488); cvxopt is a third-party dependency
// that is not part of the reference tree.  The solver here is the damped Newton iteration specified in
// DESIGN.md ("Solver"), the same algorithm and constants as the oracle's orc_newton (different elimination order and
// summation order, so iterates agree to rounding), run in a centred and scaled local polynomial basis (psi is
// invariant under affine re-parametrisation of theta).
//
// One full evaluation (psi, gradient, Hessian) of a candidate with n = 6 + M parameters:
//   every lane owns RUNS -- the (up to four) region pixels of one image row inside one aligned 4-column cell, which share the row
//   coordinate u and ONE list of G~ entries (a grid point with the float32 weights of the four pixels; 0 where the point is
//   outside a pixel's window): coalesced reads of the packed crop (4 y f64 + (row, col) + meta = 10 B / pixel) and of the run's
//   entries (16 B weights + 4 B column index per entry and lane), S of the four pixels, the logistic loss without libm, residuals r
//   and curvature weights d.  Everything that is summed over the pixels is summed in FIXED POINT: a contribution becomes an
//   integer multiple of 2^e (one fma against 1.5 * 2^(52 + e)) and is added with an integer atomic -- the polynomial (theta) parts
//   of gradient and Hessian as 21 coordinate moments in per-lane LDS slots, the xi part of the gradient (every entry of the run)
//   and of the Hessian (the run's leading entries, weight >= 10 % of the row maximum: the solver's approximate Hessian, stored as
//   an envelope) straight into their LDS entries, ONE atomic per entry and run (the four pixels' products are added first).
//   Integer sums do not depend on the order of the additions; psi -- whose terms have no a-priori bound -- is summed per chunk of
//   64 runs by a wavefront butterfly and the chunk totals are added in a fixed order.  A candidate's numbers therefore do not
//   depend on the workgroup size, the size class, the number of workgroups that share it or anything else in its launch.
//   One line-search sweep evaluates psi(x + t d) for 4 step lengths in a single pass (S is linear in t).
#include "sdsm_common.h"
#include "sdsm_logtab.h"
#include <climits>
#include <cstdlib>
#include <type_traits>

extern __shared__ __align__(16) unsigned char sdsm_smem[];

namespace {

#define LOG_DBL_MAX 709.782712893384
#define NEWTON_ABSTOL 1e-7
#define NEWTON_RELTOL 1e-6
#define LS_ALPHA 0.01
#define LS_BETA 0.5
#define LS_K 4            // step lengths per line-search sweep (8 needed 3 % fewer sweeps at twice the loss evaluations per sweep: two thirds of all of a solve's loss evaluations were line-search ones)
#define LS_SWEEPS 10      // t0 * 2^-(4 s + k): down to 2^-39
#define LS_T0_ELL 4.0     // elliptical solves start far from the optimum: longer first step
#if defined(SDSM_PROFILE) && defined(SDSM_PROFILE_FINE)
#define FINE_FENCE() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define FINE_ADD(slot) do { FINE_FENCE(); long long _t = PROF_NOW(); prof_acc[slot] += _t - ft; ft = _t; } while (0)
#define FINE_START() long long ft = PROF_NOW()
#else
#define FINE_ADD(slot) do { } while (0)
#define FINE_START() do { } while (0)
#endif
#ifndef EBATCH
#define EBATCH 4          // entries of a run requested together (16-byte weights + 4-byte index per lane and entry)
#endif
#ifndef SDSM_FACTOR_DIV
#define SDSM_FACTOR_DIV 1
#ifndef SDSM_K1_THREADS
#define SDSM_K1_THREADS 192
#endif
#ifndef SDSM_K1_WPE
#define SDSM_K1_WPE 3             // wavefronts per SIMD the throughput-mode class 1 is compiled for (4 = 128 registers: spills, measured slower)
#endif
#ifndef SDSM_K1B_THREADS
#define SDSM_K1B_THREADS 256       // (384 threads at three wavefronts per SIMD: synthetic 4096^2 93 vs 84 ms)
#define SDSM_K1B_WPE 2
#endif
#endif
#define HZREG SDSM_HZREG   // leading ('significant') entries of a run unrolled for the approximate Hessian
#define NSLOT SDSM_MSLOTS

// Compile-time LDS layout (offsets in doubles from the start of dynamic LDS).
// The Hessian lives in ENVELOPE storage (BatchParams.env_fst / env_rb): unknowns ordered xi_0 .. xi_{M-1}, theta_0 ..
// theta_5 ("logical" order i = 0 .. n-1), row i stores columns fst[i] .. i at rb[i] + column.  The xi block of the
// thresholded Hessian is banded (a pixel couples only the few grid points around it), so the envelope is a fraction of
// the dense triangle and the Cholesky factor (computed in place) fills only the envelope.
template <int NMAX_, int EMAX_, bool GLOBALH = false, int WGSIZE = 256>
struct Lay {
    static constexpr int NMAX = NMAX_, EMAX = EMAX_;
    static constexpr int WGS = WGSIZE, NWAVES = WGSIZE / 64;
    static constexpr bool GLOBAL_H = GLOBALH;      // Hessian in global memory (envelope larger than LDS)
    static constexpr int W = NMAX + 2;             // vectors are indexed up to n (right-hand-side row) inclusive
    static constexpr int LT = 0, B0 = 2 * SDSM_LOGTAB_N;   // the table of log_w (the same address in every class), then the vectors
    static constexpr int X = B0, G = B0 + W, D = B0 + 2 * W, XT = B0 + 3 * W, SC = B0 + 4 * W, YROW = B0 + 5 * W, TMP = B0 + 6 * W;
    static constexpr int RED = B0 + 7 * W, FLAG = RED + NWAVES * 32, MS = FLAG + 2, IB = MS + 42 * NSLOT;   // MS: fixed-point moment accumulators, 21 sums x 2 words x NSLOT lane slots
    static constexpr int NPANEL = NMAX / SDSM_PANEL + 2;
    static constexpr int IB_DOUBLES = (2 * W + NPANEL + 1) / 2;      // int arrays: rb[W], fst[W], rend[NPANEL]
    static constexpr int HP = IB + IB_DOUBLES;
    static constexpr int END = HP + (GLOBALH ? 0 : EMAX);
    static constexpr int TOTAL_BYTES = ((END * 8 + 15) / 16) * 16;
};

#define SD ((double *)sdsm_smem)

#ifdef SDSM_PROFILE
#define PROF_ARG , prof_acc
#define PROF_PARAM , long long *prof_acc
#else
#define PROF_ARG
#define PROF_PARAM
#endif

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef const f64x2 SDSM_GLOBAL *g_cf64x2_p;

struct Cand {                       // per-candidate global pointers (already offset) and scalars
    g_cf64x2_p crop_y2;                 // run p: elements 2 p, 2 p + 1 (four intensities)
    g_cu32_p crop_rc;                   // (row << 16) | first column of the cell
    g_cu32_p meta;                      // entries | leading entries << 12 | present pixels << 24
    g_cu32_p aux;                       // [chunk of 64 positions]: entries the rows of the chunk are padded to
    g_cf32x4_p ell_w4;                  // entry j of run position p: element j * NR + p (the four pixels' weights)
    g_cu32_p ell_im;                    // (column | leading-for-pixel bits << 16)
    double *hglob;                      // flat pointer: Hessian of the global-memory class
    int N, NR, hzmax, env_size;         // region pixels, runs
    int p_lo, p_hi;                     // the run positions this workgroup covers in a pass (all of them unless it is one of a group)
    int wg, wG;                         // member index and size of the workgroup group (wG = 1: none)
    double *wpool;                      // the group's block of BatchParams.wide_pool
    long long wtimeout;                 // BatchParams.wide_timeout
    double rmid, cmid, inv_hr, inv_hc;   // local coordinates u = (r - rmid) * inv_hr
    double scale, epsilon, alpha;
    double reg0;                        // alpha * sqrt(epsilon) * M: the regulariser's value at xi = 0 (dsm.py:325-326)
    int yexp;                           // |y| < 2^yexp over the region (setup kernel): scale of the fixed-point sums (fx_exponents)
    int boost;                          // a long chain of its launch (BatchParams.boost_pixels): passes over the pixels at a raised issue priority
};

// The evaluators are real (non-inlined) functions and receive the candidate by reference: its fields then come out of a
// stack object through vector loads and look lane-varying to the compiler (64-bit address arithmetic in VALU, every field
// in VGPRs).  They are the same for all lanes: readfirstlane moves them to SGPRs (scalar address arithmetic, SGPR-base
// global loads, ~40 VGPRs less).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double uni(double v) { return __longlong_as_double((long long)uni((unsigned long long)__double_as_longlong(v))); }
template <class T> __device__ __forceinline__ T SDSM_GLOBAL *uni(T SDSM_GLOBAL *p) { return (T SDSM_GLOBAL *)uni((unsigned long long)p); }

// The candidate's descriptor and state are the same for every lane but arrive through vector loads (the compiler cannot prove
// the workspace read-only): move every field to SGPRs once, instead of holding ~50 VGPRs for the whole solve.
__device__ __forceinline__ long long uni(long long v) { return (long long)uni((unsigned long long)v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ CandDesc uniform_desc(const CandDesc &d)
{
    CandDesc u;
    u.crop_off = uni((long long)d.crop_off); u.ell_off = uni((long long)d.ell_off); u.mask_off = uni((long long)d.mask_off); u.xi_off = uni((long long)d.xi_off);
    u.N = uni(d.N); u.r0 = uni(d.r0); u.c0 = uni(d.c0); u.h = uni(d.h); u.w = uni(d.w); u.fp_off = uni(d.fp_off); u.fp_len = uni(d.fp_len);
    u.Mcap = uni(d.Mcap); u.perm_inv = uni(d.perm_inv); u.wide_g = uni(d.wide_g);
    u.hglob_off = uni((long long)d.hglob_off); u.wide_off = uni((long long)d.wide_off); u.image = uni(d.image); u.NRcap = uni(d.NRcap);
    u.run_off = uni((long long)d.run_off); u.rows_g = uni(d.rows_g); u.pad1 = 0;
    return u;
}
__device__ __forceinline__ CandState uniform_state(const CandState &d)
{
    CandState u;
    u.M = uni(d.M); u.status = uni(d.status); u.hc = uni(d.hc); u.wc = uni(d.wc); u.npos = uni(d.npos); u.zmax = uni(d.zmax);
    u.sum_r = uni(d.sum_r); u.sum_c = uni(d.sum_c); u.sum_rr = uni(d.sum_rr); u.sum_cc = uni(d.sum_cc);
    u.hzmax = uni(d.hzmax); u.env_size = uni(d.env_size); u.nneg = uni(d.nneg); u.yexp = uni(d.yexp); u.NR = uni(d.NR);
    return u;
}
__device__ __forceinline__ Cand uniform_cand(const Cand &c)
{
    Cand u;
    u.crop_y2 = uni(c.crop_y2); u.crop_rc = uni(c.crop_rc); u.meta = uni(c.meta); u.aux = uni(c.aux); u.ell_w4 = uni(c.ell_w4); u.ell_im = uni(c.ell_im);
    u.hglob = (double *)uni((unsigned long long)c.hglob);
    u.N = uni(c.N); u.NR = uni(c.NR); u.hzmax = uni(c.hzmax); u.env_size = uni(c.env_size);
    u.p_lo = uni(c.p_lo); u.p_hi = uni(c.p_hi); u.wg = uni(c.wg); u.wG = uni(c.wG); u.wpool = (double *)uni((unsigned long long)c.wpool); u.wtimeout = uni(c.wtimeout);
    u.rmid = uni(c.rmid); u.cmid = uni(c.cmid); u.inv_hr = uni(c.inv_hr); u.inv_hc = uni(c.inv_hc);
    u.scale = uni(c.scale); u.epsilon = uni(c.epsilon); u.alpha = uni(c.alpha); u.reg0 = uni(c.reg0);
    u.yexp = uni(c.yexp); u.boost = uni(c.boost);
    return u;
}

// The thread index as the optimiser cannot see through it: the per-thread addresses of a pass (base + tid) are then formed
// inside the pass, not hoisted to the top of the kernel and kept alive (or spilled) across the whole solver.
__device__ __forceinline__ int opaque_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }

// A constant materialised where it is used: the optimiser otherwise loads the rarely used 64-bit constants of the solver logic
// into registers at the top of the kernel and keeps (or spills) them across every pass.
__device__ __forceinline__ double fresh(double v) { asm volatile("" : "+s"(v)); return v; }

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

// Sum of f(0) .. f(cnt - 1), computed by EVERY wavefront for itself: lane l adds f(l), f(l + 64), ... in that order, then the xor
// butterfly (whose result is the same in all lanes).  The same bits in every wavefront and for every workgroup size; no barrier.
template <class F>
__device__ __forceinline__ double wave_canon_sum(int cnt, F &&f)
{
    double s = 0;
    for (int j = (int)(threadIdx.x & 63); j < cnt; j += 64) s += f(j);
    return wave_sum(s);
}

template <class L> __device__ __forceinline__ double *hess_ptr(const Cand &c) { if constexpr (L::GLOBAL_H) return c.hglob; else return SD + L::HP; }
#define RBP ((int *)(SD + L::IB))              // rb[i]: entry (i, j) of the Hessian / factor is at rb[i] + j (logical order)
#define FSTP (RBP + L::W)                      // fst[i]: first stored column of row i (0 for the theta rows)
#define RENDP (FSTP + L::W)                    // rend[p]: last xi row that has entries in the columns of panel p

// ---- workgroup groups (WIDE): wG workgroups solve one very large candidate.  Every member runs the whole solver on its
// own copy of the state; only the passes over the pixels are shared: a member covers crop positions [p_lo, p_hi) and the
// partial sums are ALL-REDUCED through global memory -- every member publishes its partials, the group meets at a
// barrier, every member adds the wG partials in the same order, so all copies stay bit-identical and take the same
// branches.  Barrier: a monotonic counter, arrival = agent-scope release add by one lane after the workgroup barrier,
// wait = agent-scope acquire loads by that lane (the members sit on different XCDs, whose L2s are not coherent), with a
// time limit: a member that waits longer than ~10 s flags the group and every member gives the candidate up (status
// SDSM_CAND_GIVEN_UP: the host solves it again without a group) instead of hanging.  Two publication slots alternate: a
// member can be at most one all-reduce ahead.
// Residency: a member is not bound to its workgroup index.  Every workgroup of the group launch draws a TICKET when it
// starts (one atomic add on BatchParams.wide_ticket) and becomes entry `ticket` of the launch list, in which the members of
// a group are consecutive.  Tickets are handed out in the order in which workgroups actually start, whatever that order is,
// so at any time the running workgroups hold a dense prefix of the list: every group except at most the last one of the
// prefix is complete and finishes on its own; the incomplete one holds fewer than its size (<= 8) compute units and gets
// the next tickets as soon as any compute unit frees up.  K group launches in flight (streams, processes) can hold at most
// 7 K compute units in incomplete groups, so with K < 36 on 256 compute units somebody always progresses.
#define WIDE_PHASE (reinterpret_cast<int *>(SD + L::FLAG) + 2)   // all-reduces done so far (LDS, uniform)

template <class L>
__device__ __forceinline__ bool wide_barrier(const Cand &c, int phase0)
{
    int *sync = reinterpret_cast<int *>(c.wpool);            // [0] arrivals, [1] error flag
    __syncthreads();                                         // everybody has read the phase and published its data
    const int phase = phase0 + 1;
    if (threadIdx.x == 0) {
        *WIDE_PHASE = phase;
        __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int target = phase * c.wG;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (wall_clock64() - t0 > c.wtimeout) { __hip_atomic_store(&sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // constant 100 MHz clock; default 10 s
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    return __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// One exchange of a workgroup group.  Publication slots alternate per OPERATION (WIDE_OPS): a member can be at most one operation
// ahead of the slowest one.  A member's block of a slot: [0, WIDE_F0) integer payload, [WIDE_F0] number of psi-type sums that follow.
#define WIDE_OPS (reinterpret_cast<int *>(SD + L::FLAG) + 3)
#define WIDE_RS_MIN 512
#define WIDE_F0 (SDSM_WIDE_PBUF - SDSM_WIDE_FCAP)
template <class L>
__device__ __forceinline__ double *wide_mine(const Cand &c)
{
    return c.wpool + SDSM_WIDE_SYNC + (size_t)(*WIDE_OPS & 1) * c.wG * SDSM_WIDE_PBUF + (size_t)c.wg * SDSM_WIDE_PBUF;
}

// Completes the exchange: (i) up to three arrays of FIXED-POINT sums (raw 64-bit integers, in LDS or global memory) are all-reduced in
// place by integer addition -- order-free, so the totals do not depend on the number of members; long vectors (the Hessian envelope:
// ~10 k entries) as reduce-scatter + all-gather: member g adds the wG partials of ITS share of the entries and leaves the totals in its
// own publication block, a second group barrier, then everybody reads the totals -- 2 instead of wG reads per entry and member.
// (ii) psi-type sums: every member has published nf (at [WIDE_F0]) super-chunk sums of K values each; tot[k] = all of them, added one
// by one in position order (members cover consecutive slices): exactly the additions a single workgroup makes (run_pass).
// Returns false if the group was given up.
template <class L, int K>
__device__ __forceinline__ bool wide_finish(const Cand &c, double (&tot)[K], double *a0 = nullptr, int n0 = 0, double *a1 = nullptr, int n1 = 0, double *a2 = nullptr, int n2 = 0)
{
    typedef unsigned long long u64;
    const int phase = *WIDE_PHASE, ops = *WIDE_OPS;
    double *slotd = c.wpool + SDSM_WIDE_SYNC + (size_t)(ops & 1) * c.wG * SDSM_WIDE_PBUF;
    u64 *slot = reinterpret_cast<u64 *>(slotd);
    u64 *mine = slot + (size_t)c.wg * SDSM_WIDE_PBUF;
    u64 *b0 = reinterpret_cast<u64 *>(a0), *b1 = reinterpret_cast<u64 *>(a1), *b2 = reinterpret_cast<u64 *>(a2);
    const int tid = threadIdx.x;
    for (int e = tid; e < n0; e += L::WGS) mine[e] = b0[e];
    for (int e = tid; e < n1; e += L::WGS) mine[n0 + e] = b1[e];
    for (int e = tid; e < n2; e += L::WGS) mine[n0 + n1 + e] = b2[e];
    bool ok = wide_barrier<L>(c, phase);
    if (tid == 0) *WIDE_OPS = ops + 1;                           // (everybody read it before the barrier above)
    const int nt = n0 + n1 + n2;
    if (nt < WIDE_RS_MIN) {
        for (int e = tid; e < nt; e += L::WGS) {
            u64 v = 0;
            for (int m = 0; m < c.wG; m++) v += slot[(size_t)m * SDSM_WIDE_PBUF + e];
            if (e < n0) b0[e] = v; else if (e < n0 + n1) b1[e - n0] = v; else b2[e - n0 - n1] = v;
        }
    } else {
        const int chunk = (((nt + c.wG - 1) / c.wG) + 7) & ~7;
        const int lo = c.wg * chunk, hi = lo + chunk < nt ? lo + chunk : nt;
        for (int e = lo + tid; e < hi; e += L::WGS) {
            u64 v = 0;
            for (int m = 0; m < c.wG; m++) v += slot[(size_t)m * SDSM_WIDE_PBUF + e];
            mine[e] = v;
        }
        ok = wide_barrier<L>(c, phase + 1) && ok;
        for (int e = tid; e < nt; e += L::WGS) {
            const u64 v = slot[(size_t)(e / chunk) * SDSM_WIDE_PBUF + e];
            if (e < n0) b0[e] = v; else if (e < n0 + n1) b1[e - n0] = v; else b2[e - n0 - n1] = v;
        }
    }
    double run[K];
#pragma unroll
    for (int k = 0; k < K; k++) run[k] = 0;
    for (int m = 0; m < c.wG; m++) {
        const double *f = slotd + (size_t)m * SDSM_WIDE_PBUF + WIDE_F0;
        const int nf = (int)f[0];
        for (int j = 0; j < nf; j++) {
#pragma unroll
            for (int k = 0; k < K; k++) run[k] += f[1 + j * K + k];
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) tot[k] = ok ? run[k] : NAN;     // group given up: every member sees a non-finite value and fails the solve
    __syncthreads();
    return ok;
}

// ---- logistic loss without the math library: the passes are instruction-issue bound and a pixel costs 1 (full pass) or 8
// (line-search sweep) evaluations of log(1 + exp(-t)); libm's exp + log are ~100 FP64 instructions, these ~55.
// e^-a for 0 <= a <= 750: a = k ln2 + r, |r| <= ln2 / 2, degree-13 Taylor polynomial (1.8e-16 relative), ldexp.
__device__ __forceinline__ double exp_neg(double a)
{
    const double kf = __builtin_rint(a * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, a);
    r = fma(kf, -1.90821492927058770002e-10, r);
    const double x = -r;
    double p = 1.0 / 6227020800.0;
    p = fma(p, x, 1.0 / 479001600.0);
    p = fma(p, x, 1.0 / 39916800.0);
    p = fma(p, x, 1.0 / 3628800.0);
    p = fma(p, x, 1.0 / 362880.0);
    p = fma(p, x, 1.0 / 40320.0);
    p = fma(p, x, 1.0 / 5040.0);
    p = fma(p, x, 1.0 / 720.0);
    p = fma(p, x, 1.0 / 120.0);
    p = fma(p, x, 1.0 / 24.0);
    p = fma(p, x, 1.0 / 6.0);
    p = fma(p, x, 0.5);
    p = fma(p, x, 1.0);
    p = fma(p, x, 1.0);
    return ldexp(p, -(int)kf);
}

// 1 / x, x in the normal range: hardware estimate (4.6e-8 relative, tools/microbench/rsq_accuracy.hip) + ONE Newton step =
// 2.2e-15 relative (20 ulp); a second step would give 1 ulp.  Enough here: log_1_2 corrects its quotient with a residual
// step, and theta only needs the accuracy of the gradient.
__device__ __forceinline__ double rcp_f64(double x)
{
    const double y = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, y, 1.0), y, y);
}

// The table of log_w at the start of dynamic LDS (once per workgroup; a barrier follows before any pass over the pixels).
__device__ __forceinline__ void load_log_table(int tid)
{
    if (tid < SDSM_LOGTAB_N) reinterpret_cast<f64x2 *>(SD)[tid] = reinterpret_cast<const f64x2 *>(sdsm_logtab)[tid];
    __syncthreads();
}

// log(w), w = 1 + u, 0 <= u <= 1: w = c_j (1 + s) with c_j = 1 + j / 64 the nearest table point (|s| <= 1 / 128), log w = log c_j +
// log1p(s); the table (LDS, sdsm_logtab.h: 1 / c_j rounded and the logarithm of ITS reciprocal, so s = fma(w, 1 / c_j, -1) is exact for
// what the table means) and a degree-7 series: 1.1e-16 absolute, 5e-16 relative (w = 1 gives 0, w = 2 the rounded ln 2).  12
// instructions; the atanh series without a table took 40 and eleven 64-bit constants (round 4).
__device__ __forceinline__ double log_w(double u, double w)
{
    const int j = (int)fma(u, 64.0, 0.5);
    const f64x2 e = reinterpret_cast<const f64x2 *>(SD)[j];
    const double s = fma(w, e.x, -1.0);
    double p = 1.0 / 7.0;
    p = fma(p, s, -1.0 / 6.0);
    p = fma(p, s, 0.2);
    p = fma(p, s, -0.25);
    p = fma(p, s, 1.0 / 3.0);
    p = fma(p, s, -0.5);
    p = fma(p, s, 1.0);
    return fma(s, p, e.y);
}

// log(1 + exp(-t))   (dsm.py:298-300, 319-322: log(1 + h), h = exp(-t); -t below the exp guard -- the same value)
__device__ __forceinline__ double softplus_neg(double t)
{
    double a = fabs(t);
    a = a < 750.0 ? a : 750.0;                           // (NaN -> 750)
    const double u = exp_neg(a);                         // exp(-|t|) in [0, 1]
    return (log_w(u, 1.0 + u) + fmax(-t, 0.0)) + t * 0.0;   // (t * 0: a non-finite t gives NaN -- every caller treats a non-finite psi as a failed evaluation)
}

// loss terms of one pixel given t = y * S     (dsm.py:298-300, 306-310, 319-322, 344, 361-366).  A non-finite t makes phi NaN
// (the evaluation then counts as failed, whatever r and dcurv are).
__device__ __forceinline__ void loss_terms(double yv, double S, double *phi, double *r, double *dcurv)
{
    const double t = yv * S;
    double a = fabs(t);
    a = a < 750.0 ? a : 750.0;
    const double u = exp_neg(a);                         // exp(-|t|)
    const double w = 1.0 + u;
    *phi = (log_w(u, w) + fmax(-t, 0.0)) + t * 0.0;
    const double rw = rcp_f64(w);
    const double q = u * rw;
    const double theta = t >= 0 ? q : rw;                // h / (1 + h), h = exp(-t)
    *r = -yv * theta;
    // kappa = theta - theta^2 (dsm.py:361) = u / (1 + u)^2: the product form has no cancellation for theta -> 1 (the reference's
    // difference loses all digits there: kappa < 1e-16 comes out as 0 or 1.1e-16)
    *dcurv = (yv * yv) * (q * rw);
}

// ---------------------------------------------------------------------------------------------------------
// A pass over the runs.  `body(p, active, v)` handles run position p (active == false: a lane beyond the slice, v must come out 0)
// and returns the run's K psi-type values; everything else a pass sums goes through fixed-point atomics inside the body.
// The K values are summed per CHUNK of 64 positions by a wavefront butterfly (a chunk is the same set of runs whatever the
// workgroup size), the chunk totals in a fixed order: sequentially inside a super-chunk of SDSM_SUPER chunks, then the super-chunk
// sums sequentially -- tot does not depend on the workgroup size nor (wide_finish) on the number of workgroups sharing the
// candidate, whose slices are whole super-chunks.  Chunk totals live in L::TMP (free outside factor_solve); slices with more
// chunks than fit there are walked in segments.  wG > 1: the super-chunk sums are PUBLISHED, the caller completes the exchange
// (wide_finish); tot is then undefined here.
// ---------------------------------------------------------------------------------------------------------
#ifndef SDSM_BOOST_PRIO
#define SDSM_BOOST_PRIO 1
#endif
__device__ __forceinline__ void pass_priority(const Cand &c)
{
    if (c.boost) __builtin_amdgcn_s_setprio(SDSM_BOOST_PRIO); else __builtin_amdgcn_s_setprio(0);
}

template <class L, int K, class Body>
__device__ __forceinline__ void run_pass(const Cand &c, double (&tot)[K], Body &&body)
{
    constexpr int SEGC = ((L::W / K) / SDSM_SUPER) * SDSM_SUPER;     // chunks per segment
    static_assert(SEGC >= SDSM_SUPER, "chunk totals do not fit");
    double *ct = SD + L::TMP;
    const int tid = opaque_tid(), lane = tid & 63;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = 0;
    double *pub = c.wG > 1 ? wide_mine<L>(c) + WIDE_F0 : nullptr;
    int npub = 0;
    for (int seg0 = c.p_lo; seg0 < c.p_hi; seg0 += SEGC * 64) {
        const int seg1 = seg0 + SEGC * 64 < c.p_hi ? seg0 + SEGC * 64 : c.p_hi;
        pass_priority(c);
        for (int pb = seg0 + (tid & ~63); pb < seg1; pb += L::WGS) {
            const int p = pb + lane;
            double v[K];
            body(p, p < seg1, v);
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < K; k++) ct[((pb - seg0) >> 6) * K + k] = v[k];
            }
        }
        __builtin_amdgcn_s_setprio(2);
        __syncthreads();
        const int nch = (seg1 - seg0 + 63) >> 6;
        if (c.wG <= 1) {
            for (int sc = 0; sc < nch; sc += SDSM_SUPER) {           // (every thread: the same LDS words in the same order)
                double part[K];
#pragma unroll
                for (int k = 0; k < K; k++) part[k] = 0;
                const int e1 = sc + SDSM_SUPER < nch ? sc + SDSM_SUPER : nch;
                for (int i = sc; i < e1; i++) {
#pragma unroll
                    for (int k = 0; k < K; k++) part[k] += ct[i * K + k];
                }
#pragma unroll
                for (int k = 0; k < K; k++) acc[k] += part[k];
            }
        } else {
            // The publication block holds the super-chunk sums of a slice of the largest admissible region dealt to the largest group
            // (static_assert below); a smaller group of such a region (latency mode) does not fit: the group is given up -- flagged, every
            // member sees it at the next exchange, the host solves the candidate again without a group -- instead of losing sums.
            static_assert(SDSM_WIDE_FCAP - 8 >= LS_K * (((long long)SDSM_MAX_BBOX_DIM * (SDSM_MAX_BBOX_DIM / SDSM_RUN)) / ((long long)SDSM_WIDE_MAX_G * 64 * SDSM_SUPER)),
                          "SDSM_WIDE_FCAP: super-chunk sums of one member's slice");
            const int nsc = (nch + SDSM_SUPER - 1) / SDSM_SUPER;
            const bool fits = npub + nsc * K <= SDSM_WIDE_FCAP - 8;
            if (!fits && tid == 0) __hip_atomic_store(reinterpret_cast<int *>(c.wpool) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int t = tid; fits && t < nsc * K; t += L::WGS) {
                const int sc = (t / K) * SDSM_SUPER, k = t % K;
                const int e1 = sc + SDSM_SUPER < nch ? sc + SDSM_SUPER : nch;
                double part = 0;
                for (int i = sc; i < e1; i++) part += ct[i * K + k];
                pub[1 + npub + t] = part;
            }
            if (fits) npub += nsc * K;
        }
        __syncthreads();                                             // ct is rewritten by the next segment / the next pass
    }
    if (c.wG > 1) { if (tid == 0) pub[0] = (double)(npub / K); }
#pragma unroll
    for (int k = 0; k < K; k++) tot[k] = uni(acc[k]);
}

// The pixels of a run: local coordinates (one row: u; consecutive columns: v[k]), intensities (0 for the absent ones), presence bits.
struct RunPix { double u, v[SDSM_RUN], y[SDSM_RUN]; unsigned pm; int nnz, hz; };
// Loads through a uniform base (SGPR pair) and a 32-bit byte offset per lane (the `saddr` form of the global loads: no 64-bit address
// arithmetic in the vector ALUs).  The G~ block of a candidate is below 4 GB (sdsm_k_setup refuses larger ones).
template <class T> __device__ __forceinline__ T gld(const T SDSM_GLOBAL *base, unsigned byte_off)
{
    return *reinterpret_cast<const T SDSM_GLOBAL *>(reinterpret_cast<const char SDSM_GLOBAL *>(base) + byte_off);
}
// A lane beyond the slice reads the slice's last run (no branch around the loads) and treats it as empty: no pixels, no entries, y = 0.
__device__ __forceinline__ int load_pos(const Cand &c, int p, bool active) { return active ? p : c.p_hi - 1; }
__device__ __forceinline__ void load_run(const Cand &c, int pl, bool active, RunPix &rp)
{
    const uint32_t meta = gld(c.meta, (unsigned)pl * 4u), rc = gld(c.crop_rc, (unsigned)pl * 4u);
    const f64x2 ya = gld(c.crop_y2, (unsigned)pl * 32u), yb = gld(c.crop_y2, (unsigned)pl * 32u + 16u);
    rp.y[0] = active ? ya.x : 0.0; rp.y[1] = active ? ya.y : 0.0; rp.y[2] = active ? yb.x : 0.0; rp.y[3] = active ? yb.y : 0.0;
    rp.pm = active ? meta >> 24 : 0u; rp.nnz = active ? (int)(meta & 0xfffu) : 0; rp.hz = active ? (int)((meta >> 12) & 0xfffu) : 0;
    rp.u = ((double)(rc >> 16) - c.rmid) * c.inv_hr;
    const int col0 = (int)(rc & 0xffffu);
#pragma unroll
    for (int k = 0; k < SDSM_RUN; k++) rp.v[k] = ((double)(col0 + k) - c.cmid) * c.inv_hc;
}
__device__ __forceinline__ void poly_surface(const RunPix &rp, const double *xv, double (&S)[SDSM_RUN])
{
    const double a = (rp.u * rp.u) * xv[0] + (2 * rp.u) * xv[3] + xv[5], b2 = 2 * rp.u * xv[2] + 2 * xv[4];
#pragma unroll
    for (int k = 0; k < SDSM_RUN; k++) S[k] = (rp.v[k] * rp.v[k]) * xv[1] + rp.v[k] * b2 + a;
}

// entries the rows of the chunk of position p are padded to (uniform: one scalar load per wavefront and chunk)
__device__ __forceinline__ int chunk_entries(const Cand &c, int pb) { return (int)c.aux[pb >> 6]; }

// The entries of a run, EBATCH at a time: entry j of run position pl is element j * NR + pl of the candidate's block -- byte offsets
// (32 bits) that advance by NR elements per entry.  kh is uniform (the chunk's padded entry count): the tests on it are scalar branches.
struct EntryCursor { unsigned ow, oi, sw, si; };
__device__ __forceinline__ EntryCursor entry_cursor(const Cand &c, int pl)
{
    EntryCursor e; e.ow = (unsigned)pl * 16u; e.oi = (unsigned)pl * 4u; e.sw = (unsigned)c.NR * 16u; e.si = (unsigned)c.NR * 4u;
    return e;
}
__device__ __forceinline__ void load_entries(const Cand &c, EntryCursor &e, int j0, int kh, f32x4 (&w)[EBATCH], uint32_t (&im)[EBATCH])
{
#pragma unroll
    for (int t = 0; t < EBATCH; t++) {
        w[t].x = 0.f; w[t].y = 0.f; w[t].z = 0.f; w[t].w = 0.f; im[t] = 0;
        if (j0 + t < kh) { w[t] = gld(c.ell_w4, e.ow + (unsigned)t * e.sw); im[t] = gld(c.ell_im, e.oi + (unsigned)t * e.si); }
    }
    e.ow += EBATCH * e.sw; e.oi += EBATCH * e.si;
}

// S += G~ xi for the four pixels of the run at position pl: all loads of a batch before the first use
template <int STRIDE>
__device__ __forceinline__ void smooth_add(const Cand &c, const double *xv, int pl, int kh, double (&S)[SDSM_RUN])
{
    EntryCursor ec = entry_cursor(c, pl);
    for (int j0 = 0; j0 < kh; j0 += EBATCH) {
        f32x4 w[EBATCH];
        uint32_t im[EBATCH];
        load_entries(c, ec, j0, kh, w, im);
#pragma unroll
        for (int t = 0; t < EBATCH; t++) {
            if (j0 + t < kh) {
                const double xi = xv[STRIDE * (6 + (int)(im[t] & 0xffffu))];
                S[0] += (double)w[t].x * xi; S[1] += (double)w[t].y * xi; S[2] += (double)w[t].z * xi; S[3] += (double)w[t].w * xi;
            }
        }
    }
}

// The same for S(x) and S(d) at once: xd holds the pairs (x_j, d_j), one 16-byte LDS read per entry
__device__ __forceinline__ void smooth_add2(const Cand &c, const double *xd, int pl, int kh, double (&S0)[SDSM_RUN], double (&Sd)[SDSM_RUN])
{
    EntryCursor ec = entry_cursor(c, pl);
    for (int j0 = 0; j0 < kh; j0 += EBATCH) {
        f32x4 w[EBATCH];
        uint32_t im[EBATCH];
        load_entries(c, ec, j0, kh, w, im);
#pragma unroll
        for (int t = 0; t < EBATCH; t++) {
            if (j0 + t < kh) {
                const double *pr = xd + 2 * (6 + (int)(im[t] & 0xffffu));
                const double xi = pr[0], di = pr[1];
                const double w0 = (double)w[t].x, w1 = (double)w[t].y, w2 = (double)w[t].z, w3 = (double)w[t].w;
                S0[0] += w0 * xi; S0[1] += w1 * xi; S0[2] += w2 * xi; S0[3] += w3 * xi;
                Sd[0] += w0 * di; Sd[1] += w1 * di; Sd[2] += w2 * di; Sd[3] += w3 * di;
            }
        }
    }
}

// psi only (final step check, start values).  Result broadcast to all threads.
template <class L>
__device__ __forceinline__ double eval_value(const Cand &c, int xo, int M)
{
    const double *xv = SD + xo;
    double tot[1];
    run_pass<L, 1>(c, tot, [&](int p, bool active, double (&v)[1]) {
        RunPix rp;
        const int pl = load_pos(c, p, active);
        load_run(c, pl, active, rp);
        double S[SDSM_RUN];
        poly_surface(rp, xv, S);
        if (M > 0) smooth_add<1>(c, xv, pl, chunk_entries(c, __builtin_amdgcn_readfirstlane(p)), S);
        double ps = 0;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) if ((rp.pm >> k) & 1u) ps += softplus_neg(rp.y[k] * S[k]);
        v[0] = ps;
    });
    if (c.wG > 1) wide_finish<L, 1>(c, tot);
    double psi = tot[0];
    if (M > 0) {                                         // dsm.py:323-331
        const double s2 = wave_canon_sum(M, [&](int j) { return sqrt(xv[6 + j] * xv[6 + j] + c.epsilon); });
        const double o2 = c.alpha * s2 - c.reg0;
        psi += o2 < 0 ? 0 : o2;
    }
    return uni(psi);                                     // every thread holds the same sum: scalar from here on (uniform control flow in the solver)
}

// Line search sweep: psi(x + t_k d) for LS_K step lengths t_k = t0 * 2^-k in ONE pass over the runs.  S is linear in
// the parameters, S(x + t d) = S(x) + t S(d): the entries of a run are fetched once, applied to xi and to d_xi (interleaved pairs
// (x_j, d_j) at L::XT .. so one 16-byte LDS read serves both), and only the loss is evaluated LS_K times per pixel.
template <class L>
__device__ __forceinline__ void eval_line(const Cand &c, int M, double t0, double (&out)[LS_K])
{
    const int tid = opaque_tid(), n = 6 + M;
    const double *x = SD + L::X, *d = SD + L::D;
    double *xd = SD + L::XT;
    __syncthreads();
    for (int i = tid; i < n; i += L::WGS) { xd[2 * i] = x[i]; xd[2 * i + 1] = d[i]; }
    __syncthreads();
    double ps[LS_K];
    run_pass<L, LS_K>(c, ps, [&](int p, bool active, double (&v)[LS_K]) {
        RunPix rp;
        const int pl = load_pos(c, p, active);
        load_run(c, pl, active, rp);
        double S0[SDSM_RUN], Sd[SDSM_RUN];
        {
            const double uu = rp.u * rp.u, u2 = 2 * rp.u;
            const double a0 = uu * xd[0] + u2 * xd[6] + xd[10], a1 = uu * xd[1] + u2 * xd[7] + xd[11];
            const double b0 = u2 * xd[4] + 2 * xd[8], b1 = u2 * xd[5] + 2 * xd[9];
#pragma unroll
            for (int k = 0; k < SDSM_RUN; k++) {
                const double vv = rp.v[k] * rp.v[k];
                S0[k] = vv * xd[2] + rp.v[k] * b0 + a0;
                Sd[k] = vv * xd[3] + rp.v[k] * b1 + a1;
            }
        }
        if (M > 0) smooth_add2(c, xd, pl, chunk_entries(c, __builtin_amdgcn_readfirstlane(p)), S0, Sd);
#pragma unroll
        for (int k2 = 0; k2 < LS_K; k2++) v[k2] = 0;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
            __builtin_amdgcn_sched_barrier(0);                           // one pixel's four loss evaluations at a time (registers)
            if ((rp.pm >> k) & 1u) {
                const double a0 = rp.y[k] * S0[k], a1 = rp.y[k] * Sd[k];
                double tk = t0;
#pragma unroll
                for (int k2 = 0; k2 < LS_K; k2++) {
                    if (k2 == LS_K / 2) __builtin_amdgcn_sched_barrier(0);   // two independent loss evaluations in flight, not four (registers)
                    v[k2] += softplus_neg(a0 + tk * a1); tk *= LS_BETA;
                }
            }
        }
    });
    if (c.wG > 1) wide_finish<L, LS_K>(c, ps);
    if (M > 0) {                                         // dsm.py:323-331
        double tk = t0;
#pragma unroll
        for (int k = 0; k < LS_K; k++) {
            const double rs = wave_canon_sum(M, [&](int j) { const double xi = xd[2 * (6 + j)] + tk * xd[2 * (6 + j) + 1]; return sqrt(xi * xi + c.epsilon); });
            const double o2 = c.alpha * rs - c.reg0;
            ps[k] += o2 < 0 ? 0 : o2;
            tk *= LS_BETA;
        }
    }
#pragma unroll
    for (int k = 0; k < LS_K; k++) out[k] = uni(ps[k]);
    __syncthreads();
}

// regulariser contributions to psi, gradient and Hessian diagonal (dsm.py:323-331, 349, 372-376).  reg_mu in [0, 1] blends the
// curvature from its true value alpha eps / (xi^2 + eps)^(3/2) towards alpha / sqrt(xi^2 + eps), the curvature of the
// quadratic majoriser of alpha sqrt(xi^2 + eps) (solver approximation only, DESIGN.md "Solver"; psi and the gradient are exact)
template <class L>
__device__ __forceinline__ double add_regulariser(const Cand &c, int M, double reg_mu)
{
    const double *xv = SD + L::X;
    double *g = SD + L::G, *Hp = hess_ptr<L>(c);
    for (int j = threadIdx.x; j < M; j += L::WGS) {
        double xi = xv[6 + j], t3 = xi * xi, t2 = sqrt(t3 + c.epsilon);
        g[6 + j] += c.alpha * (xi / t2);
        double gd = c.alpha * (1 / t2 - t3 / (t2 * t2 * t2));
        gd = gd < 0 ? 0 : gd;
        gd += reg_mu * (c.alpha / t2 - gd);
        Hp[RBP[j] + j] += gd;
    }
    const double s2 = wave_canon_sum(M, [&](int j) { return sqrt(xv[6 + j] * xv[6 + j] + c.epsilon); });
    double o2 = c.alpha * s2 - c.reg0;
    return o2 < 0 ? 0 : o2;
}

// ---------------------------------------------------------------------------------------------------------
// Polynomial part of the contributions to psi, gradient and Hessian, kept as MOMENTS of the coordinates: with
// q = (u^2, v^2, 2uv, 2u, 2v, 1) the 6 gradient entries sum r q_a and the 21 Hessian entries sum d q_a q_b are small multiples
// of sum r u^a v^b (a + b <= 2: 6 sums) and sum d u^a v^b (a + b <= 4: 15 sums) -- 21 accumulators instead of 27.
//   m[1..6] r * (1, u, v, u^2, uv, v^2); m[7..21] d * (1, u, v, u^2, uv, v^2, u^3, u^2 v, u v^2, v^3, u^4, u^3 v, u^2 v^2, u v^3, v^4)
// (m[0] is not used).  The pixels of a run share u: its contribution to the moments is u^a times sum_k r_k v_k^b, sum_k d_k v_k^b.
// ---------------------------------------------------------------------------------------------------------
#define NMOM 22
// gradient entry a = GF[a] * m[GM[a]]; Hessian entry t of the packed lower triangle (a >= b, t = a (a + 1) / 2 + b) = HF[t] * m[HM[t]]
__device__ __forceinline__ double moment_grad(const double *tot, int a)
{
    const int gm[6] = {4, 6, 5, 2, 3, 1};
    const double gf[6] = {1, 1, 2, 2, 2, 1};
    int mi = 1; double f = 1;
#pragma unroll
    for (int k = 0; k < 6; k++) { mi = a == k ? gm[k] : mi; f = a == k ? gf[k] : f; }
    return f * tot[mi];
}
__device__ __forceinline__ double moment_hess(const double *tot, int t)
{
    //                 00  10  11  20  21  22  30  31  32  33  40  41  42  43  44  50  51  52  53  54  55
    const int hm[21] = {17, 19, 21, 18, 20, 19, 13, 15, 14, 10, 14, 16, 15, 11, 12, 10, 12, 11,  8,  9,  7};
    const double hf[21] = {1,  1,  1,  2,  2,  4,  2,  2,  4,  4,  2,  2,  4,  4,  4,  1,  1,  2,  2,  2,  1};
    int mi = 7; double f = 1;
#pragma unroll
    for (int k = 0; k < 21; k++) { mi = t == k ? hm[k] : mi; f = t == k ? hf[k] : f; }
    return f * tot[mi];
}
// ---------------------------------------------------------------------------------------------------------
// Fixed-point accumulation.  Sums over the pixels go through LDS atomics, and an integer add (ds_add_u64, 12 clocks per wavefront
// instruction on scattered addresses) costs half of a floating-point one (ds_add_f64: 21; tools/microbench/lds_rates.hip) -- and
// integer sums do not depend on the order of the additions, so a candidate's results do not depend on how its pixels were dealt to
// lanes, wavefronts and workgroups.  A product a * b becomes an integer multiple of the unit 2^e by ONE instruction: fma(a, b, C)
// with C = 1.5 * 2^(52 + e) rounds a * b to a multiple of 2^e and leaves it, offset by the bits of C, in the mantissa (valid for
// |a b| < 2^(51 + e)); further products of the same entry are added by fma(a', b', s) before the one atomic.  e is chosen per
// candidate from the bound on a term (|y| < 2^yexp from the setup kernel, weights <= 1, |u|, |v| <= 1) and the number of pixels,
// so that a term has up to 48 bits (a run adds four of them before it rounds), and the sum of all terms stays below 2^62: for
// N <= 2048 pixels the resolution of a term is 2^-48 of the largest possible one.
// The bound follows the iteration: r = -y theta^ and d = y^2 kappa with theta^ = h / (1 + h), kappa = theta^ (1 - theta^) <= min(1/4,
// theta^), and theta^ <= phi for t >= 0 (phi (1 + h) log(1 + h) / h >= theta^) while a pixel with t < 0 alone has phi >= ln 2: every
// theta^ <= min(1, psi / ln 2).  An evaluation is given an upper bound of psi at its point (the value the line search just
// computed there, the value at the start of a speculative step -- which is discarded if psi turns out larger --, the elliptical
// energy at the start of the DSM solve) and scales its sums by 2^b >= that: on nearly separable regions, where psi and with it all
// residuals go to 0 (SURVEY: theta -> infinity), the sums keep their relative resolution.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fx_exponents(int yexp, int N, double psi_bound, int *hh, int *gh)
{
    const int lg = N > 1 ? 32 - __builtin_clz((unsigned)(N - 1)) : 0;       // ceil(log2 N)
    const int bits = 59 - lg < 48 ? 59 - lg : 48;
    int b = 0;                                                               // theta^ <= 2^b
    const double tb = psi_bound * 1.442696484;                               // (a little more than 1 / ln 2: rounding of the psi sums)
    if (tb < 1.0) {                                                          // (false for NaN / infinity: no bound)
        int ex = ((__double2hiint(tb) >> 20) & 0x7ff) - 1022;                // tb < 2^ex
        b = ex < -400 || tb <= 0 ? -400 : ex;
    }
    const int eh = 2 * yexp + (b < -1 ? b : -1) - bits, eg = yexp + b - bits;   // Hessian terms < 2^(2 yexp + min(b, -1)), gradient terms < 2^(yexp + b)
    *hh = ((1075 + eh) << 20) | 0x80000;
    *gh = ((1075 + eg) << 20) | 0x80000;
}
__device__ __forceinline__ double fx_const(int chi) { return __hiloint2double(chi, 0); }
// s = C + (sum of products, rounded to the unit): add its integer value to *addr
__device__ __forceinline__ void fx_commit(double *addr, double s, int chi)
{
    const unsigned long long v = ((unsigned long long)(unsigned)(__double2hiint(s) - chi) << 32) | (unsigned)__double2loint(s);
    atomicAdd(reinterpret_cast<unsigned long long *>(addr), v);
}
__device__ __forceinline__ void fx_add(double *addr, double a, double b, int chi) { fx_commit(addr, fma(a, b, fx_const(chi)), chi); }
// The same to 96 bits: the product rounded to the unit goes to addr[0 ..], its exact remainder in units of 2^-48 of the unit to
// addr[NSLOT ..].  For the coordinate moments of the polynomial part: theta has no regulariser, its gradient is a difference of
// large sums and the curvature of a nearly separable region comes from pixels whose weights span many orders of magnitude --
// a single 48-bit word below the LARGEST possible term (not the largest actual one: |y| is small where theta^ is large) lost 3-4
// digits against the reference's float64 sums there, and a handful of such solves ran into the iteration cap.
#define FX_LO_SHIFT 48
// The two words are added RAW -- the bits of C + k unit are the bits of C plus the integer k --: the accumulator then holds the sum of
// the integers plus (number of additions) * (high word of C) << 32, and the number of additions into the moment slots is known (one
// per run of the slice): eval_full takes it off the slot totals.  One instruction less per word than fx_commit.
__device__ __forceinline__ void fx_add2(double *addr, double a, double b, int chi)
{
    const double c1 = fx_const(chi);
    const double s1 = fma(a, b, c1);
    const double res = fma(a, b, c1 - s1);                   // a b - (s1 - c1): exact to 2^-53 of itself, |res| <= unit / 2
    atomicAdd(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(s1));
    atomicAdd(reinterpret_cast<unsigned long long *>(addr + NSLOT), (unsigned long long)__double_as_longlong(res + fx_const(chi - (FX_LO_SHIFT << 20))));
}
__device__ __forceinline__ double fx_get2(double hi, double lo, double unit)
{
    return fma((double)__double_as_longlong(lo), unit * (1.0 / 281474976710656.0), (double)__double_as_longlong(hi) * unit);
}
__device__ __forceinline__ double fx_unit(int chi) { return __hiloint2double((chi & 0x7ff00000) - (52 << 20), 0); }
__device__ __forceinline__ double fx_get(double raw, double unit) { return (double)__double_as_longlong(raw) * unit; }

// ---------------------------------------------------------------------------------------------------------
// Full evaluation: psi, gradient and (approximate) Hessian at L::X.  M = 0: the elliptical model (the entries of the runs are
// not touched).
// ---------------------------------------------------------------------------------------------------------
template <class L>
__device__ __forceinline__ double eval_full(const Cand &c, int M, double reg_mu, double psi_bound PROF_PARAM)
{
    long long pt = PROF_NOW();
    const int tid = opaque_tid();
    const int n = 6 + M;
    int fxh_hi, fxg_hi;                                  // high words of the constants 1.5 * 2^(52 + e), Hessian / gradient scale of THIS evaluation
    fx_exponents(c.yexp, c.N, psi_bound, &fxh_hi, &fxg_hi);
    fxh_hi = uni(fxh_hi); fxg_hi = uni(fxg_hi);
    double *Hp = hess_ptr<L>(c), *g = SD + L::G;
    const double *xv = SD + L::X;
    const int *rbp = RBP;
    unsigned long long *msl = reinterpret_cast<unsigned long long *>(SD + L::MS);
    const int esz = M > 0 ? c.env_size : 21;
    for (int e = tid; e < esz; e += L::WGS) Hp[e] = 0;
    for (int i = tid; i < n; i += L::WGS) g[i] = 0;
    for (int e = tid; e < 42 * NSLOT; e += L::WGS) msl[e] = 0;
    __syncthreads();
    int rbt[6];                                          // the 6 dense theta rows (uniform)
#pragma unroll
    for (int b = 0; b < 6; b++) rbt[b] = M > 0 ? __builtin_amdgcn_readfirstlane(rbp[M + b]) : 0;
    double *ms = reinterpret_cast<double *>(msl + (tid & (NSLOT - 1)));     // this lane's slot: moment m at ms[2 (m - 1) NSLOT] (high word) and ms[(2 (m - 1) + 1) NSLOT]
    const bool hfast = c.hzmax <= HZREG;
    double tot[1];
    run_pass<L, 1>(c, tot, [&](int p, bool active, double (&pv)[1]) {
        RunPix rp;
        const int pl = load_pos(c, p, active);
        load_run(c, pl, active, rp);
        const int kh = M > 0 ? chunk_entries(c, __builtin_amdgcn_readfirstlane(p)) : 0;
        double S[SDSM_RUN];
        poly_surface(rp, xv, S);
        if (M > 0) smooth_add<1>(c, xv, pl, kh, S);
        double r[SDSM_RUN], d[SDSM_RUN], psum = 0;
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
#ifdef SDSM_LOSS_SERIAL
            __builtin_amdgcn_sched_barrier(0);
#else
            if ((k & 1) == 0) __builtin_amdgcn_sched_barrier(0);     // two independent loss evaluations in flight, not four (registers)
#endif
            r[k] = 0; d[k] = 0;
            if ((rp.pm >> k) & 1u) { double ph; loss_terms(rp.y[k], S[k], &ph, &r[k], &d[k]); psum += ph; }
        }
        pv[0] = psum;
        if (!active) return;
        // moments of the run: u^a * sum_k (r_k | d_k) v_k^b
        const double u = rp.u, uu = u * u;
        double R0 = 0, R1 = 0, R2 = 0, D0 = 0, D1 = 0, D2 = 0, D3 = 0, D4 = 0;
        bool anynz = false;                                  // (r = -y theta^ = 0 implies d = y^2 theta^ (1 - theta^) = 0)
#pragma unroll
        for (int k = 0; k < SDSM_RUN; k++) {
            const double vk = rp.v[k], vv = vk * vk, dvv = d[k] * vv;
            R0 += r[k]; R1 = fma(r[k], vk, R1); R2 = fma(r[k], vv, R2);
            D0 += d[k]; D1 = fma(d[k], vk, D1); D2 += dvv; D3 = fma(dvv, vk, D3); D4 = fma(dvv, vv, D4);
            anynz = anynz || r[k] != 0;
        }
        {
            const int cg = fxg_hi, ch = fxh_hi;
            const double u3 = uu * u, u4 = uu * uu;
            fx_add2(ms + 0 * NSLOT, R0, 1.0, cg); fx_add2(ms + 2 * NSLOT, R0, u, cg); fx_add2(ms + 4 * NSLOT, R1, 1.0, cg);
            fx_add2(ms + 6 * NSLOT, R0, uu, cg); fx_add2(ms + 8 * NSLOT, R1, u, cg); fx_add2(ms + 10 * NSLOT, R2, 1.0, cg);
            fx_add2(ms + 12 * NSLOT, D0, 1.0, ch); fx_add2(ms + 14 * NSLOT, D0, u, ch); fx_add2(ms + 16 * NSLOT, D1, 1.0, ch);
            fx_add2(ms + 18 * NSLOT, D0, uu, ch); fx_add2(ms + 20 * NSLOT, D1, u, ch); fx_add2(ms + 22 * NSLOT, D2, 1.0, ch);
            fx_add2(ms + 24 * NSLOT, D0, u3, ch); fx_add2(ms + 26 * NSLOT, D1, uu, ch); fx_add2(ms + 28 * NSLOT, D2, u, ch); fx_add2(ms + 30 * NSLOT, D3, 1.0, ch);
            fx_add2(ms + 32 * NSLOT, D0, u4, ch); fx_add2(ms + 34 * NSLOT, D1, u3, ch); fx_add2(ms + 36 * NSLOT, D2, uu, ch); fx_add2(ms + 38 * NSLOT, D3, u, ch);
            fx_add2(ms + 40 * NSLOT, D4, 1.0, ch);
        }
        if (M == 0 || !anynz) return;
        // xi part.  Gradient: every entry of the run; Hessian: its leading entries (kept in registers as they pass by)
        f32x4 lw[HZREG];
        uint32_t lim[HZREG];
#pragma unroll
        for (int a = 0; a < HZREG; a++) { lw[a].x = 0.f; lw[a].y = 0.f; lw[a].z = 0.f; lw[a].w = 0.f; lim[a] = 0; }
        int p2 = pl;
        asm volatile("" : "+v"(p2));                      // (a second load of the entries, not the first one kept alive across the loss evaluation)
        EntryCursor ec = entry_cursor(c, p2);
        const double cgd = fx_const(fxg_hi);
        auto grad_entry = [&](const f32x4 &w, uint32_t im) {
            double acc = fma(r[0], (double)w.x, cgd);
            acc = fma(r[1], (double)w.y, acc); acc = fma(r[2], (double)w.z, acc); acc = fma(r[3], (double)w.w, acc);
            fx_commit(&g[6 + (int)(im & 0xffffu)], acc, fxg_hi);
        };
        static_assert(HZREG % EBATCH == 0, "the leading entries are taken from whole batches");
#pragma unroll
        for (int jb = 0; jb < HZREG; jb += EBATCH) {          // the first batch(es) in straight-line code: they include the HZREG leading entries
            f32x4 w[EBATCH];
            uint32_t im[EBATCH];
            load_entries(c, ec, jb, kh, w, im);
#pragma unroll
            for (int t = 0; t < EBATCH; t++) {
                if (jb + t < kh) {
                    if (jb + t < rp.nnz) grad_entry(w[t], im[t]);
                    if (jb + t < HZREG) { lw[jb + t] = w[t]; lim[jb + t] = im[t]; }
                }
            }
        }
        for (int j0 = HZREG; j0 < kh; j0 += EBATCH) {
            f32x4 w[EBATCH];
            uint32_t im[EBATCH];
            load_entries(c, ec, j0, kh, w, im);
#pragma unroll
            for (int t = 0; t < EBATCH; t++) if (j0 + t < kh && j0 + t < rp.nnz) grad_entry(w[t], im[t]);
        }
        const double chd = fx_const(fxh_hi);
        const double v2[SDSM_RUN] = {rp.v[0] * rp.v[0], rp.v[1] * rp.v[1], rp.v[2] * rp.v[2], rp.v[3] * rp.v[3]};
        const double u2 = 2 * u;
        // one leading entry a (column ia, the four pixels' weights masked by "leading for that pixel"): the six theta rows through
        // three sums over the run (its pixels share u), then the leading entries b <= a
        auto theta_rows = [&](int ia, const double (&dw)[SDSM_RUN]) {
            const double m0 = (dw[0] + dw[1]) + (dw[2] + dw[3]);
            const double m1 = fma(dw[3], rp.v[3], fma(dw[2], rp.v[2], fma(dw[1], rp.v[1], dw[0] * rp.v[0])));
            const double m2 = fma(dw[3], v2[3], fma(dw[2], v2[2], fma(dw[1], v2[1], dw[0] * v2[0])));
            fx_add(&Hp[rbt[0] + ia], m0, uu, fxh_hi); fx_add(&Hp[rbt[1] + ia], m2, 1.0, fxh_hi); fx_add(&Hp[rbt[2] + ia], m1, u2, fxh_hi);
            fx_add(&Hp[rbt[3] + ia], m0, u2, fxh_hi); fx_add(&Hp[rbt[4] + ia], m1, 2.0, fxh_hi); fx_add(&Hp[rbt[5] + ia], m0, 1.0, fxh_hi);
        };
        if (hfast) {
            int rbs[HZREG];                                                          // row bases of the leading entries
#pragma unroll
            for (int a = 0; a < HZREG; a++) {
                rbs[a] = a < rp.hz ? rbp[lim[a] & 0xffffu] : 0;
                const unsigned la = lim[a] >> 16;                                    // weights of pixels for which the entry is not a leading one: 0
                lw[a].x = (la & 1u) ? lw[a].x : 0.f; lw[a].y = (la & 2u) ? lw[a].y : 0.f; lw[a].z = (la & 4u) ? lw[a].z : 0.f; lw[a].w = (la & 8u) ? lw[a].w : 0.f;
            }
#pragma unroll
            for (int a = 0; a < HZREG; a++) {
                if (a < rp.hz) {
                    const double dw[SDSM_RUN] = {d[0] * (double)lw[a].x, d[1] * (double)lw[a].y, d[2] * (double)lw[a].z, d[3] * (double)lw[a].w};
                    theta_rows((int)(lim[a] & 0xffffu), dw);
#pragma unroll
                    for (int b = 0; b <= a; b++) {                                   // leading entries are in ascending column order
                        if ((lim[a] >> 16) & (lim[b] >> 16)) {                       // some pixel couples the two
                            double acc = fma(dw[0], (double)lw[b].x, chd);
                            acc = fma(dw[1], (double)lw[b].y, acc); acc = fma(dw[2], (double)lw[b].z, acc); acc = fma(dw[3], (double)lw[b].w, acc);
                            fx_commit(&Hp[rbs[a] + (int)(lim[b] & 0xffffu)], acc, fxh_hi);
                        }
                    }
                }
            }
        } else {                                             // runs with more leading entries than the registers hold: from memory
            for (int a = 0; a < rp.hz; a++) {
                const f32x4 wa = gld(c.ell_w4, ((unsigned)a * (unsigned)c.NR + (unsigned)p2) * 16u);
                const uint32_t ima = gld(c.ell_im, ((unsigned)a * (unsigned)c.NR + (unsigned)p2) * 4u);
                const unsigned la = ima >> 16;
                const int ia = (int)(ima & 0xffffu), rba = rbp[ia];
                const double dw[SDSM_RUN] = {(la & 1u) ? d[0] * (double)wa.x : 0.0, (la & 2u) ? d[1] * (double)wa.y : 0.0,
                                             (la & 4u) ? d[2] * (double)wa.z : 0.0, (la & 8u) ? d[3] * (double)wa.w : 0.0};
                theta_rows(ia, dw);
                for (int b = 0; b <= a; b++) {
                    const f32x4 wb = gld(c.ell_w4, ((unsigned)b * (unsigned)c.NR + (unsigned)p2) * 16u);
                    const uint32_t imb = gld(c.ell_im, ((unsigned)b * (unsigned)c.NR + (unsigned)p2) * 4u);
                    const unsigned lb = imb >> 16;
                    if (!(la & lb)) continue;
                    double acc = fma(dw[0], (lb & 1u) ? (double)wb.x : 0.0, chd);
                    acc = fma(dw[1], (lb & 2u) ? (double)wb.y : 0.0, acc); acc = fma(dw[2], (lb & 4u) ? (double)wb.z : 0.0, acc); acc = fma(dw[3], (lb & 8u) ? (double)wb.w : 0.0, acc);
                    fx_commit(&Hp[rba + (int)(imb & 0xffffu)], acc, fxh_hi);
                }
            }
        }
    });
    PROF_ADD(0, pt);
    // totals of the moment slots (integers: any order), as raw integers for the exchange of a workgroup group
    unsigned long long *mraw = reinterpret_cast<unsigned long long *>(SD + L::RED);
    double *tot22 = SD + L::RED + 48;
    if (tid < 42) {
        unsigned long long sacc = 0;
        for (int i = 0; i < NSLOT; i++) sacc += msl[tid * NSLOT + i];
        // every run of the slice added the bits of its constant along with its integer (fx_add2): word tid belongs to moment tid / 2
        const unsigned chi_w = (unsigned)(tid < 12 ? fxg_hi : fxh_hi) - ((tid & 1) ? (unsigned)(FX_LO_SHIFT << 20) : 0u);
        sacc -= (unsigned long long)((unsigned)(c.p_hi - c.p_lo) * chi_w) << 32;
        mraw[tid] = sacc;
    }
    __syncthreads();
    if (c.wG > 1) wide_finish<L, 1>(c, tot, Hp, M > 0 ? c.env_size : 0, g + 6, M, reinterpret_cast<double *>(mraw), 42);
    {
        const double uh = fx_unit(fxh_hi), ug = fx_unit(fxg_hi);
        if (M > 0) {
            for (int e = tid; e < c.env_size; e += L::WGS) Hp[e] = fx_get(Hp[e], uh);
            for (int i = 6 + tid; i < n; i += L::WGS) g[i] = fx_get(g[i], ug);
        }
        if (tid < 21) tot22[1 + tid] = fx_get2(reinterpret_cast<double *>(mraw)[2 * tid], reinterpret_cast<double *>(mraw)[2 * tid + 1], tid < 6 ? ug : uh);
    }
    __syncthreads();
    if (tid < 6) g[tid] = moment_grad(tot22, tid);
    if (tid < 21) {
        if (M == 0) Hp[tid] = moment_hess(tot22, tid);           // packed lower triangle of a 6x6 matrix = the same enumeration order
        else {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= tid) a++;
            Hp[rbp[M + a] + M + (tid - a * (a + 1) / 2)] = moment_hess(tot22, tid);   // theta-theta block: columns M .. M + a of row M + a
        }
    }
    double psi = tot[0];
    __syncthreads();
    if (M > 0) psi += add_regulariser<L>(c, M, reg_mu);
    __syncthreads();
    PROF_ADD(2, pt);
    return uni(psi);
}

// Back substitution L^T z = yrow by ONE wavefront (n <= 64 KY).  The right-hand side lives in its registers (row i: lane i & 63,
// register i >> 6) together with the row bases, first columns and reciprocal pivots of those rows; column j of L^T is row j of
// the factor -- one contiguous read whose address does not depend on the solution, requested BS_DEPTH - 1 steps ahead.  A step
// is then: scale y_j, broadcast it (v_readlane), one multiply-add per register: no barrier and no memory round trip on the
// dependent chain (the blocked version with one workgroup barrier per panel took 18 us for n = 50, 44 % of factor_solve).
template <int KY>
__device__ __forceinline__ void back_substitute_wave(const double *Hp, const double *yrow, const double *dg, const int *rbp, const int *fstp, double *zl, int n, int tid)
{
    constexpr int BS_DEPTH = 4;
    if (tid < 64) {
        // scaled unknowns: with y~_i = y_i / l_ii the solution is z_j = y~_j once the columns j' > j are eliminated, and
        // eliminating column j is y~_i -= (l_ji / l_ii) z_j: the factor's row is scaled before z_j is known (off the dependent
        // chain), a step is two v_readlane and one multiply-add per register
        double yv[KY], dgv[KY];
        int rbv[KY], fsv[KY];
#pragma unroll
        for (int k = 0; k < KY; k++) {
            const int i = tid + 64 * k;
            const bool in = i < n;
            dgv[k] = in ? dg[i] : 0.0; yv[k] = in ? yrow[i] * dgv[k] : 0.0; rbv[k] = in ? rbp[i] : 0; fsv[k] = in ? fstp[i] : 0;
        }
        double buf[BS_DEPTH][KY], msk[BS_DEPTH][KY];          // raw row entries; the lane's reciprocal pivot where the row has an entry, else 0
        // The rows are walked in BLOCKS of 64 (block KB: rows 64 KB .. 64 KB + 63 live in register KB): inside a block the register of
        // row j and the registers its entries can touch (0 .. KB) are known at compile time -- no selects, and on average half of the
        // loads and multiply-adds of a version that treats every row as if it reached every register (n = 293: 58 -> 27 us per solve).
        auto fetch = [&](auto KBc, int j, double (&r)[KY], double (&m)[KY]) {     // row j of block KB: columns fst[j] .. j - 1 (0 elsewhere)
            constexpr int KB = decltype(KBc)::value;
            const int lj = j & 63;
            const int rb = __builtin_amdgcn_readlane(rbv[KB], lj), f = __builtin_amdgcn_readlane(fsv[KB], lj);
#pragma unroll
            for (int k = 0; k <= KB; k++) {
                const int i = tid + 64 * k;
                const bool in = i >= f && i < j;             // (no branch around the load: every lane reads, entry (j, j) if it has none)
                r[k] = Hp[rb + (in ? i : j)];                // used BS_DEPTH - 1 steps later: nothing here waits for it
                m[k] = in ? dgv[k] : 0.0;
            }
        };
        auto step = [&](auto KBc, int j, const double (&r)[KY], const double (&m)[KY]) {   // eliminate column j (a row of block KB)
            constexpr int KB = decltype(KBc)::value;
            double sc[KB + 1];                               // the row scaled by the lanes' reciprocal pivots: before the broadcast, off the chain
#pragma unroll
            for (int k = 0; k <= KB; k++) sc[k] = r[k] * m[k];
            const int lj = j & 63;
            const double zj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(yv[KB]), lj), __builtin_amdgcn_readlane(__double2loint(yv[KB]), lj));
#pragma unroll
            for (int k = 0; k <= KB; k++) yv[k] = fma(-sc[k], zj, yv[k]);        // rows >= j have a zero there: they keep their solution
        };
        // No branch inside a group of BS_DEPTH steps (the compiler's wait counts for the rows in flight are exact only in straight-line
        // code): the steps start at the next multiple of BS_DEPTH above n -- rows >= n have y~ = 0, so whatever is fetched for them
        // is multiplied by zero -- and the rows requested beyond row 0 at the end are row 0 again.
        static_assert(64 % BS_DEPTH == 0 && BS_DEPTH == 4, "groups of steps tile the blocks");
        const int jtop = ((n + BS_DEPTH - 1) / BS_DEPTH) * BS_DEPTH - 1;
        const int kt = jtop >> 6;
        auto block = [&](auto KBc) {
            constexpr int KB = decltype(KBc)::value;
            constexpr int KP = KB > 0 ? KB - 1 : 0;          // the block below (block 0: row 0 again)
            std::integral_constant<int, KP> KPc;
            if (KB > kt) return;
            const int jhi = KB == kt ? jtop : 64 * KB + 63;
            if (KB == kt) {
#pragma unroll
                for (int s2 = 0; s2 < BS_DEPTH - 1; s2++) fetch(KBc, jtop - s2, buf[s2], msk[s2]);
            }
            for (int jb = jhi; jb >= 64 * KB + 2 * BS_DEPTH - 1; jb -= BS_DEPTH) {
#pragma unroll
                for (int s2 = 0; s2 < BS_DEPTH; s2++) {
                    fetch(KBc, jb - s2 - (BS_DEPTH - 1), buf[(s2 + BS_DEPTH - 1) % BS_DEPTH], msk[(s2 + BS_DEPTH - 1) % BS_DEPTH]);
                    step(KBc, jb - s2, buf[s2], msk[s2]);
                }
            }
            {                                                // the last group of the block: the rows requested from its second step on are in the block below
                const int jb = 64 * KB + BS_DEPTH - 1;
                fetch(KBc, jb - (BS_DEPTH - 1), buf[BS_DEPTH - 1], msk[BS_DEPTH - 1]);
                step(KBc, jb, buf[0], msk[0]);
#pragma unroll
                for (int s2 = 1; s2 < BS_DEPTH; s2++) {
                    const int jf = jb - s2 - (BS_DEPTH - 1);
                    fetch(KPc, KB > 0 ? jf : 0, buf[(s2 + BS_DEPTH - 1) % BS_DEPTH], msk[(s2 + BS_DEPTH - 1) % BS_DEPTH]);
                    step(KBc, jb - s2, buf[s2], msk[s2]);
                }
            }
        };
        if constexpr (KY > 7) block(std::integral_constant<int, 7>{});
        if constexpr (KY > 6) block(std::integral_constant<int, 6>{});
        if constexpr (KY > 5) block(std::integral_constant<int, 5>{});
        if constexpr (KY > 4) block(std::integral_constant<int, 4>{});
        if constexpr (KY > 3) block(std::integral_constant<int, 3>{});
        if constexpr (KY > 2) block(std::integral_constant<int, 2>{});
        if constexpr (KY > 1) block(std::integral_constant<int, 1>{});
        block(std::integral_constant<int, 0>{});
#pragma unroll
        for (int k = 0; k < KY; k++) { const int i = tid + 64 * k; if (i < n) zl[i] = yv[k]; }
    }
}

// 1 / sqrt(x), x in the normal range: hardware estimate (5.2e-8 relative) + ONE Newton step = 4.2e-15 relative (37 ulp,
// tools/microbench/rsq_accuracy.hip).  The pivots of the Cholesky factorisation sit on its dependent chain, and a factor of
// an approximate Hessian does not need the last bits.
__device__ __forceinline__ double rsqrt_f64(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-0.5 * x * y, y, 0.5);
    return fma(y, e, y);
}

// ---------------------------------------------------------------------------------------------------------
// Solve (H + tau D) d = -g, D = diag(H): blocked right-looking Cholesky IN PLACE on the envelope
// (panels of NB = 4 columns, the right-hand side rides along as row n), then blocked back substitution.
// A panel only touches the rows whose envelope reaches its columns (xi rows up to rend[panel], the 6 theta rows and
// the right-hand side): about (bandwidth + 7)^2 / 2 entries per panel instead of the whole trailing triangle.
// Returns 0 ok, 1 not positive definite (the Hessian is destroyed: re-evaluate and retry with a larger shift),
// 2 non-finite input.  *lam2 = -g.d.
// ---------------------------------------------------------------------------------------------------------
template <class L>
__device__ __forceinline__ int factor_solve(const Cand &c, int M, double tau_in, double *lam2 PROF_PARAM)
{
    long long pf = PROF_NOW();
    const int tid = opaque_tid();
    const int n = 6 + M;
    double *Hp = hess_ptr<L>(c), *g = SD + L::G, *yrow = SD + L::YROW, *d = SD + L::D, *dg = SD + L::TMP, *zl = SD + L::XT;   // XT is free between line searches
    const int *rbp = RBP, *fstp = FSTP, *rendp = RENDP;
    int *flag = (int *)(SD + L::FLAG);
    if (tid == 0) *flag = 0;
    // logical order i: xi_0 .. xi_{M-1}, theta_0 .. theta_5 (yrow, dg, zl); variable order (x, g, d): theta first
    if (M == 0) {
        // ---- elliptical model: 6 x 6 system solved redundantly by every thread in registers (no barriers; every thread
        //      reads the same 21 + 6 values, so non-finite input is seen by all of them alike) ----
        bool finite = true;
#pragma unroll
        for (int i = 0; i < 6; i++) if (!isfinite(g[i])) finite = false;
#pragma unroll
        for (int e = 0; e < 21; e++) if (!isfinite(Hp[e])) finite = false;
        if (uni((int)!finite)) return 2;                        // (every thread read the same values)
        double A[6][6], bb[6], s6[6], z[6], ri6[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { double hii = Hp[tri(i, i)]; if (!(hii > 0)) hii = 1; s6[i] = rsqrt_f64(hii); }   // Jacobi scaling
        double tau = tau_in;
        bool ok = false;
        for (int attempt = 0; attempt < 12 && !ok; attempt++) {
            ok = true;
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int k = 0; k <= i; k++) A[i][k] = Hp[tri(i, k)] * s6[i] * s6[k] + (i == k ? tau : 0.0);
                bb[i] = -g[i] * s6[i];
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                double piv = A[j][j];
#pragma unroll
                for (int k = 0; k < j; k++) piv -= A[j][k] * A[j][k];
                if (!(piv > 1e-300) || !isfinite(piv)) { ok = false; piv = 1; }
                const double rj = rsqrt_f64(piv);                 // reciprocal pivot: multiplications instead of divisions
                ri6[j] = rj;
                A[j][j] = piv * rj;
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
                    double v = A[i][j];
#pragma unroll
                    for (int k = 0; k < j; k++) v -= A[i][k] * A[j][k];
                    A[i][j] = v * rj;
                }
                double v = bb[j];
#pragma unroll
                for (int k = 0; k < j; k++) v -= z[k] * A[j][k];
                z[j] = v * rj;                                    // forward substitution rides along
            }
            if (!ok) tau = tau == 0 ? 1e-12 : tau * 100;
        }
        if (uni((int)!ok)) return 2;
        double l2 = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) l2 += z[i] * z[i];
#pragma unroll
        for (int k = 5; k >= 0; k--) {                            // back substitution
            double v = z[k];
#pragma unroll
            for (int m = k + 1; m < 6; m++) v -= A[m][k] * z[m];
            z[k] = v * ri6[k];
        }
        bool fin = isfinite(l2);
#pragma unroll
        for (int i = 0; i < 6; i++) { z[i] *= s6[i]; if (!isfinite(z[i])) fin = false; }
        if (uni((int)!fin)) return 2;                             // uniform: every thread computed the same values
        if (tid < 6) d[tid] = z[tid];
        __syncthreads();
        *lam2 = uni(l2);
        return 0;
    }

    // ---- general case.  No explicit Jacobi scaling: the Cholesky factor of D^-1/2 H D^-1/2 is D^-1/2 times the factor
    //      of H, so pivots fail together and lam2 is the same; the shift tau * diag(H) is applied to the diagonal directly.
    //      ONE workgroup barrier per panel: every thread factors the panel's 4x4 diagonal block redundantly in registers
    //      (reciprocal square roots, no divisions), reads the RAW panel entries of the rows / columns it updates, solves
    //      them against the block in registers and applies the rank-4 update; the factored block and the solved panel
    //      entries are written after the barrier (nobody reads those columns again before the back substitution).
    //      WGS / SDSM_FACTOR_DIV threads take part (measured: 1 = all threads is fastest, 172 k vs 168 k (2) vs 156 k (4)
    //      candidate solves/s: the panel chain is latency bound, more threads = fewer serial entries per thread).
    constexpr int NB = SDSM_PANEL;
    constexpr int NBSH = NB == 8 ? 3 : 2;
    static_assert(NB == 4 || NB == 8, "panel width");
    constexpr int FT = L::WGS / SDSM_FACTOR_DIV;                // threads that factor
    const bool fa = tid < FT;
    int *pflag = (int *)(SD + L::FLAG);                         // [2]: panel p reports a failed pivot in slot p & 1
    for (int i = tid; i < n; i += L::WGS) {
        const int vi = i < M ? 6 + i : i - M;
        yrow[i] = -g[vi];
        if (tau_in > 0) Hp[rbp[i] + i] *= 1 + tau_in;
    }
    if (tid < 2) pflag[tid] = 0;
    __syncthreads();
    bool ok = true, nonfinite = false;
    int failed = 0;
    PROF_ADD(13, pf);
    // (the panel's row bases and last coupled row are requested one panel ahead and made uniform when the panel starts: two dependent
    // LDS round trips less at the top of the chain)
    int re_nx = rendp[0], rba_nx[NB];
#pragma unroll
    for (int a2 = 0; a2 < NB; a2++) rba_nx[a2] = rbp[a2 < n ? a2 : 0];
    for (int j0 = 0; j0 < n; j0 += NB) {
        const int nb = n - j0 < NB ? n - j0 : NB;
        const int jn = j0 + nb;
        const int re = j0 < M ? uni(re_nx) : M - 1;                  // uniform: keep the panel's index arithmetic scalar
        int rba_cur[NB];
#pragma unroll
        for (int a2 = 0; a2 < NB; a2++) rba_cur[a2] = uni(rba_nx[a2]);
        if (jn < n) {
            re_nx = rendp[jn < M ? jn >> NBSH : 0];
#pragma unroll
            for (int a2 = 0; a2 < NB; a2++) rba_nx[a2] = rbp[jn + a2 < n ? jn + a2 : jn];
        }
        const int nxi = jn < M && re >= jn ? re - jn + 1 : 0;
        const int th0 = jn > M ? jn : M;
        const int na = nxi + (n - th0) + 1;
        double t[NB][NB], rinv[NB];
        if (fa) {
        // A. diagonal block (all rows of the panel store column j0: fst is a multiple of the panel width)
#pragma unroll
        for (int a2 = 0; a2 < NB; a2++) {
            const int rba = a2 < nb ? rba_cur[a2] : 0;
#pragma unroll
            for (int b2 = 0; b2 <= a2; b2++) t[a2][b2] = a2 < nb ? Hp[rba + j0 + b2] : (a2 == b2 ? 1.0 : 0.0);
        }
#pragma unroll
        for (int cc = 0; cc < NB; cc++) {
            double piv = t[cc][cc];
#pragma unroll
            for (int m = 0; m < cc; m++) piv -= t[cc][m] * t[cc][m];
            if (cc < nb && !(piv > 0 && piv < 1.7e308)) { ok = false; if (!(piv <= 0)) nonfinite = true; piv = 1; }
            const double r = rsqrt_f64(piv);
            t[cc][cc] = piv * r;
            rinv[cc] = r;
#pragma unroll
            for (int a2 = cc + 1; a2 < NB; a2++) {
                double v = t[a2][cc];
#pragma unroll
                for (int m = 0; m < cc; m++) v -= t[a2][m] * t[cc][m];
                t[a2][cc] = v * r;
            }
        }
        PROF_ADD(8, pf);
        if (!ok && tid == 0) pflag[(j0 >> NBSH) & 1] = nonfinite ? 2 : 1;
        // active rows below the panel: xi rows jn .. re, theta rows, right-hand side (compact index -> logical row)
        // B. rank-nb update of the active rows x active columns from the raw panel entries.  The (row, column) pairs --
        //    a triangle over the active rows plus the right-hand-side row -- are dealt to the threads by a flat index, so
        //    that every wavefront gets the same share (a panel has ~400 pairs: 1-2 per thread).
        if constexpr (L::WGS >= 256) {
            //    The classes of 256 and 512 threads (long panels, one or two workgroups per compute unit): dealt in 2 x 2 TILES of (row, column) pairs: the four panel rows a tile needs are read and solved against the block once
            //    and serve four entries (14 instead of 24 multiply-adds per entry, 20 instead of 36 LDS reads per four, one index
            //    computation); per entry the operations and their order are those of the pair-by-pair loop.
            const int nr = na - 1;                                          // active rows without the right-hand side (compact 0 .. nr - 1; nr: right-hand side)
            const int NT = (na + 1) >> 1, ntile = NT * (NT + 1) / 2;
            for (int e = tid; e < ntile; e += FT) {
                int TI = (int)((__fsqrt_rn(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                TI += (TI + 1) * (TI + 2) / 2 <= e ? 1 : 0;                  // float rounding: at most one off, either way
                TI -= TI * (TI + 1) / 2 > e ? 1 : 0;
                const int TK = e - TI * (TI + 1) / 2;
                const double *pr[2], *pc[2];
                double *br[2];
                int kc[2];
                bool vr[2], vc[2];
    #pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int ti = 2 * TI + q, tk = 2 * TK + q;
                    vr[q] = ti <= nr; vc[q] = tk < nr;
                    const int i = ti < nxi ? jn + ti : th0 + (ti - nxi);      // == n for the right-hand side
                    const int k = tk < nxi ? jn + tk : th0 + (tk - nxi);
                    br[q] = vr[q] && i < n ? Hp + rbp[i] : yrow;
                    pr[q] = br[q] + j0;
                    pc[q] = vc[q] ? Hp + rbp[k] + j0 : yrow + j0;
                    kc[q] = vc[q] ? k : 0;
                }
                double lr[2][NB], lc[2][NB], acc[2][2];
    #pragma unroll
                for (int q = 0; q < 2; q++) {
    #pragma unroll
                    for (int cc = 0; cc < NB; cc++) { lr[q][cc] = cc < nb ? pr[q][cc] : 0.0; lc[q][cc] = cc < nb ? pc[q][cc] : 0.0; }
                }
                bool ve[2][2];
    #pragma unroll
                for (int q = 0; q < 2; q++)
    #pragma unroll
                    for (int c2 = 0; c2 < 2; c2++) {
                        ve[q][c2] = vr[q] && vc[c2] && 2 * TK + c2 <= 2 * TI + q;
                        acc[q][c2] = ve[q][c2] ? br[q][kc[c2]] : 0.0;
                    }
    #pragma unroll
                for (int cc = 0; cc < NB; cc++) {
    #pragma unroll
                    for (int q = 0; q < 2; q++) {
                        double a3 = lr[q][cc], b3 = lc[q][cc];
    #pragma unroll
                        for (int m = 0; m < cc; m++) { a3 -= lr[q][m] * t[cc][m]; b3 -= lc[q][m] * t[cc][m]; }
                        lr[q][cc] = a3 * rinv[cc]; lc[q][cc] = b3 * rinv[cc];
                    }
    #pragma unroll
                    for (int q = 0; q < 2; q++)
    #pragma unroll
                        for (int c2 = 0; c2 < 2; c2++) acc[q][c2] -= lr[q][cc] * lc[c2][cc];
                }
    #pragma unroll
                for (int q = 0; q < 2; q++)
    #pragma unroll
                    for (int c2 = 0; c2 < 2; c2++) if (ve[q][c2]) br[q][kc[c2]] = acc[q][c2];
            }
        } else {                                                        // class 1 (short panels, four workgroups per compute unit: pair by pair is faster there, 5.56 vs 5.72 ms)
            const int ntri = (na - 1) * na / 2, npair = ntri + (na - 1);
            for (int e = tid; e < npair; e += FT) {
                int ti, tk;
                if (e < ntri) {
                    ti = (int)((__fsqrt_rn(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                    ti += (ti + 1) * (ti + 2) / 2 <= e ? 1 : 0;                // float rounding: at most one off, either way
                    ti -= ti * (ti + 1) / 2 > e ? 1 : 0;
                    tk = e - ti * (ti + 1) / 2;
                } else { ti = na - 1; tk = e - ntri; }
                const int i = ti < nxi ? jn + ti : th0 + (ti - nxi);          // == n for the right-hand side
                const int k = tk < nxi ? jn + tk : th0 + (tk - nxi);
                const double *pi = i < n ? Hp + rbp[i] + j0 : yrow + j0;
                const double *pk = Hp + rbp[k] + j0;
                double li[NB], lk[NB];
    #pragma unroll
                for (int cc = 0; cc < NB; cc++) { li[cc] = cc < nb ? pi[cc] : 0.0; lk[cc] = cc < nb ? pk[cc] : 0.0; }
                double *Lik = (i < n ? Hp + rbp[i] : yrow) + k;
                double acc = *Lik;
    #pragma unroll
                for (int cc = 0; cc < NB; cc++) {
                    double a3 = li[cc], b3 = lk[cc];
    #pragma unroll
                    for (int m = 0; m < cc; m++) { a3 -= li[m] * t[cc][m]; b3 -= lk[m] * t[cc][m]; }
                    li[cc] = a3 * rinv[cc]; lk[cc] = b3 * rinv[cc];
                    acc -= li[cc] * lk[cc];
                }
                *Lik = acc;
            }
        }
        }
        PROF_ADD(9, pf);
        __syncthreads();
        PROF_ADD(10, pf);
        failed = uni(pflag[(j0 >> NBSH) & 1]);                    // uniform (two slots: a thread is at most one panel ahead)
        if (failed) break;
        // C. factored block, reciprocal diagonal and solved panel entries (columns j0 .. j0+3: not read again before
        //    the back substitution, so the next panel starts without another barrier)
        // (done by the LAST wavefronts: the flat pair index gives the low threads the second round of step B)
        if (tid >= FT - 64 - NB * NB && tid < FT - 64) {
            const int a2 = (tid - (FT - 64 - NB * NB)) / NB, b2 = (tid - (FT - 64 - NB * NB)) % NB;
            if (b2 <= a2 && a2 < nb) {
                double v = 0, rv = 0;
#pragma unroll
                for (int x2 = 0; x2 < NB; x2++) {
                    rv = x2 == a2 ? rinv[x2] : rv;
#pragma unroll
                    for (int y2 = 0; y2 <= x2; y2++) v = (x2 == a2 && y2 == b2) ? t[x2][y2] : v;
                }
                Hp[rbp[j0 + a2] + j0 + b2] = v;
                if (a2 == b2) dg[j0 + a2] = rv;
            }
        }
        if (fa) for (int tt = tid - (FT - 64); tt >= 0 && tt < na; tt += 64) {
            const int i = tt < nxi ? jn + tt : th0 + (tt - nxi);
            double *row = i < n ? Hp + rbp[i] + j0 : yrow + j0;
            double v[NB];
#pragma unroll
            for (int cc = 0; cc < NB; cc++) v[cc] = cc < nb ? row[cc] : 0.0;
#pragma unroll
            for (int cc = 0; cc < NB; cc++) {
                double acc = v[cc];
#pragma unroll
                for (int m = 0; m < cc; m++) acc -= v[m] * t[cc][m];
                v[cc] = acc * rinv[cc];
            }
#pragma unroll
            for (int cc = 0; cc < NB; cc++) if (cc < nb) row[cc] = v[cc];
        }
        PROF_ADD(11, pf);
    }
    __syncthreads();
    PROF_ADD(14, pf);
    if (failed) return failed;
    const double l2 = wave_canon_sum(n, [&](int i) { return yrow[i] * yrow[i]; });     // (the same bits for every workgroup size)
    // back substitution L^T z = yrow: by one wavefront in registers when n <= 512 (back_substitute_wave; which variant depends on n
    // alone, and all of them do the same arithmetic per row), else blocked like the factorisation (one barrier per panel: 64 us per solve
    // of 258 unknowns against 37 us of the wavefront version at 249)
    if (L::NMAX <= 128 || n <= 256 || (L::NMAX >= 512 && n <= 512)) {
        if (n <= 64) back_substitute_wave<1>(Hp, yrow, dg, rbp, fstp, zl, n, tid);
        else if constexpr (L::NMAX <= 128) back_substitute_wave<2>(Hp, yrow, dg, rbp, fstp, zl, n, tid);
        else if (n <= 256) back_substitute_wave<4>(Hp, yrow, dg, rbp, fstp, zl, n, tid);
        else if constexpr (L::NMAX >= 512) back_substitute_wave<8>(Hp, yrow, dg, rbp, fstp, zl, n, tid);
        __syncthreads();
    } else {
        // back substitution L^T z = yrow, blocked the same way: the threads that have a row to update (or write the block's
        // solution) solve the NB x NB block redundantly
        for (int j0 = ((n - 1) / NB) * NB; j0 >= 0; j0 -= NB) {
            const int nb = n - j0 < NB ? n - j0 : NB;
            int f4[NB];
    #pragma unroll
            for (int cc = 0; cc < NB; cc++) f4[cc] = cc < nb ? uni(fstp[j0 + cc]) : j0;
            const int start = j0 + nb - 1 >= M ? 0 : f4[0];          // fst is non-decreasing over the xi rows, 0 for theta rows
            if (tid < NB || start + tid < j0) {
                double z[NB];
    #pragma unroll
                for (int cc = NB - 1; cc >= 0; cc--) {
                    double acc = cc < nb ? yrow[j0 + cc] : 0.0;
    #pragma unroll
                    for (int m = cc + 1; m < NB; m++) if (m < nb) acc -= Hp[rbp[j0 + m] + j0 + cc] * z[m];
                    z[cc] = cc < nb ? acc * dg[j0 + cc] : 0.0;
                }
                if (tid < NB && tid < nb) {
                    double v = 0;
    #pragma unroll
                    for (int x2 = 0; x2 < NB; x2++) v = x2 == tid ? z[x2] : v;
                    zl[j0 + tid] = v;
                }
                for (int i = start + tid; i < j0; i += L::WGS) {
                    double acc = yrow[i];
    #pragma unroll
                    for (int cc = 0; cc < NB; cc++) if (cc < nb && i >= f4[cc]) acc -= Hp[rbp[j0 + cc] + i] * z[cc];
                    yrow[i] = acc;
                }
            }
            __syncthreads();
        }
    }
    bool fin = isfinite(l2);
    for (int i = tid; i < n; i += L::WGS) {
        const int vi = i < M ? 6 + i : i - M;
        const double dv = zl[i];
        d[vi] = dv;
        if (!isfinite(dv)) fin = false;
    }
    if (!fin) *flag = 1;
    __syncthreads();
    PROF_ADD(15, pf);
    if (uni(*flag)) return 2;
    *lam2 = uni(l2);
    return 0;
}

// Damped Newton on f = scale * psi from the parameters at L::X (in/out); the same algorithm and constants as the oracle's
// orc_newton (DESIGN.md "Solver").  Returns 0 optimal, 1 unknown (iteration cap / stalled line search), 2 numerical failure.
// Written as a state machine around ONE call site of every pass (full evaluation, factorisation, line-search sweep, value):
// everything is inlined into the kernel once, nothing lives on the stack.  `why` says what the next full evaluation is for:
//   EV_STEP   a new iterate (start, accepted line-search step)
//   EV_SPEC   speculative full step x + d (previous accepted step length was 1, or first iteration of a DSM solve): if it
//             meets the Armijo condition the iteration is done without a line-search pass, otherwise x is restored and the
//             sweeps start at t = 1/2
//   EV_RETRY  the factorisation hit a non-positive pivot and destroyed the Hessian: evaluate again, larger diagonal shift
//   EV_CHECK  the stop test passed with mu > 0 (inflated regulariser curvature underestimates the Newton decrement):
//             evaluate again with the true curvature
#define MU_DECAY 0.5
#define MU_MIN 1e-3
template <class L>
__device__ __forceinline__ int newton(const Cand &c, int M, int max_iters, double psi_start_bound, double *psi_out, int *iters_out, int *ev_value, int *ev_full PROF_PARAM)
{
    enum { EV_STEP = 0, EV_SPEC = 1, EV_RETRY = 2, EV_CHECK = 3 };
    const int tid = opaque_tid(), n = 6 + M;
    double *x = SD + L::X, *xt = SD + L::XT, *d = SD + L::D, *xb = SD + L::YROW;   // YROW is only live inside factor_solve
    int status = 1, iters = 0, why = EV_STEP, retries = 0;
    double mu = M > 0 ? 1.0 : 0.0, mu_spec = 0, tprev = 0, tau = 0;
    double psi = NAN, f = NAN, lam2 = 0;                     // f, lam2 describe x (kept across a speculative evaluation)
    double bound = uni(psi_start_bound);                     // an upper bound of psi at the point of the next evaluation (fx_exponents); infinity: none
    for (;;) {
        const double pe = eval_full<L>(c, M, why == EV_SPEC ? mu_spec : mu, bound PROF_ARG);
        (*ev_full)++;
        long long pt = PROF_NOW();
        bool line = false;
        double t0 = M == 0 ? LS_T0_ELL : 1.0;
        if (why == EV_SPEC) {
            if (isfinite(pe) && c.scale * pe <= f - fresh(LS_ALPHA) * lam2) { psi = pe; tprev = 1.0; mu = mu_spec; why = EV_STEP; }
            else {                                           // back to x; t = 1 is known to fail
                __syncthreads();
                for (int i = tid; i < n; i += L::WGS) x[i] = xb[i];
                __syncthreads();
                line = true; t0 = 0.5;
            }
        } else if (why != EV_RETRY) psi = pe;
        bound = uni(psi);                                    // evaluations at x again (larger shift, true curvature) and the speculative x + d, which only counts if psi does not grow
        if (!line) {
            if (why != EV_RETRY) {
                if (iters >= max_iters) { status = 1; break; }
                f = uni(c.scale * psi);
                if (!isfinite(f)) { status = 2; break; }
                tau = 0; retries = 0;
            }
            double lam2u;
            __builtin_amdgcn_s_setprio(3);
            const int fs = factor_solve<L>(c, M, tau, &lam2u PROF_ARG);
            __builtin_amdgcn_s_setprio(2);
            if (fs == 1 && retries < 11) {                   // escalating diagonal shift (same schedule as the oracle)
                retries++;
                tau = uni(tau == 0 ? fresh(1e-12) : tau * fresh(100.0));
                why = EV_RETRY;
                continue;
            }
            if (fs != 0) { status = 2; break; }
            lam2 = uni(c.scale * lam2u);
            PROF_ADD(3, pt);
            const bool conv = lam2 * 0.5 <= fresh(NEWTON_ABSTOL) + fresh(NEWTON_RELTOL) * fabs(f);
            if (conv && mu > 0) { mu = 0; why = EV_CHECK; continue; }
            iters++;
            if (conv) {
                // converged: final full step, kept only if it does not increase f (near-separable regions have a
                // vanishing Hessian and the unguarded step can be arbitrarily bad)
                for (int i = tid; i < n; i += L::WGS) xt[i] = x[i] + d[i];
                __syncthreads();
                const double psit = eval_value<L>(c, L::XT, M);
                (*ev_value)++;
                if (isfinite(psit) && c.scale * psit <= f) {
                    for (int i = tid; i < n; i += L::WGS) x[i] = xt[i];
                    __syncthreads();
                    psi = psit;
                }
                status = 0;
                break;
            }
            if (tprev == 1.0 || (iters == 1 && M > 0)) {
                mu_spec = mu * MU_DECAY;
                mu_spec = uni(mu_spec < fresh(MU_MIN) ? 0 : mu_spec);
                __syncthreads();
                for (int i = tid; i < n; i += L::WGS) { const double xi = x[i]; xb[i] = xi; x[i] = xi + d[i]; }
                __syncthreads();
                why = EV_SPEC;
                continue;
            }
        }
        // line search: sweeps of LS_K step lengths evaluated together; the accepted step is the one with the smallest f
        // among those that satisfy the Armijo condition (same rule as the oracle)
        bool accepted = false;
        double tbest = 0, fbest = INFINITY;
        for (int sweep = 0; sweep < LS_SWEEPS && !accepted; sweep++) {
            double fs[LS_K];
            eval_line<L>(c, M, t0, fs);
            (*ev_value)++;
            double tk = t0;
#pragma unroll
            for (int k = 0; k < LS_K; k++) {
                const double ft = uni(c.scale * fs[k]);
                if (isfinite(ft) && ft <= f - fresh(LS_ALPHA) * tk * lam2 && ft < fbest) { fbest = ft; tbest = tk; accepted = true; }
                tk *= LS_BETA;
            }
            t0 = tk;
        }
        PROF_ADD(4, pt);
        if (!accepted) { status = 1; break; }               // stalled; psi still is the value at x
        bound = uni(fbest / c.scale);                        // psi(x + tbest d) as the sweep computed it
        for (int i = tid; i < n; i += L::WGS) x[i] += tbest * d[i];
        __syncthreads();
        tprev = uni(tbest);
        if (M > 0) {
            if (tbest >= 1) { mu *= MU_DECAY; mu = uni(mu < fresh(MU_MIN) ? 0 : mu); }
            else mu = 1.0;
        }
        why = EV_STEP;
    }
    *psi_out = psi;
    *iters_out = iters;
    return status;
}

// theta (a1,a2,a3,b1,b2,c) expressed for coordinates z  ->  coefficients for coordinates w, where z = P w + o
__device__ __forceinline__ void reparam(const double *th, double p0, double p1, double o0, double o1, double *out)
{
    double a1 = th[0], a2 = th[1], a3 = th[2], b1 = th[3], b2 = th[4], cc = th[5];
    out[0] = a1 * p0 * p0;
    out[1] = a2 * p1 * p1;
    out[2] = a3 * p0 * p1;
    out[3] = p0 * (a1 * o0 + a3 * o1 + b1);
    out[4] = p1 * (a2 * o1 + a3 * o0 + b2);
    out[5] = a1 * o0 * o0 + a2 * o1 * o1 + 2 * a3 * o0 * o1 + 2 * b1 * o0 + 2 * b2 * o1 + cc;
}

// Affine map between the candidate's local coordinates and the full-image-normalised ones: x0 = r / (H-1) = P0 u + O0.
// Computed where it is used (by one lane, behind an opaque value) instead of being kept in registers across the solver.
struct Frame { double z0, z1, P0, P1, O0, O1; };
__device__ __forceinline__ Frame make_frame(const Cand &c, int H, int W)
{
    Frame f;
    double one = 1.0;
    asm volatile("" : "+v"(one), "+s"(H), "+s"(W));
    f.z0 = H > 1 ? H - one : one; f.z1 = W > 1 ? W - one : one;
    f.P0 = one / (c.inv_hr * f.z0); f.P1 = one / (c.inv_hc * f.z1); f.O0 = c.rmid / f.z0; f.O1 = c.cmid / f.z1;
    return f;
}

}  // namespace

// One candidate, start to finish, by one workgroup.  A candidate belongs to the FIRST class (1, 1b, 2, 2b, global memory) whose limits
// (6 + M <= NMAX and Hessian envelope <= EMAX doubles) it meets (sdsm_solve_class); a workgroup of class CLS leaves the others alone.
// Class 1 (host-built launch list: all candidates, largest first) also writes the records of trivial / failed-setup candidates.
// WIDE: the launch list holds (candidate | member << 24) for every member of every workgroup group.
// accept_any: the sweep of the "late" list by the global-memory class (any class beyond 1).
template <int NMAX, int EMAX, bool GLOBALH, int WGSIZE, bool WIDE, int CLS>
__device__ __forceinline__ void solve_candidate(const BatchParams &P, int ci, int wg, int handles_rest, bool accept_any, sdsm_record *records, uint32_t *masks, double *xi_out)
{
    using L = Lay<NMAX, EMAX, GLOBALH, WGSIZE>;
    int tid = threadIdx.x;                                   // re-derived (opaque_tid) at the start of every section: nothing per-thread is kept across the solver
    const CandDesc cd = uniform_desc(P.cand[ci]);
    const CandState st = uniform_state(P.state[ci]);
    sdsm_record *rec = &records[ci];

    if (st.status != ST_OK) {
        if (handles_rest && tid == 0) {
            sdsm_record r = {};
            r.status = st.status == ST_TRIVIAL ? SDSM_CAND_TRIVIAL : (st.status == ST_UNSUPPORTED ? SDSM_CAND_UNSUPPORTED : SDSM_CAND_ERROR);
            r.n_pixels = cd.N;
            *rec = r;
        }
        return;
    }
    int Mfull = st.M;
    bool unsupported = false;
    if (6 + Mfull > SDSM_MAX_N_GLOBAL) { unsupported = true; Mfull = 0; }    // elliptical result only (flagged)
    const int efull = Mfull > 0 ? st.env_size : 21;
    {
        const int cls = sdsm_solve_class(st.status, st.M, st.env_size, cd.N, cd.wide_g, P.k1_pixmax);
        if (accept_any ? !(cls >= SDSM_CLS_1B && cls <= SDSM_CLS_3) : cls != CLS) return;
    }
    if (GLOBALH && cd.hglob_off < 0) {                          // cannot happen (the host reserves a slot whenever Mcap admits it)
        if (tid == 0) { sdsm_record r0 = {}; r0.status = SDSM_CAND_UNSUPPORTED; r0.n_pixels = cd.N; r0.n_deform = st.M; *rec = r0; }
        return;
    }

    Cand c;
    c.N = cd.N; c.NR = st.NR; c.hzmax = Mfull > 0 ? st.hzmax : 0; c.env_size = 21;
    c.p_lo = 0; c.p_hi = st.NR; c.wg = 0; c.wG = 1; c.wpool = nullptr; c.wtimeout = P.wide_timeout;
    if (WIDE) {
        c.wG = cd.wide_g; c.wg = wg; c.wpool = P.wide_pool + cd.wide_off;
        const int sup = 64 * SDSM_SUPER;                                     // slices are whole super-chunks (run_pass: the order of the psi sums)
        const int chunk = (((st.NR + c.wG - 1) / c.wG) + sup - 1) / sup * sup;
        c.p_lo = wg * chunk < st.NR ? wg * chunk : st.NR;
        c.p_hi = c.p_lo + chunk < st.NR ? c.p_lo + chunk : st.NR;
    }
    if (tid == 0) { *WIDE_PHASE = 0; *WIDE_OPS = 0; }
    if (WIDE && wg == 0) {                                               // cleared here: the members set bits at the very end, after many group barriers
        uint32_t *mk0 = masks + cd.mask_off;
        for (int i = tid; i < (cd.h * cd.w + 31) / 32; i += L::WGS) mk0[i] = 0;
    }
    c.crop_y2 = (g_cf64x2_p)(P.crop_y + cd.run_off * SDSM_RUN); c.crop_rc = (g_cu32_p)(P.crop_rc + cd.run_off); c.meta = (g_cu32_p)(P.run_meta + cd.run_off);
    c.aux = (g_cu32_p)(P.run_aux + cd.run_off);
    c.ell_im = (g_cu32_p)(P.ell_im + cd.ell_off); c.ell_w4 = (g_cf32x4_p)(reinterpret_cast<const f32x4 *>(P.ell_w) + cd.ell_off);
    c.hglob = (GLOBALH && cd.hglob_off >= 0) ? P.hglob + cd.hglob_off : nullptr;
    c.scale = P.scale / cd.N;                                   // objects.py:380
    c.epsilon = P.epsilon; c.alpha = P.alpha; c.reg0 = P.reg_unit * Mfull;   // (only read by passes with M = Mfull > 0)
    c.yexp = st.yexp; c.boost = cd.N > P.boost_pixels;
    // local frame: centre of the bounding box, half extents
    const double half_r = 0.5 * (cd.h - 1), half_c = 0.5 * (cd.w - 1);
    c.rmid = cd.r0 + half_r; c.cmid = cd.c0 + half_c;
    c.inv_hr = 1.0 / (half_r < 1 ? 1 : half_r); c.inv_hc = 1.0 / (half_c < 1 ? 1 : half_c);
    double *x = SD + L::X, *xt = SD + L::XT;
    c = uniform_cand(c);                                        // per-candidate scalars and pointers into SGPRs (they are computed by vector instructions)

#ifdef SDSM_PROFILE
    long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long prof_t_start = PROF_NOW();
    prof_acc[7] = wall_clock64();                               // 100 MHz wall clock: when this candidate's workgroup started ...
#endif
    sdsm_record r = {};
    r.n_pixels = cd.N; r.n_deform = st.M;
    int status_final = SDSM_CAND_OPTIMAL;
    int ev_value = 0, ev_full = 0;

    // ---- phases: 0 elliptical from zeros, 1 elliptical from the moment initialisation (only if phase 0 was
    //      not optimal, objects.py:337-355), 2 deformable shape model (objects.py:394-410) ---------------
    for (int i = tid; i < NMAX; i += L::WGS) x[i] = 0;
    __syncthreads();
    double psi_ell = INFINITY, psi_final = NAN, vinit_keep = INFINITY;
    bool have = false, fallback = false;
    double keep[6] = {0, 0, 0, 0, 0, 0};
    int s_prev = 0;
    for (int phase = P.init_elliptical ? 0 : 2; phase < 3; phase++) {
        int M = 0;
        tid = opaque_tid();
        if (phase == 1) {
            if (have && s_prev == 0) continue;                                  // pass 1 was optimal
            r.flags |= 1;
            __syncthreads();
            if (tid == 0) {                                                     // one lane: the divisions and square roots of the moment model need many registers
                const Frame fr = make_frame(c, P.img[cd.image].H, P.img[cd.image].W);
                const double z0 = fr.z0, z1 = fr.z1;
                int q_n = st.npos;
                unsigned long long q_r = st.sum_r, q_c = st.sum_c, q_rr = st.sum_rr, q_cc = st.sum_cc;
                asm volatile("" : "+s"(q_n), "+s"(q_r), "+s"(q_c), "+s"(q_rr), "+s"(q_cc));
                const double cnt = (double)q_n;                                 // keeps the whole computation inside this branch (it is pure: the optimiser would hoist it to the top of the kernel for every lane)
                double mr = (double)q_r / cnt, mc = (double)q_c / cnt;
                double cr = rint(mr), cc2 = rint(mc);                           // np.round: half to even
                double c0 = cr / z0, c1 = cc2 / z1;
                // exact integer second moments: n * sum(r^2) - (sum r)^2
                double vr = ((double)q_rr * cnt - (double)q_r * (double)q_r) / (cnt * cnt);
                double vc = ((double)q_cc * cnt - (double)q_c * (double)q_c) / (cnt * cnt);
                double h0 = sqrt(vr < 0 ? 0 : vr) / z0, h1 = sqrt(vc < 0 ? 0 : vc) / z1;
                h0 = h0 < 1e-8 ? 1e-8 : h0; h1 = h1 < 1e-8 ? 1e-8 : h1;
                double e0 = 1 / (h0 * h0), e1 = 1 / (h1 * h1);
                double b0 = e0 * c0, b1 = e1 * c1, cterm = c0 * b0 + c1 * b1 - 1;
                double thg[6] = {-e0, -e1, 0, b0, b1, -cterm}, thl[6];
                reparam(thg, fr.P0, fr.P1, fr.O0, fr.O1, thl);
                for (int i = 0; i < 6; i++) xt[i] = thl[i];
            }
            __syncthreads();
            double vinit = eval_value<L>(c, L::XT, 0);
            ev_value++;
            if (vinit > psi_ell) continue;                                      // objects.py:341-342
            vinit_keep = uni(vinit);
            if (tid < 6) x[tid] = xt[tid];
            __syncthreads();
        } else if (phase == 2) {
            if (P.init_elliptical && !have) { status_final = SDSM_CAND_ERROR; break; }   // CvxprogError (objects.py:351-353)
            M = Mfull;
            __syncthreads();
            const double *x0 = (!P.init_elliptical && P.x0) ? P.x0 + (size_t)6 * ci + cd.xi_off : nullptr;   // callable dsm/init: the caller's starting point
            if (x0) {
                const Frame fr = make_frame(c, P.img[cd.image].H, P.img[cd.image].W);
                double thg[6], thl[6];
                for (int i = 0; i < 6; i++) thg[i] = x0[i];
                reparam(thg, fr.P0, fr.P1, fr.O0, fr.O1, thl);                     // (every thread: the same values)
                for (int i = 0; i < 6; i++) keep[i] = uni(thl[i]);
            }
            for (int i = tid; i < NMAX; i += L::WGS) x[i] = i < 6 ? keep[i] : (x0 && i < 6 + M ? x0[i] : 0);
            if (M > 0) {                                                         // envelope of the Hessian (setup kernel)
                int *rbp = RBP, *fstp = FSTP, *rendp = RENDP;
                const int exi = efull - 6 * M - 21;
                const int otid = opaque_tid();                                   // (addresses formed here, not at the top of the kernel)
                for (int a = otid; a < M; a += L::WGS) { rbp[a] = P.env_rb[cd.xi_off + a]; fstp[a] = P.env_fst[cd.xi_off + a]; }
                if (tid < 6) { rbp[M + tid] = exi + tid * M + tid * (tid + 1) / 2; fstp[M + tid] = 0; }
                __syncthreads();
                for (int pnl = otid; SDSM_PANEL * pnl < M; pnl += L::WGS) {      // last xi row whose envelope reaches the panel's first column
                    int lo = SDSM_PANEL * pnl, hi = M - 1;                       // fst is non-decreasing, fst[first column] <= first column
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (fstp[mid] <= SDSM_PANEL * pnl) lo = mid; else hi = mid - 1; }
                    rendp[pnl] = lo;
                }
                c.env_size = efull;
            }
            __syncthreads();
        }
        double psi; int its;
        // what is known about psi at the start: nothing (zeros: N ln 2), the value just computed (moment initialisation), the elliptical
        // energy (the DSM solve starts at [theta_ell, 0]: the same surface, the regulariser is 0 there)
        const double start_bound = phase == 0 ? INFINITY : (phase == 1 ? vinit_keep : (have ? psi_ell : INFINITY));
        int s = newton<L>(c, M, P.max_iters, start_bound, &psi, &its, &ev_value, &ev_full PROF_ARG);
        tid = opaque_tid();
#ifdef SDSM_PROFILE
        if (phase < 2) { prof_acc[6] = PROF_NOW() - prof_t_start; }
#endif
        if (phase < 2) {
            r.iters_ell += its;
            if (s != 2) { have = true; psi_ell = psi; for (int i = 0; i < 6; i++) keep[i] = uni(x[i]); s_prev = s; }
            else if (phase == 0) s_prev = 2;
            __syncthreads();
        } else {
            r.iters_dsm = its;
            psi_final = psi;
            if (s != 0) {
                // exception -> fallback; 'unknown' and worse than the start -> fallback (objects.py:399-410).  The start's energy
                // comes from the value-only evaluator, psi from the full one: the same sums in a different reduction order, so
                // "worse" allows for rounding (same rule in the oracle)
                __syncthreads();
                {
                    const double *x0 = (!P.init_elliptical && P.x0) ? P.x0 + (size_t)6 * ci + cd.xi_off : nullptr;   // the fallback is the initialisation (objects.py:409-410)
                    for (int i = tid; i < NMAX; i += L::WGS) xt[i] = i < 6 ? keep[i] : (x0 && i < 6 + Mfull ? x0[i] : 0);
                }
                __syncthreads();
                const double vi = eval_value<L>(c, L::XT, Mfull);
                ev_value++;
                if (s == 2 || psi > vi * fresh(1 + 1e-12)) {
                    fallback = true;
                    for (int i = tid; i < NMAX; i += L::WGS) x[i] = xt[i];
                    __syncthreads();
                    psi_final = vi;
                    status_final = SDSM_CAND_FALLBACK;
                }
            }
        }
    }
    r.energy_ell = P.init_elliptical ? psi_ell : NAN;
    if (unsupported && status_final != SDSM_CAND_ERROR) status_final = SDSM_CAND_UNSUPPORTED;

    // ---- mask tail (objects.py:198-209) ----------------------------------------------------------
    tid = opaque_tid();
    const int mwords = (cd.h * cd.w + 31) / 32;
    uint32_t *mk = masks + cd.mask_off;
    if (!WIDE) for (int i = opaque_tid(); i < mwords; i += L::WGS) mk[i] = 0;         // (a group's member 0 cleared it before the first all-reduce)
    __syncthreads();
    int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
    int onb = 0;
    if (status_final != SDSM_CAND_ERROR) {
        pass_priority(c);
        for (int pb = c.p_lo + (opaque_tid() & ~63); pb < c.p_hi; pb += L::WGS) {
            const int p = pb + (int)(threadIdx.x & 63);
            const bool active = p < c.p_hi;
            RunPix rp;
            const int pl = load_pos(c, p, active);
            load_run(c, pl, active, rp);
            double Sv[SDSM_RUN];
            poly_surface(rp, x, Sv);
            if (Mfull > 0) smooth_add<1>(c, x, pl, chunk_entries(c, pb), Sv);
            if (active) {
                const uint32_t rc = c.crop_rc[p];
                const int pr = rc >> 16, pc0 = rc & 0xffffu;
#pragma unroll
                for (int k = 0; k < SDSM_RUN; k++) {
                    if (((rp.pm >> k) & 1u) && Sv[k] > 0) {
                        const int pc = pc0 + k;
                        const int bit = (pr - cd.r0) * cd.w + (pc - cd.c0);
                        atomicOr(&mk[bit >> 5], 1u << (bit & 31));
                        rmin = pr < rmin ? pr : rmin; rmax = pr > rmax ? pr : rmax;
                        cmin = pc < cmin ? pc : cmin; cmax = pc > cmax ? pc : cmax;
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(2);
        // 1-px pad ring of the image, polynomial part only (G~ has no rows there)
        const int imH = P.img[cd.image].H, imW = P.img[cd.image].W;
        const int ringw = imW + 2, ringh = imH + 2;
        for (int i = tid; i < 2 * ringw + 2 * ringh; i += L::WGS) {
            int pr, pc;
            if (i < ringw) { pr = -1; pc = i - 1; }
            else if (i < 2 * ringw) { pr = imH; pc = i - ringw - 1; }
            else if (i < 2 * ringw + ringh) { pr = i - 2 * ringw - 1; pc = -1; }
            else { pr = i - 2 * ringw - ringh - 1; pc = imW; }
            double u = ((double)pr - c.rmid) * c.inv_hr, v = ((double)pc - c.cmid) * c.inv_hc;
            double Sv = u * u * x[0] + v * v * x[1] + 2 * (u * v) * x[2] + 2 * u * x[3] + 2 * v * x[4] + x[5];
            if (Sv > 0) onb = 1;
        }
    }
    int *ired = (int *)(SD + L::RED);
    rmin = block_min_i32<L::NWAVES>(rmin, ired); cmin = block_min_i32<L::NWAVES>(cmin, ired);
    rmax = -block_min_i32<L::NWAVES>(-rmax, ired); cmax = -block_min_i32<L::NWAVES>(-cmax, ired);
    onb = -block_min_i32<L::NWAVES>(-onb, ired);
    if (WIDE) {                                                              // bounding box over the members' slices
        const int phase = *WIDE_PHASE;
        double *slot = c.wpool + SDSM_WIDE_SYNC + (size_t)(*WIDE_OPS & 1) * c.wG * SDSM_WIDE_PBUF;
        int *mine = reinterpret_cast<int *>(slot + (size_t)c.wg * SDSM_WIDE_PBUF);
        if (tid == 0) { mine[0] = rmin; mine[1] = cmin; mine[2] = rmax; mine[3] = cmax; }
        const bool okw = wide_barrier<L>(c, phase);
        for (int m = 0; m < c.wG; m++) {
            const int *o = reinterpret_cast<const int *>(slot + (size_t)m * SDSM_WIDE_PBUF);
            rmin = o[0] < rmin ? o[0] : rmin; cmin = o[1] < cmin ? o[1] : cmin;
            rmax = o[2] > rmax ? o[2] : rmax; cmax = o[3] > cmax ? o[3] : cmax;
        }
        if (!okw) status_final = SDSM_CAND_GIVEN_UP;                         // the group was given up (a member waited too long): not a solver failure
        if (c.wg != 0) return;                                               // member 0 writes the record
    }

    if (xi_out) for (int j = opaque_tid(); j < st.M && j < cd.Mcap; j += L::WGS) xi_out[cd.xi_off + j] = j < Mfull ? x[6 + j] : 0;   // (st.M may be the marker of a region beyond the setup tables)
    if (tid == 0) {
        // local basis -> full-image-normalised theta:  u = (x0 - O0) / P0
        const Frame fr = make_frame(c, P.img[cd.image].H, P.img[cd.image].W);
        double thl[6] = {x[0], x[1], x[2], x[3], x[4], x[5]}, thg[6];
        reparam(thl, 1 / fr.P0, 1 / fr.P1, -fr.O0 / fr.P0, -fr.O1 / fr.P1, thg);
        for (int i = 0; i < 6; i++) r.theta[i] = thg[i];
        r.energy = (status_final == SDSM_CAND_ERROR || status_final == SDSM_CAND_GIVEN_UP) ? NAN : psi_final;
        r.status = status_final;
        r.evals_value = ev_value; r.evals_full = ev_full;
        r.n_positive = st.npos; r.n_negative = st.nneg;
        r.on_boundary = onb;
#ifdef SDSM_PROFILE
        prof_acc[5] = PROF_NOW() - prof_t_start;
        prof_acc[12] = wall_clock64();                          // ... and ended (tools/class_stats.py: who ends the launch, and when it began)
        if (P.prof) for (int i = 0; i < 16; i++) P.prof[(size_t)ci * 16 + i] = prof_acc[i];
#endif
        if (rmax >= 0) { r.fg_r0 = rmin; r.fg_c0 = cmin; r.fg_h = rmax - rmin + 1; r.fg_w = cmax - cmin + 1; }
        *rec = r;
    }
}

// The kernels.  Workgroup groups: workgroup b takes entry `ticket` of the launch list P.order (their members).  Everybody else (list 4: class 1, all
// candidates, largest first; lists 0 .. 3: the classes beyond 1): a bounded number of RESIDENT workgroups pop entries of the
// launch list (head counter BatchParams.cls_count[LIST]) until it is exhausted and solve the candidates that belong to their class.
// The host knows only an upper bound of M, so the lists of these classes are upper bounds -- candidates that COULD belong, largest
// first; most do not.  One workgroup per entry meant thousands of 512-thread workgroups that exit at once but each need a whole free
// compute unit first: they kept three hardware queues busy for milliseconds (and class 2b waited 4.5 ms behind the empty global-memory
// class).  A resident workgroup drops a foreign entry in a microsecond.
template <int NMAX, int EMAX, int WPE, bool GLOBALH = false, int WGSIZE = 256, bool WIDE = false, int CLS = SDSM_CLS_1>
__global__ __launch_bounds__(WGSIZE, WPE) void sdsm_k_solve(BatchParams P, int handles_rest, sdsm_record *records, uint32_t *masks, double *xi_out, int list)
{
    using L = Lay<NMAX, EMAX, GLOBALH, WGSIZE>;
    // Issue priority: the passes over the pixels run at 0, everything between them -- reductions, the Newton bookkeeping, mask tail --
    // at 2 and the factorisation at 3: the serial sections of a candidate (dependent LDS round trips, barriers) go first when a SIMD
    // picks an instruction, the pixel passes of the other candidates of the compute unit fill the gaps (+2 % on the 8-image launch).
    __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x;
    load_log_table(tid);
    if constexpr (!WIDE) {                                   // (one call site of the solver per kernel: the groups take entry `ticket`, everybody else pops)
        // Class 1 too (round 4): with one workgroup per candidate the hardware deals workgroup b to compute-unit die b mod 8 IN ORDER -- a die whose
        // candidates take longer holds up the dispatch for all eight, and the launch ended with half of the chip idle (8 different BBBC039-like
        // images: 494 of 1024 workgroup slots in use 1.3 ms before the end, the last thousand candidates trickled in at 0.6 per microsecond).
        // Resident workgroups that pop the next candidate balance the dies by themselves.
        int *sh = reinterpret_cast<int *>(SD + L::FLAG);
        for (;;) {
            __syncthreads();                                     // (the previous candidate is finished by all threads)
            if (tid == 0) *sh = __hip_atomic_fetch_add(&P.cls_count[list], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int i = uni(*sh);
            if (i >= P.n) return;
            const int ci = uni(P.order[i]);
            __syncthreads();
            solve_candidate<NMAX, EMAX, GLOBALH, WGSIZE, false, CLS>(P, ci, 0, handles_rest, false, records, masks, xi_out);
        }
    }
    if constexpr (WIDE) {                                    // members are claimed in start order, not by workgroup index (see wide_barrier)
        int *tk = reinterpret_cast<int *>(SD + L::FLAG);
        if (tid == 0) *tk = __hip_atomic_fetch_add(P.wide_ticket + (CLS == SDSM_CLS_WIDE2B ? 1 : 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int slot = uni(*tk);
        __syncthreads();
        if (slot >= P.n) return;                             // cannot happen: one ticket per workgroup of the launch
        const int entry = uni(P.order[slot]);
        solve_candidate<NMAX, EMAX, GLOBALH, WGSIZE, true, CLS>(P, entry & 0xffffff, (entry >> 24) & 0xff, handles_rest, false, records, masks, xi_out);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Point evaluation for parity tests (sdsm_batch_eval): psi, gradient and the polynomial block of the Hessian at
// caller-given parameters, computed by the SAME evaluators the solver uses (eval_full: loss_terms;
// eval_value: softplus_neg) on the crops and G~ rows a previous sdsm_batch_launch left in the workspace.  Parameters and
// results are in the reference's full-image-normalised basis (dsm.py:49-54); the evaluators work in the candidate's local
// basis theta_local = R theta (reparam), so grad = R^T grad_local and H = R^T H_local R.
// out: [0, 2n) psi by the full and by the value-only evaluator; [2n, 23n) lower triangle of the 6x6 polynomial Hessian
// block; [23n, ...) gradients in the layout of `params`: candidate i at 6 i + xi_off(i), theta then xi.
// ---------------------------------------------------------------------------------------------------------
template <int NMAX, int EMAX, bool GLOBALH, int WGSIZE>
__global__ __launch_bounds__(WGSIZE) void sdsm_k_eval(BatchParams P, int nprev, int eprev, const double *params, double *out)
{
    using L = Lay<NMAX, EMAX, GLOBALH, WGSIZE>;
    const int tid = threadIdx.x;
    const int ci = blockIdx.x;
    load_log_table(tid);
    const CandDesc cd = uniform_desc(P.cand[ci]);
    const CandState st = uniform_state(P.state[ci]);
    if (st.status != ST_OK || st.M < 0 || 6 + st.M > SDSM_MAX_N_GLOBAL) return;           // out stays NaN (filled by the host)
    const int M = st.M, n = 6 + M;
    const int efull = M > 0 ? st.env_size : 21;
    if (nprev > 0 && n <= nprev && efull <= eprev) return;                                 // an earlier class evaluates it
    if (!(n <= NMAX && efull <= EMAX)) return;
    if (GLOBALH && cd.hglob_off < 0) return;
    Cand c;
    c.N = cd.N; c.NR = st.NR; c.hzmax = M > 0 ? st.hzmax : 0; c.env_size = efull;
    c.p_lo = 0; c.p_hi = st.NR; c.wg = 0; c.wG = 1; c.wpool = nullptr; c.wtimeout = 0;
    c.crop_y2 = (g_cf64x2_p)(P.crop_y + cd.run_off * SDSM_RUN); c.crop_rc = (g_cu32_p)(P.crop_rc + cd.run_off); c.meta = (g_cu32_p)(P.run_meta + cd.run_off);
    c.aux = (g_cu32_p)(P.run_aux + cd.run_off);
    c.ell_im = (g_cu32_p)(P.ell_im + cd.ell_off); c.ell_w4 = (g_cf32x4_p)(reinterpret_cast<const f32x4 *>(P.ell_w) + cd.ell_off);
    c.hglob = (GLOBALH && cd.hglob_off >= 0) ? P.hglob + cd.hglob_off : nullptr;
    c.scale = P.scale / cd.N; c.epsilon = P.epsilon; c.alpha = P.alpha; c.reg0 = P.reg_unit * M;
    c.yexp = st.yexp; c.boost = 0;
    const double half_r = 0.5 * (cd.h - 1), half_c = 0.5 * (cd.w - 1);
    c.rmid = cd.r0 + half_r; c.cmid = cd.c0 + half_c;
    c.inv_hr = 1.0 / (half_r < 1 ? 1 : half_r); c.inv_hc = 1.0 / (half_c < 1 ? 1 : half_c);
    const int imH = P.img[cd.image].H, imW = P.img[cd.image].W;
    const double z0 = imH > 1 ? imH - 1.0 : 1.0, z1 = imW > 1 ? imW - 1.0 : 1.0;
    const double P0 = 1.0 / (c.inv_hr * z0), P1 = 1.0 / (c.inv_hc * z1), O0 = c.rmid / z0, O1 = c.cmid / z1;
    c = uniform_cand(c);
    double *x = SD + L::X, *g = SD + L::G, *Hp = hess_ptr<L>(c);
    const double *pin = params + (size_t)6 * ci + cd.xi_off;
    if (tid == 0) {
        *WIDE_PHASE = 0;
        double thg[6], thl[6];
        for (int i = 0; i < 6; i++) thg[i] = pin[i];
        reparam(thg, P0, P1, O0, O1, thl);
        for (int i = 0; i < 6; i++) x[i] = thl[i];
    }
    for (int j = tid; j < M; j += L::WGS) x[6 + j] = pin[6 + j];
    if (M > 0) {
        int *rbp = RBP, *fstp = FSTP, *rendp = RENDP;
        const int exi = efull - 6 * M - 21;
        for (int a = tid; a < M; a += L::WGS) { rbp[a] = P.env_rb[cd.xi_off + a]; fstp[a] = P.env_fst[cd.xi_off + a]; }
        if (tid < 6) { rbp[M + tid] = exi + tid * M + tid * (tid + 1) / 2; fstp[M + tid] = 0; }
        if (tid == 0) rendp[0] = M - 1;
    }
    __syncthreads();
#ifdef SDSM_PROFILE
    long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const double psi_value = eval_value<L>(c, L::X, M);
    const double psi_full = eval_full<L>(c, M, 0.0, psi_value PROF_ARG);
    __syncthreads();
    if (tid == 0) {
        double Rm[6][6];                                   // theta_local = Rm theta_global (reparam is linear in theta)
        for (int a = 0; a < 6; a++) {
            double e[6] = {0, 0, 0, 0, 0, 0}, col[6];
            e[a] = 1;
            reparam(e, P0, P1, O0, O1, col);
            for (int b = 0; b < 6; b++) Rm[b][a] = col[b];
        }
        double Hl[6][6];
        for (int a = 0; a < 6; a++) for (int b = 0; b <= a; b++) {
            const double v = M == 0 ? Hp[tri(a, b)] : Hp[RBP[M + a] + M + b];
            Hl[a][b] = v; Hl[b][a] = v;
        }
        out[2 * ci] = psi_full; out[2 * ci + 1] = psi_value;
        double *oh = out + (size_t)2 * P.n + (size_t)21 * ci;
        for (int a = 0; a < 6; a++) for (int b = 0; b <= a; b++) {
            double v = 0;
            for (int k = 0; k < 6; k++) for (int l = 0; l < 6; l++) v += Rm[k][a] * Hl[k][l] * Rm[l][b];
            oh[tri(a, b)] = v;
        }
        double *og = out + (size_t)23 * P.n + (size_t)6 * ci + cd.xi_off;
        for (int a = 0; a < 6; a++) {
            double v = 0;
            for (int k = 0; k < 6; k++) v += Rm[k][a] * g[k];
            og[a] = v;
        }
    }
    double *og = out + (size_t)23 * P.n + (size_t)6 * ci + cd.xi_off;
    for (int j = tid; j < M; j += L::WGS) og[6 + j] = g[6 + j];
}

extern "C" hipError_t sdsm_launch_eval(const BatchParams &P, const double *params, double *out, hipStream_t stream)
{
    if (P.n <= 0) return hipSuccess;
    hipError_t e;
    {
        auto kern = sdsm_k_eval<SDSM_K1_NMAX, SDSM_K1_EMAX, false, 256>;
        constexpr int lds = Lay<SDSM_K1_NMAX, SDSM_K1_EMAX, false, 256>::TOTAL_BYTES;
        if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(P.n), dim3(256), lds, stream, P, 0, 0, params, out);
    }
    {
        auto kern = sdsm_k_eval<SDSM_MAX_N_SOLVE, SDSM_K2_EMAX, false, 512>;
        constexpr int lds = Lay<SDSM_MAX_N_SOLVE, SDSM_K2_EMAX, false, 512>::TOTAL_BYTES;
        if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(P.n), dim3(512), lds, stream, P, SDSM_K1_NMAX, SDSM_K1_EMAX, params, out);
    }
    {
        auto kern = sdsm_k_eval<SDSM_MAX_N_GLOBAL, SDSM_MAX_N_GLOBAL * (SDSM_MAX_N_GLOBAL + 1) / 2, true, 512>;
        constexpr int lds = Lay<SDSM_MAX_N_GLOBAL, SDSM_MAX_N_GLOBAL * (SDSM_MAX_N_GLOBAL + 1) / 2, true, 512>::TOTAL_BYTES;
        if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(P.n), dim3(512), lds, stream, P, SDSM_MAX_N_SOLVE, SDSM_K2_EMAX, params, out);
    }
    return hipGetLastError();
}

// ---- launch helper (called from sdsm_api.hip) ----------------------------------------------------
// class 1:  6 + M <= 128,  envelope <= 2560 doubles, <= P.k1_pixmax pixels   256 threads, LDS ~ 30 KB  (three workgroups / CU)
// class 1b: 6 + M <= 256,  envelope <= 7168 doubles, <= P.k1_pixmax pixels   256 threads, LDS ~ 79 KB  (two workgroups / CU; 512-thread kernels need
//           ~200 registers per thread, i.e. a whole compute unit per workgroup whatever their LDS)
// class 2:  6 + M <= 1024, envelope <= 11000 doubles   512 threads, LDS ~ 157 KB (one workgroup / CU)
// class 2b: 6 + M <= 512,  envelope <= 15900 doubles   512 threads, LDS ~ 160 KB (one workgroup / CU)
// class 3:  6 + M <= 1024, any envelope                512 threads, Hessian in global memory (only launched when needed)
// The classes are independent: they run concurrently on streams forked from the caller's stream.  Measured on the synthetic 4096^2
// image (10 073 candidates; 2695 of them beyond class 1, 122 beyond class 2) before classes 1b / 2b existed: class 2 alone needed
// 10.9 s of workgroup time at ONE workgroup per compute unit (43 ms), and the global-memory class, queued behind it on the same
// stream, another 75 ms (52 ms per candidate: every atomic of the pixel pass goes to memory).
template <int NMAX, int EMAX, int WPE, bool GLOBALH = false, int WGSIZE = 256, bool WIDE = false, int CLS = SDSM_CLS_1>
static hipError_t launch_class(const BatchParams &P, int grid, int list, int handles_rest, sdsm_record *records, uint32_t *masks, double *xi_out, hipStream_t stream)
{
    auto kern = sdsm_k_solve<NMAX, EMAX, WPE, GLOBALH, WGSIZE, WIDE, CLS>;
    constexpr int lds = Lay<NMAX, EMAX, GLOBALH, WGSIZE>::TOTAL_BYTES;
    static_assert(lds <= 160 * 1024 - 512, "LDS budget");
    static bool attr_set[64] = {};                       // per instantiation and device; the attribute belongs to the function, not to the launch
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WGSIZE), lds, stream, P, handles_rest, records, masks, xi_out, list);
    return hipGetLastError();
}

extern "C" hipError_t sdsm_launch_setup_rows(const BatchParams &P, hipStream_t stream, const int32_t *order_w, int n_w);

// Gate of class 1: a 512-thread workgroup needs a whole free compute unit, and once the thousands of small workgroups of class 1 flood the
// chip none becomes free before class 1 is through -- yet the few long candidates of the large classes ARE the end of a launch.  Class 1
// therefore starts when the FIRST kernel of every side queue has its work in resident workgroups: one wavefront on the caller's stream polls the
// heads of their launch lists (a head at the end of its list: every entry has been claimed by a resident workgroup) and the ticket counters
// of the workgroup groups (all tickets drawn: every member has started), with a cap -- side queues that share a hardware queue with the
// caller's stream (GPU_MAX_HW_QUEUES too small) cannot start before this kernel ends.  (Rounds 2-3: a fixed wait of 60 / 120 us, tuned on one
// workload mix; the wait is now what the launch needs: tens of microseconds for short lists.)
__global__ void sdsm_k_gate(const int32_t *cls_count, const int32_t *ticket, int l0, int n0, int l1, int n1, int l2, int n2, long long cap_ticks, long long stall_ticks)
{
    const long long t0 = wall_clock64();
    long long t_change = t0;
    int last = 0;
    for (;;) {
        const int c0 = n0 > 0 ? __hip_atomic_load(l0 >= 0 ? &cls_count[l0] : &ticket[-1 - l0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int c1 = n1 > 0 ? __hip_atomic_load(l1 >= 0 ? &cls_count[l1] : &ticket[-1 - l1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int c2 = n2 > 0 ? __hip_atomic_load(l2 >= 0 ? &cls_count[l2] : &ticket[-1 - l2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        if (c0 >= n0 && c1 >= n1 && c2 >= n2) break;          // every list claimed, every member started
        const long long now = wall_clock64();
        if (now - t0 > cap_ticks) break;
        // ... or nothing moves any more: the resident workgroups are all busy with candidates of their own (more real candidates than resident
        // workgroups: synthetic 4096^2), the members that fit the chip have started (more members than compute units: GOWT1-like) -- whatever can
        // be resident is (round 4, first version without this: the gate ran into its cap on exactly those launches, 0.4-0.9 ms)
        const int sum = c0 + c1 + c2;
        if (sum != last) { last = sum; t_change = now; }
        else if (sum > 0 && now - t_change > stall_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
#ifndef SDSM_GATE_CAP_US
#define SDSM_GATE_CAP_US 150       // (400 until the end of round 4: launches whose group members outnumber the compute units keep drawing tickets one by one and held the gate to its cap)
#endif
#ifndef SDSM_GATE_STALL_US
#define SDSM_GATE_STALL_US 20
#endif

// Do the caller's stream and the three side streams run side by side?  Four one-wavefront kernels, one per stream, each waits (at most
// `cap_ticks`) until all four have started; *timed_out is set if one gave up: some of the streams share a hardware queue.
__global__ void sdsm_k_queue_probe(int32_t *arrived, int32_t *timed_out, long long cap_ticks)
{
    __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4) {
        if (wall_clock64() - t0 > cap_ticks) { __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(16);
    }
}
extern "C" hipError_t sdsm_launch_queue_probe(int32_t *d_words /* 2, zeroed */, hipStream_t s0, hipStream_t s1, hipStream_t s2, hipStream_t s3)
{
    hipStream_t st[4] = {s0, s1, s2, s3};
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(sdsm_k_queue_probe, dim3(1), dim3(64), 0, st[i], d_words, d_words + 1, 200000ll);   // cap: 2 ms
    return hipGetLastError();
}

// Resident workgroups of the classes beyond 1 (each pops entries of its launch list until the list is exhausted).  A 512-thread
// workgroup needs a whole free compute unit and waits for one while class 1 floods the chip: no more of them than the class can use.
#define SDSM_RESIDENT_2 256         // class 2: one per compute unit (synthetic 4096^2: hundreds of candidates)
#define SDSM_RESIDENT_2B 128        // class 2b (synthetic 4096^2: 91 candidates)
#define SDSM_RESIDENT_3 64          // global-memory class (34 there, 23-33 ms each: none may wait for another; usually none at all)
#define SDSM_RESIDENT_1B 512        // class 1b: two per compute unit
#define SDSM_RESIDENT_1 1280        // class 1: four (192 threads) or three (256 threads) per compute unit + a margin that starts as slots free up

extern "C" hipError_t sdsm_launch_solve(const BatchParams &P, sdsm_record *records, uint32_t *masks, double *xi_out,
                                        hipStream_t stream, hipStream_t side1, hipStream_t side2, hipStream_t side3, hipStream_t side4, hipEvent_t *ev /* 5 */,
                                        int n_c, int n_d, int n_w, int n_r)
{
    hipError_t e;
    // P.order = [all n, largest first | n_c candidates whose bound Mcap admits more than class 1 | the n_d that admit more than class 2 |
    // (candidate | member << 24) of the workgroup groups].  n_c / n_d only bound the lengths of the device-built work lists.
    BatchParams Pw = P;
    Pw.order = P.order + P.n + n_c + n_d; Pw.n = n_w;
    BatchParams Pc = P, Pd = P;
    Pc.order = P.order + P.n; Pc.n = n_c;
    Pd.order = P.order + P.n + n_c; Pd.n = n_d;
    const int g_c = n_c < SDSM_RESIDENT_2 ? n_c : SDSM_RESIDENT_2, g_d = n_d < SDSM_RESIDENT_2B ? n_d : SDSM_RESIDENT_2B, g_3 = n_d < SDSM_RESIDENT_3 ? n_d : SDSM_RESIDENT_3;
    const int g_b = n_c < SDSM_RESIDENT_1B ? n_c : SDSM_RESIDENT_1B;
    // rows of G~ of the very large regions (one workgroup per member of their workgroup groups), on the caller's stream right behind the
    // setup kernel: a region whose envelope turns out too large for a group belongs to class 2b or the global-memory class, whose kernels
    // therefore wait for them too
    if (n_r > 0 && (e = sdsm_launch_setup_rows(P, stream, Pw.order + n_w, n_r)) != hipSuccess) return e;
    // fork: the side streams wait for everything queued on the caller's stream so far (setup kernels)
    if (n_c > 0 || n_d > 0 || n_w > 0) { if ((e = hipEventRecord(ev[0], stream)) != hipSuccess) return e; }
    // Queues (kernels of one stream run one after the other; the longest chains first on each):
    //   side1: the global-memory class (one candidate takes tens of milliseconds), then class 2b;
    //   side2: the workgroup groups of the very large regions, then class 2;
    //   side3: class 1b (two workgroups per compute unit);
    //   caller's stream: class 1 (all candidates in the host's order, largest first; the others leave at once).
    // The driver maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with another stream in use by the caller a side
    // stream may share a queue and wait behind its neighbour -- superdsm_amd sets the variable to 8 when it is imported first.
    // An error while enqueueing (a launch that fails) must not leave forked work behind on the side streams, which all plans of the device share: the first error
    // is kept, nothing further is launched, but every side stream that was forked is still recorded and joined below.
    hipError_t first = hipSuccess;
    auto keep = [&](hipError_t r) { if (first == hipSuccess && r != hipSuccess) first = r; return first == hipSuccess; };
    bool forked1 = false, forked2 = false, forked3 = false, forked4 = false;
    // Plans WITHOUT workgroup groups (synthetic 4096^2: every class has hundreds of candidates of its own): class 2b on a queue of its own instead of behind the
    // global-memory class -- a kernel behind another one starts when class 1 already holds the chip with resident workgroups, and a 512-thread workgroup then gets no
    // compute unit before class 1 is through (start / end stamps of that launch: class 2b started at 50 of 59 ms and was its last 9 ms).  With groups the queues stay as
    // they are (a fourth stream for class 2 beside the groups was measured slower in round 3: its resident workgroups take compute units from the group members).
    // (only where the list is long: on the 8-copies step, whose list of 2b / 3 candidates holds a few dozen upper bounds and no real candidate, the 128 more resident
    // workgroups at the start cost 0.2 of 4.9 ms)
    const bool own_2b = n_w == 0 && n_d >= 2 * SDSM_RESIDENT_2B && side4 != nullptr;
    if (n_d > 0 || n_w > 0) {
        if (keep(hipStreamWaitEvent(side1, ev[0], 0))) {
            forked1 = true;
            // (the groups whose envelope needs the layout of class 2b: few and short next to the global-memory class behind them)
            if (n_w > 0 && first == hipSuccess) keep(launch_class<SDSM_K2B_NMAX, SDSM_K2B_EMAX, 2, false, 512, true, SDSM_CLS_WIDE2B>(Pw, n_w, -1, 0, records, masks, xi_out, side1));
            if (n_d > 0 && first == hipSuccess) keep(launch_class<SDSM_MAX_N_GLOBAL, SDSM_MAX_N_GLOBAL * (SDSM_MAX_N_GLOBAL + 1) / 2, 2, true, 512, false, SDSM_CLS_3>(Pd, g_3, 3, 0, records, masks, xi_out, side1));
            if (n_d > 0 && !own_2b && first == hipSuccess) keep(launch_class<SDSM_K2B_NMAX, SDSM_K2B_EMAX, 2, false, 512, false, SDSM_CLS_2B>(Pd, g_d, 2, 0, records, masks, xi_out, side1));
        }
    }
    if (own_2b && first == hipSuccess) {
        if (keep(hipStreamWaitEvent(side4, ev[0], 0))) {
            forked4 = true;
            keep(launch_class<SDSM_K2B_NMAX, SDSM_K2B_EMAX, 2, false, 512, false, SDSM_CLS_2B>(Pd, g_d, 2, 0, records, masks, xi_out, side4));
        }
    }
    if ((n_w > 0 || n_c > 0) && first == hipSuccess) {
        if (keep(hipStreamWaitEvent(side2, ev[0], 0))) {
            forked2 = true;
            if (n_w > 0 && first == hipSuccess) keep(launch_class<SDSM_MAX_N_SOLVE, SDSM_K2_EMAX, 2, false, 512, true, SDSM_CLS_WIDE>(Pw, n_w, -1, 0, records, masks, xi_out, side2));
            if (n_c > 0 && first == hipSuccess) keep(launch_class<SDSM_MAX_N_SOLVE, SDSM_K2_EMAX, 2, false, 512, false, SDSM_CLS_2>(Pc, g_c, 1, 0, records, masks, xi_out, side2));
        }
    }
    if (n_c > 0 && first == hipSuccess) {
        if (keep(hipStreamWaitEvent(side3, ev[0], 0))) {
            forked3 = true;
            keep(launch_class<SDSM_K1B_NMAX, SDSM_K1B_EMAX, SDSM_K1B_WPE, false, SDSM_K1B_THREADS, false, SDSM_CLS_1B>(Pc, g_b, 0, 0, records, masks, xi_out, side3));
        }
    }
    // class 1 runs THREE wavefronts per SIMD (168 registers).  Measured on the 8-image launch of round 2: 7.4 ms at two wavefronts
    // (240 registers), 5.5 ms at three, 6.6 ms at four (128 registers: 36 spilled) -- the solver is latency bound and a third workgroup
    // per compute unit fills its stalls.
    // Throughput mode: 192 threads per candidate, FOUR workgroups per compute unit (the same twelve wavefronts; one more independent
    // candidate per compute unit, whose barriers stall three wavefronts instead of four: 5.43 -> 5.10 ms on that launch).
    // Latency mode (one image at a time, a batch is as slow as its slowest candidate): 256 threads per candidate.
    // Head start: a 512-thread workgroup needs a whole free compute unit, and once the thousands of small workgroups of class 1 flood
    // the chip none becomes free before class 1 is through (kernel trace of 8 different BBBC039-like images: the global-memory class
    // -- no candidates -- "ran" 5.4 ms, class 2b behind it started when class 1 ended, the groups took 9.7 ms instead of 4.7; stream
    // priorities change nothing).  The few long candidates of the large classes ARE the end of the launch: class 1 starts some tens of
    // microseconds after them -- their resident workgroups are in place by then, the ones without work gone again.
    if ((n_c > 0 || n_d > 0 || n_w > 0) && first == hipSuccess) {
        // the first kernel of every side queue: side 1: groups of class-2b layout (ticket [1]), else the global-memory class (list 3); side 2: groups of
        // class-2 layout (ticket [0]), else class 2 (list 1); side 3: class 1b (list 0)
        const int l0 = n_w > 0 ? -2 : 3, c0 = n_w > 0 ? n_w : n_d;
        const int l1 = n_w > 0 ? -1 : 1, c1 = n_w > 0 ? n_w : n_c;
        hipLaunchKernelGGL(sdsm_k_gate, dim3(1), dim3(64), 0, stream, (const int32_t *)P.cls_count, (const int32_t *)P.wide_ticket, l0, c0, l1, c1, 0, n_c,
                           (long long)SDSM_GATE_CAP_US * 100, (long long)SDSM_GATE_STALL_US * 100);
    }
    static const int resident_1 = [] { const char *e = getenv("SDSM_RESIDENT_1"); const int v = e ? atoi(e) : 0; return v > 0 ? v : SDSM_RESIDENT_1; }();   // (diagnostic knob)
    const int g_1 = P.n < resident_1 ? P.n : resident_1;
    if (first == hipSuccess) {
        if (!P.latency) keep(launch_class<SDSM_K1_NMAX, SDSM_K1_EMAX, SDSM_K1_WPE, false, SDSM_K1_THREADS, false, SDSM_CLS_1>(P, g_1, 4, 1, records, masks, xi_out, stream));
        else keep(launch_class<SDSM_K1_NMAX, SDSM_K1_EMAX, 3, false, 256, false, SDSM_CLS_1>(P, g_1, 4, 1, records, masks, xi_out, stream));
    }
    // join (also after an error: whatever was forked is recorded and waited for)
    if (forked1) { hipError_t r = hipEventRecord(ev[1], side1); if (r == hipSuccess) r = hipStreamWaitEvent(stream, ev[1], 0); keep(r); }
    if (forked2) { hipError_t r = hipEventRecord(ev[2], side2); if (r == hipSuccess) r = hipStreamWaitEvent(stream, ev[2], 0); keep(r); }
    if (forked3) { hipError_t r = hipEventRecord(ev[3], side3); if (r == hipSuccess) r = hipStreamWaitEvent(stream, ev[3], 0); keep(r); }
    if (forked4) { hipError_t r = hipEventRecord(ev[4], side4); if (r == hipSuccess) r = hipStreamWaitEvent(stream, ev[4], 0); keep(r); }
    return first;
}
