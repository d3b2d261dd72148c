// Shared device-side structures and workgroup primitives (gfx950, wave64, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdsm.h"

// Explicit global address space for pointers that travel through structs / non-inlined functions: without it the
// compiler falls back to flat_load, whose results can only be awaited with vmcnt(0) (one load in flight at a time).
#define SDSM_GLOBAL __attribute__((address_space(1)))
typedef const double SDSM_GLOBAL *g_cdouble_p;
typedef const float SDSM_GLOBAL *g_cfloat_p;
typedef const uint32_t SDSM_GLOBAL *g_cu32_p;
typedef const uint16_t SDSM_GLOBAL *g_cu16_p;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 SDSM_GLOBAL *g_cf32x4_p;
typedef const u16x4 SDSM_GLOBAL *g_cu16x4_p;

#define SDSM_WG 256
#define SDSM_WAVES (SDSM_WG / 64)

// limits of this implementation (DESIGN.md "Limits")
#define SDSM_MAX_LABELS 65535      // footprint bitset in LDS
#define SDSM_MAX_BBOX_DIM 4096     // row / column rank tables in LDS
#define SDSM_PSF_LDS 4256           // words of LDS shared by the footprint bitset (2048) and the PSF table (k <= 65)
#define SDSM_MAX_GRID 2048         // grid points kept in LDS during setup
#define SDSM_K1_NMAX 128           // solve class 1: 6 + M <= 128 and envelope <= SDSM_K1_EMAX doubles (LDS ~ 30 KB)
#define SDSM_K1_EMAX 2560
#define SDSM_K1B_NMAX 256          // solve class 1b: 6 + M <= 256 and envelope <= 7168 doubles: 256 threads, LDS ~ 79 KB, TWO workgroups per compute unit
#define SDSM_K1B_EMAX 7168         //   (between class 1 and class 2, whose 157 KB leave one workgroup per compute unit: regions of ~5-15 k pixels, M ~ 100-250)
#define SDSM_K2B_NMAX 512          // solve class 2b: 6 + M <= 512 and envelope <= 15170 doubles: the LDS that class 2 spends on vectors of 1024 unknowns
#define SDSM_K2B_EMAX 15170        //   holds a larger envelope instead (~ 160 KB): keeps most of what class 2 cannot hold out of the global-memory class
// Very large regions are solved by a GROUP of workgroups (2 .. 8, one per 8192 pixels): each takes a slice of the pixels in
// every pass and the partial sums / gradient / Hessian are all-reduced through global memory (sdsm_solve.hip, WIDE).
#ifndef SDSM_WIDE_MIN_PIXELS
#define SDSM_WIDE_MIN_PIXELS 12288
#endif
#ifndef SDSM_WIDE_SLICE
#define SDSM_WIDE_SLICE 8192
#endif
#define SDSM_WIDE_MAX_G 8
#define SDSM_WIDE_SYNC 16           // doubles reserved for the group's counters at the start of its pool block
#define SDSM_WIDE_FCAP 4104         // of them for the psi-type sums of the member's super-chunks (1 count + 4 * 1024 + pad: slices of up to 2 M pixels)
#define SDSM_WIDE_PBUF (SDSM_K2B_EMAX + SDSM_MAX_N_SOLVE + 64 + SDSM_WIDE_FCAP)   // doubles one workgroup publishes per exchange (the larger envelope of the two group layouts: class 2 / class 2b)
#define SDSM_WIDE_PIXELS 3072       // latency mode: larger regions go to class 2 (512 threads per candidate; a batch is as slow as its slowest candidate)
#define SDSM_K2_EMAX 11000         // solve class 2: 6 + M <= SDSM_MAX_N_SOLVE and envelope <= 11000 doubles (LDS ~ 159 KB)
#define SDSM_K1_DENSE_N 70         // 6 + M <= 70: even a dense triangle fits class 1
#define SDSM_ENV_DENSE_N 146       // 6 + M <= 146: even a dense triangle fits class 2
#define SDSM_RUN 4                  // pixels per run: the region pixels of one image row inside one aligned 4-column cell share a G~ index list
#define SDSM_MAX_ELL_GROUPS 256    // classes of the counting sort of the runs by their number of G~ entries (zcap_run <= 1024: classes of 4 beyond 256)
#define SDSM_HZREG 8                // leading ('significant') entries of a run the solve kernel keeps in registers for the approximate Hessian
#define SDSM_MSLOTS 16              // lane slots (lane & 15) of the fixed-point moment accumulators in LDS (21 moments x 2 words each)
#define SDSM_SUPER 8                // chunks (of 64 runs) per super-chunk: psi = sequential sum over super-chunks of the sequential sum of their chunk totals
#ifndef SDSM_PANEL
#define SDSM_PANEL 4               // columns per panel (8 measured slower: 195 k vs 200 k solves/s); of the envelope Cholesky; first stored columns are multiples of it
#endif
#ifndef SDSM_ROWS_MIN_PIXELS
#define SDSM_ROWS_MIN_PIXELS 4096    // larger regions: rows of G~ by several workgroups (sdsm_k_setup_rows), one per SDSM_ROWS_SLICE pixels, at most SDSM_ROWS_MAX_G
#endif
#ifndef SDSM_ROWS_SLICE
#define SDSM_ROWS_SLICE 1536
#endif
#ifndef SDSM_ROWS_MAX_G
#define SDSM_ROWS_MAX_G 48
#endif
#define SDSM_SETUP_SMALL_PIXELS 4096   // two-launch setup: regions of at most this many pixels (that fit its tables) are set up by the 256-thread class
#define SDSM_MAX_N_SOLVE 1024      // 6 + M handled by the LDS classes and the workgroup groups
#define SDSM_MAX_N_GLOBAL 2048     // 6 + M handled by the global-memory class (vectors of that many unknowns in LDS, Hessian + factor in global memory): every grid the
                                   //   setup kernel can build (SDSM_MAX_GRID points) has a DSM solve; beyond (and regions beyond the setup tables): the elliptical result, flagged

#ifdef SDSM_PROFILE
#define PROF_NOW() ((long long)__builtin_readcyclecounter())
#define PROF_ADD(slot, t0) do { long long _t = PROF_NOW(); prof_acc[slot] += _t - (t0); (t0) = _t; } while (0)
#else
#define PROF_NOW() 0ll
#define PROF_ADD(slot, t0) do { (void)(t0); } while (0)
#endif

enum { ST_OK = 0, ST_TRIVIAL = 2, ST_ERROR = 3, ST_UNSUPPORTED = 4 };

// Host-planned description of one candidate (read-only on the device).
struct CandDesc {
    int64_t crop_off;   // first pixel of the candidate's per-pixel setup arrays (tmp_y / tmp_rc / crop_cc / dist / inv)
    int64_t ell_off;    // first entry of its G~ block (NRcap * zcap_run entries; entry j of run position p at j * NR + p)
    int64_t mask_off;   // first uint32 word of its bit-packed region-bbox mask
    int64_t xi_off;     // first entry of its grid / xi block (Mcap entries)
    int32_t N;          // region pixels (sum of the atoms' valid areas)
    int32_t r0, c0, h, w;   // region bounding box (union of the atoms' valid extents)
    int32_t fp_off, fp_len; // footprint labels
    int32_t Mcap;       // upper bound of M
    uint32_t perm_inv;  // crop position of the pixel with raster rank i is (i * perm_inv) mod N (low-discrepancy scatter)
    int32_t wide_g;     // > 0: solved by a group of this many workgroups (regions of more than SDSM_WIDE_MIN_PIXELS pixels)
    int64_t hglob_off;  // first double of its block in the global Hessian pool (a dense triangle of 6 + min(Mcap, 1018) unknowns; only if 6 + Mcap > SDSM_ENV_DENSE_N: the envelope may not fit LDS), else -1
    int64_t wide_off;   // first double of the candidate's block in the wide pool: SDSM_WIDE_SYNC (counters of the group / of sdsm_k_setup_rows) + 2 * wide_g * SDSM_WIDE_PBUF doubles; -1 if neither wide_g nor rows_g
    int32_t image;      // index into BatchParams.img (plans over several images, sdsm_plan_create_multi)
    int32_t NRcap;      // upper bound of the number of runs (min(N, rows * 4-column cells of the bounding box))
    int64_t run_off;    // first run of the candidate's packed crop (crop_y: 4 doubles per run; crop_rc, run_meta, run_q0, run_aux: one word per run)
    int32_t rows_g;     // > 0: its rows of G~ are built by this many workgroups of sdsm_k_setup_rows (regions of more than SDSM_ROWS_MIN_PIXELS pixels); sync words at wide_off
    int32_t pad1;
};

// Written by the setup kernel.
struct CandState {
    int32_t M;          // number of grid points (columns of G~)
    int32_t status;     // ST_*
    int32_t hc, wc;     // shape of the mask after deleting empty rows / columns (dsm.py:185-186)
    int32_t npos;       // region pixels with y > 0
    int32_t zmax;       // largest number of G~ entries of a run (union of the rows of its pixels)
    unsigned long long sum_r, sum_c, sum_rr, sum_cc;   // moments of the y > 0 pixels (image coordinates)
    int32_t hzmax;      // largest number of leading entries of a run ('significant', >= hess_thr * row maximum, for at least one of its pixels)
    int32_t env_size;   // doubles of the solver's Hessian in envelope storage (see env_fst / env_rb)
    int32_t nneg;       // region pixels with y < 0
    int32_t yexp;       // |y| < 2^yexp for every region pixel (clamped to +-400): scale of the solver's fixed-point sums
    int32_t NR;         // runs of the packed crop
    int32_t pad[7];
};
static_assert(sizeof(CandState) == 104, "CandState layout");
static_assert(sizeof(CandDesc) == 112, "CandDesc layout");

// One image of a plan: device pointers given at launch time, shape from the plan.
#define SDSM_MAX_IMAGES 16
struct ImageRef {
    const double *y;
    const int32_t *atoms;
    const uint8_t *valid;
    int32_t H, W;
};

struct BatchParams {
    int32_t n, n_images;
    int32_t n_total, latency;          // candidates of the plan (n is the length of the launch list of a kernel); latency mode (256 threads per class-1 candidate)
    ImageRef img[SDSM_MAX_IMAGES];
    int32_t k, R, subsample, zcap;     // PSF size, radius k/2, grid spacing, bound on the entries of one pixel's row of G~
    int32_t zcap_run, zshift;          // bound on the entries of a run (union of <= 4 rows); sort class of a run = (entries + (1 << zshift) - 1) >> zshift
    int32_t no_deform;                 // smooth_amount == inf
    int32_t no_trivial_rule;           // sdsm_dsm_config.flags bit 0: solve even a region with a single positive pixel (cvxprog called directly, c2freganal.py:58-79)
    int32_t init_elliptical, max_iters;
    int32_t k1_pixmax;                 // regions with more pixels are solved by the 512-thread classes (layout_plan: a fixed bound in latency mode, relative to the plan's pixels in throughput mode)
    double scale, epsilon, alpha;
    double reg_unit;                   // alpha * sqrt(epsilon): the regulariser's value per grid point at xi = 0 (dsm.py:325-326), computed on the host
    float hess_thr;                    // Hessian ignores row entries < hess_thr * row maximum (solver approximation)
    int32_t boost_pixels;              // throughput mode: regions with more pixels run their passes over the pixels at a raised issue priority (the long chains of a launch; layout_plan)
    int32_t rows_mcap, pad2;           // largest bound on M among the regions whose rows sdsm_k_setup_rows builds (sizes its LDS tables)
    const CandDesc *cand;
    CandState *state;
    const int32_t *fp_labels;
    const int32_t *order;              // workgroup -> candidate (largest first)
    // Packed crop, by RUN: the region pixels of one image row inside one aligned 4-column cell (1 .. 4 pixels) are one run; they
    // share u, the list of grid points around them and (to a large extent) the leading entries of their rows, so a lane of the
    // solve kernel that owns a run adds ONE fixed-point sum per gradient / Hessian entry for up to four pixels.  Final order: a
    // low-discrepancy scatter of the raster order of the runs, stably sorted by their number of G~ entries, longest first.
    double *crop_y;                    // 4 doubles per run (absent pixels 0): 8 B / pixel
    uint32_t *crop_rc;                 // (row << 16) | first column of the cell, image coordinates
    uint32_t *run_meta;                // entries | leading entries << 12 | present pixels (4 bits) << 24
    uint32_t *run_q0;                  // setup only, raster order of the runs: raster rank of the run's first pixel | present pixels << 28
    uint32_t *run_aux;                 // setup only: [chunk of 64 positions] entries of the chunk's first run (padding target of its rows)
    double *tmp_y;                     // setup only, per pixel: the crop in scan order (low-discrepancy scatter of the raster rank)
    uint32_t *tmp_rc;                  // setup only, per pixel
    uint32_t *crop_cc;                 // setup only, per pixel: compressed coordinates (scan order)
    uint32_t *dist;                    // setup only, per pixel: chessboard distance to the nearest grid point; then, per run (scan order of the runs): entries
    uint32_t *inv;                     // setup only, per run (scan order of the runs): final position
    uint32_t *grid_rc;                 // sorted grid points, compressed coordinates
    // G~ of a run: entry j of run position p is element j * NR + p of the candidate's block: one 16-byte load (the weights of the
    // run's four pixels for that grid point, float32 as the reference builds them, 0 where the point is outside a pixel's window)
    // and one 4-byte load (column index | pixels for which the entry is a leading one << 16) per lane; consecutive lanes read
    // consecutive addresses.  Leading entries first, in ascending column order.
    float *ell_w;
    uint32_t *ell_im;
    // Envelope of the solver's Hessian, unknowns ordered xi_0 .. xi_{M-1}, theta_0 .. theta_5: row a of the xi block
    // stores columns env_fst[a] .. a (env_fst: smallest column any pixel couples a with, made non-decreasing in a and a
    // multiple of SDSM_PANEL), entry (a, b) at env_rb[a] + b; the 6 theta rows are dense and follow.  Indexed by xi_off + a.
    int32_t *env_fst;
    int32_t *env_rb;
    const float *psf;
    double *wide_pool;                 // sync words and all-reduce buffers of the workgroup groups (CandDesc.wide_off)
    int32_t *wide_ticket;              // [0] / [1] (the groups of class-2 / class-2b layout): next entry of the group launch list (members are claimed in the order in which workgroups START, see sdsm_solve.hip)
    long long wide_timeout;            // ticks of the 100 MHz wall clock a group member waits for its partners before the group is given up
    int32_t *cls_count;                // [l]: next entry of launch list l that a resident workgroup of a class beyond 1 takes (sdsm_k_solve; zeroed before every launch)
    double *hglob;                     // Hessian pool of the global-memory class (envelope too large for LDS), CandDesc.hglob_off
    long long *prof;                   // diagnostic build only (-DSDSM_PROFILE): 16 cycle counters per candidate (solve kernel)
    long long *prof2;                  // diagnostic build only: 8 cycle counters per candidate (setup kernel), behind the 16 n solve counters
    const double *x0;                  // null, or the caller's starting points of the DSM solves (callable dsm/init, objects.py:385-386): candidate i's theta[6] (full-image-normalised) + xi[M] at 6 i + xi_off(i), as sdsm_batch_eval takes them; only read if init_elliptical == 0
};

// Solve class of a candidate once its setup is complete: the FIRST class whose limits (6 + M <= NMAX, Hessian envelope <= EMAX
// doubles, region <= k1_pixmax pixels for the 256-thread classes) it meets.  Results do not depend on the class.
enum { SDSM_CLS_NONE = -1, SDSM_CLS_1 = 0, SDSM_CLS_1B = 1, SDSM_CLS_2 = 2, SDSM_CLS_2B = 3, SDSM_CLS_3 = 4, SDSM_CLS_WIDE = 5, SDSM_CLS_WIDE2B = 6 };
__host__ __device__ __forceinline__ int sdsm_solve_class(int status, int M, int env_size, int N, int wide_g, int k1_pixmax)
{
    if (status != ST_OK) return SDSM_CLS_NONE;
    int Mfull = M;
    if (6 + Mfull > SDSM_MAX_N_GLOBAL) Mfull = 0;                            // elliptical result only (flagged unsupported)
    const int nfull = 6 + Mfull, efull = Mfull > 0 ? env_size : 21;
    if (wide_g > 0 && nfull <= SDSM_MAX_N_SOLVE && efull <= SDSM_K2_EMAX) return SDSM_CLS_WIDE;
    if (wide_g > 0 && nfull <= SDSM_K2B_NMAX && efull <= SDSM_K2B_EMAX) return SDSM_CLS_WIDE2B;      // groups with the LDS layout of class 2b
    if (nfull <= SDSM_K1_NMAX && efull <= SDSM_K1_EMAX && N <= k1_pixmax) return SDSM_CLS_1;
    if (nfull <= SDSM_K1B_NMAX && efull <= SDSM_K1B_EMAX && N <= k1_pixmax) return SDSM_CLS_1B;
    if (nfull <= SDSM_MAX_N_SOLVE && efull <= SDSM_K2_EMAX) return SDSM_CLS_2;
    if (nfull <= SDSM_K2B_NMAX && efull <= SDSM_K2B_EMAX) return SDSM_CLS_2B;
    return SDSM_CLS_3;
}

// ---------------------------------------------------------------------------------------------
// workgroup primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the workgroup, result broadcast to every thread.  `scratch` holds >= SDSM_WAVES doubles.
template <int WAVES = SDSM_WAVES>
__device__ __forceinline__ double block_sum(double v, double *scratch)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int i = 0; i < WAVES; i++) r += scratch[i];
    return r;
}

// Sums K values over the workgroup.  Within a wavefront a reduce-scatter butterfly: at each of the first log2(KP) steps a
// lane keeps one half of its values and sends the other half to its partner (KP/2 + KP/4 + ... + 1 shuffles instead of
// K per step), so after the 6 steps lane l holds the wavefront total of value l >> (6 - log2 KP).  One LDS hop across the
// wavefronts; totals are then read back with sum_scatter_total (any thread, any index).  scratch: WAVES * KP doubles.
template <int K> struct SdsmPow2 { static constexpr int v = K <= 1 ? 1 : 2 * SdsmPow2<(K + 1) / 2>::v; };
template <int KP> struct SdsmLog2 { static constexpr int v = KP <= 1 ? 0 : 1 + SdsmLog2<KP / 2>::v; };

template <int K, int WAVES = SDSM_WAVES>
__device__ __forceinline__ void block_sum_scatter(const double (&v)[K], double *scratch)
{
    constexpr int KP = SdsmPow2<K>::v;
    static_assert(KP <= 64, "at most 64 values");
    constexpr int SH = 6 - SdsmLog2<KP>::v;
    double t[KP];
#pragma unroll
    for (int i = 0; i < KP; i++) t[i] = i < K ? v[i] : 0.0;
    const int lane = threadIdx.x & 63;
    int c = KP;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        if (c > 1) {
            const bool bit = (lane & o) != 0;
            const int h = c / 2;
#pragma unroll
            for (int i = 0; i < KP / 2; i++) {
                if (i < h) {
                    const double lo = t[i], hi = t[i + h];
                    const double send = bit ? lo : hi, keep = bit ? hi : lo;
                    t[i] = keep + __shfl_xor(send, o);
                }
            }
            c = h;
        } else t[0] += __shfl_xor(t[0], o);
    }
    __syncthreads();                                   // scratch may still be read by the previous user
    if ((lane & ((1 << SH) - 1)) == 0) scratch[(threadIdx.x >> 6) * KP + (lane >> SH)] = t[0];
    __syncthreads();
}

template <int K, int WAVES = SDSM_WAVES>
__device__ __forceinline__ double sum_scatter_total(const double *scratch, int k)
{
    constexpr int KP = SdsmPow2<K>::v;
    double r = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) r += scratch[w * KP + k];
    return r;
}

// Same, every thread gets every total.
template <int K, int WAVES = SDSM_WAVES>
__device__ __forceinline__ void block_sum_vec(double (&v)[K], double *scratch)
{
    block_sum_scatter<K, WAVES>(v, scratch);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = sum_scatter_total<K, WAVES>(scratch, k);
}

template <int WAVES = SDSM_WAVES>
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, unsigned long long *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = scratch[0];
#pragma unroll
    for (int i = 1; i < WAVES; i++) r = scratch[i] < r ? scratch[i] : r;
    return r;
}

template <int WAVES = SDSM_WAVES>
__device__ __forceinline__ int block_min_i32(int v, int *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = scratch[0];
#pragma unroll
    for (int i = 1; i < WAVES; i++) r = scratch[i] < r ? scratch[i] : r;
    return r;
}

// The same for FOUR consecutive chunks of the workgroup's size at once (element k of thread t is item k * 64 * WAVES + t): one pair of
// barriers per four chunks.  pos[k] = number of set flags before the thread's item of chunk k; *total = all set flags.
template <int WAVES = SDSM_WAVES>
__device__ __forceinline__ void block_excl_count4(const bool (&flag)[4], int *scratch /* 4 * WAVES ints */, int (&pos)[4], int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m[4];
    int within[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { m[k] = __ballot(flag[k]); within[k] = __popcll(m[k] & ((1ull << lane) - 1ull)); }
    __syncthreads();
    if (lane < 4) {
        unsigned long long mk = m[0];
#pragma unroll
        for (int k = 1; k < 4; k++) mk = lane == k ? m[k] : mk;
        scratch[lane * WAVES + wave] = __popcll(mk);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int before = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) { const int c = scratch[k * WAVES + i]; if (i < wave) before += c; tot += c; }
        pos[k] = run + before + within[k];
        run += tot;
    }
    *total = run;
}

// Exclusive prefix sum of small unsigned values over the workgroup in thread order (Hillis-Steele inside a wavefront, one LDS
// hop across the wavefronts); *total = sum over the workgroup.  scratch: WAVES unsigned.
template <int WAVES = SDSM_WAVES>
__device__ __forceinline__ unsigned block_excl_scan_u32(unsigned v, unsigned *scratch, unsigned *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();                                   // scratch may still be read by the previous user
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    unsigned before = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < WAVES; i++) { const unsigned c = scratch[i]; if (i < wave) before += c; tot += c; }
    *total = tot;
    return before + inc - v;
}

// Exclusive prefix count of `flag` over the workgroup in thread order; *total = number of set flags.
__device__ __forceinline__ int block_excl_count(bool flag, int *scratch /* SDSM_WAVES ints */, int *total)
{
    unsigned long long m = __ballot(flag);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int within = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) scratch[wave] = __popcll(m);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SDSM_WAVES; i++) { if (i < wave) before += scratch[i]; tot += scratch[i]; }
    *total = tot;
    return before + within;
}
