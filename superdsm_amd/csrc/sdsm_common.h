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
#define SDSM_K1B_NMAX 256          // solve class 1b: 6 + M <= 256 and envelope <= 6144 doubles: 256 threads, LDS ~ 68 KB, TWO workgroups per compute unit
#define SDSM_K1B_EMAX 6144         //   (between class 1 and class 2, whose 157 KB leave one workgroup per compute unit: regions of ~5-15 k pixels, M ~ 100-250)
#define SDSM_K2B_NMAX 512          // solve class 2b: 6 + M <= 512 and envelope <= 15900 doubles: the LDS that class 2 spends on vectors of 1024 unknowns
#define SDSM_K2B_EMAX 15900        //   holds a larger envelope instead (~ 160 KB): keeps most of what class 2 cannot hold out of the global-memory class
// Very large regions are solved by a GROUP of workgroups (2 .. 8, one per 8192 pixels): each takes a slice of the pixels in
// every pass and the partial sums / gradient / Hessian are all-reduced through global memory (sdsm_solve.hip, WIDE).
#define SDSM_WIDE_MIN_PIXELS 12288
#define SDSM_WIDE_SLICE 8192
#define SDSM_WIDE_MAX_G 8
#define SDSM_WIDE_SYNC 16           // doubles reserved for the group's counters at the start of its pool block
#define SDSM_WIDE_PBUF (SDSM_K2_EMAX + SDSM_MAX_N_SOLVE + 64)   // doubles one workgroup publishes per all-reduce
#define SDSM_WIDE_PIXELS 3072       // latency mode: larger regions go to class 2 (512 threads per candidate; a batch is as slow as its slowest candidate)
#define SDSM_K2_EMAX 11000         // solve class 2: 6 + M <= SDSM_MAX_N_SOLVE and envelope <= 11000 doubles (LDS ~ 157 KB)
#define SDSM_K1_DENSE_N 70         // 6 + M <= 70: even a dense triangle fits class 1
#define SDSM_ENV_DENSE_N 146       // 6 + M <= 146: even a dense triangle fits class 2
#define SDSM_ELL_GROUPS_REG 7       // groups of 4 G~ row entries the solve kernel keeps in registers (rows of <= 28 entries)
#define SDSM_MAX_ELL_GROUPS 256    // zcap <= 1024 entries per row of G~ (a solvable candidate has M <= 1018 columns)
#ifndef SDSM_PANEL
#define SDSM_PANEL 4               // columns per panel (8 measured slower: 195 k vs 200 k solves/s); of the envelope Cholesky; first stored columns are multiples of it
#endif
#define SDSM_MAX_N_SOLVE 1024      // 6 + M handled by the largest solve class (Hessian + factor in global memory)

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
    int64_t crop_off;   // first pixel of the candidate's packed crop (crop_y / crop_rc / crop_cc / ell_meta)
    int64_t ell_off;    // first entry of its ELL block (N * zcap4 entries: groups of 4 slots, group-major)
    int64_t mask_off;   // first uint32 word of its bit-packed region-bbox mask
    int64_t xi_off;     // first entry of its grid / xi block (Mcap entries)
    int32_t N;          // region pixels (sum of the atoms' valid areas)
    int32_t r0, c0, h, w;   // region bounding box (union of the atoms' valid extents)
    int32_t fp_off, fp_len; // footprint labels
    int32_t Mcap;       // upper bound of M
    uint32_t perm_inv;  // crop position of the pixel with raster rank i is (i * perm_inv) mod N (low-discrepancy scatter)
    int32_t wide_g;     // > 0: solved by a group of this many workgroups (regions of more than SDSM_WIDE_MIN_PIXELS pixels)
    int64_t hglob_off;  // first double of its block in the global Hessian pool (a dense triangle of 6 + min(Mcap, 1018) unknowns; only if 6 + Mcap > SDSM_ENV_DENSE_N: the envelope may not fit LDS), else -1
    int64_t wide_off;   // first double of the group's block in the wide pool: SDSM_WIDE_SYNC + 2 * wide_g * SDSM_WIDE_PBUF doubles; else -1
    int32_t image;      // index into BatchParams.img (plans over several images, sdsm_plan_create_multi)
    int32_t pad0;
};

// Written by the setup kernel.
struct CandState {
    int32_t M;          // number of grid points (columns of G~)
    int32_t status;     // ST_*
    int32_t hc, wc;     // shape of the mask after deleting empty rows / columns (dsm.py:185-186)
    int32_t npos;       // region pixels with y > 0
    int32_t zmax;       // largest number of non-zeros in a row of G~
    unsigned long long sum_r, sum_c, sum_rr, sum_cc;   // moments of the y > 0 pixels (image coordinates)
    int32_t hzmax;      // largest number of 'significant' entries (>= hess_thr * row maximum) in a row of G~
    int32_t env_size;   // doubles of the solver's Hessian in envelope storage (see env_fst / env_rb)
    int32_t nneg;       // region pixels with y < 0
    int32_t yexp;       // |y| < 2^yexp for every region pixel (clamped to +-400): scale of the solver's fixed-point sums
    int32_t gcount[8];  // gcount[j] = crop positions whose row has more than 4 j entries (positions are sorted by that)
};
static_assert(sizeof(CandState) == 104 && SDSM_ELL_GROUPS_REG <= 8, "CandState layout");
static_assert(sizeof(CandDesc) == 96, "CandDesc layout");

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
    ImageRef img[SDSM_MAX_IMAGES];
    int32_t k, R, subsample, zcap;     // PSF size, radius k/2, grid spacing, ELL slots per pixel (a multiple of 4)
    int32_t no_deform;                 // smooth_amount == inf
    int32_t no_trivial_rule;           // sdsm_dsm_config.flags bit 0: solve even a region with a single positive pixel (cvxprog called directly, c2freganal.py:58-79)
    int32_t init_elliptical, max_iters;
    int32_t k1_pixmax;                 // regions with more pixels are solved by class 2 (INT_MAX: throughput mode)
    double scale, epsilon, alpha;
    float hess_thr; int32_t pad1;      // Hessian ignores row entries < hess_thr * row maximum (solver approximation)
    const CandDesc *cand;
    CandState *state;
    const int32_t *fp_labels;
    const int32_t *order;              // workgroup -> candidate (largest first)
    double *crop_y;                    // 8 B / pixel, final crop order (rows with the most G~ entries first)
    uint32_t *crop_rc;                 // (row << 16) | col, image coordinates: 4 B / pixel, final crop order
    double *tmp_y;                     // setup only: the crop in scan order (low-discrepancy scatter of the raster rank)
    uint32_t *tmp_rc;                  // setup only
    uint32_t *crop_cc;                 // compressed coordinates (setup only, scan order)
    uint32_t *dist;                    // setup only: chessboard distance to the nearest grid point, then the final crop position
    uint32_t *inv;                     // setup only: scan-order index of a final crop position
    uint32_t *grid_rc;                 // sorted grid points, compressed coordinates
    // G~ rows: entry s of the row of crop position p is element ((s / 4) * N + p) * 4 + s % 4 of the candidate's block:
    // one 16-byte (weights) + one 8-byte (column indices) load per lane fetches 4 entries, consecutive lanes read
    // consecutive addresses
    float *ell_w;
    uint16_t *ell_idx;
    uint32_t *ell_meta;                // entries of the row | entries used by the solver's approximate Hessian << 16
    // Envelope of the solver's Hessian, unknowns ordered xi_0 .. xi_{M-1}, theta_0 .. theta_5: row a of the xi block
    // stores columns env_fst[a] .. a (env_fst: smallest column any pixel couples a with, made non-decreasing in a and a
    // multiple of SDSM_PANEL), entry (a, b) at env_rb[a] + b; the 6 theta rows are dense and follow.  Indexed by xi_off + a.
    int32_t *env_fst;
    int32_t *env_rb;
    const float *psf;
    double *wide_pool;                 // sync words and all-reduce buffers of the workgroup groups (CandDesc.wide_off)
    int32_t *wide_ticket;              // [0]: next entry of the group launch list (members are claimed in the order in which workgroups START, see sdsm_solve.hip)
    long long wide_timeout;            // ticks of the 100 MHz wall clock a group member waits for its partners before the group is given up
    double *hglob;                     // Hessian pool of the global-memory class (envelope too large for LDS), CandDesc.hglob_off
    long long *prof;                   // diagnostic build only (-DSDSM_PROFILE): 16 cycle counters per candidate (solve kernel)
    long long *prof2;                  // diagnostic build only: 8 cycle counters per candidate (setup kernel), behind the 16 n solve counters
};

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
