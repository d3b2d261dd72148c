// Per-image kernels: bounded exact Euclidean distance tests, per-atom statistics, and the Gaussian
// scale-space preprocessing (separable stencils staged through LDS).
//
// Reference behaviour restated here (never its code):
//   Preprocessing.process                 superdsm/preprocess.py:39-68  (SciPy gaussian_filter: mode='reflect',
//                                         truncate=4; symmetric correlate1d association order)
//   EDT(y <= 0) <= background_margin      superdsm/objects.py:126-127   (candidate independent)
#include "sdsm_common.h"
#include <vector>

#pragma clang fp contract(off)   // keep SciPy's / numpy's unfused association (bit-level parity of y)

extern __shared__ __align__(16) unsigned char sdsm_dyn_lds[];

namespace {

// ---- target masks --------------------------------------------------------------------------------
// mode 0: target = y > 0                (EDT(y <= 0): distance to the nearest pixel with y > 0)
// mode 1: target = g > scal[2]          (clip_area, preprocess.py:55)
__global__ void k_target(const double *__restrict__ src, size_t n, int mode, const double *__restrict__ scal,
                         uint8_t *__restrict__ target, int *__restrict__ any)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool t = false;
    if (i < n) {
        double thr = mode == 0 ? 0.0 : scal[2];
        t = src[i] > thr;
        target[i] = t;
    }
    if (__syncthreads_or(t) && threadIdx.x == 0) *any = 1;       // every writer stores the same value: no atomic needed
}

// horizontal distance to the nearest target in the same row, capped at radius + 1
__global__ void k_hdist(const uint8_t *__restrict__ target, int H, int W, int radius, uint16_t *__restrict__ hd)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const uint8_t *row = target + (size_t)r * W;
    int d = radius + 1;
    for (int k = 0; k <= radius; k++) {
        if ((c - k >= 0 && row[c - k]) || (c + k < W && row[c + k])) { d = k; break; }
    }
    hd[(size_t)r * W + c] = (uint16_t)d;
}

// squared Euclidean distance to the nearest target if it is <= radius^2 (exact), else "far".
// out_valid != NULL: valid = y_mask & (d2 <= m2)            (image prepare)
// out_t     != NULL: t = max(sigma2 - sqrt(d2), 0)           (preprocess.py:56-57)
__global__ void k_vdist(const uint16_t *__restrict__ hd, int H, int W, int radius, const int *__restrict__ any,
                        double m2, const uint8_t *__restrict__ y_mask, uint8_t *__restrict__ out_valid,
                        double sigma2, double *__restrict__ out_t)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    long long best = -1;
    if (!*any) best = (long long)(r + 1) * (r + 1) + (long long)c * c;   // SciPy's behaviour without any background pixel
    else {
        int lo = r - radius < 0 ? 0 : r - radius, hi = r + radius >= H ? H - 1 : r + radius;
        for (int rr = lo; rr <= hi; rr++) {
            int h = hd[(size_t)rr * W + c];
            if (h > radius) continue;
            long long d2 = (long long)(rr - r) * (rr - r) + (long long)h * h;
            if (best < 0 || d2 < best) best = d2;
        }
    }
    size_t p = (size_t)r * W + c;
    if (out_valid) out_valid[p] = (best >= 0 && (double)best <= m2) && (y_mask ? y_mask[p] != 0 : true);
    if (out_t) {
        double t = best < 0 ? 0.0 : sigma2 - sqrt((double)best);
        out_t[p] = t < 0 ? 0.0 : t;
    }
}

// ---- the same two passes for radius <= 63, the ones that normally run -------------------------------------------------
// Row pass with the target test fused in (k_target + k_hdist): a wavefront turns the target flags of 64 consecutive columns into
// ONE 64-bit word (ballot); a workgroup holds the words of its 256 columns and of 64 columns on either side in LDS, and the
// distance of a column to the nearest target of its row is a count of leading / trailing zeros of two shifted words -- no loop
// over the radius, no per-pixel byte reads.
// mode 0: target = src > 0; mode 1: target = src > scal[2] (see k_target).
__global__ __launch_bounds__(256) void k_hdist_bits(const double *__restrict__ src, int H, int W, int mode, const double *__restrict__ scal,
                                                    int radius, uint16_t *__restrict__ hd, int *__restrict__ any)
{
    __shared__ unsigned long long words[6];                                // columns c0 - 64 .. c0 + 319
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.y, c0 = blockIdx.x * 256;
    const double thr = mode == 0 ? 0.0 : scal[2];
    const double *row = src + (size_t)r * W;
    const int c = c0 + tid;
    const bool t = c < W && row[c] > thr;
    const unsigned long long own = __ballot(t);
    if (lane == 0) words[1 + wave] = own;
    if (wave < 2) {                                                         // the halo words: wave 0 the left one, wave 1 the right one
        const int ch = wave == 0 ? c0 - 64 + lane : c0 + 256 + lane;
        const unsigned long long hw = __ballot(ch >= 0 && ch < W && row[ch] > thr);
        if (lane == 0) words[wave == 0 ? 0 : 5] = hw;
    }
    if (__syncthreads_or(t) && tid == 0) *any = 1;                         // (every writer stores the same value)
    if (c >= W) return;
    const unsigned long long w0 = words[wave], w1 = words[1 + wave], w2 = words[2 + wave];
    // bit 63 of L = column c, bit 62 = column c - 1, ...; bit 0 of Rt = column c, bit 1 = column c + 1, ...
    const unsigned long long L = (w1 << (63 - lane)) | (lane == 63 ? 0ull : w0 >> (lane + 1));
    const unsigned long long Rt = (w1 >> lane) | (lane == 0 ? 0ull : w2 << (64 - lane));
    const int dl = L ? __clzll((long long)L) : 64 + 64, dr = Rt ? __ffsll((long long)Rt) - 1 : 64 + 64;
    int d = dl < dr ? dl : dr;
    d = d > radius ? radius + 1 : d;
    hd[(size_t)r * W + c] = (uint16_t)d;
}

// Column pass (k_vdist) on a tile of 64 columns x 64 rows with its halo rows of horizontal distances staged in LDS.
#define VT_ROWS 64
__global__ __launch_bounds__(256) void k_vdist_t(const uint16_t *__restrict__ hd, int H, int W, int radius, const int *__restrict__ any,
                                                 double m2, const uint8_t *__restrict__ y_mask, uint8_t *__restrict__ out_valid,
                                                 double sigma2, double *__restrict__ out_t)
{
    uint16_t *tile = (uint16_t *)sdsm_dyn_lds;                             // (VT_ROWS + 2 radius) x 64
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int c = blockIdx.x * 64 + tx, r0 = blockIdx.y * VT_ROWS;
    const int cs = c < W ? c : W - 1;
    const int rows = VT_ROWS + 2 * radius;
    for (int i = ty; i < rows; i += 4) {
        const int rr = r0 - radius + i;
        tile[i * 64 + tx] = rr >= 0 && rr < H ? hd[(size_t)rr * W + cs] : (uint16_t)(radius + 1);     // outside the image: no target
    }
    __syncthreads();
    if (c >= W) return;
    const bool has_any = *any != 0;
    for (int k = 0; k < VT_ROWS / 4; k++) {
        const int rl = ty + 4 * k, r = r0 + rl;
        if (r >= H) break;
        long long best = -1;
        if (!has_any) best = (long long)(r + 1) * (r + 1) + (long long)c * c;   // SciPy's behaviour without any background pixel
        else {
            int b = 0x7fffffff;
            for (int i = 0; i <= 2 * radius; i++) {                         // tile row rl + i is image row r - radius + i
                const int h = tile[(rl + i) * 64 + tx];
                const int dr = i - radius;
                const int d2 = dr * dr + h * h;
                b = h <= radius && d2 < b ? d2 : b;
            }
            if (b != 0x7fffffff) best = b;
        }
        const size_t p = (size_t)r * W + c;
        if (out_valid) out_valid[p] = (best >= 0 && (double)best <= m2) && (y_mask ? y_mask[p] != 0 : true);
        if (out_t) {
            double t = best < 0 ? 0.0 : sigma2 - sqrt((double)best);
            out_t[p] = t < 0 ? 0.0 : t;
        }
    }
}

// target test + the two distance passes; radius <= 63 (a column sees 63 neighbours on either side in two shifted words) runs the bit / tile kernels
static void launch_bounded_edt(const double *src, int H, int W, int mode, const double *scal, int radius, uint8_t *target, uint16_t *hd, int *any,
                               double m2, const uint8_t *y_mask, uint8_t *out_valid, double sigma2, double *out_t, hipStream_t stream)
{
    const size_t n = (size_t)H * W;
    dim3 b(256), g2((W + 255) / 256, H);
    if (radius <= 63) {
        hipLaunchKernelGGL(k_hdist_bits, g2, b, 0, stream, src, H, W, mode, scal, radius, hd, any);
        hipLaunchKernelGGL(k_vdist_t, dim3((W + 63) / 64, (H + VT_ROWS - 1) / VT_ROWS), b, (size_t)(VT_ROWS + 2 * radius) * 64 * 2, stream,
                           (const uint16_t *)hd, H, W, radius, (const int *)any, m2, y_mask, out_valid, sigma2, out_t);
    } else {
        hipLaunchKernelGGL(k_target, dim3((n + 255) / 256), b, 0, stream, src, n, mode, scal, target, any);
        hipLaunchKernelGGL(k_hdist, g2, b, 0, stream, (const uint8_t *)target, H, W, radius, hd);
        hipLaunchKernelGGL(k_vdist, g2, b, 0, stream, (const uint16_t *)hd, H, W, radius, (const int *)any, m2, y_mask, out_valid, sigma2, out_t);
    }
}

__global__ void k_stats_init(int32_t *stats, int n_atoms)
{
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > n_atoms) return;
    int32_t *s = stats + (size_t)l * SDSM_ATOM_STATS_STRIDE;
    s[0] = 0; s[1] = 0x7fffffff; s[2] = -1; s[3] = 0x7fffffff; s[4] = -1; s[5] = 0;
}

// Per-atom area and extents.  A wavefront covers 64 consecutive columns of one row, which mostly belong to one or two
// atoms: the lanes of each label present are aggregated with ballots and ONE lane issues the 5 atomics for them.
__global__ void k_stats(const int32_t *__restrict__ atoms, const uint8_t *__restrict__ valid, int H, int W, int n_atoms, int32_t *stats)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    const int lane = threadIdx.x & 63;
    int l = 0;
    if (c < W) {
        const size_t p = (size_t)r * W + c;
        l = atoms[p];
        if (l < 1 || l > n_atoms || !valid[p]) l = 0;
    }
    unsigned long long todo = __ballot(l != 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int l0 = __shfl(l, src);
        const unsigned long long same = __ballot(l == l0);
        if (lane == src) {
            int32_t *s = stats + (size_t)l0 * SDSM_ATOM_STATS_STRIDE;
            const int c0 = c - lane;                                       // column of lane 0
            atomicAdd(&s[0], __popcll(same));
            atomicMin(&s[1], r); atomicMax(&s[2], r);
            atomicMin(&s[3], c0 + __ffsll((long long)same) - 1); atomicMax(&s[4], c0 + 63 - __clzll((long long)same));
        }
        todo &= ~same;
    }
}

// ---- reductions for np.std (preprocess.py:52) ------------------------------------------------------
// mode 0: sum(x), mode 1: sum((x - scal[0])^2).  Deterministic two-level tree (not numpy's pairwise order:
// the mean / std may differ from numpy's in the last ulp).
__global__ void k_partial(const double *__restrict__ x, size_t n, int mode, const double *__restrict__ scal, double *__restrict__ partial)
{
    __shared__ double red[SDSM_WAVES];
    double mean = mode ? scal[0] : 0.0;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = x[i];
        if (mode) { v = v - mean; v = v * v; }
        s += v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void k_final(const double *__restrict__ partial, int np, double n, int mode, double offset_clip, double *__restrict__ scal)
{
    __shared__ double red[SDSM_WAVES];
    double s = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        if (mode == 0) scal[0] = s / n;                                  // mean
        else { scal[1] = sqrt(s / n); scal[2] = offset_clip * scal[1]; }  // std, clip_abs
    }
}

__global__ void k_clip(const double *__restrict__ g, size_t n, const double *__restrict__ scal, double *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double hi = scal[2], v = g[i];
    out[i] = v < 0 ? 0 : (v > hi ? hi : v);                               // g_raw.clip(0, clip_abs)
}

// ---- separable Gaussian, reflect boundary, SciPy's symmetric association order -----------------------
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if ((unsigned)i < (unsigned)n) return i;                 // inside the image: no integer division (only tiles at the border pay for it)
    int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

// axis 1: one workgroup = 256 consecutive pixels of one row (+ halo) staged in LDS
__global__ __launch_bounds__(256) void k_gauss_rows(const double *__restrict__ in, int H, int W, const double *__restrict__ w, int R, double *__restrict__ out)
{
    double *lds = (double *)sdsm_dyn_lds;
    const int r = blockIdx.y, c0 = blockIdx.x * 256, tid = threadIdx.x;
    const double *row = in + (size_t)r * W;
    for (int i = tid; i < 256 + 2 * R; i += 256) lds[i] = row[reflect_idx(c0 - R + i, W)];
    __syncthreads();
    int c = c0 + tid;
    if (c >= W) return;
    const double *ctr = lds + tid + R;
    double acc = ctr[0] * w[R];
    for (int j = R; j >= 1; j--) acc += (ctr[-j] + ctr[j]) * w[R - j];
    out[(size_t)r * W + c] = acc;
}

// axis 0: one workgroup = 32 columns x 64 rows (+ halo rows) staged in LDS; thread = (column, 8 rows)
#define GC_COLS_MAX 32
#define GC_ROWS 64
template <int GC_COLS>
__global__ __launch_bounds__(256) void k_gauss_cols(const double *__restrict__ in, int H, int W, const double *__restrict__ w, int R, double *__restrict__ out)
{
    double *lds = (double *)sdsm_dyn_lds;
    const int c0 = blockIdx.x * GC_COLS, r0 = blockIdx.y * GC_ROWS, tid = threadIdx.x;
    constexpr int LR = 256 / GC_COLS;                          // 32 x 8 (or 8 x 32 for very long filters: the tile must fit the LDS)
    const int lc = tid % GC_COLS, lr = tid / GC_COLS;
    const int rows = GC_ROWS + 2 * R;
    const int c = c0 + lc;
    const int cs = c < W ? c : W - 1;
    for (int i = lr; i < rows; i += LR) lds[i * GC_COLS + lc] = in[(size_t)reflect_idx(r0 - R + i, H) * W + cs];
    __syncthreads();
    if (c >= W) return;
    for (int k = 0; k < GC_ROWS / LR; k++) {
        int rl = lr + LR * k, r = r0 + rl;
        if (r >= H) break;
        const double *ctr = lds + (rl + R) * GC_COLS + lc;
        double acc = ctr[0] * w[R];
        for (int j = R; j >= 1; j--) acc += (ctr[-j * GC_COLS] + ctr[j * GC_COLS]) * w[R - j];
        out[(size_t)r * W + c] = acc;
    }
}

// ---- register-tiled versions (the ones that run; the two kernels above remain for filters too long for their LDS tiles) -----
// A thread computes GT_K consecutive outputs ALONG the filtered axis and keeps the two windows of inputs that tap j needs --
// x[p - j .. p - j + K - 1] and x[p + j .. p + j + K - 1] -- in registers: going from tap j to tap j - 1 slides each window by
// one element, i.e. ONE new LDS read per window instead of K (the loop is unrolled by K so that the slide is a renaming of
// registers, not a copy).  With one output per thread the filters were bound by LDS bandwidth (2 reads of 8 bytes per tap pair
// and output: 2.8 KB per pixel of the three filters of the preprocessing); now the FP64 pipes are.  Every output is still
// accumulated in SciPy's order (centre, then the pairs from the outermost inwards), bit for bit as before.
#define GT_K 8
struct CombineArgs {               // epilogue of the last pass of the preprocessing (k_combine fused in); off == nullptr: none
    const double *off, *offc, *tbuf, *scal;
    const int *any;
    double sigma2;
    int use_clip, lower_clip_mean;
};

__device__ __forceinline__ double clip_on_load(double v, const double *__restrict__ clip_scal)
{
    if (!clip_scal) return v;
    const double hi = clip_scal[2];
    return v < 0 ? 0 : (v > hi ? hi : v);                                  // g_raw.clip(0, clip_abs), preprocess.py:53
}

template <class LoadAt>
__device__ __forceinline__ void taps_sliding(double (&acc)[GT_K], const double *__restrict__ w, int R, LoadAt x /* x(i): input at offset i from the thread's first output, -R <= i <= GT_K - 1 + R */)
{
    double U[GT_K], D[GT_K];
#pragma unroll
    for (int k = 0; k < GT_K; k++) { acc[k] = x(k) * w[R]; U[k] = x(k - R); D[k] = x(k + R); }
    // blocks of GT_K taps: the weights of a block (uniform: scalar loads) and the 2 x GT_K elements that enter the windows during
    // it are requested together at its start, the taps themselves are then arithmetic only
    for (int j = R; j >= 1; j -= GT_K) {
        const int nb = j < GT_K ? j : GT_K;                                 // taps of this block: j, j - 1, ..., j - nb + 1
        double wj[GT_K], NU[GT_K], ND[GT_K];
#pragma unroll
        for (int s = 0; s < GT_K; s++) {
            const int js = j - s > 1 ? j - s : 1;                           // (clamped: the values of taps beyond the block are not used)
            wj[s] = w[R - js];
            NU[s] = x(GT_K - js);                                           // windows of tap js - 1: x[k - (js - 1)], x[k + (js - 1)]
            ND[s] = x(js - 1);
        }
#pragma unroll
        for (int s = 0; s < GT_K; s++) {
            if (s < nb) {                                                   // (uniform)
#pragma unroll
                for (int k = 0; k < GT_K; k++) acc[k] += (U[(k + s) % GT_K] + D[(k - s + GT_K) % GT_K]) * wj[s];
                U[s % GT_K] = NU[s];
                D[(GT_K - s - 1) % GT_K] = ND[s];
            }
        }
    }
}

// axis 0: a workgroup = TX columns x (TY * GT_K) rows (+ halo rows) in LDS; thread = (column, GT_K consecutive rows)
template <int TX>
__global__ __launch_bounds__(256) void k_gauss_cols_t(const double *__restrict__ in, int H, int W, const double *__restrict__ w, int R,
                                                      const double *__restrict__ clip_scal, double *__restrict__ out)
{
    constexpr int TY = 256 / TX, TR = TY * GT_K;
    double *lds = (double *)sdsm_dyn_lds;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int c = blockIdx.x * TX + tx, r0 = blockIdx.y * TR;
    const int cs = c < W ? c : W - 1;
    for (int i = ty; i < TR + 2 * R + 1; i += TY) lds[i * TX + tx] = clip_on_load(in[(size_t)reflect_idx(r0 - R + i, H) * W + cs], clip_scal);
    __syncthreads();
    const double *ctr = lds + (ty * GT_K + R) * TX + tx;
    double acc[GT_K];
    taps_sliding(acc, w, R, [&](int i) { return ctr[i * TX]; });
    if (c >= W) return;
#pragma unroll
    for (int k = 0; k < GT_K; k++) {
        const int r = r0 + ty * GT_K + k;
        if (r < H) out[(size_t)r * W + c] = acc[k];
    }
}

// axis 1: a workgroup = 8 rows x 256 columns (+ halo columns) in LDS, one double of padding after every 8 (a thread owns 8
// consecutive columns: the lanes of a wavefront then read 9 doubles apart -- different banks); thread = (row, GT_K consecutive columns)
#define GR_ROWS 8
#define GR_COLS (32 * GT_K)
__global__ __launch_bounds__(256) void k_gauss_rows_t(const double *__restrict__ in, int H, int W, const double *__restrict__ w, int R,
                                                      double *__restrict__ out, CombineArgs cmb)
{
    double *lds = (double *)sdsm_dyn_lds;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int c0 = blockIdx.x * GR_COLS, r0 = blockIdx.y * GR_ROWS;
    const int span = GR_COLS + 2 * R + 1, pitch = span + (span >> 3) + 1;
    for (int rr = 0; rr < GR_ROWS; rr++) {
        const int r = r0 + rr < H ? r0 + rr : H - 1;
        const double *row = in + (size_t)r * W;
        double *dst = lds + rr * pitch;
        for (int i = tid; i < span; i += 256) dst[i + (i >> 3)] = row[reflect_idx(c0 - R + i, W)];
    }
    __syncthreads();
    const double *line = lds + ty * pitch;
    const int q0 = tx * GT_K + R;                                           // LDS column of the thread's first output
    double acc[GT_K];
    taps_sliding(acc, w, R, [&](int i) { const int q = q0 + i; return line[q + (q >> 3)]; });
    // results back through the tile (a thread holds 8 consecutive columns: stored directly, the lanes of a wavefront would write
    // 64 bytes apart), then rows of 256 consecutive columns leave with one lane per column
    __syncthreads();
    {
        double *dst = lds + ty * pitch;
#pragma unroll
        for (int k = 0; k < GT_K; k++) { const int q = tx * GT_K + k; dst[q + (q >> 3)] = acc[k]; }
    }
    __syncthreads();
    const int c = c0 + tid;
    if (c >= W) return;
    double tmax = 0, mean = 0;
    if (cmb.off) {
        tmax = *cmb.any ? cmb.sigma2 : (cmb.sigma2 - 1.0 < 0 ? 0.0 : cmb.sigma2 - 1.0);
        if (cmb.lower_clip_mean) mean = cmb.scal[0];
    }
    for (int rr = 0; rr < GR_ROWS; rr++) {
        const int r = r0 + rr;
        if (r >= H) break;
        const size_t p = (size_t)r * W + c;
        double v = lds[rr * pitch + tid + (tid >> 3)];
        if (cmb.off) {                                                      // y = gauss(g, sigma1) - offset_combined (preprocess.py:57-64), as k_combine
            double comb;
            if (cmb.use_clip) {
                double t = cmb.tbuf[p] / tmax;
                t = t * t;
                comb = (1 - t) * cmb.offc[p] + t * cmb.off[p];
            } else comb = cmb.off[p];
            if (cmb.lower_clip_mean) { if (!(comb > mean)) comb = mean; }
            v = v - comb;
        }
        out[p] = v;
    }
}

// y = gauss(g, sigma1) - ((1 - t) * offset_clipped + t * offset_original),  t = (t / tmax)^2   (preprocess.py:57-64)
__global__ void k_combine(const double *__restrict__ g1, const double *__restrict__ off, const double *__restrict__ offc,
                          const double *__restrict__ tbuf, size_t n, int use_clip, int lower_clip_mean,
                          const double *__restrict__ scal, const int *__restrict__ any, double sigma2, double *__restrict__ y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double comb;
    if (use_clip) {
        // max of (sigma2 - d).clip(0): d == 0 on the clipped pixels; without any, SciPy's virtual pixel gives d_min = 1
        double tmax = *any ? sigma2 : (sigma2 - 1.0 < 0 ? 0.0 : sigma2 - 1.0);
        double t = tbuf[i] / tmax;
        t = t * t;
        comb = (1 - t) * offc[i] + t * off[i];
    } else comb = off[i];
    if (lower_clip_mean) { double mean = scal[0]; if (!(comb > mean)) comb = mean; }
    y[i] = g1[i] - comb;
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t sdsm_image_workspace_bytes(int H, int W)
{
    size_t n = (size_t)H * W;
    return align_up(n, 256) /* target */ + align_up(n * 2, 256) /* hd */ + 256 /* flags */;
}

extern "C" hipError_t sdsm_image_prepare_impl(const double *d_y, const uint8_t *d_y_mask, const int32_t *d_atoms, int H, int W,
                                              double margin, int n_atoms, uint8_t *d_valid, int32_t *d_atom_stats,
                                              void *d_ws, hipStream_t stream)
{
    size_t n = (size_t)H * W;
    uint8_t *target = (uint8_t *)d_ws;
    uint16_t *hd = (uint16_t *)((uint8_t *)d_ws + align_up(n, 256));
    int *any = (int *)((uint8_t *)hd + align_up(n * 2, 256));
    hipError_t e = hipMemsetAsync(any, 0, 256, stream);
    if (e != hipSuccess) return e;
    int radius = (int)ceil(margin);
    if (radius < 0) radius = 0;
    dim3 b(256), g2((W + 255) / 256, H);
    launch_bounded_edt(d_y, H, W, 0, nullptr, radius, target, hd, any, margin * margin, d_y_mask, d_valid, 0.0, nullptr, stream);
    hipLaunchKernelGGL(k_stats_init, dim3((n_atoms + 256) / 256), b, 0, stream, d_atom_stats, n_atoms);
    hipLaunchKernelGGL(k_stats, g2, b, 0, stream, d_atoms, d_valid, H, W, n_atoms, d_atom_stats);
    return hipGetLastError();
}

static int gauss_radius(double sigma) { return (int)(4.0 * sigma + 0.5); }

extern "C" size_t sdsm_preprocess_workspace_bytes(int H, int W, double sigma1, double sigma2)
{
    size_t n = (size_t)H * W;
    size_t r1 = gauss_radius(sigma1), r2 = gauss_radius(sigma2);
    return 4 * align_up(n * 8, 256) + align_up(n, 256) + align_up(n * 2, 256) + align_up((2 * r1 + 1 + 2 * r2 + 1) * 8, 256)
           + align_up(1024 * 8, 256) + 512;
}

// numpy pairwise sum (PW_BLOCKSIZE 128), host
static double np_pairwise(const double *a, long n)
{
    if (n < 8) { double r = 0; for (long i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8]; long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2; n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

// scipy.ndimage._filters._gaussian_kernel1d
extern "C" void sdsm_gauss_kernel_host(double sigma, int radius, double *w)
{
    double s2 = sigma * sigma;
    for (int i = 0; i <= 2 * radius; i++) { double x = i - radius; w[i] = exp(-0.5 / s2 * (x * x)); }
    double s = np_pairwise(w, 2 * radius + 1);
    for (int i = 0; i <= 2 * radius; i++) w[i] = w[i] / s;
}

// hipFuncSetAttribute costs microseconds per call: the dynamic LDS limit of a kernel is raised only when a launch needs more than before
static hipError_t need_lds(const void *fn, size_t bytes)
{
    struct Seen { const void *fn; int dev; size_t bytes; };
    static thread_local Seen seen[32];
    static thread_local int nseen = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    for (int i = 0; i < nseen; i++) if (seen[i].fn == fn && seen[i].dev == dev) {
        if (seen[i].bytes >= bytes) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) seen[i].bytes = bytes;
        return e;
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && nseen < 32 && dev >= 0) { seen[nseen].fn = fn; seen[nseen].dev = dev; seen[nseen].bytes = bytes; nseen++; }
    return e;
}

// axis 0 with weights (d_w0, R0) into tmp, then axis 1 with (d_w1, R1) into out: scipy.ndimage's order of the axes.
// clip_scal != nullptr: the input is clipped to [0, clip_scal[2]] as it is read; cmb: epilogue of the second pass (or none).
#define GT_LDS_MAX (64 * 1024)          // tiles of the register-tiled kernels: at least two workgroups per compute unit
static hipError_t separable2d(const double *in, int H, int W, const double *d_w0, int R0, const double *d_w1, int R1, double *tmp, double *out, hipStream_t stream,
                              const double *clip_scal = nullptr, const CombineArgs *cmb = nullptr)
{
    hipError_t e;
    CombineArgs none = {};
    const size_t lds_c32 = (size_t)(8 * GT_K + 2 * R0 + 1) * 32 * 8, lds_c16 = (size_t)(16 * GT_K + 2 * R0 + 1) * 16 * 8;
    if (lds_c32 <= GT_LDS_MAX) {
        if ((e = need_lds((const void *)k_gauss_cols_t<32>, lds_c32)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_gauss_cols_t<32>, dim3((W + 31) / 32, (H + 8 * GT_K - 1) / (8 * GT_K)), dim3(256), lds_c32, stream, in, H, W, d_w0, R0, clip_scal, tmp);
    } else if (lds_c16 <= GT_LDS_MAX) {
        if ((e = need_lds((const void *)k_gauss_cols_t<16>, lds_c16)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_gauss_cols_t<16>, dim3((W + 15) / 16, (H + 16 * GT_K - 1) / (16 * GT_K)), dim3(256), lds_c16, stream, in, H, W, d_w0, R0, clip_scal, tmp);
    } else {
        if (clip_scal) return hipErrorInvalidValue;                         // (callers clip beforehand when the filter is this long)
        size_t lds_c = (size_t)(GC_ROWS + 2 * R0) * 32 * 8;
        if (lds_c <= 160 * 1024 - 1024) {
            if ((e = need_lds((const void *)k_gauss_cols<32>, lds_c)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_gauss_cols<32>, dim3((W + 31) / 32, (H + GC_ROWS - 1) / GC_ROWS), dim3(256), lds_c, stream, in, H, W, d_w0, R0, tmp);
        } else {
            lds_c = (size_t)(GC_ROWS + 2 * R0) * 8 * 8;
            if (lds_c > 160 * 1024 - 1024) return hipErrorInvalidValue;
            if ((e = need_lds((const void *)k_gauss_cols<8>, lds_c)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_gauss_cols<8>, dim3((W + 7) / 8, (H + GC_ROWS - 1) / GC_ROWS), dim3(256), lds_c, stream, in, H, W, d_w0, R0, tmp);
        }
    }
    const int span = GR_COLS + 2 * R1 + 1;
    const size_t lds_rt = (size_t)GR_ROWS * (span + (span >> 3) + 1) * 8;
    if (lds_rt <= GT_LDS_MAX) {
        if ((e = need_lds((const void *)k_gauss_rows_t, lds_rt)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_gauss_rows_t, dim3((W + GR_COLS - 1) / GR_COLS, (H + GR_ROWS - 1) / GR_ROWS), dim3(256), lds_rt, stream, (const double *)tmp, H, W, d_w1, R1, out,
                           cmb ? *cmb : none);
    } else {
        if (cmb) return hipErrorInvalidValue;
        const size_t lds_r = (size_t)(256 + 2 * R1) * 8;
        if (lds_r > 160 * 1024 - 1024) return hipErrorInvalidValue;
        if ((e = need_lds((const void *)k_gauss_rows, lds_r)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_gauss_rows, dim3((W + 255) / 256, H), dim3(256), lds_r, stream, (const double *)tmp, H, W, d_w1, R1, out);
    }
    return hipGetLastError();
}
static bool tiled_fits(int R) { const int span = GR_COLS + 2 * R + 1; return (size_t)(16 * GT_K + 2 * R + 1) * 16 * 8 <= GT_LDS_MAX && (size_t)GR_ROWS * (span + (span >> 3) + 1) * 8 <= GT_LDS_MAX; }

static hipError_t gauss2d(const double *in, int H, int W, const double *d_w, int R, double *tmp, double *out, hipStream_t stream)
{
    return separable2d(in, H, W, d_w, R, d_w, R, tmp, out, stream);
}

extern "C" hipError_t sdsm_preprocess_impl(const double *d_g, int H, int W, double sigma1, double sigma2, double offset_clip,
                                           int lower_clip_mean, double *d_y, void *d_ws, hipStream_t stream)
{
    size_t n = (size_t)H * W, nb = align_up(n * 8, 256);
    uint8_t *base = (uint8_t *)d_ws;
    double *off = (double *)base, *offc = (double *)(base + nb), *tmpA = (double *)(base + 2 * nb), *tmpB = (double *)(base + 3 * nb);
    uint8_t *target = base + 4 * nb;
    uint16_t *hd = (uint16_t *)(target + align_up(n, 256));
    int R1 = gauss_radius(sigma1), R2 = gauss_radius(sigma2);
    double *w1 = (double *)((uint8_t *)hd + align_up(n * 2, 256));
    double *w2 = w1 + (2 * R1 + 1);
    double *partial = (double *)((uint8_t *)w1 + align_up((2 * R1 + 1 + 2 * R2 + 1) * 8, 256));
    double *scal = partial + 1024;
    int *any = (int *)(scal + 8);
    // weights: computed on the host exactly as SciPy does, then copied (small, pageable -> staged synchronously by HIP)
    static thread_local double *hw = nullptr; static thread_local size_t hw_cap = 0;
    size_t nw = (size_t)(2 * R1 + 1 + 2 * R2 + 1);
    if (hw_cap < nw) { free(hw); hw = (double *)malloc(nw * 8); hw_cap = nw; }
    sdsm_gauss_kernel_host(sigma1, R1, hw);
    sdsm_gauss_kernel_host(sigma2, R2, hw + 2 * R1 + 1);
    hipError_t e = hipMemcpyAsync(w1, hw, nw * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(scal, 0, 128, stream);
    if (e != hipSuccess) return e;
    const int use_clip = !std::isinf(offset_clip);
    dim3 b(256), g1((n + 255) / 256), g2((W + 255) / 256, H);
    int npart = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    e = gauss2d(d_g, H, W, w2, R2, tmpA, off, stream);                                       // offset_original (:47)
    if (e != hipSuccess) return e;
    if (use_clip || lower_clip_mean) {
        hipLaunchKernelGGL(k_partial, dim3(npart), b, 0, stream, d_g, n, 0, (const double *)scal, partial);
        hipLaunchKernelGGL(k_final, dim3(1), b, 0, stream, (const double *)partial, npart, (double)n, 0, offset_clip, scal);
    }
    const bool fused = tiled_fits(R1) && tiled_fits(R2);       // clip on load and the combination as the epilogue of the last pass
    if (use_clip) {
        hipLaunchKernelGGL(k_partial, dim3(npart), b, 0, stream, d_g, n, 1, (const double *)scal, partial);
        hipLaunchKernelGGL(k_final, dim3(1), b, 0, stream, (const double *)partial, npart, (double)n, 1, offset_clip, scal);
        if (fused) e = separable2d(d_g, H, W, w2, R2, w2, R2, tmpA, offc, stream, scal);      // offset_clipped (:53), g clipped as it is read
        else {
            hipLaunchKernelGGL(k_clip, g1, b, 0, stream, d_g, n, (const double *)scal, tmpB);
            e = gauss2d(tmpB, H, W, w2, R2, tmpA, offc, stream);
        }
        if (e != hipSuccess) return e;
        int radius = (int)ceil(sigma2);
        launch_bounded_edt(d_g, H, W, 1, scal, radius, target, hd, any, 0.0, nullptr, nullptr, sigma2, tmpB, stream);      // t = max(sigma2 - d, 0)
    }
    if (fused) {
        CombineArgs cmb = {off, offc, tmpB, scal, any, sigma2, use_clip, lower_clip_mean};
        return separable2d(d_g, H, W, w1, R1, w1, R1, tmpA, d_y, stream, nullptr, &cmb);     // gauss(g, sigma1) - offset_combined (:64)
    }
    // denoised image into tmpA via d_y as the intermediate
    e = gauss2d(d_g, H, W, w1, R1, d_y, tmpA, stream);                                       // gauss(g, sigma1) (:64)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_combine, g1, b, 0, stream, (const double *)tmpA, (const double *)off, (const double *)offc, (const double *)tmpB,
                       n, use_clip, lower_clip_mean, (const double *)scal, (const int *)any, sigma2, d_y);
    return hipGetLastError();
}

// ---- plain Gaussian filter (scipy.ndimage.gaussian_filter defaults), for the smoothed images of the post-processing stage ----
extern "C" size_t sdsm_gaussian_workspace_bytes(int H, int W, double sigma)
{
    return align_up((size_t)H * W * 8, 256) + align_up((size_t)(2 * gauss_radius(sigma) + 1) * 8, 256);
}

extern "C" hipError_t sdsm_gaussian_filter_impl(const double *d_in, int H, int W, double sigma, double *d_out, void *d_ws, hipStream_t stream)
{
    const int R = gauss_radius(sigma);
    double *tmp = (double *)d_ws;
    double *w = (double *)((uint8_t *)d_ws + align_up((size_t)H * W * 8, 256));
    std::vector<double> hw(2 * R + 1);
    sdsm_gauss_kernel_host(sigma, R, hw.data());
    hipError_t e = hipMemcpyAsync(w, hw.data(), hw.size() * 8, hipMemcpyHostToDevice, stream);   // (pageable source: staged before the call returns)
    if (e != hipSuccess) return e;
    return gauss2d(d_in, H, W, w, R, tmp, d_out, stream);
}

// ---- separable filter with caller-given SYMMETRIC weights per axis (host arrays of 2 R + 1 doubles, centre at R): the
//      derivative-of-Gaussian filters of scipy.ndimage.gaussian_laplace in the scale estimation (superdsm/automation.py:52) ----
extern "C" size_t sdsm_separable_workspace_bytes(int H, int W, int R0, int R1)
{
    return align_up((size_t)H * W * 8, 256) + align_up((size_t)(2 * R0 + 1 + 2 * R1 + 1) * 8, 256);
}

extern "C" hipError_t sdsm_separable_filter_impl(const double *d_in, int H, int W, const double *h_w0, int R0, const double *h_w1, int R1,
                                                 double *d_out, void *d_ws, hipStream_t stream)
{
    double *tmp = (double *)d_ws;
    double *w0 = (double *)((uint8_t *)d_ws + align_up((size_t)H * W * 8, 256)), *w1 = w0 + (2 * R0 + 1);
    hipError_t e = hipMemcpyAsync(w0, h_w0, (size_t)(2 * R0 + 1) * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemcpyAsync(w1, h_w1, (size_t)(2 * R1 + 1) * 8, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    return separable2d(d_in, H, W, w0, R0, w1, R1, tmp, d_out, stream);
}
