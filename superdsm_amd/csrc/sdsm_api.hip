// Host side of the C ABI (include/sdsm.h): planning, workspace layout, launches.  No device allocation
// happens here: every device buffer is provided by the caller.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "sdsm_common.h"
#include <climits>

extern "C" hipError_t sdsm_launch_solve(const BatchParams &P, sdsm_record *records, uint32_t *masks, double *xi_out, hipStream_t stream,
                                        hipStream_t side1, hipStream_t side2, hipStream_t side3, hipStream_t side4, hipEvent_t *ev, int n_c, int n_d, int n_w, int n_r);
extern "C" hipError_t sdsm_image_prepare_impl(const double *, const uint8_t *, const int32_t *, int, int, double, int, uint8_t *, int32_t *, void *, hipStream_t);
extern "C" hipError_t sdsm_preprocess_impl(const double *, int, int, double, double, double, int, double *, void *, hipStream_t);
extern "C" void sdsm_gauss_kernel_host(double sigma, int radius, double *w);
extern "C" hipError_t sdsm_launch_setup(const BatchParams &P, hipStream_t stream, const int32_t *order_w, int n_w, int cls, const int32_t *order_small, int n_small, const int32_t *order_big, int n_big);
extern "C" int sdsm_setup_fits_small(int h, int w, int mcap, int max_label, int k);
extern "C" int sdsm_setup_class(int max_dim, int max_mcap, int max_label, int k);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
static int hipfail(hipError_t e, const char *what) { return fail(SDSM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }

extern "C" int sdsm_version(void) { return SDSM_VERSION; }
extern "C" const char *sdsm_last_error(void) { return g_err.c_str(); }

extern "C" int sdsm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { hipfail(e, "hipGetDeviceCount"); return 0; }
    return n;
}

extern "C" int sdsm_set_device(int device)
{
    hipError_t e = hipSetDevice(device);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "hipSetDevice");
}

extern "C" int sdsm_stream_synchronize(void *stream)
{
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "hipStreamSynchronize");
}

// ---- PSF of G~ (dsm.py:137-142): delta image filtered with SciPy's gaussian_filter, float32 ----------
static inline long reflect_h(long i, long n) { long p = 2 * n, m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m; }

static void correlate_line(const double *in, long n, long stride, const double *w, int R, double *out)
{
    std::vector<double> buf(n + 2 * R);
    for (long i = -R; i < n + R; i++) buf[i + R] = in[reflect_h(i, n) * stride];
    for (long l = 0; l < n; l++) {
        const double *c = buf.data() + l + R;
        double acc = c[0] * w[R];
        for (int j = R; j >= 1; j--) acc += (c[-j] + c[j]) * w[R - j];
        out[l * stride] = acc;
    }
}

static int make_psf(double sigma, double mult, std::vector<float> &psf)
{
    int k = (int)std::nearbyint(1 + sigma * 4 * mult);
    if (k < 1) k = 1;
    int R = (int)(4.0 * sigma + 0.5);
    std::vector<double> w(2 * R + 1), a((size_t)k * k, 0.0), b((size_t)k * k), c((size_t)k * k);
    sdsm_gauss_kernel_host(sigma, R, w.data());
    a[(size_t)(k / 2) * k + k / 2] = 1;
    for (int col = 0; col < k; col++) correlate_line(a.data() + col, k, k, w.data(), R, b.data() + col);
    for (int row = 0; row < k; row++) correlate_line(b.data() + (size_t)row * k, k, 1, w.data(), R, c.data() + (size_t)row * k);
    psf.resize((size_t)k * k);
    for (size_t i = 0; i < psf.size(); i++) psf[i] = (float)c[i];
    return k;
}

extern "C" int sdsm_psf(double sigma, double multiplier, float *out)
{
    if (!(sigma > 0) || std::isinf(sigma)) return fail(SDSM_ERR_ARGUMENT, "sdsm_psf: sigma must be finite and positive");
    std::vector<float> psf;
    int k = make_psf(sigma, multiplier, psf);
    if (out) memcpy(out, psf.data(), psf.size() * sizeof(float));
    return k;
}

// ---- preprocessing / image prepare --------------------------------------------------------------------
extern "C" size_t sdsm_preprocess_workspace_bytes(int H, int W, double sigma1, double sigma2);
extern "C" size_t sdsm_image_workspace_bytes(int H, int W);

extern "C" int sdsm_preprocess(const double *d_g, int H, int W, double sigma1, double sigma2, double offset_clip, int lower_clip_mean,
                               double *d_y, void *d_ws, size_t ws_bytes, void *stream)
{
    if (!d_g || !d_y || H < 1 || W < 1 || !(sigma1 > 0) || !(sigma2 > 0)) return fail(SDSM_ERR_ARGUMENT, "sdsm_preprocess: bad argument");
    if (ws_bytes < sdsm_preprocess_workspace_bytes(H, W, sigma1, sigma2) || !d_ws) return fail(SDSM_ERR_WORKSPACE, "sdsm_preprocess: workspace too small");
    hipError_t e = sdsm_preprocess_impl(d_g, H, W, sigma1, sigma2, offset_clip, lower_clip_mean, d_y, d_ws, (hipStream_t)stream);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "sdsm_preprocess");
}

extern "C" int sdsm_image_prepare(const double *d_y, const uint8_t *d_y_mask, const int32_t *d_atoms, int H, int W, double margin,
                                  int n_atoms, uint8_t *d_valid, int32_t *d_atom_stats, void *d_ws, size_t ws_bytes, void *stream)
{
    if (!d_y || !d_atoms || !d_valid || !d_atom_stats || H < 2 || W < 2 || H > 65535 || W > 65535 || n_atoms < 0)
        return fail(SDSM_ERR_ARGUMENT, "sdsm_image_prepare: bad argument (image must be 2..65535 pixels per side)");
    if (n_atoms > SDSM_MAX_LABELS) return fail(SDSM_ERR_UNSUPPORTED, "sdsm_image_prepare: more than 65535 atom labels");
    if (!(margin >= 0)) return fail(SDSM_ERR_ARGUMENT, "sdsm_image_prepare: background_margin must be >= 0");
    if (ws_bytes < sdsm_image_workspace_bytes(H, W) || !d_ws) return fail(SDSM_ERR_WORKSPACE, "sdsm_image_prepare: workspace too small");
    hipError_t e = sdsm_image_prepare_impl(d_y, d_y_mask, d_atoms, H, W, margin, n_atoms, d_valid, d_atom_stats, d_ws, (hipStream_t)stream);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "sdsm_image_prepare");
}

// ---- plan ------------------------------------------------------------------------------------------------
struct PlanImage { int H, W, n_atoms; };

// Side streams / fork-join events of the solve classes.  A plan borrows a set at its first launch and gives it back when it is
// destroyed: creating and destroying HIP streams costs milliseconds (hipStreamDestroy synchronises), a reference-style caller
// builds a plan per batch.  A set carries no state between users (events are recorded before they are waited for).
#include <mutex>
struct SideSet { hipStream_t side[4]; hipEvent_t fj[5]; int device; std::mutex enqueue; int queues = 0; int32_t *probe_words = nullptr; };   // queues: 0 not probed yet, 1 the four streams run side by side, -1 they share hardware queues
struct sdsm_plan {
    int n = 0;
    std::vector<PlanImage> images;             // one entry for sdsm_plan_create, several for sdsm_plan_create_multi
    sdsm_dsm_config cfg{};
    int k = 1, R = 0, zcap = 1, zcap_run = 1, zshift = 0, no_deform = 0;
    std::vector<CandDesc> cand;
    std::vector<int32_t> fp_labels, order;     // order: all candidates (largest first), then those whose bound on M admits more than solve class 1, then more than class 2
    int n_order_c = 0, n_order_d = 0, n_order_w = 0;   // the last list: (candidate | member << 24) of the workgroup groups
    int n_order_r = 0;                                  // then: (candidate | member << 24) of sdsm_k_setup_rows
    int rows_mcap = 1;                                  // largest bound on M among them
    int n_setup_small = 0, n_setup_big = 0, max_label = 1;   // setup in two launches (plans that mix small and large regions): lists behind the others
    int setup_class = 2;         // LDS limits of the setup kernel that hold this plan (sdsm_setup_class)
    int mode = 0;                // sdsm_plan_set_latency_mode: 0 throughput, 1 latency, 2 no workgroup groups
    int wide_pixels = INT_MAX;   // throughput mode by default
    int boost_pixels = INT_MAX;  // regions above run their pixel passes at a raised priority (throughput mode, layout_plan)
    std::vector<float> psf;
    std::vector<int32_t> mask_info, n_pixels;
    std::vector<int64_t> mask_off_bytes, xi_off;
    int64_t total_pixels = 0, total_runs = 0, total_ell = 0, total_xi = 0, total_mask_words = 0, n_hglob = 0, n_wide = 0;
    size_t off_cand = 0, off_state = 0, off_fp = 0, off_order = 0, off_crop_y = 0, off_crop_rc = 0, off_crop_cc = 0, off_dist = 0, off_tmp_y = 0, off_tmp_rc = 0, off_inv = 0, off_run_meta = 0,
           off_run_q0 = 0, off_run_aux = 0, off_grid = 0, off_ell_im = 0, off_ell_w = 0, off_psf = 0, off_env_fst = 0, off_env_rb = 0, off_hglob = 0, off_wide = 0, off_ticket = 0, total = 0;
    // The launch lists and CandDesc.wide_* live in the workspace (sdsm_batch_upload): a layout change after the upload
    // (sdsm_plan_set_latency_mode) would leave stale tables on the device, so launches check the generation they were uploaded at.
    uint64_t layout_gen = 0;
    mutable const double *x0 = nullptr;   // sdsm_plan_set_start: device pointer to the starting points of the DSM solves, or null
    mutable uint64_t uploaded_gen = 0;
    mutable const void *uploaded_ws = nullptr;
    // side streams / fork-join events of the solve classes, owned by the plan (created at its first launch)
    mutable SideSet *sides = nullptr;
};

#include <mutex>
// ONE set of side streams per device, shared by all plans and created at the first launch that needs it: the runtime maps streams onto
// GPU_MAX_HW_QUEUES hardware queues in the order of their creation, and a set created late (a set per live plan: the tenth stream of
// the process and later) came to share queues with the caller's stream or within itself -- the classes of one launch then ran one after
// the other (NIH3T3-like launch inside the full bench run 8.3 instead of 6.4 ms).  Launches enqueue under the set's mutex (fork and join
// events belong to the set); launches from different caller streams share the side queues.
static std::mutex g_pool_mutex;
static std::vector<SideSet *> g_side_sets;

static hipError_t acquire_sides(const sdsm_plan *p, hipStream_t caller)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    // the set belongs to a device: the caller's stream must live on the calling thread's current device (as PyTorch arranges it) -- a stream of another device
    // would fork work onto side streams of the wrong card
    int sdev = dev;
    if (caller && hipStreamGetDevice(caller, &sdev) == hipSuccess && sdev != dev) return hipErrorInvalidDevice;
    (void)hipGetLastError();
    if (p->sides) return p->sides->device == dev ? hipSuccess : hipErrorInvalidDevice;   // (a plan stays with the device of its first launch)
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (SideSet *q : g_side_sets) if (q->device == dev) { p->sides = q; return hipSuccess; }
    SideSet *s = new SideSet();
    s->device = dev;
    // Default priority.  (High priority for these queues -- the few long candidates of a launch are on them -- changes nothing where the
    // launch is short and costs 3 ms of the 71 of the synthetic 4096^2 launch, where every class has work: measured, round 3.  A fourth
    // side stream for class 2, beside the groups instead of behind them: 8 different BBBC039-like images 8.2 -> 8.5 ms, GOWT1-like
    // 4.9 -> 5.2: its 256 resident workgroups take compute units from the group members at the start.)
    for (int i = 0; i < 4; i++) if ((e = hipStreamCreateWithFlags(&s->side[i], hipStreamNonBlocking)) != hipSuccess) { delete s; return e; }   // (the fourth: class 2b of plans without groups)
    for (int i = 0; i < 5; i++) if ((e = hipEventCreateWithFlags(&s->fj[i], hipEventDisableTiming)) != hipSuccess) { delete s; return e; }
    g_side_sets.push_back(s);
    p->sides = s;
    return hipSuccess;
}

static size_t al(size_t v) { return (v + 255) / 256 * 256; }

// Multiplier A ~ 0.618 N coprime to N defines the scatter "position q holds raster rank (q * A) mod N"; the setup
// kernel needs the inverse map rank -> position, i.e. A^-1 mod N.
static uint32_t perm_inverse(uint32_t N)
{
    if (N <= 2) return 1;
    uint64_t A = (uint64_t)(0.6180339887498949 * N);
    if (A < 1) A = 1;
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; };
    while (gcd(A, N) != 1) A++;
    // extended Euclid: A * x = 1 (mod N)
    int64_t t = 0, newt = 1, r = N, newr = (int64_t)(A % N);
    while (newr != 0) { int64_t q = r / newr; int64_t tmp = t - q * newt; t = newt; newt = tmp; tmp = r - q * newr; r = newr; newr = tmp; }
    if (t < 0) t += N;
    return (uint32_t)t;
}

#ifndef SDSM_LAT_SLICE
#define SDSM_LAT_SLICE 2048         // latency mode: pixels per member of the group of a mid-size region
#endif
#ifndef SDSM_LAT_GMAX
#define SDSM_LAT_GMAX 4
#endif
#ifndef SDSM_ROWS_MAX_REGIONS
#define SDSM_ROWS_MAX_REGIONS 1024     // plans with more regions above SDSM_ROWS_MIN_PIXELS build all rows in the setup kernel
#endif
#ifndef SDSM_WIDE_TP_MIN_PIXELS
#define SDSM_WIDE_TP_MIN_PIXELS 8192   // throughput mode: no groups below this many pixels
#endif
#ifndef SDSM_K1_PIXDIV
#define SDSM_K1_PIXDIV 2048             // throughput mode: regions above (pixels of all candidates) / SDSM_K1_PIXDIV pixels leave the 192 / 256-thread classes
#endif
#ifndef SDSM_WIDE_FILL
#define SDSM_WIDE_FILL 1024              // throughput mode: groups only for regions of more than (pixels of all candidates) / SDSM_WIDE_FILL pixels
#endif
#ifndef SDSM_WIDE_MAX_MEMBERS
#define SDSM_WIDE_MAX_MEMBERS 256    // throughput mode: group members of one launch (= compute units of an MI355X); the largest regions first
#endif
// Workgroup groups, launch lists and workspace layout (repeated when the scheduling mode changes).
static void layout_plan(sdsm_plan *p)
{
    const int n = p->n;
    const bool latency = p->mode == 1, groups = p->mode != 2;
    p->wide_pixels = latency ? SDSM_WIDE_PIXELS : INT_MAX;
    p->boost_pixels = INT_MAX;
    p->layout_gen++;
    p->n_wide = 0;
    // Groups shorten the longest chains of a launch; every member holds a whole compute unit and a group only advances while all of its
    // members are resident.  In throughput mode the LARGEST regions get groups until their members add up to the compute units of the
    // chip; very large regions beyond that (synthetic 4096^2 image: 434 of them, 1296 members -- they waited for each other behind the
    // small classes) are solved as ordinary class-2 candidates, one workgroup each.
    long long wide_thr = SDSM_WIDE_MIN_PIXELS;       // regions with more pixels get a group of one member per SDSM_WIDE_SLICE pixels (throughput mode: set below)
    if (groups && !latency) {
        // ... and only regions whose chain (~ its pixels) is long next to the time the whole plan keeps the chip busy (~ all pixels / compute units):
        // the synthetic 4096^2 plan (10 073 candidates, 50 M pixels) gets no groups at all -- its 31 k-pixel regions end long before the rest does,
        // and groups there cost compute units that the other candidates wait for (75 -> 97 ms with groups for everything above 12 288 pixels)
        long long all_pixels = 0;
        for (int i = 0; i < n; i++) all_pixels += p->cand[i].N;
        long long tp_min = SDSM_WIDE_TP_MIN_PIXELS;
        if (const char *e = getenv("SDSM_WIDE_TP_MIN_PIXELS")) { const long long v = atoll(e); if (v > 0) tp_min = v; }   // diagnostic knob
        wide_thr = std::max<long long>(tp_min, all_pixels / SDSM_WIDE_FILL);
    }
    if (!latency) {
        // Throughput mode: a candidate whose chain (~ its pixels) is long next to the time the plan keeps the chip busy (~ pixels of all
        // candidates / 1024 workgroup slots of class 1) runs its passes over the pixels at a raised issue priority: regions above 1 /
        // SDSM_K1_PIXDIV of all pixels.  8 different BBBC039-like images: class 1 took 6.8 ms beside the other classes while its workgroup
        // time was 3.6 ms of the chip -- its 4-5 ms chains (6-7 k pixels, 95-114 unknowns; 2.8 ms on an idle chip) ended it.  (Moving
        // them to the 512-thread classes instead, a compute unit each: 7.6 -> 8.7-9.2 ms per step, measured in round 4.)
        long long all_pixels = 0;
        for (int i = 0; i < n; i++) all_pixels += p->cand[i].N;
        long long div = SDSM_K1_PIXDIV;
        if (const char *e = getenv("SDSM_K1_PIXDIV")) { const long long v = atoll(e); if (v > 0) div = v; else if (v < 0) div = 0; }   // diagnostic knob (negative: no bound)
        p->boost_pixels = div > 0 ? (int)std::min<long long>(INT_MAX, std::max<long long>(SDSM_WIDE_PIXELS, all_pixels / div)) : INT_MAX;
    }
    long wide_slice = SDSM_WIDE_SLICE;
    if (const char *e = getenv("SDSM_WIDE_SLICE")) { const long v = atol(e); if (v > 0) wide_slice = v; }                // diagnostic knob
    auto group_size = [&](long N) -> long {              // members of the group a region of N pixels would get (0: none)
        if (!groups || n >= (1 << 24)) return 0;
        if (N > wide_thr) return std::min<long>(SDSM_WIDE_MAX_G, std::max<long>(2, (N + wide_slice - 1) / wide_slice));
        if (latency && N > SDSM_WIDE_PIXELS) return std::min<long>(SDSM_LAT_GMAX, std::max<long>(2, (N + SDSM_LAT_SLICE - 1) / SDSM_LAT_SLICE));   // latency mode: the largest regions of an ordinary image too
        return 0;
    };
    // the member budget (both modes: a single image with many large clusters -- 636 candidates, 16 k-pixel regions -- asked for several hundred
    // members in latency mode and took 15 ms instead of 12)
    std::vector<char> grouped(n, 1);
    {
        std::vector<int> big;
        for (int i = 0; i < n; i++) if (group_size(p->cand[i].N) > 0) big.push_back(i);
        std::stable_sort(big.begin(), big.end(), [&](int a, int b) { return p->cand[a].N > p->cand[b].N; });
        long members = 0;
        int cutoff = 0;                                  // regions of at most this many pixels get no group (equal regions are treated alike)
        for (int i : big) {
            members += group_size(p->cand[i].N);
            if (members > SDSM_WIDE_MAX_MEMBERS) { cutoff = p->cand[i].N; break; }
        }
        for (int i : big) if (p->cand[i].N <= cutoff) grouped[i] = 0;
    }
    long rows_regions = 0;
    for (int i = 0; i < n; i++) rows_regions += p->cand[i].N > SDSM_ROWS_MIN_PIXELS;
    const bool rows_multi = rows_regions <= SDSM_ROWS_MAX_REGIONS;
    for (int i = 0; i < n; i++) {
        CandDesc &c = p->cand[i];
        const long G = grouped[i] ? group_size(c.N) : 0;
        // rows of G~ by several workgroups for every large region, whether or not a workgroup group solves it (a 12 k-pixel region took a
        // single setup workgroup 0.5-1 ms: the end of the setup kernel)
        // (only while such regions are few enough to be the END of the setup kernel: with the 5000 of the synthetic 4096^2 plan the setup kernel is
        // busy throughout and builds the rows itself at a little more than half the cost -- 5.2 instead of 9.2 ms there)
        const long RG = rows_multi && c.N > SDSM_ROWS_MIN_PIXELS && n < (1 << 24) ? std::min<long>(SDSM_ROWS_MAX_G, (c.N + SDSM_ROWS_SLICE - 1) / SDSM_ROWS_SLICE) : 0;
        c.rows_g = (int32_t)std::max<long>(RG, G > 0 ? 1 : 0);
        c.pad1 = 0;
        c.wide_g = (int32_t)G;
        if (G > 0 || c.rows_g > 0) { c.wide_off = p->n_wide; p->n_wide += SDSM_WIDE_SYNC + (int64_t)2 * G * SDSM_WIDE_PBUF; }
        else c.wide_off = -1;
    }
    p->n_order_c = p->n_order_d = p->n_order_w = p->n_order_r = 0;
    p->rows_mcap = 1;
    p->order.resize(n);
    std::iota(p->order.begin(), p->order.end(), 0);
    std::stable_sort(p->order.begin(), p->order.end(), [&](int a, int b) { return p->cand[a].N > p->cand[b].N; });
    // a candidate can only belong to a larger size class if its upper bound Mcap allows it: the larger classes get
    // their own (shorter) launch lists instead of n workgroups that exit immediately
    for (int k = 0; k < n; k++) if (6 + p->cand[p->order[k]].Mcap > SDSM_K1_DENSE_N || p->cand[p->order[k]].N > SDSM_WIDE_PIXELS) { p->order.push_back(p->order[k]); p->n_order_c++; }
    for (int k = 0; k < n; k++) if (6 + p->cand[p->order[k]].Mcap > SDSM_ENV_DENSE_N) { p->order.push_back(p->order[k]); p->n_order_d++; }
    for (int k = 0; k < n; k++) {
        const int ci = p->order[k];
        for (int g = 0; g < p->cand[ci].wide_g; g++) { p->order.push_back(ci | (g << 24)); p->n_order_w++; }
    }
    for (int k = 0; k < n; k++) {
        const int ci = p->order[k];
        for (int g = 0; g < p->cand[ci].rows_g; g++) { p->order.push_back(ci | (g << 24)); p->n_order_r++; }
        if (p->cand[ci].rows_g > 0) p->rows_mcap = std::max(p->rows_mcap, (int)std::min<int64_t>(p->cand[ci].Mcap, SDSM_MAX_N_GLOBAL));
    }
    // Setup: a plan whose largest region needs the large tables of the setup kernel (1024 threads per candidate) but whose candidates are
    // mostly small (an image set with a few big clusters) sets the small ones up with the 256-thread class in a launch of its own
    p->n_setup_small = p->n_setup_big = 0;
    if (p->setup_class != 0) {
        std::vector<int32_t> small, big;
        for (int k = 0; k < n; k++) {
            const CandDesc &c = p->cand[p->order[k]];
            (c.N <= SDSM_SETUP_SMALL_PIXELS && sdsm_setup_fits_small(c.h, c.w, c.Mcap, p->max_label, p->k) ? small : big).push_back(p->order[k]);
        }
        if (small.size() >= 256) {
            p->n_setup_small = (int)small.size(); p->n_setup_big = (int)big.size();
            p->order.insert(p->order.end(), small.begin(), small.end());
            p->order.insert(p->order.end(), big.begin(), big.end());
        }
    }
    // workspace layout
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    p->off_cand = take(sizeof(CandDesc) * std::max(n, 1));
    p->off_state = take(sizeof(CandState) * std::max(n, 1));
    p->off_fp = take(4 * std::max<size_t>(p->fp_labels.size(), 1));
    p->off_order = take(4 * std::max<size_t>(p->order.size(), 1));
    p->off_psf = take(4 * p->psf.size());
    size_t np = (size_t)std::max<int64_t>(p->total_pixels, 1), nr = (size_t)std::max<int64_t>(p->total_runs, 1);
    p->off_crop_y = take(8 * SDSM_RUN * nr);
    p->off_crop_rc = take(4 * nr);
    p->off_run_meta = take(4 * nr);
    p->off_run_q0 = take(4 * nr);
    p->off_run_aux = take(4 * nr);
    p->off_crop_cc = take(4 * np);
    p->off_dist = take(4 * np);
    p->off_tmp_y = take(8 * np);
    p->off_tmp_rc = take(4 * np);
    p->off_inv = take(4 * np);
    p->off_grid = take(4 * (size_t)std::max<int64_t>(p->total_xi, 1));
    p->off_ell_im = take(4 * (size_t)std::max<int64_t>(p->total_ell, 1));
    p->off_ell_w = take(16 * (size_t)std::max<int64_t>(p->total_ell, 1));
    p->off_env_fst = take(4 * (size_t)std::max<int64_t>(p->total_xi, 1));
    p->off_env_rb = take(4 * (size_t)std::max<int64_t>(p->total_xi, 1));
    p->off_hglob = take(8 * (size_t)std::max<int64_t>(p->n_hglob, 1));      // n_hglob counts doubles
    p->off_wide = take(8 * (size_t)std::max<int64_t>(p->n_wide, 1));        // n_wide counts doubles
    p->off_ticket = take(256);                                                // [0] ticket of the workgroup groups; [16 ..] heads of the launch lists of the classes beyond 1 (zeroed before every launch)
    p->total = o;
}
extern "C" sdsm_plan *sdsm_plan_create_multi(int n_images, const int32_t *H, const int32_t *W, const int32_t *n_atoms, const int32_t *const *atom_stats,
                                             const sdsm_dsm_config *cfg, int n, const int32_t *offsets, const int32_t *labels, const int32_t *image_of)
{
    if (n_images < 1 || n_images > SDSM_MAX_IMAGES || !H || !W || !n_atoms || !atom_stats || !cfg || n < 0 || (n > 0 && (!offsets || !labels)) || (n > 0 && n_images > 1 && !image_of)) {
        fail(SDSM_ERR_ARGUMENT, "sdsm_plan_create: bad argument (1 .. 16 images per plan)"); return nullptr;
    }
    for (int i = 0; i < n_images; i++) if (H[i] < 2 || W[i] < 2 || !atom_stats[i] || n_atoms[i] < 0) { fail(SDSM_ERR_ARGUMENT, "sdsm_plan_create: bad image"); return nullptr; }
    if (!(cfg->epsilon > 0) || !(cfg->alpha >= 0) || !(cfg->scale > 0) || cfg->smooth_subsample < 1 || !(cfg->smooth_amount > 0)) {
        fail(SDSM_ERR_ARGUMENT, "sdsm_plan_create: epsilon > 0, alpha >= 0, scale > 0, smooth_subsample >= 1, smooth_amount > 0 required (dsm.py:275-285)");
        return nullptr;
    }
    sdsm_plan *p = new sdsm_plan();
    p->n = n; p->cfg = *cfg;
    for (int i = 0; i < n_images; i++) p->images.push_back({H[i], W[i], n_atoms[i]});
    if (p->cfg.max_iters <= 0) p->cfg.max_iters = 100;
    p->no_deform = std::isinf(cfg->smooth_amount) ? 1 : 0;
    if (!p->no_deform) {
        p->k = make_psf(cfg->smooth_amount, cfg->gaussian_shape_multiplier, p->psf);
        if (p->k % 2 == 0) {   // _convmat asserts an odd filter size (dsm.py:147)
            fail(SDSM_ERR_ARGUMENT, "sdsm_plan_create: round(1 + 4 * smooth_amount * gaussian_shape_multiplier) must be odd (dsm.py:147)");
            delete p; return nullptr;
        }
        p->R = p->k / 2;
        long per = (2 * p->R) / cfg->smooth_subsample + 1;
        long z = per * per;
        p->zcap = (int)std::min<long>(z, 4 * SDSM_MAX_ELL_GROUPS);   // grid points are at least `subsample` apart (chessboard): at most per^2 inside a pixel's window; a solvable row has <= M <= 1018 entries
        // a run spans up to SDSM_RUN columns: the union of its pixels' windows is SDSM_RUN - 1 columns wider
        const long per_c = (2 * p->R + SDSM_RUN - 1) / cfg->smooth_subsample + 1;
        p->zcap_run = (int)std::min<long>(per * per_c, 4 * SDSM_MAX_ELL_GROUPS);
        p->zshift = p->zcap_run > SDSM_MAX_ELL_GROUPS ? 2 : 0;
        if (p->zshift) p->zcap_run = (p->zcap_run + 3) / 4 * 4;                   // sort classes of 4 entries: rows are padded to whole classes
    } else { p->psf.assign(1, 1.f); p->k = 1; p->R = 0; p->zcap = 1; p->zcap_run = 1; p->zshift = 0; }
    const int s = cfg->smooth_subsample;
    p->cand.resize(n); p->mask_info.resize((size_t)4 * n); p->mask_off_bytes.resize(n); p->xi_off.resize(n); p->n_pixels.resize(n);
    p->fp_labels.assign(labels, labels + (n > 0 ? offsets[n] : 0));
    for (int i = 0; i < n; i++) {
        CandDesc &c = p->cand[i];
        const int im = image_of ? image_of[i] : 0;
        if (im < 0 || im >= n_images) { fail(SDSM_ERR_ARGUMENT, "sdsm_plan_create: image index out of range"); delete p; return nullptr; }
        const int Hi = H[im], Wi = W[im];
        long N = 0; int r0 = Hi, r1 = -1, c0 = Wi, c1 = -1;
        for (int e = offsets[i]; e < offsets[i + 1]; e++) {
            int l = labels[e];
            if (l < 1 || l > n_atoms[im]) continue;
            const int32_t *st = atom_stats[im] + (size_t)l * SDSM_ATOM_STATS_STRIDE;
            if (st[0] <= 0) continue;
            N += st[0];
            r0 = std::min(r0, st[1]); r1 = std::max(r1, st[2]); c0 = std::min(c0, st[3]); c1 = std::max(c1, st[4]);
        }
        if (r1 < 0) { r0 = c0 = 0; r1 = c1 = 0; N = 0; }
        c.image = im; c.pad1 = 0; c.rows_g = 0;
        c.N = (int32_t)N; c.r0 = r0; c.c0 = c0; c.h = r1 - r0 + 1; c.w = c1 - c0 + 1;
        c.fp_off = offsets[i]; c.fp_len = offsets[i + 1] - offsets[i];
        long mc = p->no_deform ? 1 : (long)((c.h + s - 1) / s) * ((c.w + s - 1) / s);
        c.Mcap = (int32_t)std::max<long>(1, std::min<long>(mc, std::max<long>(N, 1)));
        // runs: the region pixels of one image row inside one aligned 4-column cell; at most one per pixel and one per cell of the bounding box
        const long ncw = c.w > 0 ? ((c0 + c.w - 1) >> 2) - (c0 >> 2) + 1 : 1;
        c.NRcap = (int32_t)std::max<long>(1, std::min<long>(N, (long)c.h * ncw));
        c.crop_off = p->total_pixels; p->total_pixels += N;
        c.run_off = p->total_runs; p->total_runs += c.NRcap;
        c.ell_off = p->total_ell; p->total_ell += (int64_t)c.NRcap * p->zcap_run;
        c.xi_off = p->total_xi; p->total_xi += c.Mcap;
        c.mask_off = p->total_mask_words; p->total_mask_words += ((int64_t)c.h * c.w + 31) / 32;
        if (6 + c.Mcap > SDSM_ENV_DENSE_N) {
            const int64_t nn = 6 + std::min<int64_t>(c.Mcap, SDSM_MAX_N_GLOBAL - 6);
            c.hglob_off = p->n_hglob; p->n_hglob += nn * (nn + 1) / 2;
        } else c.hglob_off = -1;
        c.perm_inv = perm_inverse((uint32_t)std::max<long>(N, 1));
        p->mask_info[4 * i] = r0; p->mask_info[4 * i + 1] = c0; p->mask_info[4 * i + 2] = c.h; p->mask_info[4 * i + 3] = c.w;
        p->mask_off_bytes[i] = c.mask_off * 4; p->xi_off[i] = c.xi_off; p->n_pixels[i] = c.N;
    }
    {
        int max_dim = 1, max_mcap = 1, max_label = 1;
        for (const CandDesc &c : p->cand) { max_dim = std::max(max_dim, std::max(c.h, c.w)); max_mcap = std::max(max_mcap, c.Mcap); }
        for (int im = 0; im < n_images; im++) max_label = std::max(max_label, n_atoms[im]);
        p->setup_class = sdsm_setup_class(max_dim, max_mcap, max_label, p->k);
        p->max_label = max_label;
    }
    layout_plan(p);
    return p;
}

extern "C" sdsm_plan *sdsm_plan_create(int H, int W, int n_atoms, const int32_t *atom_stats, const sdsm_dsm_config *cfg,
                                       int n, const int32_t *offsets, const int32_t *labels)
{
    const int32_t h = H, w = W, na = n_atoms;
    return sdsm_plan_create_multi(1, &h, &w, &na, &atom_stats, cfg, n, offsets, labels, nullptr);
}

extern "C" void sdsm_plan_destroy(sdsm_plan *plan)
{
    if (!plan) return;
    delete plan;                                         // (the side streams belong to the device, not to the plan)
}
extern "C" int sdsm_plan_set_latency_mode(sdsm_plan *p, int on)
{
    if (!p) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_set_latency_mode: null plan");
    if (on < 0 || on > 2) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_set_latency_mode: mode must be 0, 1 or 2");
    p->mode = on;
    layout_plan(p);                                       // new launch lists / workspace size / generation: upload again before the next launch
    return SDSM_OK;
}
extern "C" size_t sdsm_plan_workspace_bytes(const sdsm_plan *p) { return p ? p->total : 0; }
extern "C" size_t sdsm_plan_mask_bytes(const sdsm_plan *p) { return p ? (size_t)std::max<int64_t>(p->total_mask_words, 1) * 4 : 0; }
extern "C" int64_t sdsm_plan_total_pixels(const sdsm_plan *p) { return p ? p->total_pixels : 0; }
extern "C" int64_t sdsm_plan_xi_count(const sdsm_plan *p) { return p ? std::max<int64_t>(p->total_xi, 1) : 0; }

extern "C" int sdsm_plan_describe(const sdsm_plan *p, int32_t *mask_info, int64_t *mask_offset, int32_t *n_pixels)
{
    if (!p) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_describe: null plan");
    if (mask_info) memcpy(mask_info, p->mask_info.data(), p->mask_info.size() * 4);
    if (mask_offset) memcpy(mask_offset, p->mask_off_bytes.data(), p->mask_off_bytes.size() * 8);
    if (n_pixels) memcpy(n_pixels, p->n_pixels.data(), p->n_pixels.size() * 4);
    return SDSM_OK;
}

extern "C" int sdsm_plan_schedule(const sdsm_plan *p, int32_t *group_members, int32_t *rows_workgroups)
{
    if (!p) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_schedule: null plan");
    for (int i = 0; i < p->n; i++) {
        if (group_members) group_members[i] = p->cand[i].wide_g;
        if (rows_workgroups) rows_workgroups[i] = p->cand[i].rows_g;
    }
    return SDSM_OK;
}

extern "C" int sdsm_plan_xi_offsets(const sdsm_plan *p, int64_t *xi_offset)
{
    if (!p || !xi_offset) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_xi_offsets: null argument");
    memcpy(xi_offset, p->xi_off.data(), p->xi_off.size() * 8);
    return SDSM_OK;
}

extern "C" int sdsm_plan_layout(const sdsm_plan *p, int64_t *out)
{
    if (!p || !out) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_layout: null argument");
    int64_t v[16] = {(int64_t)p->off_cand, (int64_t)p->off_state, (int64_t)p->off_crop_y, (int64_t)p->off_crop_rc, (int64_t)p->off_crop_cc,
                     (int64_t)p->off_run_meta, (int64_t)p->off_grid, (int64_t)p->off_ell_im, (int64_t)p->off_ell_w, p->zcap_run, p->k,
                     (int64_t)sizeof(CandDesc), p->total_pixels, p->total_ell, (int64_t)sizeof(CandState), p->total_runs};
    memcpy(out, v, sizeof(v));
    return SDSM_OK;
}

extern "C" int sdsm_batch_upload(const sdsm_plan *p, void *d_ws, size_t ws_bytes, void *stream)
{
    if (!p || !d_ws) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_upload: null argument");
    if (ws_bytes < p->total) return fail(SDSM_ERR_WORKSPACE, "sdsm_batch_upload: workspace too small");
    if (p->n == 0) { p->uploaded_gen = p->layout_gen; p->uploaded_ws = d_ws; return SDSM_OK; }
    uint8_t *b = (uint8_t *)d_ws;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if ((e = hipMemcpyAsync(b + p->off_cand, p->cand.data(), sizeof(CandDesc) * p->n, hipMemcpyHostToDevice, s)) != hipSuccess) return hipfail(e, "upload cand");
    if (!p->fp_labels.empty() && (e = hipMemcpyAsync(b + p->off_fp, p->fp_labels.data(), 4 * p->fp_labels.size(), hipMemcpyHostToDevice, s)) != hipSuccess) return hipfail(e, "upload footprints");
    if ((e = hipMemcpyAsync(b + p->off_order, p->order.data(), 4 * p->order.size(), hipMemcpyHostToDevice, s)) != hipSuccess) return hipfail(e, "upload order");
    if ((e = hipMemcpyAsync(b + p->off_psf, p->psf.data(), 4 * p->psf.size(), hipMemcpyHostToDevice, s)) != hipSuccess) return hipfail(e, "upload psf");
    p->uploaded_gen = p->layout_gen; p->uploaded_ws = d_ws;
    return SDSM_OK;
}

static thread_local long long *g_prof = nullptr;
// Time a member of a workgroup group waits for its partners at one exchange before the group is given up (the candidate is then solved again
// without a group: correct, one more launch).  50 ms: a partner that is not resident by then is waiting for a compute unit behind other work --
// several processes sharing the card --, and the retry costs less than the wait (round 3 waited 10 s).
#define SDSM_GROUP_TIMEOUT_TICKS 5000000ll
static thread_local long long g_wide_timeout = SDSM_GROUP_TIMEOUT_TICKS;   // ticks of the 100 MHz wall clock
extern "C" int sdsm_set_group_timeout_us(double us) { g_wide_timeout = us > 0 ? (long long)(us * 100.0) : SDSM_GROUP_TIMEOUT_TICKS; return SDSM_OK; }

// Whether the caller's stream and the side streams of the last probed launch ran side by side: 1 yes, 0 they share hardware queues (the solve
// classes of a launch then run one after the other: correct, slower -- GPU_MAX_HW_QUEUES was too small when the process first touched the GPU),
// -1 not probed yet (no launch needed side streams so far).
static int g_queues_distinct = -1;
extern "C" int sdsm_side_queues_distinct(void) { return g_queues_distinct; }
extern "C" hipError_t sdsm_launch_queue_probe(int32_t *d_words, hipStream_t s0, hipStream_t s1, hipStream_t s2, hipStream_t s3);
// Once per device (first launch that needs the side streams): four one-wavefront kernels, one per stream, wait for each other (<= 2 ms).
static void probe_queues(SideSet *q, hipStream_t caller)
{
    if (q->queues != 0) return;
    q->queues = -1;
    if (hipMalloc((void **)&q->probe_words, 8) != hipSuccess) { q->probe_words = nullptr; (void)hipGetLastError(); return; }
    int32_t res[2] = {0, 1};
    if (hipMemsetAsync(q->probe_words, 0, 8, caller) == hipSuccess && hipStreamSynchronize(caller) == hipSuccess
        && sdsm_launch_queue_probe(q->probe_words, caller, q->side[0], q->side[1], q->side[2]) == hipSuccess) {
        (void)hipStreamSynchronize(q->side[0]); (void)hipStreamSynchronize(q->side[1]); (void)hipStreamSynchronize(q->side[2]); (void)hipStreamSynchronize(caller);
        if (hipMemcpy(res, q->probe_words, 8, hipMemcpyDeviceToHost) != hipSuccess) res[1] = 1;
    }
    q->queues = res[1] == 0 ? 1 : -1;
    g_queues_distinct = q->queues == 1 ? 1 : 0;
    if (q->queues != 1) {
        const char *v = getenv("GPU_MAX_HW_QUEUES");
        fprintf(stderr, "superdsm_amd: the streams of a launch share hardware queues (GPU_MAX_HW_QUEUES=%s): its solve classes run one after the other -- "
                        "correct, slower.  Set GPU_MAX_HW_QUEUES=8 (or import superdsm_amd) before the process first touches the GPU.\n", v ? v : "unset: 4");
    }
}
// Diagnostic builds (-DSDSM_PROFILE): device buffer of 8 int64 cycle counters per candidate, see DESIGN.md.
extern "C" int sdsm_set_debug_buffer(void *d_buf) { g_prof = (long long *)d_buf; return SDSM_OK; }


#ifndef SDSM_HESS_THR
#define SDSM_HESS_THR 0.1f    // same constant as the oracle's ORC_HESS_THR
#endif
// Kernel arguments of a plan laid out in the caller's workspace.
static BatchParams make_params(const sdsm_plan *p, void *d_ws)
{
    uint8_t *b = (uint8_t *)d_ws;
    BatchParams P{};
    P.n = p->n; P.n_total = p->n; P.latency = p->mode == 1; P.n_images = (int)p->images.size();
    for (size_t i = 0; i < p->images.size(); i++) { P.img[i].H = p->images[i].H; P.img[i].W = p->images[i].W; }   // device pointers: filled by the launch
    P.k = p->k; P.R = p->R; P.subsample = p->cfg.smooth_subsample; P.zcap = p->zcap; P.zcap_run = p->zcap_run; P.zshift = p->zshift; P.no_deform = p->no_deform; P.no_trivial_rule = p->cfg.flags & 1;
    P.init_elliptical = p->cfg.init_elliptical; P.max_iters = p->cfg.max_iters; P.k1_pixmax = p->wide_pixels; P.boost_pixels = p->boost_pixels; P.rows_mcap = p->rows_mcap; P.pad2 = 0;
    P.scale = p->cfg.scale; P.epsilon = p->cfg.epsilon; P.alpha = p->cfg.alpha; P.reg_unit = p->cfg.alpha * std::sqrt(p->cfg.epsilon);
    P.cand = (const CandDesc *)(b + p->off_cand); P.state = (CandState *)(b + p->off_state);
    P.fp_labels = (const int32_t *)(b + p->off_fp); P.order = (const int32_t *)(b + p->off_order);
    P.crop_y = (double *)(b + p->off_crop_y); P.crop_rc = (uint32_t *)(b + p->off_crop_rc); P.crop_cc = (uint32_t *)(b + p->off_crop_cc);
    P.dist = (uint32_t *)(b + p->off_dist); P.grid_rc = (uint32_t *)(b + p->off_grid);
    P.ell_im = (uint32_t *)(b + p->off_ell_im); P.ell_w = (float *)(b + p->off_ell_w); P.run_meta = (uint32_t *)(b + p->off_run_meta);
    P.run_q0 = (uint32_t *)(b + p->off_run_q0); P.run_aux = (uint32_t *)(b + p->off_run_aux);
    P.tmp_y = (double *)(b + p->off_tmp_y); P.tmp_rc = (uint32_t *)(b + p->off_tmp_rc); P.inv = (uint32_t *)(b + p->off_inv);
    P.hess_thr = SDSM_HESS_THR;
    P.psf = (const float *)(b + p->off_psf);
    P.env_fst = (int32_t *)(b + p->off_env_fst); P.env_rb = (int32_t *)(b + p->off_env_rb);
    P.hglob = (double *)(b + p->off_hglob); P.wide_pool = (double *)(b + p->off_wide);
    P.wide_ticket = (int32_t *)(b + p->off_ticket); P.wide_timeout = g_wide_timeout;
    P.cls_count = (int32_t *)(b + p->off_ticket) + 16;
    P.prof = g_prof; P.prof2 = g_prof ? g_prof + (size_t)16 * p->n : nullptr;
    P.x0 = p->x0;
    return P;
}

static thread_local int g_timing = 0;
static thread_local hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
static thread_local int g_ev_valid = 0;

extern "C" int sdsm_enable_kernel_timing(int enable)
{
    g_timing = enable;
    if (enable && !g_ev[0]) {
        for (int i = 0; i < 3; i++) { hipError_t e = hipEventCreate(&g_ev[i]); if (e != hipSuccess) return hipfail(e, "hipEventCreate"); }
    }
    return SDSM_OK;
}

static double elapsed(int a, int b)
{
    if (!g_ev_valid) return -1.0;
    if (hipEventSynchronize(g_ev[b]) != hipSuccess) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, g_ev[a], g_ev[b]) != hipSuccess) return -1.0;
    return ms;
}
extern "C" double sdsm_last_setup_kernel_ms(void) { return elapsed(0, 1); }
extern "C" double sdsm_last_solve_kernel_ms(void) { return elapsed(1, 2); }

extern "C" int sdsm_batch_launch_multi(const sdsm_plan *p, const double *const *d_y, const int32_t *const *d_atoms, const uint8_t *const *d_valid,
                                       void *d_ws, size_t ws_bytes, sdsm_record *d_records, uint32_t *d_masks, double *d_xi, void *stream)
{
    if (!p || !d_y || !d_atoms || !d_valid || !d_ws || !d_records || !d_masks) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_launch: null argument");
    for (size_t i = 0; i < p->images.size(); i++) if (!d_y[i] || !d_atoms[i] || !d_valid[i]) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_launch: null image pointer");
    if (p->uploaded_gen != p->layout_gen || p->uploaded_ws != d_ws)
        return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_launch: the plan's tables in this workspace are missing or stale (sdsm_batch_upload must follow sdsm_plan_create and every sdsm_plan_set_latency_mode)");
    if (ws_bytes < p->total) return fail(SDSM_ERR_WORKSPACE, "sdsm_batch_launch: workspace too small");
    if (p->n == 0) return SDSM_OK;
    hipStream_t s = (hipStream_t)stream;
    BatchParams P = make_params(p, d_ws);
    for (size_t i = 0; i < p->images.size(); i++) { P.img[i].y = d_y[i]; P.img[i].atoms = d_atoms[i]; P.img[i].valid = d_valid[i]; }
    hipError_t e;
    if (g_timing && (e = hipEventRecord(g_ev[0], s)) != hipSuccess) return hipfail(e, "hipEventRecord");
    if ((e = hipMemsetAsync((uint8_t *)d_ws + p->off_ticket, 0, 256, s)) != hipSuccess) return hipfail(e, "hipMemsetAsync");   // ticket, work-list counters
    {
        const int32_t *lists = P.order + p->n + p->n_order_c + p->n_order_d;
        if ((e = sdsm_launch_setup(P, s, lists, p->n_order_w, p->setup_class, lists + p->n_order_w + p->n_order_r, p->n_setup_small, lists + p->n_order_w + p->n_order_r + p->n_setup_small, p->n_setup_big)) != hipSuccess)
            return hipfail(e, "launch setup");
    }
    if (g_timing && (e = hipEventRecord(g_ev[1], s)) != hipSuccess) return hipfail(e, "hipEventRecord");
    hipStream_t s1 = nullptr, s2 = nullptr, s3 = nullptr;
    hipEvent_t *fj = nullptr;
    if (p->n_order_c > 0 || p->n_order_d > 0 || p->n_order_w > 0) {
        if ((e = acquire_sides(p, s)) != hipSuccess) return hipfail(e, "side streams (the stream must belong to the current device; a plan stays with the device of its first launch)");
        s1 = p->sides->side[0]; s2 = p->sides->side[1]; s3 = p->sides->side[2]; fj = p->sides->fj;
        if (p->sides->queues == 0) { std::lock_guard<std::mutex> lock(p->sides->enqueue); probe_queues(p->sides, s); }
    }
    {
        std::unique_lock<std::mutex> enq;
        if (p->sides) enq = std::unique_lock<std::mutex>(p->sides->enqueue);
        if ((e = sdsm_launch_solve(P, d_records, d_masks, d_xi, s, s1, s2, s3, p->sides ? p->sides->side[3] : nullptr, fj, p->n_order_c, p->n_order_d, p->n_order_w, p->n_order_r)) != hipSuccess) return hipfail(e, "launch solve");
    }
    if (g_timing) { if ((e = hipEventRecord(g_ev[2], s)) != hipSuccess) return hipfail(e, "hipEventRecord"); g_ev_valid = 1; }
    return SDSM_OK;
}

// ---- callable dsm/init (objects.py:385-386: params = init(number of columns of G~)) -----------------------------------------
// The number of columns of a candidate's G~ (its grid points, dsm.py:159-181) is known once the setup kernel has run: this runs
// it alone and returns the counts to the HOST (it synchronises the stream): n_deform[i] = M of candidate i, -1 for a candidate
// without a solve (trivial, failed setup).  The caller then builds the starting points and hands them over with sdsm_plan_set_start.
extern "C" int sdsm_batch_deform_counts(const sdsm_plan *p, const double *const *d_y, const int32_t *const *d_atoms, const uint8_t *const *d_valid,
                                        void *d_ws, size_t ws_bytes, int32_t *n_deform, void *stream)
{
    if (!p || !d_y || !d_atoms || !d_valid || !d_ws || !n_deform) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_deform_counts: null argument");
    for (size_t i = 0; i < p->images.size(); i++) if (!d_y[i] || !d_atoms[i] || !d_valid[i]) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_deform_counts: null image pointer");
    if (p->uploaded_gen != p->layout_gen || p->uploaded_ws != d_ws) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_deform_counts: the plan's tables in this workspace are missing or stale (sdsm_batch_upload)");
    if (ws_bytes < p->total) return fail(SDSM_ERR_WORKSPACE, "sdsm_batch_deform_counts: workspace too small");
    if (p->n == 0) return SDSM_OK;
    hipStream_t s = (hipStream_t)stream;
    BatchParams P = make_params(p, d_ws);
    for (size_t i = 0; i < p->images.size(); i++) { P.img[i].y = d_y[i]; P.img[i].atoms = d_atoms[i]; P.img[i].valid = d_valid[i]; }
    hipError_t e;
    const int32_t *lists = P.order + p->n + p->n_order_c + p->n_order_d;
    if ((e = sdsm_launch_setup(P, s, lists, p->n_order_w, p->setup_class, lists + p->n_order_w + p->n_order_r, p->n_setup_small, lists + p->n_order_w + p->n_order_r + p->n_setup_small, p->n_setup_big)) != hipSuccess)
        return hipfail(e, "launch setup");
    std::vector<CandState> st(p->n);
    if ((e = hipMemcpyAsync(st.data(), (const uint8_t *)d_ws + p->off_state, sizeof(CandState) * p->n, hipMemcpyDeviceToHost, s)) != hipSuccess) return hipfail(e, "hipMemcpyAsync");
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return hipfail(e, "hipStreamSynchronize");
    for (int i = 0; i < p->n; i++) n_deform[i] = st[i].status == ST_OK ? (6 + st[i].M > SDSM_MAX_N_GLOBAL ? 0 : st[i].M) : -1;   // (beyond the solver's limit: elliptical model only)
    return SDSM_OK;
}

// Starting points of the DSM solves of the following launches of this plan: d_x0 holds sdsm_plan_eval_param_count() doubles in the layout of
// sdsm_batch_eval's d_params (candidate i: theta[6] in full-image-normalised coordinates, then xi[M], at 6 i + xi_offset[i]) and must stay
// valid until those launches have completed; null = none.  Only plans created with init_elliptical = 0 read it; a fallback
// (SDSM_CAND_FALLBACK) then returns this initialisation, as the reference does (objects.py:409-410).
extern "C" int sdsm_plan_set_start(const sdsm_plan *p, const double *d_x0)
{
    if (!p) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_set_start: null plan");
    if (d_x0 && p->cfg.init_elliptical) return fail(SDSM_ERR_ARGUMENT, "sdsm_plan_set_start: the plan solves the elliptical model first (init_elliptical = 1)");
    p->x0 = d_x0;
    return SDSM_OK;
}

extern "C" int sdsm_batch_launch(const sdsm_plan *p, const double *d_y, const int32_t *d_atoms, const uint8_t *d_valid,
                                 void *d_ws, size_t ws_bytes, sdsm_record *d_records, uint32_t *d_masks, double *d_xi, void *stream)
{
    if (p && p->images.size() != 1) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_launch: the plan covers several images, use sdsm_batch_launch_multi");
    return sdsm_batch_launch_multi(p, &d_y, &d_atoms, &d_valid, d_ws, ws_bytes, d_records, d_masks, d_xi, stream);
}

// ---- point evaluation for parity tests ---------------------------------------------------------------------
extern "C" hipError_t sdsm_launch_eval(const BatchParams &P, const double *params, double *out, hipStream_t stream);

extern "C" int64_t sdsm_plan_eval_param_count(const sdsm_plan *p) { return p ? (int64_t)6 * p->n + std::max<int64_t>(p->total_xi, 1) : 0; }
extern "C" int64_t sdsm_plan_eval_out_count(const sdsm_plan *p) { return p ? (int64_t)29 * p->n + std::max<int64_t>(p->total_xi, 1) : 0; }

extern "C" int sdsm_batch_eval(const sdsm_plan *p, void *d_ws, size_t ws_bytes, const double *d_params, double *d_out, void *stream)
{
    if (!p || !d_ws || !d_params || !d_out) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_eval: null argument");
    if (ws_bytes < p->total) return fail(SDSM_ERR_WORKSPACE, "sdsm_batch_eval: workspace too small");
    if (p->n == 0) return SDSM_OK;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    // results of candidates that cannot be evaluated (trivial, failed setup, beyond the solver's limits) stay NaN
    if ((e = hipMemsetAsync(d_out, 0xff, sizeof(double) * (size_t)sdsm_plan_eval_out_count(p), s)) != hipSuccess) return hipfail(e, "hipMemsetAsync");
    if (p->uploaded_gen != p->layout_gen || p->uploaded_ws != d_ws) return fail(SDSM_ERR_ARGUMENT, "sdsm_batch_eval: needs the workspace of a previous sdsm_batch_launch of this plan");
    const BatchParams P = make_params(p, d_ws);
    if ((e = sdsm_launch_eval(P, d_params, d_out, s)) != hipSuccess) return hipfail(e, "launch eval");
    return SDSM_OK;
}

// ---- host helper: foreground fragments out of the bit-packed masks (objects.py:148-174 applied to the region-bbox masks) -------
// One byte per pixel of every fragment into `out` (fragment i: fg_h * fg_w bytes, row-major, at out_offset[i]); candidates
// without a foreground (fg_h <= 0, trivial, failed) get the single byte 0 = [[False]] (objects.py:172-174, 185-186).
// Returns the number of bytes written (or needed when out == NULL), < 0 on error.
extern "C" int64_t sdsm_unpack_fragments(const sdsm_record *records, const int32_t *mask_info, const int64_t *mask_offset, const uint8_t *masks,
                                         int n, uint8_t *out, int64_t *out_offset)
{
    if (n < 0 || (n > 0 && (!records || !mask_info || !mask_offset || !masks))) return fail(SDSM_ERR_ARGUMENT, "sdsm_unpack_fragments: null argument");
    int64_t pos = 0;
    for (int i = 0; i < n; i++) {
        const sdsm_record &r = records[i];
        const bool empty = r.fg_h <= 0 || r.status == SDSM_CAND_TRIVIAL || r.status == SDSM_CAND_ERROR || r.status == SDSM_CAND_GIVEN_UP;
        if (out_offset) out_offset[i] = pos;
        if (empty) { if (out) out[pos] = 0; pos += 1; continue; }
        const int r0 = mask_info[4 * i], c0 = mask_info[4 * i + 1], h = mask_info[4 * i + 2], w = mask_info[4 * i + 3];
        const int fr = r.fg_r0 - r0, fc = r.fg_c0 - c0;
        if (fr < 0 || fc < 0 || fr + r.fg_h > h || fc + r.fg_w > w) return fail(SDSM_ERR_ARGUMENT, "sdsm_unpack_fragments: fragment outside its mask box");
        if (out) {
            const uint32_t *words = reinterpret_cast<const uint32_t *>(masks + mask_offset[i]);
            uint8_t *o = out + pos;
            for (int a = 0; a < r.fg_h; a++) {
                int64_t bit = (int64_t)(fr + a) * w + fc;
                for (int b = 0; b < r.fg_w; b++, bit++) *o++ = (uint8_t)((words[bit >> 5] >> (bit & 31)) & 1u);
            }
        }
        pos += (int64_t)r.fg_h * r.fg_w;
    }
    return pos;
}

// ---- post-processing, per-object work ----------------------------------------------------------------------------------------
extern "C" hipError_t sdsm_launch_post(const double *g, const double *gs, const uint8_t *bg, int H, int W, int n, const int32_t *boxes,
                                       const int64_t *bits_off, const uint32_t *bits, const int64_t *new_off, uint32_t *new_bits,
                                       uint32_t *boundary_pool, const int64_t *bpool_off, double exterior_scale, double exterior_offset,
                                       double contrast_epsilon, double inv_gstd, int max_distance, double stdamp, sdsm_post_record *out, hipStream_t stream);
extern "C" hipError_t sdsm_gaussian_filter_impl(const double *d_in, int H, int W, double sigma, double *d_out, void *d_ws, hipStream_t stream);
extern "C" size_t sdsm_gaussian_workspace_bytes(int H, int W, double sigma);

extern "C" int sdsm_post_objects(const double *d_g, const double *d_gs, const uint8_t *d_bg, int H, int W, int n, const int32_t *d_boxes,
                                 const int64_t *d_bits_off, const uint32_t *d_bits, const int64_t *d_new_off, uint32_t *d_new_bits,
                                 uint32_t *d_boundary_pool, const int64_t *d_bpool_off, double exterior_scale, double exterior_offset,
                                 double contrast_epsilon, double inv_gstd, int max_distance, double stdamp, sdsm_post_record *d_out, void *stream)
{
    if (n < 0 || H < 1 || W < 1 || H > 65535 || W > 65535) return fail(SDSM_ERR_ARGUMENT, "sdsm_post_objects: bad shape");
    if (n == 0) return SDSM_OK;
    if (!d_g || !d_gs || !d_bg || !d_boxes || !d_bits_off || !d_bits || !d_out) return fail(SDSM_ERR_ARGUMENT, "sdsm_post_objects: null argument");
    if (!(exterior_scale > 0) || !(exterior_offset >= 0) || max_distance < 0 || max_distance > 16) return fail(SDSM_ERR_ARGUMENT, "sdsm_post_objects: exterior_scale > 0, exterior_offset >= 0, 0 <= max_distance <= 16 required");
    if (max_distance > 0 && stdamp > 0 && (!d_new_off || !d_new_bits)) return fail(SDSM_ERR_ARGUMENT, "sdsm_post_objects: refinement needs the output mask buffers");
    if ((d_boundary_pool == nullptr) != (d_bpool_off == nullptr)) return fail(SDSM_ERR_ARGUMENT, "sdsm_post_objects: boundary pool and its offsets go together");
    hipError_t e = sdsm_launch_post(d_g, d_gs, d_bg, H, W, n, d_boxes, d_bits_off, d_bits, d_new_off, d_new_bits, d_boundary_pool, d_bpool_off,
                                    exterior_scale, exterior_offset, contrast_epsilon, inv_gstd, max_distance, stdamp, d_out, (hipStream_t)stream);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "sdsm_post_objects");
}

extern "C" int sdsm_gaussian_filter(const double *d_in, int H, int W, double sigma, double *d_out, void *d_ws, size_t ws_bytes, void *stream)
{
    if (!d_in || !d_out || !d_ws || H < 1 || W < 1 || !(sigma > 0)) return fail(SDSM_ERR_ARGUMENT, "sdsm_gaussian_filter: bad argument");
    if (ws_bytes < sdsm_gaussian_workspace_bytes(H, W, sigma)) return fail(SDSM_ERR_WORKSPACE, "sdsm_gaussian_filter: workspace too small");
    hipError_t e = sdsm_gaussian_filter_impl(d_in, H, W, sigma, d_out, d_ws, (hipStream_t)stream);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "sdsm_gaussian_filter");
}

extern "C" size_t sdsm_separable_workspace_bytes(int H, int W, int R0, int R1);
extern "C" hipError_t sdsm_separable_filter_impl(const double *d_in, int H, int W, const double *h_w0, int R0, const double *h_w1, int R1,
                                                 double *d_out, void *d_ws, hipStream_t stream);
extern "C" int sdsm_separable_filter(const double *d_in, int H, int W, const double *h_w0, int R0, const double *h_w1, int R1,
                                     double *d_out, void *d_ws, size_t ws_bytes, void *stream)
{
    if (!d_in || !d_out || !d_ws || !h_w0 || !h_w1 || H < 1 || W < 1 || R0 < 0 || R1 < 0) return fail(SDSM_ERR_ARGUMENT, "sdsm_separable_filter: bad argument");
    for (int j = 1; j <= R0; j++) if (h_w0[R0 - j] != h_w0[R0 + j]) return fail(SDSM_ERR_ARGUMENT, "sdsm_separable_filter: weights must be symmetric");
    for (int j = 1; j <= R1; j++) if (h_w1[R1 - j] != h_w1[R1 + j]) return fail(SDSM_ERR_ARGUMENT, "sdsm_separable_filter: weights must be symmetric");
    if (ws_bytes < sdsm_separable_workspace_bytes(H, W, R0, R1)) return fail(SDSM_ERR_WORKSPACE, "sdsm_separable_filter: workspace too small");
    hipError_t e = sdsm_separable_filter_impl(d_in, H, W, h_w0, R0, h_w1, R1, d_out, d_ws, (hipStream_t)stream);
    return e == hipSuccess ? SDSM_OK : hipfail(e, "sdsm_separable_filter (a filter radius beyond ~2500 does not fit the LDS tiles)");
}
