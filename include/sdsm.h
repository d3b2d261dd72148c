/*
 * sdsm.h -- C ABI of the MI355X-native SuperDSM hot path (libsdsm_hip.so).
 *
 * The reference (BMCV/SuperDSM @ 2024_08_07) has no FFI for this path: its only native boundary is MKL via
 * ctypes (superdsm/_mkl.py:1-8, superdsm/_libs/sparse_dot_mkl/_mkl_interface.py:6-135) and cvxopt's C
 * extension behind `cvxopt.solvers.cp` (superdsm/dsm.py:488).  This header is the boundary a maintainer
 * would bind instead (see INTEGRATION.md for the ctypes stub): it replaces, per entry point,
 *
 *   sdsm_preprocess            Preprocessing.process                superdsm/preprocess.py:39-68
 *   sdsm_image_prepare         the candidate-independent half of    superdsm/objects.py:95-128
 *                              Object.get_cvxprog_region (EDT(y<=0) <= margin, per-atom extents)
 *   sdsm_plan_* / sdsm_batch_* compute_objects / _compute_object    superdsm/objects.py:177-284
 *                              cvxprog + Energy + CP.solve          superdsm/objects.py:361-412, dsm.py:253-490
 *                              SmoothMatrixFactory.get              superdsm/dsm.py:137-237
 *   sdsm_dsm_config            DSM_CONFIG_DEFAULTS                  superdsm/dsmcfg.py:6-21
 *
 * Conventions: extern "C", plain pointers and sizes, no C++ / torch types.  Every function returns 0 on
 * success and a negative sdsm_status on failure; sdsm_last_error() returns a thread-local message.
 * All device buffers are allocated and freed by the caller (PyTorch-ROCm tensors on the Python side) and
 * passed as raw device pointers; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Calls on one stream are ordered; nothing here synchronises the device except where stated.
 * There is no CPU backend: without a HIP device every compute entry point fails with SDSM_ERR_DEVICE.
 */
#ifndef SDSM_H
#define SDSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDSM_VERSION 200

typedef enum {
    SDSM_OK = 0,
    SDSM_ERR_ARGUMENT = -1,
    SDSM_ERR_DEVICE = -2,      /* HIP runtime error / no device */
    SDSM_ERR_WORKSPACE = -3,   /* caller-provided buffer too small */
    SDSM_ERR_UNSUPPORTED = -4
} sdsm_status;

/* Hyper-parameters of the operator: DSM_CONFIG_DEFAULTS (dsmcfg.py:6-21) minus the keys that are
 * meaningless on the GPU (cachesize, cachetest, smooth_mat_max_allocations, smooth_mat_dtype = float32
 * construction is always used, cp_timeout is replaced by the iteration cap). */
typedef struct {
    double scale;                      /* dsm/scale (1000) */
    double epsilon;                    /* dsm/epsilon (1.0) */
    double alpha;                      /* dsm/alpha */
    double smooth_amount;              /* dsm/smooth_amount, sigma_G; +inf = elliptical models only (c2freganal.py:126) */
    double gaussian_shape_multiplier;  /* dsm/gaussian_shape_multiplier (2) */
    double background_margin;          /* dsm/background_margin */
    int32_t smooth_subsample;          /* dsm/smooth_subsample */
    int32_t init_elliptical;           /* dsm/init == 'elliptical' */
    int32_t max_iters;                 /* Newton iteration cap per solve (100 = cvxopt's maxiters) */
    int32_t flags;                     /* bit 0: no "single positive pixel" shortcut (objects.py:184-191) -- for callers that mirror a direct
                                          cvxprog call, e.g. the normalised energies of c2freganal.py:58-79 */
} sdsm_dsm_config;

/* One candidate's result (objects.py:198-211).  128 bytes, written by the device. */
typedef struct {
    double energy;          /* psi(result), unscaled (objects.py:208) */
    double theta[6];        /* a1,a2,a3,b1,b2,c in full-image-normalised coordinates (dsm.py:49-54) */
    double energy_ell;      /* psi of the elliptical solution the DSM solve started from */
    int32_t status;         /* sdsm_cand_status */
    int32_t flags;          /* bit0: elliptical retry from the moment initialisation (objects.py:337-355) */
    int32_t n_pixels;       /* N = region pixels */
    int32_t n_deform;       /* M = deformation parameters (grid points) */
    int32_t iters_ell, iters_dsm;   /* Newton iterations */
    int32_t evals_value;    /* pixel passes that computed psi only */
    int32_t evals_full;     /* pixel passes that computed psi, gradient and Hessian */
    int32_t on_boundary;    /* S > 0 anywhere on the 1-px pad ring (objects.py:209) */
    int32_t fg_r0, fg_c0, fg_h, fg_w;   /* bounding box of the foreground fragment; fg_h == 0: empty */
    int32_t n_positive;     /* region pixels with y > 0 */
    int32_t n_negative;     /* region pixels with y < 0 (C2F: a region whose pixels are all positive or all negative has no energy, c2freganal.py:67-68) */
    int32_t reserved;
} sdsm_record;

typedef enum {
    SDSM_CAND_OPTIMAL = 0,      /* is_optimal = True */
    SDSM_CAND_FALLBACK = 1,     /* DSM solve failed, elliptical result returned (objects.py:406-410) */
    SDSM_CAND_TRIVIAL = 2,      /* single positive pixel (objects.py:184-191): energy 0, is_optimal False */
    SDSM_CAND_ERROR = 3,        /* CvxprogError (objects.py:351-353) or malformed G~ (dsm.py:194) */
    SDSM_CAND_UNSUPPORTED = 4,  /* exceeds an implementation limit; elliptical result returned */
    SDSM_CAND_GIVEN_UP = 5      /* scheduling, not arithmetic: the workgroup group of a very large region waited too long for a member (GPU
                                   oversubscribed by other work) and gave the candidate up; no result.  Solve it again with groups
                                   disabled (sdsm_plan_set_latency_mode(plan, 2)); superdsm_amd.objects.compute_objects does. */
} sdsm_cand_status;

/* Per-atom statistics written by sdsm_image_prepare: for label l (1..n_atoms) six int32 at [6*l]:
 * area (pixels with atoms == l, y_mask and EDT(y<=0) <= margin), rmin, rmax, cmin, cmax, reserved. */
#define SDSM_ATOM_STATS_STRIDE 6

/* ---- library --------------------------------------------------------------------------------- */
int sdsm_version(void);
const char *sdsm_last_error(void);
int sdsm_device_count(void);
int sdsm_set_device(int device);
/* Blocks until all work queued on `stream` has finished. */
int sdsm_stream_synchronize(void *stream);

/* Gaussian PSF of G~ (dsm.py:137-142, float32 cast dsm.py:226).  Host function.  Returns k (the PSF is
 * k x k); writes k*k floats if `out` != NULL. */
int sdsm_psf(double sigma, double multiplier, float *out);

/* ---- preprocessing (preprocess.py:39-68) ------------------------------------------------------- */
size_t sdsm_preprocess_workspace_bytes(int H, int W, double sigma1, double sigma2);
/* d_g: H*W float64 in [0,1] (pipeline.py:192), d_y: H*W float64 out.  Device pointers. */
int sdsm_preprocess(const double *d_g, int H, int W, double sigma1, double sigma2, double offset_clip,
                    int lower_clip_mean, double *d_y, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- per-image preparation --------------------------------------------------------------------- */
size_t sdsm_image_workspace_bytes(int H, int W);
/* d_y float64 H*W, d_y_mask uint8 H*W (NULL = all True), d_atoms int32 H*W (labels 1..n_atoms, 0 = none).
 * Outputs: d_valid uint8 H*W = y_mask & (EDT(y <= 0) <= margin)   (objects.py:126-127; image.py:82)
 *          d_atom_stats int32 (n_atoms+1)*SDSM_ATOM_STATS_STRIDE. */
int sdsm_image_prepare(const double *d_y, const uint8_t *d_y_mask, const int32_t *d_atoms, int H, int W,
                       double background_margin, int n_atoms, uint8_t *d_valid, int32_t *d_atom_stats,
                       void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- batch of candidates (one compute_objects call) ------------------------------------------- */
typedef struct sdsm_plan sdsm_plan;   /* host-side plan: offsets into the workspace, launch order */

/* atom_stats: HOST copy of d_atom_stats.  offsets[n+1]/labels[]: footprints in CSR form (objects.py:53). */
sdsm_plan *sdsm_plan_create(int H, int W, int n_atoms, const int32_t *atom_stats, const sdsm_dsm_config *cfg,
                            int n, const int32_t *offsets, const int32_t *labels);
/* The same for candidates of SEVERAL images in one plan (1 <= n_images <= 16): one launch then fills the GPU even when the batches
 * of the single images are small (the generations of globalenergymin.py:228-263 are tens of candidates; an image set as in
 * examples/NIH3T3).  H / W / n_atoms / atom_stats: one entry per image; image_of[i]: the image of candidate i (NULL: all 0); the
 * labels of a footprint refer to that image's atoms.  All images share the hyper-parameters. */
sdsm_plan *sdsm_plan_create_multi(int n_images, const int32_t *H, const int32_t *W, const int32_t *n_atoms, const int32_t *const *atom_stats,
                                  const sdsm_dsm_config *cfg, int n, const int32_t *offsets, const int32_t *labels, const int32_t *image_of);
void sdsm_plan_destroy(sdsm_plan *plan);
size_t sdsm_plan_workspace_bytes(const sdsm_plan *plan);
size_t sdsm_plan_mask_bytes(const sdsm_plan *plan);       /* total size of the bit-packed region-bbox masks */
/* Per candidate i: mask_info[4*i..] = r0, c0, h, w of the region bounding box its mask bits cover (row-major,
 * LSB-first within uint32 words), mask_offset[i] = byte offset into the mask buffer; n_pixels[i] = N. */
int sdsm_plan_describe(const sdsm_plan *plan, int32_t *mask_info, int64_t *mask_offset, int32_t *n_pixels);
/* Algorithmic bytes of the setup phase (crop reads + writes), for the roofline bookkeeping. */
int64_t sdsm_plan_total_pixels(const sdsm_plan *plan);

/* Copies the plan tables into the workspace (H2D, asynchronous on `stream`). */
int sdsm_batch_upload(const sdsm_plan *plan, void *d_workspace, size_t workspace_bytes, void *stream);
/* Region crops, G~ rows, elliptical + DSM solves, masks, records.  Everything stays on the device:
 * d_records: n * sizeof(sdsm_record); d_masks: sdsm_plan_mask_bytes(); d_xi (optional, may be NULL):
 * float64, sdsm_plan_xi_count() entries, candidate i's xi at xi_offset[i] (debug / tests). */
int sdsm_batch_launch(const sdsm_plan *plan, const double *d_y, const int32_t *d_atoms, const uint8_t *d_valid,
                      void *d_workspace, size_t workspace_bytes, sdsm_record *d_records, uint32_t *d_masks,
                      double *d_xi, void *stream);
/* d_y / d_atoms / d_valid: HOST arrays of n_images device pointers (plans of sdsm_plan_create_multi). */
int sdsm_batch_launch_multi(const sdsm_plan *plan, const double *const *d_y, const int32_t *const *d_atoms, const uint8_t *const *d_valid,
                            void *d_workspace, size_t workspace_bytes, sdsm_record *d_records, uint32_t *d_masks, double *d_xi, void *stream);
int64_t sdsm_plan_xi_count(const sdsm_plan *plan);
/* Workspace layout for inspection (parity tests of the crops / grid / G~ rows).  out[16], byte offsets into the
 * workspace: 0 cand table, 1 cand state (M, status, ...), 2 crop_y f64, 3 crop_rc u32, 4 crop_cc u32 (setup order),
 * 5 ell_meta u32 (row entries | Hessian entries << 16), 6 grid u32, 7 ell_idx u16, 8 ell_w f32; then 9 zcap (entries
 * per row, a multiple of 4), 10 k, 11 sizeof(cand entry), 12 total_pixels, 13 total_ell entries, 14 sizeof(cand state);
 * 15 reserved.  Candidate i's blocks start at crop offset sum(n_pixels[:i]), G~ offset that * zcap -- entry s of crop
 * position p at element ((s / 4) * N + p) * 4 + s % 4 of the candidate's block -- and grid offset xi_offset[i]. */
int sdsm_plan_layout(const sdsm_plan *plan, int64_t *out);
/* Scheduling of one batch, mode =
 *   0  throughput (default): every candidate whose system fits is solved by a 192- or 256-thread workgroup, two to four per compute
 *      unit -- most candidate solves per second when the GPU is full (plans over several images, several batches in flight); only
 *      regions of more than max(8192, pixels of all candidates of the plan / 1024) pixels are solved by a GROUP of cooperating
 *      512-thread workgroups (the largest first, until the members add up to 256);
 *   1  latency: regions of more than 3072 pixels are solved by groups of 2-4 workgroups too (same member budget), which shortens
 *      the slowest candidates and with them the wall clock of a single batch (the reference waits for all candidates of an image
 *      before the set-cover step, globalenergymin.py:131-137);
 *   2  no groups: every candidate by a single workgroup (slow for very large regions; the way to solve candidates again whose
 *      group was given up, SDSM_CAND_GIVEN_UP).
 * Results do not depend on the mode.  It changes the launch lists and the workspace size: call it before
 * sdsm_plan_workspace_bytes / sdsm_batch_upload; a launch on a workspace uploaded before the change fails with SDSM_ERR_ARGUMENT. */
int sdsm_plan_set_latency_mode(sdsm_plan *plan, int mode);
/* What the current mode schedules per candidate (host only, diagnostics and tests): group_members[i] = workgroups of the group that
 * solves candidate i (0: a single workgroup), rows_workgroups[i] = workgroups that build its rows of G~ (0: the setup workgroup itself).
 * Either pointer may be NULL. */
int sdsm_plan_schedule(const sdsm_plan *plan, int32_t *group_members, int32_t *rows_workgroups);
int sdsm_plan_xi_offsets(const sdsm_plan *plan, int64_t *xi_offset);

/* ---- post-processing, per-object work (superdsm/postprocess.py:254-337) ------------------------------------- */
/* One object's results.  64 bytes, written by the device. */
typedef struct {
    double contrast;        /* (interior_mean + eps) / (exterior_mean + eps)                     postprocess.py:254-266 */
    double interior_mean;   /* mean of g / g.std() over the object's mask */
    double exterior_mean;   /* weighted mean over the exterior neighbourhood */
    double fg_mean, fg_std; /* mean / population std of the smoothed intensities over the mask   postprocess.py:323-325 */
    int32_t area;           /* mask pixels */
    int32_t status;         /* 0 ok, 1 boundary list too long and no pool slot, 2 empty mask, 3 ok, no refinement requested */
    int32_t r0, c0, h, w;   /* bounding box of the refined mask (h == 0: empty) */
} sdsm_post_record;
/* d_g: raw intensities (H*W float64); d_gs: Gaussian-smoothed intensities used by the mask refinement (postprocess.py:165);
 * d_bg: background_mask uint8 (postprocess.py:152-155).  Objects: d_boxes n*4 int32 (r0, c0, h, w of each fragment), fragments
 * bit-packed (row-major, LSB first in uint32 words) at d_bits + d_bits_off[i] (in words).  Outputs: d_out n records; the refined
 * masks over the windows box +- max_distance (clamped to the image), bit-packed at d_new_bits + d_new_off[i] (only written when
 * max_distance > 0 and stdamp > 0; hole filling stays on the host).  d_boundary_pool / d_bpool_off (may be NULL): global
 * boundary lists for objects whose mask boundary exceeds 12288 pixels (d_bpool_off[i] < 0: none).  inv_gstd = 1 / g.std(). */
int sdsm_post_objects(const double *d_g, const double *d_gs, const uint8_t *d_bg, int H, int W, int n, const int32_t *d_boxes,
                      const int64_t *d_bits_off, const uint32_t *d_bits, const int64_t *d_new_off, uint32_t *d_new_bits,
                      uint32_t *d_boundary_pool, const int64_t *d_bpool_off, double exterior_scale, double exterior_offset,
                      double contrast_epsilon, double inv_gstd, int max_distance, double stdamp, sdsm_post_record *d_out, void *stream);
/* Separable Gaussian filter with SciPy's defaults (mode 'reflect', truncate 4): the smoothing of postprocess.py:165-166 and the
 * building block of sdsm_preprocess. */
size_t sdsm_gaussian_workspace_bytes(int H, int W, double sigma);
int sdsm_gaussian_filter(const double *d_in, int H, int W, double sigma, double *d_out, void *d_workspace, size_t workspace_bytes, void *stream);
/* The same with caller-given SYMMETRIC weights per axis (HOST arrays of 2 R + 1 doubles): axis 0 with h_w0, then axis 1 with h_w1,
 * 'reflect' boundary -- e.g. the derivative-of-Gaussian filters of scipy.ndimage.gaussian_laplace used by the scale estimation
 * (superdsm/automation.py:52). */
size_t sdsm_separable_workspace_bytes(int H, int W, int R0, int R1);
int sdsm_separable_filter(const double *d_in, int H, int W, const double *h_w0, int R0, const double *h_w1, int R1,
                          double *d_out, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- host-side combinatorial steps of the stage (no device access) -------------------------------------------------------------
 * Approximate min-weight set cover (superdsm/minsetcover.py:4-88: greedy + merge phase, retried with beta * gamma on up to max_iter
 * levels) and greedy max-weight set packing (superdsm/maxsetpack.py:8-24) over n objects whose footprints are bit sets of `words`
 * uint64 each (bit = atom of the cluster); same decisions, arithmetic and tie-breaking as the reference's Python.  selected
 * receives the indices of the solution in the reference's list order, n_selected their number. */
int sdsm_minsetcover(int n, int words, const uint64_t *footprints, const double *energies, double beta, int merge, int max_iter,
                     double gamma, int32_t *selected, int32_t *n_selected);
int sdsm_maxsetpack(int n, int words, const uint64_t *footprints, const double *energies, int32_t *selected, int32_t *n_selected);
/* sdsm_minsetcover for several independent families in one call (MinSetCover.update, superdsm/minsetcover.py:142-153: one cover per touched
 * cluster): family f has n[f] objects of words[f] uint64; footprints / energies / selected of the families follow each other (n[f] *
 * words[f] / n[f] / n[f] entries); n_selected[f] = size of family f's solution. */
int sdsm_minsetcover_multi(int n_families, const int32_t *n, const int32_t *words, const uint64_t *footprints, const double *energies, double beta,
                           int merge, int max_iter, double gamma, int32_t *selected, int32_t *n_selected);

/* Host helper (no device access): size of the search space of the stage's iterations, per cluster -- what the reference's
 * _estimate_progress (superdsm/globalenergymin.py:310-323) enumerates footprint by footprint in Python from the generation of the
 * atoms on: the number of footprints that growing by one adjacent atom at a time produces (_iterate_generation,
 * globalenergymin.py:292-307; skip_last: the universe of a cluster is not counted).  Atoms of cluster k: offsets[k] .. offsets[k+1]-1;
 * adj / compat: per atom, the bit set (local indices within its cluster) of its adjacent atoms / of the atoms within
 * max_seed_distance of it (NULL: no limit).  counts[k] = -1 for clusters of more than 64 atoms.  Counting stops once the total
 * exceeds max_amount. */
int sdsm_count_growth(int n_clusters, const int32_t *offsets, const uint64_t *adj, const uint64_t *compat, int skip_last,
                      int64_t max_amount, int64_t *counts);

/* Host helper (no device access): the foreground fragments (objects.py:148-174) of a batch out of the downloaded records and
 * bit-packed masks, one byte per pixel: fragment i (fg_h x fg_w, row-major) at out + out_offset[i]; candidates without a
 * foreground get the single byte 0 ([[False]], objects.py:172-174).  Returns the bytes written -- or needed, when out == NULL. */
int64_t sdsm_unpack_fragments(const sdsm_record *records, const int32_t *mask_info, const int64_t *mask_offset, const uint8_t *masks,
                              int n, uint8_t *out, int64_t *out_offset);

/* Parity / debug: psi, its gradient and the polynomial (theta) block of its Hessian at caller-given parameters, computed
 * by the evaluators of the solve kernels (Energy.__call__ / grad / hessian, superdsm/dsm.py:312-385) on the crops and G~
 * rows that a previous sdsm_batch_launch of the same plan left in the workspace.  d_params: sdsm_plan_eval_param_count()
 * doubles, candidate i's vector (theta[6] in full-image-normalised coordinates, then xi[M]) at 6 * i + xi_offset[i].
 * d_out: sdsm_plan_eval_out_count() doubles: [2 i], [2 i + 1] psi by the full and by the value-only evaluator;
 * [2 n + 21 i ..] lower triangle (row-major) of the 6x6 theta block of the Hessian; [23 n + 6 i + xi_offset[i] ..] the
 * gradient in the layout of d_params.  Candidates without a solve (trivial, failed, beyond the limits) get NaN. */
/* Callable dsm/init (reference: superdsm/objects.py:385-386, `params = init(J.smooth_mat.shape[1])`): the caller's own starting point of
 * the DSM solve instead of the elliptical model's optimum.  The number of columns of a candidate's G~ -- its grid points -- is a result
 * of the setup kernel: sdsm_batch_deform_counts runs it alone (arguments as sdsm_batch_launch_multi), SYNCHRONISES the stream and writes
 * n_deform[i] = M of candidate i to host memory (-1: no solve -- trivial region, failed setup; 0 also for a system beyond the solver's
 * limit, which gets the elliptical model only).  sdsm_plan_set_start then names the starting points of the following launches: d_x0 =
 * sdsm_plan_eval_param_count() doubles on the device, candidate i's theta[6] (full-image-normalised) + xi[M] at 6 * i + xi_offset[i]
 * (the layout of sdsm_batch_eval's d_params); it must stay valid until those launches have completed; NULL = none.  Only for plans with
 * init_elliptical = 0 (SDSM_ERR_ARGUMENT otherwise); a candidate whose solve falls back (SDSM_CAND_FALLBACK) returns the initialisation. */
int sdsm_batch_deform_counts(const sdsm_plan *plan, const double *const *d_y, const int32_t *const *d_atoms, const uint8_t *const *d_valid,
                             void *d_workspace, size_t workspace_bytes, int32_t *n_deform, void *stream);
int sdsm_plan_set_start(const sdsm_plan *plan, const double *d_x0);
int64_t sdsm_plan_eval_param_count(const sdsm_plan *plan);
int64_t sdsm_plan_eval_out_count(const sdsm_plan *plan);
int sdsm_batch_eval(const sdsm_plan *plan, void *d_workspace, size_t workspace_bytes, const double *d_params, double *d_out, void *stream);

/* Timing of the dominant kernel with HIP events on the launch stream: after sdsm_batch_launch returns,
 * sdsm_last_solve_kernel_ms() synchronises on the recorded events and returns the solve kernels' duration. */
int sdsm_enable_kernel_timing(int enable);
double sdsm_last_solve_kernel_ms(void);
double sdsm_last_setup_kernel_ms(void);
/* Diagnostic builds only (-DSDSM_PROFILE): device buffer receiving 8 int64 cycle counters per candidate
 * (phase A, phase B, reductions, factor+solve, line search, total, elliptical total, reserved). */
int sdsm_set_debug_buffer(void *d_buf);
/* Microseconds a member of a workgroup group waits for its partners at one exchange before the group gives its candidate up
 * (SDSM_CAND_GIVEN_UP: the caller solves it again in a plan without groups, sdsm_plan_set_latency_mode(plan, 2)); <= 0 restores the
 * default of 50 ms.  Applies to the launches of the calling thread. */
int sdsm_set_group_timeout_us(double us);
/* A launch runs its solve classes on the caller's stream and three side streams of the library.  1: they ran side by side when the
 * library probed them (first launch that needed them); 0: they share hardware queues (GPU_MAX_HW_QUEUES was too small when the process
 * first touched the GPU) -- launches are correct but their classes run one after the other, a warning went to stderr once; -1: not
 * probed yet.  No reference counterpart (the reference's parallelism is Ray, objects.py:270-284). */
int sdsm_side_queues_distinct(void);

#ifdef __cplusplus
}
#endif
#endif /* SDSM_H */
