// Per-object work of the post-processing stage, one workgroup per object, one launch per image.
//
// Reference behaviour restated here (never its code):
//   contrast response           superdsm/postprocess.py:254-266  (_compute_contrast)
//   mask refinement             superdsm/postprocess.py:316-337  (_process_mask, up to the hole filling, which stays on the host)
// The reference evaluates both on FULL-IMAGE arrays per object (a Euclidean distance transform of the whole image for every
// object).  Neither needs more than a window around the object: the exterior weights vanish beyond exterior_offset + 5 *
// exterior_scale pixels from the mask, the refinement only touches pixels within mask_max_distance of its boundary.
//   exterior distance of a pixel = sqrt of the smallest integer squared distance to a mask pixel -- what
//   scipy.ndimage.distance_transform_edt returns; the nearest mask pixel of an outside pixel is a boundary pixel of the mask, so
//   the minimum runs over the boundary list (LDS) only.
#include "sdsm_common.h"

namespace {

#define POST_WG 256
#define POST_MAX_BOUNDARY 12288          // boundary pixels kept in LDS (48 KB); larger objects use the global list

struct PostParams {
    int32_t H, W, n;
    int32_t max_distance;                // postprocess/mask_max_distance (disk radius), 0: no refinement
    double exterior_scale, exterior_offset, contrast_epsilon, inv_gstd;   // inv_gstd = 1 / g.std() (postprocess.py:255)
    double stdamp;
    const double *g, *gs;                // raw intensities; Gaussian-smoothed intensities of the refinement (postprocess.py:165)
    const uint8_t *bg;                   // background_mask (postprocess.py:152-155)
    const int32_t *boxes;                // n x 4: r0, c0, h, w of the fragments
    const int64_t *bits_off;             // first uint32 word of each fragment's bits (row-major, LSB first)
    const uint32_t *bits;
    const int64_t *new_off;              // first word of each refined mask (window = box +- max_distance, clamped to the image)
    uint32_t *new_bits;
    uint32_t *boundary_pool;             // global boundary lists for objects beyond POST_MAX_BOUNDARY: bpool_off[i] .. (may be null)
    const int64_t *bpool_off;
    sdsm_post_record *out;
};

__device__ __forceinline__ bool frag_bit(const uint32_t *bits, int h, int w, int r, int c)
{
    if (r < 0 || c < 0 || r >= h || c >= w) return false;
    const int b = r * w + c;
    return (bits[b >> 5] >> (b & 31)) & 1u;
}

}  // namespace

__global__ __launch_bounds__(POST_WG) void sdsm_k_post(PostParams P)
{
    __shared__ uint32_t bnd[POST_MAX_BOUNDARY];
    __shared__ double red[POST_WG / 64 * 8];
    __shared__ int nb_sh, ired[POST_WG / 64];
    const int tid = threadIdx.x, i = blockIdx.x;
    const int r0 = P.boxes[4 * i], c0 = P.boxes[4 * i + 1], h = P.boxes[4 * i + 2], w = P.boxes[4 * i + 3];
    const uint32_t *bits = P.bits + P.bits_off[i];
    if (tid == 0) nb_sh = 0;
    __syncthreads();

    // ---- A. interior statistics and the boundary list ------------------------------------------------------------------------
    double s_g = 0, s_gs = 0, s_gs2 = 0;
    int cnt = 0;
    uint32_t *blist = bnd;
    const bool use_pool = P.boundary_pool != nullptr && P.bpool_off[i] >= 0;
    if (use_pool) blist = P.boundary_pool + P.bpool_off[i];
    for (int e = tid; e < h * w; e += POST_WG) {
        const int r = e / w, c = e - r * w;
        if (!frag_bit(bits, h, w, r, c)) continue;
        const size_t p = (size_t)(r0 + r) * P.W + (c0 + c);
        const double gv = P.g[p], sv = P.gs[p];
        s_g += gv; s_gs += sv; s_gs2 += sv * sv; cnt++;
        if (!(frag_bit(bits, h, w, r - 1, c) && frag_bit(bits, h, w, r + 1, c) && frag_bit(bits, h, w, r, c - 1) && frag_bit(bits, h, w, r, c + 1))) {
            const int k = atomicAdd(&nb_sh, 1);
            if (use_pool || k < POST_MAX_BOUNDARY) blist[k] = ((uint32_t)(r0 + r) << 16) | (uint32_t)(c0 + c);
        }
    }
    double v4[4] = {s_g, s_gs, s_gs2, (double)cnt};
#pragma unroll
    for (int k = 0; k < 4; k++) v4[k] = wave_sum(v4[k]);
    __syncthreads();
    if ((tid & 63) == 0) for (int k = 0; k < 4; k++) red[(tid >> 6) * 8 + k] = v4[k];
    __syncthreads();
    double tot[4] = {0, 0, 0, 0};
    for (int wv = 0; wv < POST_WG / 64; wv++) for (int k = 0; k < 4; k++) tot[k] += red[wv * 8 + k];
    const int nb = nb_sh;
    const double area = tot[3];
    sdsm_post_record rec = {};
    rec.area = (int32_t)area;
    if (!use_pool && nb > POST_MAX_BOUNDARY) {           // the host did not reserve a global list for this object
        if (tid == 0) { rec.status = 1; P.out[i] = rec; }
        return;
    }
    if (area == 0) { if (tid == 0) { rec.status = 2; P.out[i] = rec; } return; }
    const double interior_mean = (tot[0] / area) * P.inv_gstd;
    const double fg_mean = tot[1] / area;
    double var = tot[2] / area - fg_mean * fg_mean;       // population variance (numpy std)
    var = var < 0 ? 0 : var;
    const double fg_amp = sqrt(var) * P.stdamp;
    __syncthreads();

    // ---- B. exterior mean (postprocess.py:260-265): weights exp(-max(0, d - offset) / scale) where that exponent is <= 5 ------
    const double reach = P.exterior_offset + 5.0 * P.exterior_scale;
    const int D = (int)ceil(reach);
    const int wr0 = r0 - D < 0 ? 0 : r0 - D, wc0 = c0 - D < 0 ? 0 : c0 - D;
    const int wr1 = r0 + h + D > P.H ? P.H : r0 + h + D, wc1 = c0 + w + D > P.W ? P.W : c0 + w + D;
    const int ww = wc1 - wc0, wh = wr1 - wr0;
    double s_w = 0, s_gw = 0;
    for (int e = tid; e < wh * ww; e += POST_WG) {
        const int r = wr0 + e / ww, c = wc0 + e % ww;
        if (frag_bit(bits, h, w, r - r0, c - c0)) continue;              // xor with the mask: mask pixels have distance 0 <= 5
        const size_t p = (size_t)r * P.W + c;
        if (!P.bg[p]) continue;
        long long best = 1ll << 60;
        for (int k = 0; k < nb; k++) {
            const uint32_t q = blist[k];
            const long long dr = (long long)(q >> 16) - r, dc = (long long)(q & 0xffffu) - c;
            const long long d2 = dr * dr + dc * dc;
            best = d2 < best ? d2 : best;
        }
        double t = sqrt((double)best) - P.exterior_offset;
        t = (t < 0 ? 0 : t) / P.exterior_scale;
        if (t <= 5.0) {
            const double wgt = exp(-t);
            s_w += wgt; s_gw += wgt * (P.g[p] * P.inv_gstd);
        }
    }
    double v2[2] = {s_w, s_gw};
#pragma unroll
    for (int k = 0; k < 2; k++) v2[k] = wave_sum(v2[k]);
    __syncthreads();
    if ((tid & 63) == 0) for (int k = 0; k < 2; k++) red[(tid >> 6) * 8 + k] = v2[k];
    __syncthreads();
    double sw = 0, sgw = 0;
    for (int wv = 0; wv < POST_WG / 64; wv++) { sw += red[wv * 8]; sgw += red[wv * 8 + 1]; }
    const double exterior_mean = sgw / sw;               // 0 / 0 = NaN when nothing qualifies, as the reference's division does
    rec.interior_mean = interior_mean; rec.exterior_mean = exterior_mean;
    rec.contrast = (interior_mean + P.contrast_epsilon) / (exterior_mean + P.contrast_epsilon);
    rec.fg_mean = fg_mean; rec.fg_std = sqrt(var);

    // ---- C. mask refinement (postprocess.py:316-337): pixels within disk(max_distance) of the mask boundary join the mask iff
    //      their smoothed intensity lies within stdamp standard deviations of the mask's mean ------------------------------------
    const int m = P.max_distance;
    int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
    if (m > 0 && P.stdamp > 0) {
        const int nr0 = r0 - m < 0 ? 0 : r0 - m, nc0 = c0 - m < 0 ? 0 : c0 - m;
        const int nr1 = r0 + h + m > P.H ? P.H : r0 + h + m, nc1 = c0 + w + m > P.W ? P.W : c0 + w + m;
        const int nw = nc1 - nc0, nh = nr1 - nr0;
        uint32_t *nbits = P.new_bits + P.new_off[i];
        __syncthreads();
        for (int e = tid; e < (nh * nw + 31) / 32; e += POST_WG) nbits[e] = 0;
        __syncthreads();
        for (int e = tid; e < nh * nw; e += POST_WG) {
            const int r = nr0 + e / nw, c = nc0 + e % nw;
            const bool in = frag_bit(bits, h, w, r - r0, c - c0);
            bool any = false, all = true;                                 // dilation / erosion by the disk footprint
            for (int dr = -m; dr <= m; dr++) for (int dc = -m; dc <= m; dc++) {
                if (dr * dr + dc * dc > m * m) continue;
                const int rr = r + dr, cc = c + dc;
                const bool inside_image = rr >= 0 && cc >= 0 && rr < P.H && cc < P.W;
                const bool b = inside_image ? frag_bit(bits, h, w, rr - r0, cc - c0) : false;
                any |= b;
                all &= inside_image ? b : true;                            // erosion: outside the image counts as foreground
            }
            bool v = in;
            if (any != all) {                                              // dilation xor erosion (erosion implies dilation)
                const double sv = P.gs[(size_t)r * P.W + c];
                v = fg_mean - fg_amp <= sv && sv <= fg_mean + fg_amp;
            }
            if (v) {
                atomicOr(&nbits[e >> 5], 1u << (e & 31));
                rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax; cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
            }
        }
        rmin = block_min_i32<POST_WG / 64>(rmin, ired); cmin = block_min_i32<POST_WG / 64>(cmin, ired);
        rmax = -block_min_i32<POST_WG / 64>(-rmax, ired); cmax = -block_min_i32<POST_WG / 64>(-cmax, ired);
        if (rmax >= 0) { rec.r0 = rmin; rec.c0 = cmin; rec.h = rmax - rmin + 1; rec.w = cmax - cmin + 1; }
    } else { rec.r0 = r0; rec.c0 = c0; rec.h = h; rec.w = w; rec.status = 3; }   // no refinement on the device
    if (tid == 0) P.out[i] = rec;
}

extern "C" hipError_t sdsm_launch_post(const double *g, const double *gs, const uint8_t *bg, int H, int W, int n, const int32_t *boxes,
                                       const int64_t *bits_off, const uint32_t *bits, const int64_t *new_off, uint32_t *new_bits,
                                       uint32_t *boundary_pool, const int64_t *bpool_off, double exterior_scale, double exterior_offset,
                                       double contrast_epsilon, double inv_gstd, int max_distance, double stdamp, sdsm_post_record *out, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    PostParams P{};
    P.H = H; P.W = W; P.n = n; P.max_distance = max_distance;
    P.exterior_scale = exterior_scale; P.exterior_offset = exterior_offset; P.contrast_epsilon = contrast_epsilon; P.inv_gstd = inv_gstd; P.stdamp = stdamp;
    P.g = g; P.gs = gs; P.bg = bg; P.boxes = boxes; P.bits_off = bits_off; P.bits = bits; P.new_off = new_off; P.new_bits = new_bits;
    P.boundary_pool = boundary_pool; P.bpool_off = bpool_off; P.out = out;
    hipLaunchKernelGGL(sdsm_k_post, dim3(n), dim3(POST_WG), 0, stream, P);
    return hipGetLastError();
}
