// Per-candidate convex solve: elliptical model, deformable shape model, final energy, mask, record.
// One persistent 256-thread workgroup per candidate; all solver state lives in LDS.
//
// Reference behaviour restated here (never its code):
//   surface S, energy psi, gradient, Hessian   superdsm/dsm.py:86-94, 291-385   
//   solve protocol (elliptical, retry, DSM, fallback)   superdsm/objects.py:321-412   
//   moment initialisation                       superdsm/objects.py:287-296, dsm.py:96-111 (A7)
//   mask tail                                   superdsm/objects.py:198-209, dsm.py:113-128 (A9)
// The reference hands scale*psi to cvxopt.solvers.cp (dsm.py:488); cvxopt is a third-party dependency
// that is not part of the reference tree.  The solver here is the damped Newton iteration specified in
// DESIGN.md ("Solver"), identical step for step to the oracle's orc_newton, but run in a centred and
// scaled local polynomial basis (psi is invariant under affine re-parametrisation of theta).
//
// Work decomposition of one full evaluation (psi, gradient, Hessian):
//   phase A  lane = pixel: coalesced read of the packed crop (y f64 + (row,col) u16x2) and of the pixel's
//            ELL row of G~, S, exp/log, residual r and curvature weight d; the pixel's dense Jacobian row is
//            staged in LDS (float32, column-major, conflict-free).
//   phase B  lane = (4x4 Hessian tile, pixel slice): register-tiled accumulation of J^T diag(d) J and J^T r
//            over the staged pixels in float64; slices are combined with wave shuffles.
//   psi and the 6 polynomial gradient entries are reduced with wavefront shuffles + one LDS hop.
#include "sdsm_common.h"

extern __shared__ __align__(16) unsigned char sdsm_smem[];

namespace {

#define LOG_DBL_MAX 709.782712893384
#define NEWTON_ABSTOL 1e-7
#define NEWTON_RELTOL 1e-6
#define LS_ALPHA 0.01
#define LS_BETA 0.5
#define LS_MAX 40

// Compile-time LDS layout (offsets in doubles from the start of dynamic LDS).
template <int NMAX, int B, bool INPLACE>
struct Lay {
    static constexpr int NP = NMAX * (NMAX + 1) / 2;
    static constexpr int X = 0, G = NMAX, D = 2 * NMAX, XT = 3 * NMAX, SC = 4 * NMAX, YROW = 5 * NMAX, TMP = 6 * NMAX;
    static constexpr int RED = 7 * NMAX + 2, FLAG = RED + 8, DW = FLAG + 2, RW = DW + B, HP = RW + B;
    static constexpr int LP = INPLACE ? HP : HP + NP;
    static constexpr int END = LP + NP;
    static constexpr int V_BYTES = ((END * 8 + 15) / 16) * 16;
    static constexpr int LDV = ((NMAX + 3) / 4) * 4;
    static constexpr int TOTAL_BYTES = V_BYTES + LDV * B * 4;
};

#define SD ((double *)sdsm_smem)

struct Cand {                       // per-candidate global pointers (already offset) and scalars
    const double *crop_y;
    const uint32_t *crop_rc;
    const uint16_t *ell_idx;
    const float *ell_w;
    const uint16_t *ell_nnz;
    double *hsave;
    int N;
    double rmid, cmid, inv_hr, inv_hc;   // local coordinates u = (r - rmid) * inv_hr
    double scale, epsilon, alpha;
};

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

struct PixelEval { double q0, q1, q2, q3, q4, phi, r, dcurv; int nnz; };

// One pixel's surface value and loss terms.  xo: LDS offset (doubles) of the parameter vector (local basis).
template <bool NEED_DERIV>
__device__ __forceinline__ PixelEval eval_pixel(const Cand &c, int xo, int M, int p)
{
    PixelEval e;
    const double *xv = SD + xo;
    double yv = c.crop_y[p];
    uint32_t rc = c.crop_rc[p];
    double u = ((double)(rc >> 16) - c.rmid) * c.inv_hr, v = ((double)(rc & 0xffffu) - c.cmid) * c.inv_hc;
    e.q0 = u * u; e.q1 = v * v; e.q2 = 2 * (u * v); e.q3 = 2 * u; e.q4 = 2 * v;
    double S = e.q0 * xv[0] + e.q1 * xv[1] + e.q2 * xv[2] + e.q3 * xv[3] + e.q4 * xv[4] + xv[5];
    int nnz = 0;
    if (M > 0) {
        nnz = c.ell_nnz[p];
        double gx = 0;
        for (int s = 0; s < nnz; s++) {
            size_t o = (size_t)s * c.N + p;
            gx += (double)c.ell_w[o] * xv[6 + c.ell_idx[o]];
        }
        S += gx;
    }
    e.nnz = nnz;
    double t = yv * S, theta;
    if (t >= -LOG_DBL_MAX) {            // dsm.py:298-300
        double h = exp(-t);
        e.phi = log(1 + h);             // dsm.py:321
        theta = h / (1 + h);            // dsm.py:310
    } else { e.phi = -t; theta = 1; }   // dsm.py:322,309
    if (NEED_DERIV) {
        e.r = -yv * theta;                              // dsm.py:344
        e.dcurv = yv * yv * (theta - theta * theta);    // y^2 kappa, dsm.py:361-366
    }
    return e;
}

// psi only (line search, final energy).  Result broadcast to all threads.
template <class L>
__device__ __noinline__ double eval_value(const Cand &c, int xo, int M)
{
    double psi = 0;
    for (int p = threadIdx.x; p < c.N; p += SDSM_WG) psi += eval_pixel<false>(c, xo, M, p).phi;
    psi = block_sum(psi, SD + L::RED);
    if (M > 0) {                                         // dsm.py:323-331
        const double *xv = SD + xo;
        double s2 = 0;
        for (int j = threadIdx.x; j < M; j += SDSM_WG) s2 += sqrt(xv[6 + j] * xv[6 + j] + c.epsilon);
        s2 = block_sum(s2, SD + L::RED);
        double o2 = c.alpha * s2 - c.alpha * sqrt(c.epsilon) * M;
        psi += o2 < 0 ? 0 : o2;
    }
    return psi;
}

// Full evaluation at the parameters stored at L::X: psi returned (broadcast); gradient at L::G and packed lower
// Hessian at L::HP filled, unscaled.
template <class L, int TPL, int B>
__device__ __forceinline__ double eval_full(const Cand &c, int M)
{
    const int tid = threadIdx.x;
    const int n = 6 + M, nb = (n + 3) / 4, ldv = nb * 4;
    const int Th = nb * (nb + 1) / 2, T = Th + nb;
    float *V = (float *)(sdsm_smem + L::V_BYTES);
    double *dW = SD + L::DW, *rW = SD + L::RW;
    // lane -> (tile, slice): few tiles -> several pixel slices per tile (power of two, consecutive lanes);
    // many tiles -> TPL tiles per lane.
    const bool sliced = T <= SDSM_WG;
    int S = 1;
    if (sliced) { while (S * 2 * T <= SDSM_WG && S < 64) S *= 2; }
    const int slice = tid % S;
    int tileJ[TPL], tileK[TPL];
    bool tileOn[TPL];
#pragma unroll
    for (int q = 0; q < TPL; q++) {
        int t = sliced ? (q == 0 ? tid / S : T) : tid + q * SDSM_WG;
        tileOn[q] = t < T;
        if (t < Th) {
            int J = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while (J * (J + 1) / 2 > t) J--;
            while ((J + 1) * (J + 2) / 2 <= t) J++;
            tileJ[q] = J; tileK[q] = t - J * (J + 1) / 2;
        } else { tileJ[q] = t - Th; tileK[q] = -1; }       // gradient tile
        if (!tileOn[q]) { tileJ[q] = 0; tileK[q] = 0; }
    }
    double acc[TPL][16];
#pragma unroll
    for (int q = 0; q < TPL; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[q][e] = 0;

    double psi = 0, gq0 = 0, gq1 = 0, gq2 = 0, gq3 = 0, gq4 = 0, gq5 = 0;
    for (int base = 0; base < c.N; base += B) {
        // ---- phase A: lane = pixel ---------------------------------------------------------------
        for (int lp = tid; lp < B; lp += SDSM_WG) {
            int p = base + lp;
            if (p < c.N) {
                PixelEval e = eval_pixel<true>(c, L::X, M, p);
                psi += e.phi;
                gq0 += e.r * e.q0; gq1 += e.r * e.q1; gq2 += e.r * e.q2; gq3 += e.r * e.q3; gq4 += e.r * e.q4; gq5 += e.r;
                V[0 * B + lp] = (float)e.q0; V[1 * B + lp] = (float)e.q1; V[2 * B + lp] = (float)e.q2;
                V[3 * B + lp] = (float)e.q3; V[4 * B + lp] = (float)e.q4; V[5 * B + lp] = 1.f;
                for (int col = 6; col < ldv; col++) V[col * B + lp] = 0.f;
                for (int s = 0; s < e.nnz; s++) {
                    size_t o = (size_t)s * c.N + p;
                    V[(6 + c.ell_idx[o]) * B + lp] = c.ell_w[o];
                }
                dW[lp] = e.dcurv; rW[lp] = e.r;
            } else {
                for (int col = 0; col < ldv; col++) V[col * B + lp] = 0.f;
                dW[lp] = 0; rW[lp] = 0;
            }
        }
        __syncthreads();
        // ---- phase B: lane = (tile, slice); quads of 4 staged pixels ----------------------------
#pragma unroll
        for (int q = 0; q < TPL; q++) {
            if (!tileOn[q]) continue;
            const int J = tileJ[q], K = tileK[q];
            for (int quad = slice; quad < B / 4; quad += S) {
                const int p4 = quad * 4;
                const float4 a0 = *(const float4 *)&V[(4 * J + 0) * B + p4];
                const float4 a1 = *(const float4 *)&V[(4 * J + 1) * B + p4];
                const float4 a2 = *(const float4 *)&V[(4 * J + 2) * B + p4];
                const float4 a3 = *(const float4 *)&V[(4 * J + 3) * B + p4];
                if (K >= 0) {
                    const double2 w01 = *(const double2 *)&dW[p4], w23 = *(const double2 *)&dW[p4 + 2];
                    const float4 b0 = *(const float4 *)&V[(4 * K + 0) * B + p4];
                    const float4 b1 = *(const float4 *)&V[(4 * K + 1) * B + p4];
                    const float4 b2 = *(const float4 *)&V[(4 * K + 2) * B + p4];
                    const float4 b3 = *(const float4 *)&V[(4 * K + 3) * B + p4];
                    const float4 av[4] = {a0, a1, a2, a3}, bv[4] = {b0, b1, b2, b3};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double ax = (double)av[i].x * w01.x, ay = (double)av[i].y * w01.y, az = (double)av[i].z * w23.x, aw = (double)av[i].w * w23.y;
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            acc[q][i * 4 + j] += ax * (double)bv[j].x + ay * (double)bv[j].y + az * (double)bv[j].z + aw * (double)bv[j].w;
                    }
                } else {
                    const double2 r01 = *(const double2 *)&rW[p4], r23 = *(const double2 *)&rW[p4 + 2];
                    acc[q][0] += (double)a0.x * r01.x + (double)a0.y * r01.y + (double)a0.z * r23.x + (double)a0.w * r23.y;
                    acc[q][1] += (double)a1.x * r01.x + (double)a1.y * r01.y + (double)a1.z * r23.x + (double)a1.w * r23.y;
                    acc[q][2] += (double)a2.x * r01.x + (double)a2.y * r01.y + (double)a2.z * r23.x + (double)a2.w * r23.y;
                    acc[q][3] += (double)a3.x * r01.x + (double)a3.y * r01.y + (double)a3.z * r23.x + (double)a3.w * r23.y;
                }
            }
        }
        __syncthreads();
    }
    // ---- combine slices (consecutive lanes of one wave), scatter into the packed Hessian / gradient ----
    double *Hp = SD + L::HP, *g = SD + L::G;
#pragma unroll
    for (int q = 0; q < TPL; q++) {
        if (sliced && q > 0) break;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            double v = acc[q][e];
            for (int o = 1; o < S; o <<= 1) v += __shfl_xor(v, o);
            acc[q][e] = v;
        }
        if (tileOn[q] && slice == 0) {
            const int J = tileJ[q], K = tileK[q];
            if (K >= 0) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        int row = 4 * J + i, col = 4 * K + j;
                        if (row < n && col <= row) Hp[tri(row, col)] = acc[q][i * 4 + j];
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) { int row = 4 * J + i; if (row >= 6 && row < n) g[row] = acc[q][i]; }
            }
        }
    }
    double *red = SD + L::RED;
    psi = block_sum(psi, red);
    gq0 = block_sum(gq0, red); gq1 = block_sum(gq1, red); gq2 = block_sum(gq2, red);
    gq3 = block_sum(gq3, red); gq4 = block_sum(gq4, red); gq5 = block_sum(gq5, red);
    if (tid == 0) { g[0] = gq0; g[1] = gq1; g[2] = gq2; g[3] = gq3; g[4] = gq4; g[5] = gq5; }
    __syncthreads();
    if (M > 0) {                                          // regulariser, dsm.py:323-331, 349, 372-376
        const double *xv = SD + L::X;
        double s2 = 0;
        for (int j = tid; j < M; j += SDSM_WG) {
            double xi = xv[6 + j], t3 = xi * xi, t2 = sqrt(t3 + c.epsilon);
            s2 += t2;
            g[6 + j] += c.alpha * (xi / t2);
            double gd = c.alpha * (1 / t2 - t3 / (t2 * t2 * t2));
            Hp[tri(6 + j, 6 + j)] += gd < 0 ? 0 : gd;
        }
        s2 = block_sum(s2, red);
        double o2 = c.alpha * s2 - c.alpha * sqrt(c.epsilon) * M;
        psi += o2 < 0 ? 0 : o2;
    }
    __syncthreads();
    return psi;
}

// Solve H d = -g with Jacobi scaling and an escalating diagonal shift (same schedule as the oracle).
// Returns false on numerical failure.  *lam2 = -g.d (unscaled).  INPLACE: the factor overwrites the Hessian;
// a copy is kept in global memory (c.hsave) so that a failed factorisation can be retried.
template <class L, bool INPLACE>
__device__ __forceinline__ bool factor_solve(const Cand &c, int n, double *lam2)
{
    const int tid = threadIdx.x;
    const int np = n * (n + 1) / 2;
    double *Hp = SD + L::HP, *Lp = SD + L::LP, *g = SD + L::G, *sc = SD + L::SC, *yrow = SD + L::YROW, *tmp = SD + L::TMP, *d = SD + L::D;
    int *flag = (int *)(SD + L::FLAG);
    bool finite = true;
    for (int i = tid; i < n; i += SDSM_WG) {
        double hii = Hp[tri(i, i)];
        if (!(hii > 0) || !isfinite(hii)) hii = 1;
        sc[i] = 1 / sqrt(hii);
        if (!isfinite(g[i])) finite = false;
    }
    for (int e = tid; e < np; e += SDSM_WG) {
        double v = Hp[e];
        if (!isfinite(v)) finite = false;
        if (INPLACE) c.hsave[e] = v;
    }
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (!finite) *flag = 1;
    __syncthreads();
    if (*flag) return false;

    double tau = 0;
    bool ok = false;
    for (int attempt = 0; attempt < 12 && !ok; attempt++) {
        for (int e = tid; e < np; e += SDSM_WG) {
            int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while (i * (i + 1) / 2 > e) i--;
            while ((i + 1) * (i + 2) / 2 <= e) i++;
            int j = e - i * (i + 1) / 2;
            double v = (INPLACE ? c.hsave[e] : Hp[e]) * sc[i] * sc[j];
            if (i == j) v += tau;
            Lp[e] = v;
        }
        __syncthreads();
        ok = true;
        for (int j = 0; j < n; j++) {
            for (int i = j + tid; i <= n; i += SDSM_WG) {     // row n = right-hand side -g * sc
                double s;
                const double *Lj = Lp + tri(j, 0);
                if (i < n) {
                    const double *Li = Lp + tri(i, 0);
                    s = Li[j];
                    for (int k = 0; k < j; k++) s -= Li[k] * Lj[k];
                } else {
                    s = -g[j] * sc[j];
                    for (int k = 0; k < j; k++) s -= yrow[k] * Lj[k];
                }
                tmp[i] = s;
            }
            __syncthreads();
            double piv = tmp[j];
            if (!(piv > 1e-300) || !isfinite(piv)) ok = false;
            if (ok) {
                double ljj = sqrt(piv);
                for (int i = j + tid; i <= n; i += SDSM_WG) {
                    if (i == j) Lp[tri(j, j)] = ljj;
                    else if (i < n) Lp[tri(i, j)] = tmp[i] / ljj;
                    else yrow[j] = tmp[n] / ljj;
                }
            }
            __syncthreads();
            if (!ok) break;
        }
        if (!ok) tau = tau == 0 ? 1e-12 : tau * 100;
    }
    if (!ok) return false;
    double l2 = 0;
    for (int i = tid; i < n; i += SDSM_WG) l2 += yrow[i] * yrow[i];
    l2 = block_sum(l2, SD + L::RED);
    for (int k = n - 1; k >= 0; k--) {                        // back substitution L^T z = yrow
        if (tid == 0) d[k] = yrow[k] / Lp[tri(k, k)];
        __syncthreads();
        double dk = d[k];
        const double *Lk = Lp + tri(k, 0);
        for (int i = tid; i < k; i += SDSM_WG) yrow[i] -= Lk[i] * dk;
        __syncthreads();
    }
    bool fin = isfinite(l2);
    for (int i = tid; i < n; i += SDSM_WG) { d[i] *= sc[i]; if (!isfinite(d[i])) fin = false; }
    if (!fin) *flag = 1;
    __syncthreads();
    if (*flag) return false;
    *lam2 = l2;
    return true;
}

// Damped Newton on f = scale * psi from the parameters at L::X (in/out).
// Returns 0 optimal, 1 unknown (iteration cap / stalled line search), 2 numerical failure.
template <class L, int TPL, int B, bool INPLACE>
__device__ __forceinline__ int newton(const Cand &c, int M, int max_iters, double *psi_out, int *iters_out, int *ev_value, int *ev_full)
{
    const int tid = threadIdx.x, n = 6 + M;
    double *x = SD + L::X, *xt = SD + L::XT, *d = SD + L::D;
    double tprev = 1;
    int status = 1, iters = 0;
    for (;;) {
        double psi = eval_full<L, TPL, B>(c, M);
        (*ev_full)++;
        double f = c.scale * psi;
        if (iters >= max_iters) { status = 1; break; }
        if (!isfinite(f)) { status = 2; break; }
        double lam2u;
        if (!factor_solve<L, INPLACE>(c, n, &lam2u)) { status = 2; break; }
        double lam2 = c.scale * lam2u;
        iters++;
        if (lam2 * 0.5 <= NEWTON_ABSTOL + NEWTON_RELTOL * fabs(f)) {
            for (int i = tid; i < n; i += SDSM_WG) x[i] += d[i];          // final full step
            __syncthreads();
            status = 0;
            break;
        }
        double t = tprev * 2 < 1 ? tprev * 2 : 1;
        bool accepted = false;
        for (int ls = 0; ls < LS_MAX; ls++) {
            for (int i = tid; i < n; i += SDSM_WG) xt[i] = x[i] + t * d[i];
            __syncthreads();
            double ft = c.scale * eval_value<L>(c, L::XT, M);
            (*ev_value)++;
            if (isfinite(ft) && ft <= f - LS_ALPHA * t * lam2) { accepted = true; break; }
            t *= LS_BETA;
        }
        if (!accepted) { status = 1; break; }
        tprev = t;
        for (int i = tid; i < n; i += SDSM_WG) x[i] = xt[i];
        __syncthreads();
    }
    *psi_out = eval_value<L>(c, L::X, M);
    (*ev_value)++;
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

}  // namespace

// NMAX: largest 6 + M this instantiation handles; candidates with 6 + M in (nmin_excl, NMAX] are processed,
// the others are left to the other classes.  The smallest class also writes the records of trivial /
// failed-setup candidates.
template <int NMAX, int TPL, int B, bool INPLACE, int WPE>
__global__ __launch_bounds__(SDSM_WG, WPE) void sdsm_k_solve(BatchParams P, int nmin_excl, int handles_rest, sdsm_record *records,
                                                         uint32_t *masks, double *xi_out)
{
    using L = Lay<NMAX, B, INPLACE>;
    const int tid = threadIdx.x;
    const int ci = P.order[blockIdx.x];
    const CandDesc cd = P.cand[ci];
    const CandState st = P.state[ci];
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
    if (6 + Mfull > SDSM_MAX_N_SOLVE) { unsupported = true; Mfull = 0; }     // elliptical result only (flagged)
    const int nfull = 6 + Mfull;
    if (!(nfull > nmin_excl && nfull <= NMAX)) return;

    Cand c;
    c.N = cd.N;
    c.crop_y = P.crop_y + cd.crop_off; c.crop_rc = P.crop_rc + cd.crop_off; c.ell_nnz = P.ell_nnz + cd.crop_off;
    c.ell_idx = P.ell_idx + cd.ell_off; c.ell_w = P.ell_w + cd.ell_off;
    c.hsave = (INPLACE && cd.hsave_slot >= 0) ? P.hsave + (int64_t)cd.hsave_slot * P.hsave_stride : nullptr;
    c.scale = P.scale / cd.N;                                   // objects.py:380
    c.epsilon = P.epsilon; c.alpha = P.alpha;
    // local frame: centre of the bounding box, half extents
    const double half_r = 0.5 * (cd.h - 1), half_c = 0.5 * (cd.w - 1);
    c.rmid = cd.r0 + half_r; c.cmid = cd.c0 + half_c;
    c.inv_hr = 1.0 / (half_r < 1 ? 1 : half_r); c.inv_hc = 1.0 / (half_c < 1 ? 1 : half_c);
    // normalised image coordinate x0 = r / (H-1) = P0 * u + O0
    const double z0 = P.H > 1 ? P.H - 1.0 : 1.0, z1 = P.W > 1 ? P.W - 1.0 : 1.0;
    const double P0 = 1.0 / (c.inv_hr * z0), P1 = 1.0 / (c.inv_hc * z1), O0 = c.rmid / z0, O1 = c.cmid / z1;
    double *x = SD + L::X, *xt = SD + L::XT;

    sdsm_record r = {};
    r.n_pixels = cd.N; r.n_deform = st.M;
    int status_final = SDSM_CAND_OPTIMAL;
    int ev_value = 0, ev_full = 0;

    // ---- phases: 0 elliptical from zeros, 1 elliptical from the moment initialisation (only if phase 0 was
    //      not optimal, objects.py:337-355), 2 deformable shape model (objects.py:394-410) ---------------
    for (int i = tid; i < NMAX; i += SDSM_WG) x[i] = 0;
    __syncthreads();
    double psi_ell = INFINITY, psi_final = NAN;
    bool have = false, fallback = false;
    double keep[6] = {0, 0, 0, 0, 0, 0};
    int s_prev = 0;
    for (int phase = P.init_elliptical ? 0 : 2; phase < 3; phase++) {
        int M = 0;
        if (phase == 1) {
            if (have && s_prev == 0) continue;                                  // pass 1 was optimal
            r.flags |= 1;
            double cnt = (double)st.npos;
            double mr = (double)st.sum_r / cnt, mc = (double)st.sum_c / cnt;
            double cr = rint(mr), cc2 = rint(mc);                               // np.round: half to even
            double c0 = cr / z0, c1 = cc2 / z1;
            // exact integer second moments: n * sum(r^2) - (sum r)^2
            double vr = ((double)st.sum_rr * cnt - (double)st.sum_r * (double)st.sum_r) / (cnt * cnt);
            double vc = ((double)st.sum_cc * cnt - (double)st.sum_c * (double)st.sum_c) / (cnt * cnt);
            double h0 = sqrt(vr < 0 ? 0 : vr) / z0, h1 = sqrt(vc < 0 ? 0 : vc) / z1;
            h0 = h0 < 1e-8 ? 1e-8 : h0; h1 = h1 < 1e-8 ? 1e-8 : h1;
            double e0 = 1 / (h0 * h0), e1 = 1 / (h1 * h1);
            double b0 = e0 * c0, b1 = e1 * c1, cterm = c0 * b0 + c1 * b1 - 1;
            double thg[6] = {-e0, -e1, 0, b0, b1, -cterm}, thl[6];
            reparam(thg, P0, P1, O0, O1, thl);
            __syncthreads();
            if (tid < 6) xt[tid] = thl[tid];
            __syncthreads();
            double vinit = eval_value<L>(c, L::XT, 0);
            ev_value++;
            if (vinit > psi_ell) continue;                                      // objects.py:341-342
            if (tid < 6) x[tid] = thl[tid];
            __syncthreads();
        } else if (phase == 2) {
            if (P.init_elliptical && !have) { status_final = SDSM_CAND_ERROR; break; }   // CvxprogError (objects.py:351-353)
            M = Mfull;
            __syncthreads();
            for (int i = tid; i < NMAX; i += SDSM_WG) x[i] = i < 6 ? keep[i] : 0;
            __syncthreads();
        }
        double psi; int its;
        int s = newton<L, TPL, B, INPLACE>(c, M, P.max_iters, &psi, &its, &ev_value, &ev_full);
        if (phase < 2) {
            r.iters_ell += its;
            if (s != 2) { have = true; psi_ell = psi; for (int i = 0; i < 6; i++) keep[i] = x[i]; s_prev = s; }
            else if (phase == 0) s_prev = 2;
            __syncthreads();
        } else {
            r.iters_dsm = its;
            psi_final = psi;
            if (s == 2) fallback = true;                                        // exception -> fallback
            else if (s == 1) {                                                  // 'unknown' and worse than the start
                __syncthreads();
                for (int i = tid; i < NMAX; i += SDSM_WG) xt[i] = i < 6 ? keep[i] : 0;
                __syncthreads();
                double vi = eval_value<L>(c, L::XT, Mfull);
                ev_value++;
                if (psi > vi) fallback = true;
            }
            if (fallback) {
                __syncthreads();
                for (int i = tid; i < NMAX; i += SDSM_WG) x[i] = i < 6 ? keep[i] : 0;
                __syncthreads();
                psi_final = eval_value<L>(c, L::X, Mfull);
                ev_value++;
                status_final = SDSM_CAND_FALLBACK;
            }
        }
    }
    r.energy_ell = P.init_elliptical ? psi_ell : NAN;
    if (unsupported && status_final != SDSM_CAND_ERROR) status_final = SDSM_CAND_UNSUPPORTED;

    // ---- mask tail (objects.py:198-209) ----------------------------------------------------------
    const int mwords = (cd.h * cd.w + 31) / 32;
    uint32_t *mk = masks + cd.mask_off;
    for (int i = tid; i < mwords; i += SDSM_WG) mk[i] = 0;
    __syncthreads();
    int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
    int onb = 0;
    if (status_final != SDSM_CAND_ERROR) {
        for (int p = tid; p < c.N; p += SDSM_WG) {
            uint32_t rc = c.crop_rc[p];
            int pr = rc >> 16, pc = rc & 0xffffu;
            double u = ((double)pr - c.rmid) * c.inv_hr, v = ((double)pc - c.cmid) * c.inv_hc;
            double Sv = u * u * x[0] + v * v * x[1] + 2 * (u * v) * x[2] + 2 * u * x[3] + 2 * v * x[4] + x[5];
            if (Mfull > 0) {
                int nnz = c.ell_nnz[p];
                double gx = 0;
                for (int s = 0; s < nnz; s++) { size_t o = (size_t)s * c.N + p; gx += (double)c.ell_w[o] * x[6 + c.ell_idx[o]]; }
                Sv += gx;
            }
            if (Sv > 0) {
                int bit = (pr - cd.r0) * cd.w + (pc - cd.c0);
                atomicOr(&mk[bit >> 5], 1u << (bit & 31));
                rmin = pr < rmin ? pr : rmin; rmax = pr > rmax ? pr : rmax;
                cmin = pc < cmin ? pc : cmin; cmax = pc > cmax ? pc : cmax;
            }
        }
        // 1-px pad ring of the image, polynomial part only (G~ has no rows there)
        const int ringw = P.W + 2, ringh = P.H + 2;
        for (int i = tid; i < 2 * ringw + 2 * ringh; i += SDSM_WG) {
            int pr, pc;
            if (i < ringw) { pr = -1; pc = i - 1; }
            else if (i < 2 * ringw) { pr = P.H; pc = i - ringw - 1; }
            else if (i < 2 * ringw + ringh) { pr = i - 2 * ringw - 1; pc = -1; }
            else { pr = i - 2 * ringw - ringh - 1; pc = P.W; }
            double u = ((double)pr - c.rmid) * c.inv_hr, v = ((double)pc - c.cmid) * c.inv_hc;
            double Sv = u * u * x[0] + v * v * x[1] + 2 * (u * v) * x[2] + 2 * u * x[3] + 2 * v * x[4] + x[5];
            if (Sv > 0) onb = 1;
        }
    }
    int *ired = (int *)(SD + L::RED);
    rmin = block_min_i32(rmin, ired); cmin = block_min_i32(cmin, ired);
    rmax = -block_min_i32(-rmax, ired); cmax = -block_min_i32(-cmax, ired);
    onb = -block_min_i32(-onb, ired);

    if (xi_out) for (int j = tid; j < st.M; j += SDSM_WG) xi_out[cd.xi_off + j] = j < Mfull ? x[6 + j] : 0;
    if (tid == 0) {
        // local basis -> full-image-normalised theta:  u = (x0 - O0) / P0
        double thl[6] = {x[0], x[1], x[2], x[3], x[4], x[5]}, thg[6];
        reparam(thl, 1 / P0, 1 / P1, -O0 / P0, -O1 / P1, thg);
        for (int i = 0; i < 6; i++) r.theta[i] = thg[i];
        r.energy = status_final == SDSM_CAND_ERROR ? NAN : psi_final;
        r.status = status_final;
        r.evals_value = ev_value; r.evals_full = ev_full;
        r.on_boundary = onb;
        if (rmax >= 0) { r.fg_r0 = rmin; r.fg_c0 = cmin; r.fg_h = rmax - rmin + 1; r.fg_w = cmax - cmin + 1; }
        *rec = r;
    }
}

// ---- launch helper (called from sdsm_api.hip) ----------------------------------------------------
// class A: n <= 40  (B 256, separate factor)           LDS ~ 60 KB
// class B: n <= 84  (B 128, separate factor)           LDS ~ 107 KB
// class C: n <= 172 (4 tiles per lane, B 32, in place) LDS ~ 151 KB
template <int NMAX, int TPL, int B, bool INPLACE, int WPE>
static hipError_t launch_class(const BatchParams &P, int nmin_excl, int handles_rest, sdsm_record *records, uint32_t *masks, double *xi_out, hipStream_t stream)
{
    auto kern = sdsm_k_solve<NMAX, TPL, B, INPLACE, WPE>;
    constexpr int lds = Lay<NMAX, B, INPLACE>::TOTAL_BYTES;
    static_assert(lds <= 160 * 1024 - 256, "LDS budget");
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(P.n), dim3(SDSM_WG), lds, stream, P, nmin_excl, handles_rest, records, masks, xi_out);
    return hipGetLastError();
}

extern "C" hipError_t sdsm_launch_solve(const BatchParams &P, sdsm_record *records, uint32_t *masks, double *xi_out, hipStream_t stream)
{
    hipError_t e;
    if ((e = launch_class<172, 4, 32, true, 1>(P, 84, 0, records, masks, xi_out, stream)) != hipSuccess) return e;
    if ((e = launch_class<84, 1, 128, false, 1>(P, 40, 0, records, masks, xi_out, stream)) != hipSuccess) return e;
    return launch_class<40, 1, 256, false, 2>(P, 0, 1, records, masks, xi_out, stream);
}
