// band.hip -- Householder reduction of a batch of symmetric matrices to BAND form (bandwidth BW = 4), the basis the Eigen path's
// kappa search works in (eigen.hip; reference src/pyimcom/lakernel.py:154-223, routine.py:341-430).
//
//   A = Q B Q^T,   Q = H_0 H_1 ... ,  H_r = I - tau_r v_r v_r^T with v_r zero above its pivot row r + BW,   B[i][j] = 0 for |i - j| > BW
//
// Why a band and not the tridiagonal form of tridiag.hip: the one-stage tridiagonalisation needs one pass over the trailing
// matrix PER COLUMN (the product A v_r: 4 N^3 / 3 bytes = 34 GB per cfg-3 stamp, HBM bound, 79 % of the Eigen path after
// round 3's first step).  With the pivot BW rows below the diagonal, reflector r + 1 does not wait for that product: the
// update of column r + 1 by H_r is  M[:, r+1] -= v_r w_r[r+1]  (v_r[r+1] = 0), and w_r[r+1] = tau_r (M v_r)[r+1] is a dot
// product inside the N x BW panel.  So BW reflectors are formed from the panel alone, and ONE pass over the trailing matrix
// multiplies it by all BW of them: a quarter of the passes.  The kappa search then solves banded instead of tridiagonal
// systems per output pixel (eigen.hip) -- a few times the (small) work of the tridiagonal sweeps.
//
// Structure (per stamp; all kernels run the whole batch):
//   band_step_kernel   one workgroup per stamp, the N x BW panel in LDS.  Phase W finishes the w vectors of the previous group
//                      (w_c = tau (p_c - corrections) - 1/2 tau^2 (p_c . v_c) v_c, the lazy two-sided update of LAPACK's latrd with
//                      the products M v_c from symv4); phase P forms the next group's BW reflectors from its panel
//   symv4_kernel       the one pass over the trailing lower triangle: Z = At [v_0 .. v_3] (row sums + transposed partials per strip)
//   GEMM               every TPL = 64 reflectors the trailing matrix gets its rank-2 x 64 update (the tile engine)
// Verified step by step against a numpy restatement of exactly this decomposition (band to 3e-15, eigenvalues to 6e-15).
#include <algorithm>

#include "common.h"
#include "launchers.h"

namespace imcom {

constexpr int BW = BAND_BW;      // bandwidth = reflectors per group
constexpr int BTPL = 64;         // reflectors per lazy super-panel (multiple of BW)
constexpr int BTHREADS = 1024;   // band_step_kernel: one workgroup per stamp
constexpr int BSTRIP = 32;       // rows per strip of the symmetric product

struct BHouse {
    double beta, tau, scale;
};
__device__ inline BHouse bhouse(double a0, double xn2)
{
    BHouse h;
    if (xn2 == 0.0) { h.beta = a0; h.tau = 0.0; h.scale = 0.0; }
    else {
        h.beta = -copysign(sqrt(a0 * a0 + xn2), a0);
        h.tau = (h.beta - a0) / h.beta;
        h.scale = 1.0 / (a0 - h.beta);
    }
    return h;
}

// sums of NV values over the workgroup (1024 threads), returned to every thread; red: [16][NV] doubles of LDS
template <int NV>
__device__ inline void block_sums(double (&v)[NV], double *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
    }
    __syncthreads();  // red may still be read from the previous reduction
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < NV; q++) red[wave * NV + q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; q++) {
        double t = 0.0;
        for (int w = 0; w < BTHREADS / 64; w++) t += red[w * NV + q];
        v[q] = t;
    }
}

// ps: first reflector of the current lazy super-panel (Wp row k - ps holds w_k); g0 >= 0: finish the w vectors of the group of
// columns g0 .. g0+BW-1 (needs Z4 / part4 from symv4_kernel); r0 >= 0: form the reflectors of columns r0 .. r0+BW-1.
__global__ __launch_bounds__(BTHREADS) void band_step_kernel(const double *__restrict__ At, double *__restrict__ Vall, double *__restrict__ Wp,
                                                             double *__restrict__ tau, double *__restrict__ band, double *__restrict__ Z4,
                                                             const double *__restrict__ part4, const int *__restrict__ n, int ld, int ps, int g0,
                                                             int r0)
{
    extern __shared__ double sm[];
    double *vl = sm;                        // [BW][ld]: phase W the group's v_c, phase P the panel's columns
    double *coef = sm + (size_t)BW * ld;    // [2][BTPL][BW]
    double *red = coef + 2 * BTPL * BW;     // [16][4]
    double *sc = red + 16 * 4;              // [BW][BW] w_c' . v_c, then [BW][BW] v_c' . v_c
    const int s = blockIdx.x, ns = n[s], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long so = (long)s * ld * ld;
    const double *A = At + so;
    double *V = Vall + so, *W = Wp + (long)s * BTPL * ld, *tv = tau + (long)s * ld, *bd = band + (long)s * (BW + 1) * ld;
    double *Z = Z4 + (long)s * BW * ld;
    const double *part = part4 + (long)s * (ld / BSTRIP) * BW * ld;

    // ------------------------------------------------------------------ phase W
    if (g0 >= 0 && g0 + BW + 1 < ns) {  // (a group whose first pivot has nothing below it has no reflector: tau = 0, w = 0)
        const int G = min(BW, ns - g0), kc = g0 - ps;
        for (int i = g0 + tid; i < ns; i += BTHREADS)
#pragma unroll
            for (int c = 0; c < BW; c++) vl[c * ld + i] = c < G ? V[(long)(g0 + c) * ld + i] : 0.0;
        __syncthreads();
        // dots of the group's v_c with the super-panel's earlier reflectors and their w's: one wave per earlier reflector
        for (int k = wave; k < kc; k += BTHREADS / 64) {
            double dv[BW], dw[BW];
#pragma unroll
            for (int c = 0; c < BW; c++) dv[c] = dw[c] = 0.0;
            const double *vk = V + (long)(ps + k) * ld, *wk = W + (long)k * ld;
            for (int i = g0 + BW + lane; i < ns; i += 64) {
                const double a = vk[i], b = wk[i];
#pragma unroll
                for (int c = 0; c < BW; c++) { const double x = vl[c * ld + i]; dv[c] += a * x; dw[c] += b * x; }
            }
#pragma unroll
            for (int c = 0; c < BW; c++) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { dv[c] += __shfl_xor(dv[c], off, 64); dw[c] += __shfl_xor(dw[c], off, 64); }
            }
            if (lane == 0)
#pragma unroll
                for (int c = 0; c < BW; c++) { coef[k * BW + c] = dv[c]; coef[(BTPL + k) * BW + c] = dw[c]; }
        }
        // v_c' . v_c inside the group (c' < c): wave q takes pair q
        if (wave < BW * (BW - 1) / 2) {
            int c1 = 0, c2 = 1, q = wave;
            while (q >= BW - 1 - c1) { q -= BW - 1 - c1; c1++; }
            c2 = c1 + 1 + q;
            double d = 0.0;
            for (int i = g0 + BW + lane; i < ns; i += 64) d += vl[c1 * ld + i] * vl[c2 * ld + i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
            if (lane == 0) sc[BW * BW + c1 * BW + c2] = d;
        }
        __syncthreads();
        // p_c without the group's own corrections: row sums + transposed partials of the strips below - earlier reflectors
        const int slast = (ns - 1) / BSTRIP;
        for (int i = g0 + 1 + tid; i < ns; i += BTHREADS) {
            double p[BW];
#pragma unroll
            for (int c = 0; c < BW; c++) p[c] = Z[c * ld + i];
            for (int st = i / BSTRIP + 1; st <= slast; st++)
#pragma unroll
                for (int c = 0; c < BW; c++) p[c] += part[((long)st * BW + c) * ld + i];
            for (int k = 0; k < kc; k++) {
                const double a = V[(long)(ps + k) * ld + i], b = W[(long)k * ld + i];
#pragma unroll
                for (int c = 0; c < BW; c++) p[c] -= a * coef[(BTPL + k) * BW + c] + b * coef[k * BW + c];
            }
#pragma unroll
            for (int c = 0; c < BW; c++) Z[c * ld + i] = p[c];
        }
        // the group's w vectors, one after the other
        for (int c = 0; c < G; c++) {
            const double tc = tv[g0 + c];
            double *wc = W + (long)(kc + c) * ld;
            double a_[BW], b_[BW];
#pragma unroll
            for (int q = 0; q < BW; q++) { a_[q] = q < c ? sc[q * BW + c] : 0.0; b_[q] = q < c ? sc[BW * BW + q * BW + c] : 0.0; }
            double dot[1] = {0.0};
            for (int i = g0 + 1 + tid; i < ns; i += BTHREADS) {
                double p = Z[c * ld + i];
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q < c) p -= vl[q * ld + i] * a_[q] + W[(long)(kc + q) * ld + i] * b_[q];
                const double wprime = tc * p;
                wc[i] = wprime;
                dot[0] += wprime * vl[c * ld + i];
            }
            block_sums<1>(dot, red);
            const double alpha = -0.5 * tc * dot[0];
            double d3[BW];
#pragma unroll
            for (int q = 0; q < BW; q++) d3[q] = 0.0;
            for (int i = g0 + 1 + tid; i < ns; i += BTHREADS) {
                const double w = wc[i] + alpha * vl[c * ld + i];
                wc[i] = w;
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q > c) d3[q] += w * vl[q * ld + i];
            }
            block_sums<BW>(d3, red);
            if (tid == 0)
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q > c) sc[c * BW + q] = d3[q];
            __syncthreads();
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase P
    if (r0 >= 0 && r0 < ns) {
        const int G = min(BW, ns - r0), kc = r0 - ps;
        for (int e = tid; e < kc * BW; e += BTHREADS) {
            const int k = e / BW, c = e % BW;
            const bool in = c < G;
            coef[k * BW + c] = in ? V[(long)(ps + k) * ld + r0 + c] : 0.0;         // v_k[column]
            coef[(BTPL + k) * BW + c] = in ? W[(long)k * ld + r0 + c] : 0.0;      // w_k[column]
        }
        __syncthreads();
        for (int i = r0 + tid; i < ns; i += BTHREADS) {
            double x[BW];
#pragma unroll
            for (int c = 0; c < BW; c++) x[c] = c < G ? A[(long)(r0 + c) * ld + i] : 0.0;  // row r0+c read as column (symmetric)
            for (int k = 0; k < kc; k++) {
                const double a = V[(long)(ps + k) * ld + i], b = W[(long)k * ld + i];
#pragma unroll
                for (int c = 0; c < BW; c++) x[c] -= a * coef[(BTPL + k) * BW + c] + b * coef[k * BW + c];
            }
#pragma unroll
            for (int c = 0; c < BW; c++) vl[c * ld + i] = x[c];
        }
        __syncthreads();
        for (int c = 0; c < G; c++) {
            const int r = r0 + c, piv = r + BW;
            if (tid < BW) bd[(long)tid * ld + r] = r + tid < ns ? vl[c * ld + r + tid] : 0.0;  // final: later reflectors start below
            if (piv < ns) {
                double x2[1] = {0.0};
                for (int i = piv + 1 + tid; i < ns; i += BTHREADS) x2[0] += vl[c * ld + i] * vl[c * ld + i];
                block_sums<1>(x2, red);
                const BHouse h = bhouse(vl[c * ld + piv], x2[0]);
                __syncthreads();  // everybody has read the pivot entry
                if (tid == 0) { bd[(long)BW * ld + r] = h.beta; tv[r] = h.tau; }
                double d3[BW];
#pragma unroll
                for (int q = 0; q < BW; q++) d3[q] = 0.0;
                for (int i = piv + tid; i < ns; i += BTHREADS) {
                    const double v = i == piv ? 1.0 : h.scale * vl[c * ld + i];
                    V[(long)r * ld + i] = v;
                    vl[c * ld + i] = v;
#pragma unroll
                    for (int q = 0; q < BW; q++)
                        if (q > c) d3[q] += vl[q * ld + i] * v;  // (M v_r)[r0 + q]: the panel column against the new reflector
                }
                block_sums<BW>(d3, red);
                for (int i = piv + tid; i < ns; i += BTHREADS) {
                    const double v = vl[c * ld + i];
#pragma unroll
                    for (int q = 0; q < BW; q++)
                        if (q > c && q < G) vl[q * ld + i] -= v * (h.tau * d3[q]);
                }
                __syncthreads();
            } else if (tid == 0) {
                bd[(long)BW * ld + r] = 0.0;
                tv[r] = 0.0;
            }
        }
    }
}

// Z[c][i] = (At v_c)[i] for the BW reflectors of columns r0 .. r0+BW-1, rows i > r0: one pass over the trailing lower triangle.
// A workgroup takes a strip of 32 rows (4 waves x 8 rows), walks its columns in chunks of 128 (a double2 per lane) and
// leaves (a) the row sums over the columns up to the strip's diagonal block and (b), from the same loads, the strip's
// contributions to the rows left of it (part4[strip][c][column]); band_step_kernel adds the strips up.
__global__ __launch_bounds__(256) void symv4_kernel(const double *__restrict__ At, const double *__restrict__ Vall, const int *__restrict__ n,
                                                    int ld, int r0, int nrowtiles, double *__restrict__ Z4, double *__restrict__ part4)
{
    __shared__ double y2s[2][4][BW][128];
    const int s = blockIdx.y, ns = n[s];
    if (r0 + BW + 1 >= ns) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long so = (long)s * ld * ld;
    // longest strips (bottom of the matrix) first: they bound the launch's critical path
    const int strip = ((r0 + 1) / BSTRIP) + (nrowtiles - 1 - (int)blockIdx.x), rb = strip * BSTRIP;
    if (rb >= ns) return;
    const int rw = rb + wave * 8;
    const double *v0 = Vall + so + (long)r0 * ld;
    double ur[BW][8];  // the reflectors at this wave's rows (wave-uniform)
#pragma unroll
    for (int c = 0; c < BW; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) ur[c][i] = rw + i < ns ? v0[(long)c * ld + rw + i] : 0.0;
    const double *A = At + so + (long)rw * ld;
    double *part_s = part4 + (((long)s * (ld / BSTRIP) + strip) * BW) * ld;
    double acc[BW][8];
#pragma unroll
    for (int c = 0; c < BW; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[c][i] = 0.0;
    const int cend = rb + BSTRIP, cfirst = (r0 + 1) & ~127;
    int buf = 0;
    double2 an[8], un[BW];
    auto fetch = [&](int c0) {
        const int cc = c0 + 2 * lane;
#pragma unroll
        for (int c = 0; c < BW; c++) {
            un[c] = *(const double2 *)(v0 + (long)c * ld + cc);
            if (cc >= cend) un[c].x = 0.0;
            if (cc + 1 >= cend) un[c].y = 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) an[i] = *(const double2 *)(A + (long)i * ld + cc);
    };
    fetch(cfirst);
    for (int c0 = cfirst; c0 < cend; c0 += 128, buf ^= 1) {
        double2 a[8], uu[BW];
#pragma unroll
        for (int c = 0; c < BW; c++) uu[c] = un[c];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = an[i];
        if (c0 + 128 < cend) fetch(c0 + 128);
        double2 y2[BW];
#pragma unroll
        for (int c = 0; c < BW; c++) y2[c] = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int c = 0; c < BW; c++) {
                acc[c][i] += a[i].x * uu[c].x + a[i].y * uu[c].y;
                y2[c].x += a[i].x * ur[c][i];
                y2[c].y += a[i].y * ur[c][i];
            }
        }
        if (c0 < rb) {  // columns strictly left of the diagonal block receive the transposed contributions
#pragma unroll
            for (int c = 0; c < BW; c++) *(double2 *)&y2s[buf][wave][c][2 * lane] = y2[c];
            __syncthreads();
            for (int e = threadIdx.x; e < BW * 128; e += 256) {
                const int c = e >> 7, t = e & 127;
                if (c0 + t < rb) part_s[(long)c * ld + c0 + t] = (y2s[buf][0][c][t] + y2s[buf][1][c][t]) + (y2s[buf][2][c][t] + y2s[buf][3][c][t]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < BW; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[c][i] += __shfl_xor(acc[c][i], off, 64);
        }
    if (lane == 0) {
        double *Zs = Z4 + (long)s * BW * ld;
#pragma unroll
        for (int c = 0; c < BW; c++)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (rw + i < ns) Zs[(long)c * ld + rw + i] = acc[c][i];
    }
}

// At = A on the leading n x n (zero elsewhere)
__global__ void band_init_kernel(const double *__restrict__ A, long lda, long strideA, const int *__restrict__ n, double *__restrict__ At, int ld)
{
    const int s = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const int ns = n[s];
    At[(long)s * ld * ld + (long)i * ld + j] = (i < ns && j < ns) ? A[s * strideA + (long)i * lda + j] : 0.0;
}

bool band_basis_fits(int ld) { return ((size_t)BW * ld + 2 * BTPL * BW + 16 * 4 + 2 * BW * BW) * 8 <= 160 * 1024; }

size_t band_basis_ws_bytes(int batch, int ld, int mp)
{
    const int npanels = (ld + NB - 1) / NB;
    size_t t = 0;
    auto add = [&](size_t b) { t = align_up(t, 256) + b; };
    add((size_t)batch * ld * ld * 8);                       // Vall
    add((size_t)batch * ld * 8);                            // tau
    add((size_t)batch * (BW + 1) * ld * 8);                 // band
    add((size_t)batch * 4);                                 // n
    add((size_t)batch * npanels * NB * NB * 8);             // T factors of all panels
    add((size_t)batch * NB * NB * 8);                       // S = V V^T of one panel
    add((size_t)batch * NB * mp * 8 * 2);                   // W1, W2
    add((size_t)batch * ld * ld * 8);                       // At
    add((size_t)batch * BTPL * ld * 8);                     // Wp
    add((size_t)batch * BW * ld * 8);                       // Z4
    add((size_t)batch * (ld / BSTRIP) * BW * ld * 8);       // part4
    return t + 8192;
}

int trd_panel_factors(imcom_ctx *ctx, TrdBasis *out, int batch);  // tridiag.hip

// A -> band B (out->band: [batch][BW+1][ld], band[t][i] = B[i+t][i]) and the reflectors of Q (out->Vall, out->tauvec), ready for
// trd_apply_q.  Same contract as trd_basis_device; the caller has reserved band_basis_ws_bytes().
int band_basis_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, int mp, const double *A, long lda, long strideA, TrdBasis *out)
{
    IMCOM_REQUIRE(ld % NB == 0 && ld >= NB && mp % NB == 0 && band_basis_fits(ld), "band reduction: ld=%d, mp=%d", ld, mp);
    const size_t mat = (size_t)batch * ld * ld * 8, vecb = (size_t)batch * ld * 8;
    const int npanels_max = (ld + NB - 1) / NB;
    out->Vall = (double *)ws_take(ctx, mat);
    out->tauvec = (double *)ws_take(ctx, vecb);
    out->band = (double *)ws_take(ctx, (size_t)batch * (BW + 1) * ld * 8);
    out->n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    out->Tm = (double *)ws_take(ctx, (size_t)batch * npanels_max * NB * NB * 8);
    out->Sm = (double *)ws_take(ctx, (size_t)batch * NB * NB * 8);
    out->W1 = (double *)ws_take(ctx, (size_t)batch * NB * mp * 8);
    out->W2 = (double *)ws_take(ctx, (size_t)batch * NB * mp * 8);
    out->dvec = out->evec = nullptr;
    out->ld = ld;
    out->bw = BW;
    out->nmax = 0;
    for (int s = 0; s < batch; s++) out->nmax = std::max(out->nmax, n_host[s]);
    const int nmax = out->nmax;
    out->npanels = (std::max(nmax - BW - 1, 0) + NB - 1) / NB;
    if (!out->Vall || !out->tauvec || !out->band || !out->n_dev || !out->Tm || !out->Sm || !out->W1 || !out->W2) {
        set_error("internal: band workspace");
        return IMCOM_ERR_NOMEM;
    }
    hipStream_t st = ctx->stream;
    IMCOM_TRY(upload(ctx, out->n_dev, n_host, (size_t)batch));
    const size_t mark = ctx->ws_used;
    double *At = (double *)ws_take(ctx, mat);
    double *Wp = (double *)ws_take(ctx, (size_t)batch * BTPL * ld * 8);
    double *Z4 = (double *)ws_take(ctx, (size_t)batch * BW * ld * 8);
    double *part4 = (double *)ws_take(ctx, (size_t)batch * (ld / BSTRIP) * BW * ld * 8);
    if (!At || !Wp || !Z4 || !part4) { set_error("internal: band workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_HIP_CHECK(hipMemsetAsync(out->Vall, 0, mat, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(out->tauvec, 0, vecb, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(out->band, 0, (size_t)batch * (BW + 1) * ld * 8, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(Wp, 0, (size_t)batch * BTPL * ld * 8, st));
    hipLaunchKernelGGL(band_init_kernel, dim3((ld + 255) / 256, ld, batch), dim3(256), 0, st, A, lda, strideA, out->n_dev, At, ld);
    IMCOM_TRY(check_launch("band_init_kernel"));
    const size_t lds = ((size_t)BW * ld + 2 * BTPL * BW + 16 * 4 + 2 * BW * BW) * 8;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)band_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ProfScope ps_(ctx, "eigen_trd", nmax);
        int ps = 0;
        for (int r0 = 0; r0 < nmax; r0 += BW) {
            if (r0 - ps >= BTPL) {
                // the previous group's w vectors, then the trailing two-sided update A[pe:, pe:] -= V W^T + W V^T; a new lazy panel
                hipLaunchKernelGGL(band_step_kernel, dim3(batch), dim3(BTHREADS), lds, st, At, out->Vall, Wp, out->tauvec, out->band, Z4, part4,
                                   out->n_dev, ld, ps, r0 - BW, -1);
                IMCOM_TRY(check_launch("band_step_kernel"));
                // The GEMM tiles are 128-aligned: start at the tile boundary at or below the panel's end.  The extra rows / columns
                // it touches are already reduced (their band entries have been taken out) and are never read again.
                const int pa = r0 / NB * NB, rem = ld - pa;
                const double *Vp = out->Vall + (long)ps * ld + pa, *Wq = Wp + pa;
                double *C = At + (long)pa * ld + pa;
                IMCOM_TRY(launch_gemm(ctx, true, true, rem, rem, BTPL, batch, Vp, ld, (long)ld * ld, Wq, ld, (long)BTPL * ld, C, ld, (long)ld * ld, -1.0, 1.0));
                IMCOM_TRY(launch_gemm(ctx, true, true, rem, rem, BTPL, batch, Wq, ld, (long)BTPL * ld, Vp, ld, (long)ld * ld, C, ld, (long)ld * ld, -1.0, 1.0));
                IMCOM_HIP_CHECK(hipMemsetAsync(Wp, 0, (size_t)batch * BTPL * ld * 8, st));
                ps = r0;
                hipLaunchKernelGGL(band_step_kernel, dim3(batch), dim3(BTHREADS), lds, st, At, out->Vall, Wp, out->tauvec, out->band, Z4, part4,
                                   out->n_dev, ld, ps, -1, r0);
            } else
                hipLaunchKernelGGL(band_step_kernel, dim3(batch), dim3(BTHREADS), lds, st, At, out->Vall, Wp, out->tauvec, out->band, Z4, part4,
                                   out->n_dev, ld, ps, r0 > 0 ? r0 - BW : -1, r0);
            if (r0 + BW + 1 < nmax) {
                const int nrowtiles = (nmax - 1) / BSTRIP - (r0 + 1) / BSTRIP + 1;
                hipLaunchKernelGGL(symv4_kernel, dim3(nrowtiles, batch), dim3(256), 0, st, At, out->Vall, out->n_dev, ld, r0, nrowtiles, Z4, part4);
            }
            IMCOM_TRY(check_launch("band step"));
        }
    }
    ctx->ws_used = mark;  // the scratch is free again (same stream: everything queued so far runs before whatever reuses it)
    return trd_panel_factors(ctx, out, batch);
}

}  // namespace imcom
