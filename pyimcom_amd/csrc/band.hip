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
// Structure (all kernels run the whole batch), per group of BW columns:
//   symv4_kernel       the one pass over the trailing lower triangle: Z = At [v_0 .. v_3] (row sums + transposed partials per strip
//                      of 32 rows); extra "dot blocks" of the same launch form V_k . v_c, W_k . v_c for the lazy super-panel's earlier
//                      reflectors and the group's own v_c' . v_c
//   band_apply_kernel  many workgroups per stamp: p_c = Z + the strips' partials - sum_k (v_k (w_k . v_c) + w_k (v_k . v_c))
//   band_step_kernel   ONE workgroup per stamp.  Phase W: what is sequential inside the group -- w_c = tau (p_c - in-group
//                      corrections) - 1/2 tau^2 (p_c . v_c) v_c (the lazy two-sided update of LAPACK's latrd).  Phase P: the next
//                      group's N x BW panel into LDS with the super-panel's corrections, its BW reflectors one after the other
//   syr2k (tile engine) every BTPL = 64 reflectors the trailing matrix gets its rank-2 x 64 update, lower 128-tiles only
// Verified step by step against a numpy restatement of exactly this decomposition (band to 3e-15, eigenvalues to 6e-15).
#include <algorithm>

#include "common.h"
#include "launchers.h"

namespace imcom {

constexpr int BW = BAND_BW;      // bandwidth = reflectors per group
#ifndef IMCOM_BTPL
#define IMCOM_BTPL 64
#endif
constexpr int BTPL = IMCOM_BTPL; // reflectors per lazy super-panel (multiple of BW; build-time: make EXTRA=-DIMCOM_BTPL=128 for A/B runs)
constexpr int BTHREADS = 1024;   // band_step_kernel: one workgroup per stamp
constexpr int RTHREADS = 512;    // band_step_reg_kernel (256 registers per thread)
constexpr int BSTRIP = 64;       // rows per strip of the symmetric product
constexpr int SLPW = 64 / (BSTRIP / 8);  // symv4_kernel: lanes (column pairs) per group of 8 rows inside a wave
constexpr int SWCOLS = 2 * SLPW;        // columns per wave and chunk;  4 waves: chunks of 4 * SWCOLS columns
constexpr int SCHUNK = 4 * SWCOLS;
static_assert(BSTRIP == 32 || BSTRIP == 64, "strip of 4 or 8 row groups per wave");
static_assert(BW == 4, "the panel loads of band_step_kernel take four columns as two double2");

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

// sums of NV values over the workgroup (NT threads), returned to every thread; red: [NT / 64][NV] doubles of LDS
template <int NV, int NT = BTHREADS>
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
#pragma unroll
        for (int w = 0; w < NT / 64; w++) t += red[w * NV + q];
        v[q] = t;
    }
}

// The same with ONE barrier: consecutive calls alternate between two buffers (red: [2][NT / 64][NV]; flip toggles) -- a buffer is
// written again only behind the barrier of the call in between, which every thread passes after its reads.
template <int NV, int NT>
__device__ inline void block_sums_alt(double (&v)[NV], double *red, int &flip)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *r = red + flip * (NT / 64) * NV;
    flip ^= 1;
#pragma unroll
    for (int q = 0; q < NV; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
    }
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < NV; q++) r[wave * NV + q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; q++) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) t += r[w * NV + q];
        v[q] = t;
    }
}

// ps: first reflector of the current lazy super-panel (Wp row k - ps holds w_k); g0 >= 0: finish the w vectors of the group of
// columns g0 .. g0+BW-1 (needs Z4 / part4 from symv4_kernel); r0 >= 0: form the reflectors of columns r0 .. r0+BW-1.
__global__ __launch_bounds__(BTHREADS) void band_step_kernel(const double *__restrict__ At, double *__restrict__ Vall, double *__restrict__ Wp,
                                                             double *__restrict__ tau, double *__restrict__ band, double *__restrict__ Z4,
                                                             const double *__restrict__ gramG, const double *__restrict__ Xp,
                                                             const int *__restrict__ n, int ld, int ps, int g0, int r0)
{
    extern __shared__ double sm[];
    double *vl = sm;                        // [BW][ld]: phase W the group's v_c, phase P the panel's columns
    double *coef = sm + (size_t)BW * ld;    // [2][BTPL][BW]
    double *red = coef + 2 * BTPL * BW;     // [16][4]
    double *sc = red + 16 * 4;              // [BW][BW] w_c' . v_c
    const int s = blockIdx.x, ns = n[s], tid = threadIdx.x;
    const long so = (long)s * ld * ld;
    const double *A = At + so;
    double *V = Vall + so, *W = Wp + (long)s * BTPL * ld, *tv = tau + (long)s * ld, *bd = band + (long)s * (BW + 1) * ld;
    double *Z = Z4 + (long)s * BW * ld;

    // ------------------------------------------------------------------ phase W
    // p_c = M v_c without the group's own corrections arrives in Z (symv4_kernel + band_apply_kernel: row sums, the strips'
    // transposed partials and the super-panel's earlier reflectors are already in); what is left is the part that is
    // sequential inside the group: w_c = tau_c (p_c - sum_{c' < c} (v_c' (w_c' . v_c) + w_c' (v_c' . v_c))) + alpha_c v_c.
    if (g0 >= 0 && g0 + BW + 1 < ns) {  // (a group whose first pivot has nothing below it has no reflector: tau = 0, w = 0)
        const int G = min(BW, ns - g0), kc = g0 - ps;
        const double *gram = gramG + (long)s * BW * BW;  // v_c' . v_c (c' < c), from the dot blocks of symv4_kernel
        for (int c = 0; c < G; c++) {
            const double tc = tv[g0 + c];
            double *wc = W + (long)(kc + c) * ld;
            const double *vc = V + (long)(g0 + c) * ld;
            double a_[BW], b_[BW];
#pragma unroll
            for (int q = 0; q < BW; q++) { a_[q] = q < c ? sc[q * BW + c] : 0.0; b_[q] = q < c ? gram[q * BW + c] : 0.0; }
            double dot[1] = {0.0};
            for (int i = g0 + 1 + tid; i < ns; i += BTHREADS) {
                double p = Z[c * ld + i];
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q < c) p -= V[(long)(g0 + q) * ld + i] * a_[q] + W[(long)(kc + q) * ld + i] * b_[q];
                const double wprime = tc * p;
                wc[i] = wprime;
                dot[0] += wprime * vc[i];
            }
            block_sums<1>(dot, red);
            const double alpha = -0.5 * tc * dot[0];
            double d3[BW];
#pragma unroll
            for (int q = 0; q < BW; q++) d3[q] = 0.0;
            for (int i = g0 + 1 + tid; i < ns; i += BTHREADS) {
                const double w = wc[i] + alpha * vc[i];
                wc[i] = w;
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q > c && q < G) d3[q] += w * V[(long)(g0 + q) * ld + i];
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
        // the panel arrives from band_apply_kernel with the corrections of the super-panel's reflectors before the previous group;
        // that group's own (their w vectors come from phase W of this launch) are added here.  First group of a super-panel: At itself.
        const int k0 = kc > 0 ? kc - BW : 0;
        for (int e = tid; e < (kc - k0) * BW; e += BTHREADS) {
            const int k = k0 + e / BW, c = e % BW;
            const bool in = c < G;
            coef[e] = in ? V[(long)(ps + k) * ld + r0 + c] : 0.0;            // v_k[column]
            coef[BTPL * BW + e] = in ? W[(long)k * ld + r0 + c] : 0.0;      // w_k[column]
        }
        __syncthreads();
        const double *X = Xp + (long)s * BW * ld;
        for (int i = r0 + tid; i < ns; i += BTHREADS) {
            double x[BW];
            if (kc > 0) {
#pragma unroll
                for (int c = 0; c < BW; c++) x[c] = X[(long)c * ld + i];
            } else {
                // columns r0 .. r0+BW-1 of row i: the lower triangle (the trailing updates leave the upper tiles behind), 32 bytes per row
                const double2 *ap = (const double2 *)(A + (long)i * ld + r0);
                const double2 a01 = ap[0], a23 = ap[1];
                x[0] = a01.x; x[1] = a01.y; x[2] = a23.x; x[3] = a23.y;
#pragma unroll
                for (int c = 0; c < BW; c++)
                    if (c >= G) x[c] = 0.0;
            }
            for (int k = k0; k < kc; k++) {
                const double a = V[(long)(ps + k) * ld + i], b = W[(long)k * ld + i];
#pragma unroll
                for (int c = 0; c < BW; c++) x[c] -= a * coef[(BTPL + k - k0) * BW + c] + b * coef[(k - k0) * BW + c];
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

// The same step with every vector in REGISTERS or the thread's own LDS slots (512 threads, ld <= NR x 512: thread t owns the rows
// base + t + 512 r): the group's reflectors,
// w vectors and the panel never touch LDS or memory between the passes, and the algebra is arranged so that each column needs
// ONE reduction over the workgroup -- phase W: sum_i w'_c v_q for q >= c gives the dot (q = c) and, with the group's Gram
// matrix, w_c . v_q = w'_c . v_q + alpha v_c . v_q; phase P: S_q = sum_{i > piv} x_q x_c and the pivot row x_q[piv] give the norm
// (q = c) and (M v)[q] = x_q[piv] + scale S_q.  8 reductions per launch instead of 16, no passes over memory in between.
template <int NR>
__global__ __launch_bounds__(RTHREADS) void band_step_reg_kernel(const double *__restrict__ At, double *__restrict__ Vall, double *__restrict__ Wp,
                                                                 double *__restrict__ tau, double *__restrict__ band, const double *__restrict__ Z4,
                                                                 const double *__restrict__ gramG, const double *__restrict__ Xp,
                                                                 const int *__restrict__ n, int ld, int ps, int g0, int r0)
{
    extern __shared__ double sm[];
    double *vs = sm;                                  // [BW][NR][512]: the group's reflectors at this thread's rows (only their owner touches them)
    double *red = sm + (size_t)BW * NR * RTHREADS;    // [2][8][2 BW]: two buffers, block_sums_alt
    double (*cw)[BW][BW] = (double (*)[BW][BW])(red + 2 * 16 * 2 * BW);  // v_q[r0 + c], w_q[r0 + c] of the group whose w vectors phase W has just made
#define VS_(c, r) vs[((c) * NR + (r)) * RTHREADS + tid]
    const int s = blockIdx.x, ns = n[s], tid = threadIdx.x;
    int flip = 0;
    const long so = (long)s * ld * ld;
    const double *A = At + so;
    double *V = Vall + so, *W = Wp + (long)s * BTPL * ld, *tv = tau + (long)s * ld, *bd = band + (long)s * (BW + 1) * ld;
    const int base = g0 >= 0 ? g0 + 1 : r0;
    double w[BW][NR];
#pragma unroll
    for (int c = 0; c < BW; c++)
#pragma unroll
        for (int r = 0; r < NR; r++) { VS_(c, r) = 0.0; w[c][r] = 0.0; }
    // ------------------------------------------------------------------ phase W (see band_step_kernel)
    if (g0 >= 0 && g0 + BW + 1 < ns) {
        const int kc = g0 - ps;
        const double *Z = Z4 + (long)s * BW * ld, *gram = gramG + (long)s * BW * BW;
        double z[BW][NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = base + tid + r * RTHREADS;
#pragma unroll
            for (int c = 0; c < BW; c++) {
                z[c][r] = i < ns ? Z[(long)c * ld + i] : 0.0;
                VS_(c, r) = i < ns ? V[(long)(g0 + c) * ld + i] : 0.0;
            }
        }
        double gq[BW][BW], wv[BW][BW];  // v_q . v_c and w_q . v_c (q < c)
#pragma unroll
        for (int q = 0; q < BW; q++)
#pragma unroll
            for (int c = 0; c < BW; c++) { gq[q][c] = q < c ? gram[q * BW + c] : 0.0; wv[q][c] = 0.0; }
#pragma unroll
        for (int c = 0; c < BW; c++) {
            const double tc = tv[g0 + c];
            double sums[BW];
#pragma unroll
            for (int q = 0; q < BW; q++) sums[q] = 0.0;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                double p = z[c][r];
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q < c) p -= VS_(q, r) * wv[q][c] + w[q][r] * gq[q][c];
                const double wp = tc * p;
                w[c][r] = wp;
#pragma unroll
                for (int q = 0; q < BW; q++)
                    if (q >= c) sums[q] += wp * VS_(q, r);
            }
            block_sums_alt<BW, RTHREADS>(sums, red, flip);
            const double alpha = -0.5 * tc * sums[c];
#pragma unroll
            for (int r = 0; r < NR; r++) w[c][r] += alpha * VS_(c, r);
#pragma unroll
            for (int q = 0; q < BW; q++)
                if (q > c) wv[c][q] = sums[q] + alpha * gq[c][q];
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = base + tid + r * RTHREADS;
            if (i < ns)
#pragma unroll
                for (int c = 0; c < BW; c++) W[(long)(kc + c) * ld + i] = w[c][r];
        }
    }
    // ------------------------------------------------------------------ phase P
    if (r0 >= 0 && r0 < ns) {
        const int G = min(BW, ns - r0), kc = r0 - ps;
        // The panel arrives from band_apply_kernel with the corrections of the super-panel's reflectors before the previous group;
        // that group's own are in this thread's registers, their entries at the panel's columns come from the rows' owners.
        // (kc > 0 implies that phase W ran in this launch; a group without reflectors left zeros.)  kc = 0: At itself.
        if (kc > 0) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int i = base + tid + r * RTHREADS;
                if (i >= r0 && i < r0 + BW)
#pragma unroll
                    for (int q = 0; q < BW; q++) { cw[0][q][i - r0] = VS_(q, r); cw[1][q][i - r0] = w[q][r]; }
            }
        }
        __syncthreads();
        const double *X = Xp + (long)s * BW * ld;
        double x[BW][NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = base + tid + r * RTHREADS;
            const bool act = i >= r0 && i < ns;
            if (kc > 0) {
#pragma unroll
                for (int c = 0; c < BW; c++) x[c][r] = act && c < G ? X[(long)c * ld + i] : 0.0;
#pragma unroll
                for (int q = 0; q < BW; q++)
#pragma unroll
                    for (int c = 0; c < BW; c++) x[c][r] -= VS_(q, r) * cw[1][q][c] + w[q][r] * cw[0][q][c];
#pragma unroll
                for (int c = 0; c < BW; c++)
                    if (!(act && c < G)) x[c][r] = 0.0;
            } else if (act) {
                const double2 *ap = (const double2 *)(A + (long)i * ld + r0);  // the lower triangle: 32 bytes per row
                const double2 a01 = ap[0], a23 = ap[1];
                x[0][r] = a01.x; x[1][r] = a01.y; x[2][r] = a23.x; x[3][r] = a23.y;
#pragma unroll
                for (int c = 0; c < BW; c++)
                    if (c >= G) x[c][r] = 0.0;
            } else {
#pragma unroll
                for (int c = 0; c < BW; c++) x[c][r] = 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < BW; c++) {
            if (c >= G) break;
            const int rr = r0 + c, piv = rr + BW;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int i = base + tid + r * RTHREADS;
                if (i >= rr && i < rr + BW && i < ns) bd[(long)(i - rr) * ld + rr] = x[c][r];  // final: later reflectors start below
            }
            if (piv < ns) {
                double sums[2 * BW];
#pragma unroll
                for (int q = 0; q < 2 * BW; q++) sums[q] = 0.0;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const int i = base + tid + r * RTHREADS;
#pragma unroll
                    for (int q = 0; q < BW; q++)
                        if (q >= c) {
                            if (i > piv) sums[q] += x[q][r] * x[c][r];
                            if (i == piv) sums[BW + q] = x[q][r];
                        }
                }
                block_sums_alt<2 * BW, RTHREADS>(sums, red, flip);
                const BHouse h = bhouse(sums[BW + c], sums[c]);
                if (tid == 0) { bd[(long)BW * ld + rr] = h.beta; tv[rr] = h.tau; }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const int i = base + tid + r * RTHREADS;
                    const double vv = i == piv ? 1.0 : (i > piv ? h.scale * x[c][r] : 0.0);  // (x is zero beyond the stamp)
                    if (i >= piv && i < ns) V[(long)rr * ld + i] = vv;
#pragma unroll
                    for (int q = 0; q < BW; q++)
                        if (q > c) x[q][r] -= vv * (h.tau * (sums[BW + q] + h.scale * sums[q]));
                }
            } else if (tid == 0) {
                bd[(long)BW * ld + rr] = 0.0;
                tv[rr] = 0.0;
            }
        }
    }
}
#undef VS_

// dot blocks of the symmetric-product launches, one wave per task: (V_k . v_c, W_k . v_c) for the super-panel's earlier reflectors k
// (coefG[s][0 / 1][k][c]), then the group's own v_c' . v_c (gramG[s][c'][c])
__device__ __forceinline__ void symv4_dot_blocks(const double *__restrict__ Vall, const double *__restrict__ Wp, long so, int s, int ns, int ld, int r0, int ps,
                                                 int task, int lane, double *__restrict__ coefG, double *__restrict__ gramG)
{
    const int kc = r0 - ps;
    const double *v0 = Vall + so + (long)r0 * ld;
    if (task < kc) {
        const double *vk = Vall + so + (long)(ps + task) * ld, *wk = Wp + ((long)s * BTPL + task) * ld;
        double dv[BW], dw[BW];
#pragma unroll
        for (int c = 0; c < BW; c++) dv[c] = dw[c] = 0.0;
        for (int i = r0 + BW + lane; i < ns; i += 64) {
            const double a = vk[i], b = wk[i];
#pragma unroll
            for (int c = 0; c < BW; c++) { const double x = v0[(long)c * ld + i]; dv[c] += a * x; dw[c] += b * x; }
        }
#pragma unroll
        for (int c = 0; c < BW; c++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { dv[c] += __shfl_xor(dv[c], off, 64); dw[c] += __shfl_xor(dw[c], off, 64); }
        }
        if (lane == 0)
#pragma unroll
            for (int c = 0; c < BW; c++) {
                coefG[(((long)s * 2 + 0) * BTPL + task) * BW + c] = dv[c];
                coefG[(((long)s * 2 + 1) * BTPL + task) * BW + c] = dw[c];
            }
    } else if (task - kc < BW * (BW - 1) / 2) {
        int c1 = 0, q = task - kc;
        while (q >= BW - 1 - c1) { q -= BW - 1 - c1; c1++; }
        const int c2 = c1 + 1 + q;
        double d = 0.0;
        for (int i = r0 + BW + lane; i < ns; i += 64) d += v0[(long)c1 * ld + i] * v0[(long)c2 * ld + i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
        if (lane == 0) gramG[(long)s * BW * BW + c1 * BW + c2] = d;
    }
}

// The same pass with BOTH products on the fp64 matrix pipe (v_mfma_f64_16x16x4), which the path's one bandwidth-bound kernel left
// idle (SQ_VALU_MFMA_BUSY 0.000 while the waves waited on memory a third of their time and on their own arithmetic -- 16 flops per
// 8-byte element, three cross-lane steps per chunk for the transposed sums -- for the rest).  A wave takes a 64-row x 16-column slab
// of the strip per chunk as four 16 x 16 tiles.  A tile is loaded ONCE, in the layout of the MFMA's B operand (lane = (row 4 kk + lane / 16,
// column lane % 16): 4 rows x 128 contiguous bytes per instruction), and serves
//   the transposed product  Zt[c][j] += sum_i V[i][c] T[i][j]  as that B operand (A = V^T at the strip's rows, padded from 4 to 16
//                           reflectors: the pipe is idle, the padding costs nothing that was in use) -- the sum over the rows is the
//                           instruction's own k sum: no cross-lane step, one coalesced 8-byte store per lane and chunk;
//   the row product         Z[i][c]  += sum_j T[i][j] V[j][c]  as the A operand of a second MFMA, for which the k index has to be the
//                           COLUMN: the tile goes through 2.3 KB of the wave's own LDS (written row-major, read transposed, stride 18:
//                           conflict-free both ways; no barrier -- a wave's LDS instructions execute in order).
// Per tile: 4 global loads, 4 + 4 LDS instructions, 8 MFMAs, where the VALU form issued 44 instructions; the 32 MFMAs of a chunk take
// 2048 cycles of a SIMD's pipe per 8 KB = 9.4 TB/s chip-wide, twice the rate HBM delivers.
typedef double sv_f64x4 __attribute__((ext_vector_type(4)));
constexpr int SV_TLD = 18;
template <int ABL>  // ablation (timing runs only, results are then garbage): 1 = loads alone, 2 = + transposed product and its stores, 3 = + LDS round trip without the row MFMAs
__global__ __launch_bounds__(256) void symv4_mfma_kernel(const double *__restrict__ At, const double *__restrict__ Vall, const double *__restrict__ Wp,
                                                         const int *__restrict__ n, int ld, int r0, int ps, int nrowtiles, double *__restrict__ Z4,
                                                         double *__restrict__ part4, double *__restrict__ coefG, double *__restrict__ gramG)
{
    static_assert(BSTRIP == 64 && BW == 4, "four 16-row tiles per slab, four reflectors");
    __shared__ __attribute__((aligned(16))) double tl[4][16 * SV_TLD];
    __shared__ double racc[4][BSTRIP][BW];
    const int s = blockIdx.y, ns = n[s];
    if (r0 + BW + 1 >= ns) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long so = (long)s * ld * ld;
    if ((int)blockIdx.x >= nrowtiles) {
        symv4_dot_blocks(Vall, Wp, so, s, ns, ld, r0, ps, ((int)blockIdx.x - nrowtiles) * 4 + wave, lane, coefG, gramG);
        return;
    }
    const int strip = ((r0 + 1) / BSTRIP) + (nrowtiles - 1 - (int)blockIdx.x), rb = strip * BSTRIP;  // longest strips first
    if (rb >= ns) return;
    const int lr = lane >> 4, lc = lane & 15;
    double *part_s = part4 + (((long)s * (ld / BSTRIP) + strip) * BW) * ld;
    double *tile = tl[wave];
    // one buffer descriptor per operand; a lane that has nothing to load (reflector index >= BW) points beyond the buffer and gets 0.0
    const int nrec = (int)((long)ld * ld * 8);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(At + so), 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(Vall + so), 0, nrec, 0x00020000);
    auto ld1 = [&](const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
        typedef int v2i_ __attribute__((ext_vector_type(2)));
        const v2i_ q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        return __builtin_bit_cast(double, q);
    };
    const int oob = 0x7ffffff0;
    const int voff_a = (lr * ld + wave * 16 + lc) * 8;                           // tile element (row 4 kk + lr, column lc) of the wave's slab
    const int voff_vc = lc < BW ? ((r0 + lc) * ld + wave * 16 + lr) * 8 : oob;  // B operand of the row product: V[column 4 kk + lr][reflector lc]
    const int voff_vt = lc < BW ? ((r0 + lc) * ld + lr) * 8 : oob;              // A operand of the transposed product: V[row 4 kk + lr][reflector lc]
    // V^T at the strip's rows (rows beyond the stamp: the reflectors are zero there)
    double vt[16];
#pragma unroll
    for (int q = 0; q < 16; q++) vt[q] = ld1(rv, voff_vt, (rb + 4 * q) * 8);
    sv_f64x4 drow[4];
#pragma unroll
    for (int t = 0; t < 4; t++) drow[t] = sv_f64x4{0.0, 0.0, 0.0, 0.0};
    const int cend = rb + BSTRIP, cfirst = (r0 + 1) & ~(SCHUNK - 1), clast = cend - SCHUNK;
    // The slab of a chunk is fetched and worked on in two HALVES of 32 rows, TWO halves ahead of the one being worked on.  Loads and
    // stores leave a wave's vmcnt counter in order: with one chunk in flight (the first version, and the VALU form) the wait for the
    // next chunk's loads also waited for the previous chunk's store of the transposed sums to be acknowledged -- 100 of 590 ms at batch
    // 256, the whole gap to the loads-only ablation.  With two halves in flight the store of chunk k is younger than the loads the next
    // two halves wait for, and long done when a wait first reaches it.  (Always the same loads in flight, past the strip's end the last
    // chunk once more -- a fetch under a branch loses the prefetch in the machine code, see symv4_kernel --, nothing consumes a loaded
    // value inside fetch.)
    auto fetch = [&](auto half_c, double (&rn)[8], double (&vn)[4], int c0) {
        constexpr int H = decltype(half_c)::value;
        const int cc = min(c0, clast);
        if (H == 0)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) vn[kk] = ld1(rv, voff_vc, (cc + 4 * kk) * 8);
#pragma unroll
        for (int q = 0; q < 8; q++) rn[q] = ld1(ra, voff_a, ((rb + 32 * H + 4 * q) * ld + cc) * 8);
    };
    sv_f64x4 dt = {0.0, 0.0, 0.0, 0.0};
    auto process = [&](auto half_c, const double (&r)[8], const double (&vc)[4], int c0) {
        constexpr int H = decltype(half_c)::value;
        const int cw = c0 + wave * 16;
        if (ABL == 1) {
#pragma unroll
            for (int q = 0; q < 8; q++) drow[q & 3][0] += r[q] + vc[q & 3];
            return;
        }
        if (cw < rb) {  // (wave-uniform) columns left of the diagonal block receive the transposed contributions
            if (H == 0) dt = sv_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 8; q++) dt = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[8 * H + q], r[q], dt, 0, 0, 0);
            if (H == 1) part_s[(long)lr * ld + cw + lc] = dt[0];  // D[reflector lr + 4 reg][column lc]: register 0 holds the four reflectors
        }
        if (ABL == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) drow[q][0] += vc[q];
            return;
        }
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 4; kk++) tile[(4 * kk + lr) * SV_TLD + lc] = r[4 * tt + kk];
            __builtin_amdgcn_wave_barrier();
            double rt[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) rt[kk] = tile[lc * SV_TLD + 4 * kk + lr];
            if (ABL == 3) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) drow[2 * H + tt][kk] += rt[kk] * vc[kk];
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) drow[2 * H + tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(rt[kk], vc[kk], drow[2 * H + tt], 0, 0, 0);
            }
        }
    };
    const std::integral_constant<int, 0> h0c;
    const std::integral_constant<int, 1> h1c;
    double a0[8], a1[8], b0[8], b1[8], u0[4], u1[4];
    fetch(h0c, a0, u0, cfirst);
    fetch(h1c, a1, u0, cfirst);
    for (int c0 = cfirst; c0 < cend; c0 += 2 * SCHUNK) {
        fetch(h0c, b0, u1, c0 + SCHUNK);
        process(h0c, a0, u0, c0);
        fetch(h1c, b1, u1, c0 + SCHUNK);
        process(h1c, a1, u0, c0);
        if (c0 + SCHUNK >= cend) break;
        fetch(h0c, a0, u0, c0 + 2 * SCHUNK);
        process(h0c, b0, u1, c0 + SCHUNK);
        fetch(h1c, a1, u0, c0 + 2 * SCHUNK);
        process(h1c, b1, u1, c0 + SCHUNK);
    }
    // row sums of the four waves (each has walked its own columns): D[row lr + 4 reg][reflector lc] of tile t
    if (lc < BW)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) racc[wave][16 * t + lr + 4 * rg][lc] = drow[t][rg];
    __syncthreads();
    if (threadIdx.x < BSTRIP * BW) {
        const int i = threadIdx.x / BW, c = threadIdx.x % BW;
        if (rb + i < ns) Z4[((long)s * BW + c) * ld + rb + i] = (racc[0][i][c] + racc[1][i][c]) + (racc[2][i][c] + racc[3][i][c]);
    }
}

// Z[c][i] = (At v_c)[i] for the BW reflectors of columns r0 .. r0+BW-1, rows i > r0: one pass over the trailing lower triangle.
// A workgroup takes a strip of 32 rows, walks its columns in chunks of 128 (32 per wave, a double2 per lane and row) and
// leaves (a) the row sums over the columns up to the strip's diagonal block and (b), from the same loads, the strip's
// contributions to the rows left of it (part4[strip][c][column]); band_apply_kernel adds the strips up.
__global__ __launch_bounds__(256) void symv4_kernel(const double *__restrict__ At, const double *__restrict__ Vall, const double *__restrict__ Wp,
                                                    const int *__restrict__ n, int ld, int r0, int ps, int nrowtiles, double *__restrict__ Z4,
                                                    double *__restrict__ part4, double *__restrict__ coefG, double *__restrict__ gramG)
{
    __shared__ __attribute__((aligned(16))) double urs[BSTRIP][BW];
    __shared__ double racc[4][BSTRIP][BW];
    const int s = blockIdx.y, ns = n[s];
    if (r0 + BW + 1 >= ns) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long so = (long)s * ld * ld;
    if ((int)blockIdx.x >= nrowtiles) {
        symv4_dot_blocks(Vall, Wp, so, s, ns, ld, r0, ps, ((int)blockIdx.x - nrowtiles) * 4 + wave, lane, coefG, gramG);
        return;
    }
    // longest strips (bottom of the matrix) first: they bound the launch's critical path
    const int strip = ((r0 + 1) / BSTRIP) + (nrowtiles - 1 - (int)blockIdx.x), rb = strip * BSTRIP;
    if (rb >= ns) return;
    if (threadIdx.x < BSTRIP * BW) {
        const int i = threadIdx.x / BW, c = threadIdx.x % BW;
        urs[i][c] = rb + i < ns ? Vall[so + (long)(r0 + c) * ld + rb + i] : 0.0;  // the reflectors at the strip's rows
    }
    __syncthreads();
    // A chunk is BSTRIP rows x SCHUNK columns; wave w takes ALL rows of its SWCOLS columns: lane = (row group g of 8 rows,
    // column pair lp).  The transposed contributions of a column are then complete inside the wave (two shuffles over g) and go
    // straight to memory -- the main loop has no LDS exchange and no barrier (the first version split the chunk by rows: one
    // barrier per chunk, 3.6 TB/s at best).  The row sums meet once, at the end of the strip.
    const int g = lane / SLPW, lp = lane % SLPW;
    const int colw = wave * SWCOLS + lp * 2;
    double *part_s = part4 + (((long)s * (ld / BSTRIP) + strip) * BW) * ld;
    double acc[BW][8];
#pragma unroll
    for (int c = 0; c < BW; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[c][i] = 0.0;
    const int cend = rb + BSTRIP, cfirst = (r0 + 1) & ~(SCHUNK - 1);
    // Buffer loads: one descriptor per operand (wave-uniform), the chunk's part of the address in a scalar offset, the lane's in ONE
    // vector register (with 64-bit pointers the 12 loads of a chunk hold 24 registers of addresses per register set)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(At + so), 0, (int)((long)ld * ld * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(Vall + so), 0, (int)((long)ld * ld * 8), 0x00020000);
    const int voff_a = ((g * 8) * ld + colw) * 8, voff_u = colw * 8;
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const v4i_ q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return __builtin_bit_cast(double2, q);
    };
    // Always twelve loads, past the strip's end the last chunk once more (unused): with the fetch under a branch the compiler
    // merges "twelve more loads in flight" with "none" at the join and lets the arithmetic of chunk k wait for chunk k+1's loads
    // (vmcnt(7) .. vmcnt(0) where 19 .. 12 were due) -- the prefetch was there in the source and not in the machine code.  For the
    // same reason nothing may consume a loaded value here (a mask on the reflector columns waited for the load just issued);
    // a chunk never reaches past the strip's diagonal block, so there is nothing to mask.
    const int clast = cend - SCHUNK;
    auto fetch = [&](double2 (&an)[8], double2 (&un)[BW], int c0) {
        const int cc = min(c0, clast);
#pragma unroll
        for (int c = 0; c < BW; c++) un[c] = ld2(rv, voff_u, ((r0 + c) * ld + cc) * 8);
#pragma unroll
        for (int i = 0; i < 8; i++) an[i] = ld2(ra, voff_a, ((rb + i) * ld + cc) * 8);
    };
    auto process = [&](const double2 (&a)[8], const double2 (&uu)[BW], int c0) {
        double2 y2[BW];
#pragma unroll
        for (int c = 0; c < BW; c++) y2[c] = make_double2(0.0, 0.0);
        asm volatile("" ::: "memory");  // the reflector rows are re-read from LDS in every chunk: hoisted out of the loop they take 64 registers
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const double2 u01 = *(const double2 *)&urs[g * 8 + i][0], u23 = *(const double2 *)&urs[g * 8 + i][2];
            const double ur[BW] = {u01.x, u01.y, u23.x, u23.y};
#pragma unroll
            for (int c = 0; c < BW; c++) {
                acc[c][i] += a[i].x * uu[c].x + a[i].y * uu[c].y;
                y2[c].x += a[i].x * ur[c];
                y2[c].y += a[i].y * ur[c];
            }
        }
        if (c0 + wave * SWCOLS < rb) {  // (wave-uniform) columns left of the diagonal block receive the transposed contributions
#pragma unroll
            for (int c = 0; c < BW; c++) {
#pragma unroll
                for (int off = SLPW; off < 64; off <<= 1) {
                    y2[c].x += __shfl_xor(y2[c].x, off, 64);
                    y2[c].y += __shfl_xor(y2[c].y, off, 64);
                }
            }
            if (g == 0)
#pragma unroll
                for (int c = 0; c < BW; c++) *(double2 *)(part_s + (long)c * ld + c0 + colw) = y2[c];
        }
    };
    double2 s0[8], s1[8], u0[BW], u1[BW];
    fetch(s0, u0, cfirst);
    for (int c0 = cfirst; c0 < cend; c0 += 2 * SCHUNK) {
        fetch(s1, u1, c0 + SCHUNK);
        process(s0, u0, c0);
        if (c0 + SCHUNK >= cend) break;
        fetch(s0, u0, c0 + 2 * SCHUNK);
        process(s1, u1, c0 + SCHUNK);
    }
    // row sums: over the 16 column pairs of the wave, then over the four waves
#pragma unroll
    for (int c = 0; c < BW; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int off = SLPW / 2; off > 0; off >>= 1) acc[c][i] += __shfl_xor(acc[c][i], off, 64);
        }
    if (lp == 0)
#pragma unroll
        for (int c = 0; c < BW; c++)
#pragma unroll
            for (int i = 0; i < 8; i++) racc[wave][g * 8 + i][c] = acc[c][i];
    __syncthreads();
    if (threadIdx.x < BSTRIP * BW) {
        const int i = threadIdx.x / BW, c = threadIdx.x % BW;
        if (rb + i < ns) Z4[((long)s * BW + c) * ld + rb + i] = (racc[0][i][c] + racc[1][i][c]) + (racc[2][i][c] + racc[3][i][c]);
    }
}

// Z[c][i] <- row sums + the transposed partials of the strips below row i - the super-panel's earlier reflectors:
// p_c = (At - sum_k (v_k w_k^T + w_k v_k^T)) v_c on the rows i > g0, many workgroups per stamp (in the first version ONE
// workgroup per stamp did this inside band_step_kernel: 8.8 MB per group and stamp through one CU, 147 us per launch).
// From the same loads of v_k, w_k: the NEXT group's panel, Xp[c][i] = (At - sum_k (v_k w_k^T + w_k v_k^T))[i][g0 + BW + c] for the
// reflectors k of the super-panel before group g0 (band_step_kernel adds group g0's own four when its w vectors exist; it
// used to read all of them, up to 2 x 60 vectors per stamp through one CU).
__global__ __launch_bounds__(256) void band_apply_kernel(const double *__restrict__ At, const double *__restrict__ Vall, const double *__restrict__ Wp,
                                                         const double *__restrict__ part4, const double *__restrict__ coefG, const int *__restrict__ n,
                                                         int ld, int g0, int ps, double *__restrict__ Z4, double *__restrict__ Xp)
{
    __shared__ double cf[4 * BTPL * BW];
    const int s = blockIdx.y, ns = n[s], r0 = g0 + BW;
    if (r0 >= ns) return;
    const bool doz = g0 + BW + 1 < ns;  // (otherwise the group has no reflectors: only the last panel is due)
    const int kc = g0 - ps, G = min(BW, ns - r0);
    const long so = (long)s * ld * ld;
    const double *V = Vall + so + (long)ps * ld, *W = Wp + (long)s * BTPL * ld;
    for (int e = threadIdx.x; e < kc * BW; e += 256) {
        const int k = e / BW, c = e % BW;
        cf[e] = coefG[((long)s * 2 + 0) * BTPL * BW + e];
        cf[BTPL * BW + e] = coefG[((long)s * 2 + 1) * BTPL * BW + e];
        cf[2 * BTPL * BW + e] = c < G ? V[(long)k * ld + r0 + c] : 0.0;  // v_k[column]
        cf[3 * BTPL * BW + e] = c < G ? W[(long)k * ld + r0 + c] : 0.0;  // w_k[column]
    }
    __syncthreads();
    const int i = g0 + 1 + blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const double *part = part4 + (long)s * (ld / BSTRIP) * BW * ld;
    double *Z = Z4 + (long)s * BW * ld;
    double p[BW], x[BW];
#pragma unroll
    for (int c = 0; c < BW; c++) p[c] = doz ? Z[(long)c * ld + i] : 0.0;
    {
        // columns r0 .. r0+BW-1 of row i: the lower triangle (the trailing updates leave the upper tiles behind), 32 bytes per row
        const double2 *ap = (const double2 *)(At + so + (long)i * ld + r0);
        const double2 a01 = ap[0], a23 = ap[1];
        x[0] = a01.x; x[1] = a01.y; x[2] = a23.x; x[3] = a23.y;
#pragma unroll
        for (int c = 0; c < BW; c++)
            if (c >= G) x[c] = 0.0;
    }
    if (doz) {
        const int slast = (ns - 1) / BSTRIP;
        for (int st = i / BSTRIP + 1; st <= slast; st++)
#pragma unroll
            for (int c = 0; c < BW; c++) p[c] += part[((long)st * BW + c) * ld + i];
    }
    for (int k = 0; k < kc; k++) {
        const double a = V[(long)k * ld + i], b = W[(long)k * ld + i];
#pragma unroll
        for (int c = 0; c < BW; c++) {
            p[c] -= a * cf[(BTPL + k) * BW + c] + b * cf[k * BW + c];
            x[c] -= a * cf[(3 * BTPL + k) * BW + c] + b * cf[(2 * BTPL + k) * BW + c];
        }
    }
    double *X = Xp + (long)s * BW * ld;
#pragma unroll
    for (int c = 0; c < BW; c++) {
        if (doz) Z[(long)c * ld + i] = p[c];
        if (i >= r0) X[(long)c * ld + i] = x[c];
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

bool band_basis_fits(int ld) { return ((size_t)BW * ld + 2 * BTPL * BW + 16 * 4 + BW * BW) * 8 <= 160 * 1024; }

// what band_basis_device leaves behind for trd_apply_q (the rest of band_basis_ws_bytes is scratch, free again when it returns)
size_t band_basis_keep_bytes(int batch, int ld, int mp)
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
    add((size_t)batch * 2 * NB * mp * 8 * 2);               // W1, W2 (256 rows: pairs of panels)
    add((size_t)batch * (npanels / 2) * 4 * NB * NB * 8);   // factors of the pairs
    return align_up(t, 256);
}

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
    add((size_t)batch * 2 * NB * mp * 8 * 2);               // W1, W2 (256 rows: pairs of panels)
    add((size_t)batch * (npanels / 2) * 4 * NB * NB * 8);   // factors of the pairs
    add((size_t)batch * ld * ld * 8);                       // At
    add((size_t)batch * BTPL * ld * 8);                     // Wp
    add((size_t)batch * BW * ld * 8);                       // Z4
    add((size_t)batch * BW * ld * 8);                       // Xp
    add((size_t)batch * (ld / BSTRIP) * BW * ld * 8);       // part4
    add((size_t)batch * (2 * BTPL * BW + BW * BW) * 8);     // dots with the super-panel's reflectors, the group's Gram matrix
    return t + 8192;
}

int trd_panel_factors(imcom_ctx *ctx, TrdBasis *out, int batch);  // tridiag.hip

// A -> band B (out->band: [batch][BW+1][ld], band[t][i] = B[i+t][i]) and the reflectors of Q (out->Vall, out->tauvec), ready for
// trd_apply_q.  Same contract as trd_basis_device; the caller has reserved band_basis_ws_bytes().
int band_basis_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, int mp, const double *A, long lda, long strideA, TrdBasis *out,
                      const std::function<int(int)> &on_panel)
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
    out->W1 = (double *)ws_take(ctx, (size_t)batch * 2 * NB * mp * 8);
    out->W2 = (double *)ws_take(ctx, (size_t)batch * 2 * NB * mp * 8);
    out->T2 = (double *)ws_take(ctx, (size_t)batch * (npanels_max / 2) * 4 * NB * NB * 8);
    out->dvec = out->evec = nullptr;
    out->ld = ld;
    out->bw = BW;
    out->nmax = 0;
    for (int s = 0; s < batch; s++) out->nmax = std::max(out->nmax, n_host[s]);
    const int nmax = out->nmax;
    out->npanels = (std::max(nmax - BW - 1, 0) + NB - 1) / NB;
    if (!out->Vall || !out->tauvec || !out->band || !out->n_dev || !out->Tm || !out->Sm || !out->W1 || !out->W2 || (npanels_max >= 2 && !out->T2)) {
        set_error("internal: band workspace");
        return IMCOM_ERR_NOMEM;
    }
    hipStream_t st = ctx->stream;
    IMCOM_TRY(upload(ctx, out->n_dev, n_host, (size_t)batch));
    const size_t mark = ctx->ws_used;
    double *At = (double *)ws_take(ctx, mat);
    double *Wp = (double *)ws_take(ctx, (size_t)batch * BTPL * ld * 8);
    double *Z4 = (double *)ws_take(ctx, (size_t)batch * BW * ld * 8), *Xp = (double *)ws_take(ctx, (size_t)batch * BW * ld * 8);
    double *part4 = (double *)ws_take(ctx, (size_t)batch * (ld / BSTRIP) * BW * ld * 8);
    double *coefG = (double *)ws_take(ctx, (size_t)batch * 2 * BTPL * BW * 8), *gramG = (double *)ws_take(ctx, (size_t)batch * BW * BW * 8);
    if (!At || !Wp || !Z4 || !Xp || !part4 || !coefG || !gramG) { set_error("internal: band workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_HIP_CHECK(hipMemsetAsync(out->Vall, 0, mat, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(out->tauvec, 0, vecb, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(out->band, 0, (size_t)batch * (BW + 1) * ld * 8, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(Wp, 0, (size_t)batch * BTPL * ld * 8, st));
    hipLaunchKernelGGL(band_init_kernel, dim3((ld + 255) / 256, ld, batch), dim3(256), 0, st, A, lda, strideA, out->n_dev, At, ld);
    IMCOM_TRY(check_launch("band_init_kernel"));
    const size_t lds = ((size_t)BW * ld + 2 * BTPL * BW + 16 * 4 + BW * BW) * 8;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)band_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ProfScope ps_(ctx, "eigen_trd", nmax);
        int ps = 0;
        auto reg_lds = [](int nr) { return (size_t)(BW * nr * RTHREADS + 2 * 16 * 2 * BW + 2 * BW * BW) * 8; };
        IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)band_step_reg_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)reg_lds(4)));
        IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)band_step_reg_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)reg_lds(6)));
        auto step = [&](int ps_, int g0, int r0) {
            if (ld <= 4 * RTHREADS)
                hipLaunchKernelGGL(band_step_reg_kernel<4>, dim3(batch), dim3(RTHREADS), reg_lds(4), st, At, out->Vall, Wp, out->tauvec, out->band, Z4, gramG, Xp,
                                   out->n_dev, ld, ps_, g0, r0);
            else if (ld <= 6 * RTHREADS)
                hipLaunchKernelGGL(band_step_reg_kernel<6>, dim3(batch), dim3(RTHREADS), reg_lds(6), st, At, out->Vall, Wp, out->tauvec, out->band, Z4, gramG, Xp,
                                   out->n_dev, ld, ps_, g0, r0);
            else
                hipLaunchKernelGGL(band_step_kernel, dim3(batch), dim3(BTHREADS), lds, st, At, out->Vall, Wp, out->tauvec, out->band, Z4, gramG,
                                   Xp, out->n_dev, ld, ps_, g0, r0);
        };
        for (int r0 = 0; r0 < nmax; r0 += BW) {
            const int g0 = r0 - BW;  // the group whose w vectors are due
            if (g0 >= 0)
                hipLaunchKernelGGL(band_apply_kernel, dim3((nmax - g0 - 1 + 255) / 256, batch), dim3(256), 0, st, At, out->Vall, Wp, part4, coefG, out->n_dev,
                                   ld, g0, ps, Z4, Xp);
            if (r0 - ps >= BTPL) {
                // the previous group's w vectors, then the trailing two-sided update A[pe:, pe:] -= V W^T + W V^T; a new lazy panel
                step(ps, g0, -1);
                IMCOM_TRY(check_launch("band_step_kernel"));
                // The GEMM tiles are 128-aligned: start at the tile boundary at or below the panel's end.  The extra rows / columns
                // it touches are already reduced (their band entries have been taken out) and are never read again.
                const int pa = r0 / NB * NB, rem = ld - pa;
                const double *Vp = out->Vall + (long)ps * ld + pa, *Wq = Wp + pa;
                double *C = At + (long)pa * ld + pa;
                // (lower tiles only: the symmetric product and the panels read the trailing matrix from its lower triangle)
                IMCOM_TRY(launch_syr2k_lower(ctx, rem, BTPL, batch, Vp, ld, (long)ld * ld, Wq, ld, (long)BTPL * ld, C, ld, (long)ld * ld, -1.0));
                IMCOM_HIP_CHECK(hipMemsetAsync(Wp, 0, (size_t)batch * BTPL * ld * 8, st));
                ps = r0;
                step(ps, -1, r0);
            } else
                step(ps, g0, r0);
            if (r0 + BW + 1 < nmax) {
                const int nrowtiles = (nmax - 1) / BSTRIP - (r0 + 1) / BSTRIP + 1, ndot = (r0 - ps + BW * (BW - 1) / 2 + 3) / 4;
                ProfScope pf(ctx, "symv4", 1, true);  // (profile level 2: HIP events around this launch alone, for the HBM roofline of the pass)
                // IMCOM_SYMV4=mfma: both products on the matrix pipe (symv4_mfma_kernel).  Built in round 5 and measured: alone on one stream
                // as fast as the VALU form (537-552 against 527-555 ms per 256 cfg-3 stamps), 1.5 % slower beside a second sub-batch's
                // stream (5.39-5.44 against 5.31-5.34 ms per stamp at batch 256, 7.15-7.28 against 7.13-7.17 at 32) -- the pass is bounded by
                // its access pattern and its partial stores, not by arithmetic (profiles/r05_negative_results.txt item 1).  Not the default.
                // (read per call -- a test runs both forms in one process; two getenv beside a launch are nothing)
                const char *form = getenv("IMCOM_SYMV4"), *ablv = getenv("IMCOM_SYMV4_ABL");
                const bool use_mfma = form && strcmp(form, "mfma") == 0;
                const int abl = ablv ? atoi(ablv) : 0;  // (timing runs: parts of the kernel taken out)
#define IMCOM_SYMV4_LAUNCH(A_) hipLaunchKernelGGL(symv4_mfma_kernel<A_>, dim3(nrowtiles + ndot, batch), dim3(256), 0, st, At, out->Vall, Wp, out->n_dev, ld, r0, ps, nrowtiles, Z4, part4, coefG, gramG)
                if (use_mfma && abl == 1) IMCOM_SYMV4_LAUNCH(1);
                else if (use_mfma && abl == 2) IMCOM_SYMV4_LAUNCH(2);
                else if (use_mfma && abl == 3) IMCOM_SYMV4_LAUNCH(3);
                else if (use_mfma) IMCOM_SYMV4_LAUNCH(0);
#undef IMCOM_SYMV4_LAUNCH
                else
                    hipLaunchKernelGGL(symv4_kernel, dim3(nrowtiles + ndot, batch), dim3(256), 0, st, At, out->Vall, Wp, out->n_dev, ld, r0, ps, nrowtiles, Z4,
                                       part4, coefG, gramG);
            }
            IMCOM_TRY(check_launch("band step"));
            // the reflectors of columns < r0 + BW are final (phase P of this group has been queued): a whole panel of 128?
            if (on_panel && (r0 + BW) % NB == 0 && (r0 + BW) / NB - 1 < out->npanels) IMCOM_TRY(on_panel((r0 + BW) / NB - 1));
        }
        if (on_panel && out->npanels > 0 && ((nmax + BW - 1) / BW * BW) / NB - 1 < out->npanels - 1) IMCOM_TRY(on_panel(out->npanels - 1));  // the last, partial panel
    }
    ctx->ws_used = mark;  // the scratch is free again (same stream: everything queued so far runs before whatever reuses it)
    return on_panel ? IMCOM_OK : trd_panel_factors(ctx, out, batch);
}

}  // namespace imcom
