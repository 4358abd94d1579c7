// eigen.hip -- the eigendecomposition LA path (reference src/pyimcom/lakernel.py:141-223 EigenKernel)
// and the public batched eigensolver entry point.
//
// The reference diagonalises A = Q diag(lam) Q^T, takes P = (-B/2) Q and evaluates, per output pixel a and trial kappa,
//     Sigma_a = sum_i P_ai^2 / (lam_i + kappa)^2,    UC_a = 1 - sum_i (lam_i + 2 kappa) P_ai^2 / (lam_i + kappa)^2 / C
// (single kappa 154-172; multi kappa 174-223: routine.lakernel1 bisects kappa per pixel, then kappa *= C, line 222),
// and T = (P / (lam + kappa)) Q^T.  Both sums are invariants of b_a = (-B/2)[a] under any orthogonal change of basis:
//     D(kappa) = b^T (A + kappa)^-1 b,   S(kappa) = b^T (A + kappa)^-2 b,   Sigma = S,   UC = 1 - (D + kappa S) / C,
// so the eigenvectors are never needed.  Here the work is done in the TRIDIAGONAL basis A = Qh T Qh^T (tridiag.hip: the
// Householder reduction alone, Qh kept as reflectors):
//     c_a = Qh^T b_a                                  block reflectors applied to -B/2 [N][m]: fp64 MFMA GEMMs (trd_apply_q)
//     D, S at a trial kappa                           ONE forward sweep of the LDL^T of T + kappa I per pixel, O(N), with the
//                                                     derivative of the sweep carried along: S = -dD/dkappa  (tri_search_kernel)
//     y_a = (T + kappa_a I)^-1 c_a at the final kappa forward + backward substitution                     (tri_solve_kernel)
//     T_a = Qh y_a                                    block reflectors again, stored float32
// which removes the implicit QR iteration, the rotations of the eigenvector matrix and the explicit Qh (a third of the
// old path's time at N = 2.9k) and all of their workspace.  T + kappa I is positive definite for kappa > -lam_min, and then the
// LDL^T needs no pivoting (it is a Cholesky factorisation in disguise: backward stable).  The positive semi-definite A of this
// problem guarantees that for the whole bracket kappa >= kappaC[0] C > 0 -- but the reference serves ANY symmetric A: it
// diagonalises with numpy.linalg.eigh and divides by lam + kappa whatever its sign (lakernel.py:154-172, 199-223), and it is
// the kernel users turn to when the Cholesky factorisation fails.  So every stamp is CHECKED, not assumed: one sweep of the
// reduced matrix at the lowest kappa of the call (pd_check kernels; every kappa the bisection visits is larger) decides
// whether all pivots are positive.  Stamps that fail -- A + kappa_min I not positive definite, where an unpivoted LDL^T is only
// conditionally stable -- are re-solved, in the same call, through the eigendecomposition itself (eigen_fallback: tridiag.hip's
// eigensolver, P = (-B/2) Q, the reference's formulas / lakernel1 in the eigenbasis, T = (P / (lam + kappa)) Q^T) and reported
// with info[s] = 1.
// The bisection takes the same decisions as lakernel1 except where udc or sum2 sits within rounding of its bound.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "launchers.h"

namespace imcom {

// Cb[s][i][a] = B[s][a][i] (reference layout -B/2 [m][ldb] -> input-pixel-major [np][mp]); zero for a >= m, i >= n[s]
__global__ __launch_bounds__(256) void tri_pack_kernel(const double *__restrict__ B, long ldb, int m, const int *__restrict__ n,
                                                       double *__restrict__ Cb, int np, int mp)
{
    __shared__ double tile[32][33];
    const int s = blockIdx.z, i0 = blockIdx.y * 32, a0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ns = n[s];
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, i = i0 + tx;
        tile[r][tx] = (a < m && i < ns) ? B[(long)s * m * ldb + (long)a * ldb + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, a = a0 + tx;
        if (i < np && a < mp) Cb[((long)s * np + i) * mp + a] = tile[tx][r];
    }
}

// 1 / x for the pivots of the search sweeps: hardware reciprocal + two Newton steps (relative error ~1e-16; the dependent
// chain of the sweep is a third shorter than with the IEEE division sequence).  The final solve divides exactly.
__device__ __forceinline__ double fast_recip(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// One forward sweep of the LDL^T of T + kappa I against c, with its kappa-derivative:
//   delta_0 = d_0 + kappa, l_i = e_i / delta_i, delta_{i+1} = d_{i+1} + kappa - l_i e_i, z_{i+1} = c_{i+1} - l_i z_i
//   D = sum z_i^2 / delta_i = c^T (T + kappa)^-1 c,     S = -dD/dkappa = c^T (T + kappa)^-2 c
__device__ __forceinline__ void tri_sweep(const double *__restrict__ d, const double *__restrict__ e, const double *__restrict__ c,
                                          long cstride, int ns, double kap, double &D, double &S)
{
    double delta = d[0] + kap, dp = 1.0, z = c[0], zp = 0.0, cn = ns > 1 ? c[cstride] : 0.0;
    D = 0.0;
    S = 0.0;
    for (int i = 0; i < ns; i++) {
        const double cnext = i + 2 < ns ? c[(long)(i + 2) * cstride] : 0.0;  // one row ahead of its use
        const double r = fast_recip(delta), t = z * r;
        D += z * t;
        S += t * (t * dp - 2.0 * zp);
        if (i + 1 < ns) {
            const double b = e[i], l = b * r, lp = -l * r * dp;
            const double zn = cn - l * z, zpn = -lp * z - l * zp;
            delta = d[i + 1] + kap - l * b;
            dp = 1.0 - lp * b;
            z = zn;
            zp = zpn;
            cn = cnext;
        }
    }
}

// multi kappa: routine.lakernel1's bisection (routine.py:405-419) per output pixel, D and S from tri_sweep
__global__ __launch_bounds__(64) void tri_search_kernel(const double *__restrict__ dvec, const double *__restrict__ evec,
                                                        const double *__restrict__ Cb, int np, int mp, int m,
                                                        const int *__restrict__ n, const double *__restrict__ Cs,
                                                        const double *__restrict__ kmin, const double *__restrict__ kmax,
                                                        double targetleak, double smax, int nbis, double *__restrict__ kap_out)
{
    const int s = blockIdx.y, a = blockIdx.x * 64 + threadIdx.x;
    const int ns = n[s];
    if (a >= m || ns == 0) return;
    const double *d = dvec + (long)s * np, *e = evec + (long)s * np, *c = Cb + (long)s * np * mp + a;
    const double C = Cs[s], kCmin = kmin[s], kCmax = kmax[s];
    double factor = sqrt(kCmax / kCmin), kap = sqrt(kCmax * kCmin);
    for (int it = 0; it < nbis; it++) {
        double D, S;
        tri_sweep(d, e, c, mp, ns, kap, D, S);
        const double udc = 1.0 - (D + kap * S) / C;
        factor = sqrt(factor);
        kap *= (udc > targetleak && S < smax) ? 1.0 / factor : factor;
    }
    kap_out[(long)s * m + a] = kap;
}

// y = (T + kappa I)^-1 c per output pixel, in place (Cb: c -> y), with the maps.  kap_pix == null: one kappa per stamp
// (kap_stamp, single-kappa path); else kappa per pixel (multi: kappa stored as float32 and THEN multiplied by C, the
// quirk of lakernel.py:216-222).  Lb [np][mp]: the l_i of every pixel's factorisation between the two passes.
__global__ __launch_bounds__(64) void tri_solve_kernel(const double *__restrict__ dvec, const double *__restrict__ evec,
                                                       double *__restrict__ Cb, double *__restrict__ Lb, int np, int mp, int m,
                                                       const int *__restrict__ n, const double *__restrict__ Cs,
                                                       const double *__restrict__ kap_stamp, const double *__restrict__ kap_pix,
                                                       float *__restrict__ UC, float *__restrict__ Sigma, float *__restrict__ kappa)
{
    const int s = blockIdx.y, a = blockIdx.x * 64 + threadIdx.x;
    const int ns = n[s];
    if (a >= m) return;
    const long pa = (long)s * m + a;
    if (ns == 0) { UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f; return; }  // lakernel.py:110-119
    const double *d = dvec + (long)s * np, *e = evec + (long)s * np;
    double *c = Cb + (long)s * np * mp + a, *lb = Lb + (long)s * np * mp + a;
    const double C = Cs[s], kap = kap_pix ? kap_pix[pa] : kap_stamp[s];
    double delta = d[0] + kap, z = c[0], D = 0.0;
    for (int i = 0; i < ns; i++) {
        const double r = 1.0 / delta, t = z * r;
        D += z * t;
        c[(long)i * mp] = t;  // w_i = z_i / delta_i
        if (i + 1 < ns) {
            const double b = e[i], l = b * r;
            lb[(long)i * mp] = l;
            z = c[(long)(i + 1) * mp] - l * z;
            delta = d[i + 1] + kap - l * b;
        }
    }
    double y = c[(long)(ns - 1) * mp], S = y * y;
    for (int i = ns - 2; i >= 0; i--) {
        y = c[(long)i * mp] - lb[(long)i * mp] * y;
        c[(long)i * mp] = y;
        S += y * y;
    }
    const double udc = 1.0 - (D + kap * S) / C;
    Sigma[pa] = (float)S;
    UC[pa] = (float)udc;
    kappa[pa] = kap_pix ? (float)((double)(float)kap * C) : (float)kap;
}

// ---- the same in the BAND basis of band.hip (bandwidth BW = BAND_BW: B[i+t][i] = band[t][i], t = 0..BW) ------------------------------
// LDL^T of B + kappa I, left-looking over the last BW columns kept in registers:
//   d_i = B_ii + kappa - sum_q l_{i,j}^2 d_j,   l_{i+t,i} = (B_{i+t,i} - sum_q l_{i+t,j} l_{i,j} d_j) / d_i   (j = i-1-q, q < BW),
//   z_i = c_i - sum_q l_{i,j} z_j,   D = sum z_i^2 / d_i,   S = -dD/dkappa with every quantity's kappa-derivative carried along.
// The last BW columns live in a ring of PHYSICAL slots: column j in slot j mod BW -- dh, lh[t] = l_{j+t,j} (t = 1..BW), zh, the
// products xh[t] = l_{j+t,j} d_j that both sums need, and the kappa-derivatives of all of them.  The rows are processed BW at a
// time with the row's phase i mod BW a template parameter, so every slot index is a compile-time constant: the ring never moves
// (round 3 shifted it by one slot per row: 84 register moves next to ~120 arithmetic instructions; and formed l d three times).
// Slots of columns before the first hold d = 1 and zeros, so the sums need no guards.
template <int B_>
struct BandState {
    double dh[B_], dph[B_], zh[B_], zph[B_], lh[B_][B_ + 1], lph[B_][B_ + 1], xh[B_][B_ + 1], xph[B_][B_ + 1];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int q = 0; q < B_; q++) {
            dh[q] = 1.0; dph[q] = 0.0; zh[q] = 0.0; zph[q] = 0.0;
#pragma unroll
            for (int t = 0; t <= B_; t++) { lh[q][t] = 0.0; lph[q][t] = 0.0; xh[q][t] = 0.0; xph[q][t] = 0.0; }
        }
    }
};

// Row i (i mod B_ == PHI) of the sweep with derivative; bd[t] = B[i+t][i] (zero beyond the matrix), ci = c_i; adds the row's
// contributions to D and S; the row's own column (d_i, z_i, l_{i+t,i}) is left in slot PHI.
template <int B_, bool FAST, int PHI>
__device__ __forceinline__ void band_row(BandState<B_> &h, const double (&bd)[B_ + 1], double ci, double kap, double &D, double &S)
{
    double d = bd[0] + kap, dp = 1.0, z = ci, zp = 0.0;
#pragma unroll
    for (int q = 0; q < B_; q++) {
        const int p = (PHI - 1 - q + 2 * B_) % B_;  // slot of column j = i - 1 - q
        const double l = h.lh[p][q + 1], lp = h.lph[p][q + 1], x = h.xh[p][q + 1], xp = h.xph[p][q + 1];
        d -= l * x;                  // l^2 d_j
        dp -= lp * x + l * xp;       // d/dkappa (l^2 d_j) = 2 l l' d_j + l^2 d_j'
        z -= l * h.zh[p];
        zp -= lp * h.zh[p] + l * h.zph[p];
    }
    const double r = FAST ? fast_recip(d) : 1.0 / d, t = z * r;
    D += z * t;
    S += t * (t * dp - 2.0 * zp);
    double ln[B_ + 1], lpn[B_ + 1];
    ln[0] = lpn[0] = 0.0;
#pragma unroll
    for (int tt = 1; tt <= B_; tt++) {
        double num = bd[tt], nump = 0.0;
#pragma unroll
        for (int q = 0; q < B_; q++)
            if (q + 1 + tt <= B_) {
                const int p = (PHI - 1 - q + 2 * B_) % B_;
                const double a = h.lh[p][q + 1 + tt], ap = h.lph[p][q + 1 + tt], x = h.xh[p][q + 1], xp = h.xph[p][q + 1];
                num -= a * x;              // l_{i+tt,j} l_{i,j} d_j
                nump -= ap * x + a * xp;
            }
        ln[tt] = num * r;
        lpn[tt] = (nump - ln[tt] * dp) * r;
    }
    h.dh[PHI] = d; h.dph[PHI] = dp; h.zh[PHI] = z; h.zph[PHI] = zp;
#pragma unroll
    for (int tt = 0; tt <= B_; tt++) {
        h.lh[PHI][tt] = ln[tt];
        h.lph[PHI][tt] = lpn[tt];
        h.xh[PHI][tt] = ln[tt] * d;
        h.xph[PHI][tt] = lpn[tt] * d + ln[tt] * dp;
    }
}

// rows i0 .. i0 + B_ - 1 (those below ns): f(phase as an integral constant, row) for every one of them, phases in order
template <int B_, typename F>
__device__ __forceinline__ void band_rows(int i0, int ns, F &&f)
{
    static_assert(B_ >= 1 && B_ <= 4, "band_rows unrolls up to four phases");
    if (i0 < ns) f(std::integral_constant<int, 0>{}, i0);
    if constexpr (B_ > 1) { if (i0 + 1 < ns) f(std::integral_constant<int, 1>{}, i0 + 1); }
    if constexpr (B_ > 2) { if (i0 + 2 < ns) f(std::integral_constant<int, 2>{}, i0 + 2); }
    if constexpr (B_ > 3) { if (i0 + 3 < ns) f(std::integral_constant<int, 3>{}, i0 + 3); }
}

template <int B_>
__device__ __forceinline__ void band_sweep(const double *__restrict__ band, int np, const double *__restrict__ c, long cstride, int ns, double kap,
                                           double &D, double &S)
{
    BandState<B_> h;
    h.clear();
    D = 0.0;
    S = 0.0;
    double cv[B_];
#pragma unroll
    for (int q = 0; q < B_; q++) cv[q] = q < ns ? c[(long)q * cstride] : 0.0;
    for (int i0 = 0; i0 < ns; i0 += B_) {
        double cn[B_];  // the next group's rows of c: in flight while this group is swept
#pragma unroll
        for (int q = 0; q < B_; q++) cn[q] = i0 + B_ + q < ns ? c[(long)(i0 + B_ + q) * cstride] : 0.0;
        band_rows<B_>(i0, ns, [&](auto phi, int i) {
            constexpr int PHI = decltype(phi)::value;
            double bd[B_ + 1];
#pragma unroll
            for (int t = 0; t <= B_; t++) bd[t] = band[(long)t * np + i];  // wave-uniform
            band_row<B_, true, PHI>(h, bd, cv[PHI], kap, D, S);
        });
#pragma unroll
        for (int q = 0; q < B_; q++) cv[q] = cn[q];
    }
}

template <int B_>
__global__ __launch_bounds__(64) void band_search_kernel(const double *__restrict__ band, const double *__restrict__ Cb, int np, int mp, int m,
                                                         const int *__restrict__ n, const double *__restrict__ Cs, const double *__restrict__ kmin,
                                                         const double *__restrict__ kmax, double targetleak, double smax, int nbis,
                                                         double *__restrict__ kap_out)
{
    const int s = blockIdx.y, a = blockIdx.x * 64 + threadIdx.x;
    const int ns = n[s];
    if (a >= m || ns == 0) return;
    const double *bs = band + (long)s * (B_ + 1) * np, *c = Cb + (long)s * np * mp + a;
    const double C = Cs[s], kCmin = kmin[s], kCmax = kmax[s];
    double factor = sqrt(kCmax / kCmin), kap = sqrt(kCmax * kCmin);
    for (int it = 0; it < nbis; it++) {
        double D, S;
        band_sweep<B_>(bs, np, c, mp, ns, kap, D, S);
        const double udc = 1.0 - (D + kap * S) / C;
        factor = sqrt(factor);
        kap *= (udc > targetleak && S < smax) ? 1.0 / factor : factor;
    }
    kap_out[(long)s * m + a] = kap;
}

// y = (B + kappa I)^-1 c per output pixel, in place, with the maps (tri_solve_kernel's contract).  Lb [B_][np][mp]: l_{i+t,i} of
// every pixel's factorisation between the forward and the backward pass.
// A launch takes the output pixels a0 .. a0 + mc - 1 (a chunk of columns; Lb [B_][np][mc] then holds the factors of that chunk only:
// the launches of a call reuse it one after the other -- four chunks: a quarter of the 4 [np][mp] arrays that used to be the largest
// item of the Eigen path's workspace).
template <int B_>
__global__ __launch_bounds__(64) void band_solve_kernel(const double *__restrict__ band, double *__restrict__ Cb, double *__restrict__ Lb, int np,
                                                        int mp, int m, const int *__restrict__ n, const double *__restrict__ Cs,
                                                        const double *__restrict__ kap_stamp, const double *__restrict__ kap_pix,
                                                        float *__restrict__ UC, float *__restrict__ Sigma, float *__restrict__ kappa, int a0, int mc)
{
    const int s = blockIdx.y, al = blockIdx.x * 64 + threadIdx.x, a = a0 + al;
    const int ns = n[s];
    if (al >= mc || a >= m) return;
    const long pa = (long)s * m + a;
    if (ns == 0) { UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f; return; }  // lakernel.py:110-119
    const double *bs = band + (long)s * (B_ + 1) * np;
    double *c = Cb + (long)s * np * mp + a, *lb = Lb + (long)s * B_ * np * mc + al;
    const double C = Cs[s], kap = kap_pix ? kap_pix[pa] : kap_stamp[s];
    BandState<B_> h;
    h.clear();
    double D = 0.0, Sd = 0.0;
    for (int i0 = 0; i0 < ns; i0 += B_) {
        band_rows<B_>(i0, ns, [&](auto phi, int i) {
            constexpr int PHI = decltype(phi)::value;
            double bd[B_ + 1];
#pragma unroll
            for (int t = 0; t <= B_; t++) bd[t] = bs[(long)t * np + i];
            band_row<B_, false, PHI>(h, bd, c[(long)i * mp], kap, D, Sd);
            c[(long)i * mp] = h.zh[PHI] / h.dh[PHI];  // w_i = z_i / d_i
#pragma unroll
            for (int t = 1; t <= B_; t++) lb[((long)(t - 1) * np + i) * mc] = h.lh[PHI][t];
        });
    }
    double yh[B_];  // y_{i+1} .. y_{i+B_}
#pragma unroll
    for (int q = 0; q < B_; q++) yh[q] = 0.0;
    double S = 0.0;
    for (int i = ns - 1; i >= 0; i--) {
        double y = c[(long)i * mp];
#pragma unroll
        for (int t = 1; t <= B_; t++) y -= lb[((long)(t - 1) * np + i) * mc] * yh[t - 1];  // l_{i+t,i} is zero beyond the matrix
        c[(long)i * mp] = y;
        S += y * y;
#pragma unroll
        for (int q = B_ - 1; q > 0; q--) yh[q] = yh[q - 1];
        yh[0] = y;
    }
    const double udc = 1.0 - (D + kap * S) / C;
    Sigma[pa] = (float)S;
    UC[pa] = (float)udc;
    kappa[pa] = kap_pix ? (float)((double)(float)kap * C) : (float)kap;
}

// Is M + kappa_min I positive definite?  One thread per stamp sweeps the LDL^T of the reduced matrix at the lowest kappa of the
// call: flag[s] = 1 at the first pivot that is not positive (NaN included), minpiv[s] = the smallest pivot met.
template <int B_>
__global__ __launch_bounds__(64) void band_pd_kernel(const double *__restrict__ band, int np, int batch, const int *__restrict__ n,
                                                     const double *__restrict__ kap_stamp, int *__restrict__ flag, double *__restrict__ minpiv)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= batch) return;
    const double *bs = band + (long)s * (B_ + 1) * np;
    const double kap = kap_stamp[s];
    BandState<B_> h;
    h.clear();
    double D = 0.0, S = 0.0, mn = 1.0 / 0.0;
    int bad = 0;
    const int ns = n[s];
    for (int i0 = 0; i0 < ns && !bad; i0 += B_) {
        band_rows<B_>(i0, ns, [&](auto phi, int i) {
            constexpr int PHI = decltype(phi)::value;
            if (bad) return;
            double bd[B_ + 1];
#pragma unroll
            for (int t = 0; t <= B_; t++) bd[t] = bs[(long)t * np + i];
            band_row<B_, false, PHI>(h, bd, 0.0, kap, D, S);
            if (!(h.dh[PHI] > 0.0)) { bad = 1; mn = h.dh[PHI]; }
            else mn = fmin(mn, h.dh[PHI]);
        });
    }
    flag[s] = bad;
    minpiv[s] = mn;
}

__global__ __launch_bounds__(64) void tri_pd_kernel(const double *__restrict__ dvec, const double *__restrict__ evec, int np, int batch,
                                                    const int *__restrict__ n, const double *__restrict__ kap_stamp, int *__restrict__ flag,
                                                    double *__restrict__ minpiv)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= batch) return;
    const double *d = dvec + (long)s * np, *e = evec + (long)s * np;
    const double kap = kap_stamp[s];
    const int ns = n[s];
    double mn = 1.0 / 0.0;
    int bad = 0;
    double delta = ns > 0 ? d[0] + kap : 1.0;
    for (int i = 0; i < ns; i++) {
        mn = fmin(mn, delta);
        if (!(delta > 0.0)) { bad = 1; mn = delta; break; }
        if (i + 1 < ns) delta = d[i + 1] + kap - e[i] * e[i] / delta;
    }
    flag[s] = bad;
    minpiv[s] = mn;
}

// ---- the eigenbasis route (EigenKernel as the reference writes it), for the stamps the check above turns away ---------------
// single kappa (lakernel.py:154-172): one wave per output pixel; S[a][i] = P[a][i] / (lam_i + kappa) (zero for i >= n), the maps
// go to the stamp's own slot (map[f] = its index in the caller's batch)
__global__ __launch_bounds__(256) void eigen_single_kernel(const double *__restrict__ lam, const double *__restrict__ P, int mp, int np, int m,
                                                           const int *__restrict__ map, const int *__restrict__ n,
                                                           const double *__restrict__ kap, const double *__restrict__ Cs,
                                                           double *__restrict__ S, float *__restrict__ UC, float *__restrict__ Sigma,
                                                           float *__restrict__ kappa)
{
    const int f = blockIdx.y, a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a >= mp) return;
    const int s = map[f], ns = n[s];
    const double k = kap[s], C = Cs[s];
    const double *l = lam + (long)f * np, *p = P + ((long)f * mp + a) * np;
    double *o = S + ((long)f * mp + a) * np;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < np; i += 64) {
        double v = 0.0;
        if (i < ns && a < m) {
            const double li = l[i];
            v = p[i] / (li + k);
            s2 += v * v;
            s1 += (li + 2.0 * k) * v * v;
        }
        o[i] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    if (lane == 0 && a < m) {
        const long pa = (long)s * m + a;
        kappa[pa] = (float)k;
        Sigma[pa] = (float)s2;
        UC[pa] = (float)(1.0 - s1 / C);
    }
}

// multi kappa: float32 stores as the reference makes them (lakernel.py:216-222: kappa stored float32, THEN multiplied by C)
__global__ void eigen_multi_store_kernel(const double *__restrict__ k64, const double *__restrict__ S64, const double *__restrict__ U64, int m,
                                         double C, float *__restrict__ UC, float *__restrict__ Sigma, float *__restrict__ kappa)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    kappa[a] = (float)((double)(float)k64[a] * C);
    Sigma[a] = (float)S64[a];
    UC[a] = (float)U64[a];
}

// X [np][mp] float64 -> resident layout Tt [ldn][ldm] float32 (rows >= n[s], columns >= m zero)
__global__ void tri_store_resident_kernel(const double *__restrict__ X, int np, int mp, int m, const int *__restrict__ n,
                                          float *__restrict__ Tt, int ldn, int ldm)
{
    const int s = blockIdx.z, i = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= ldm) return;
    Tt[((long)s * ldn + i) * ldm + a] = (i < n[s] && a < m) ? (float)X[((long)s * np + i) * mp + a] : 0.0f;
}

// X [np][mp] float64 -> reference layout T [m][ldt] float32 (columns >= n[s] zero)
__global__ __launch_bounds__(256) void tri_store_ref_kernel(const double *__restrict__ X, int np, int mp, int m,
                                                            const int *__restrict__ n, float *__restrict__ T, long ldt)
{
    __shared__ double tile[32][33];
    const int s = blockIdx.z, i0 = blockIdx.y * 32, a0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ns = n[s];
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, a = a0 + tx;
        tile[r][tx] = (i < ns && i < np && a < mp) ? X[((long)s * np + i) * mp + a] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, i = i0 + tx;
        if (a < m && i < ldt) T[(long)s * m * ldt + (long)a * ldt + i] = (float)tile[tx][r];
    }
}

}  // namespace imcom

using namespace imcom;

static int ctx_ok(imcom_ctx *ctx)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    return IMCOM_OK;
}

extern "C" int imcom_eigh(imcom_ctx *ctx, int batch, const int *n, int ldn, const double *A, double *lam, double *Q,
                          int memspace)
{
    IMCOM_TRY(ctx_ok(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && A && lam && Q && ldn >= 1, "bad arguments");
    int nmax = 0;
    for (int s = 0; s < batch; s++) { IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn", s, n[s]); nmax = std::max(nmax, n[s]); }
    const bool host = memspace == IMCOM_MEM_HOST;
    const int ld = (int)align_up((size_t)std::max(nmax, 1), NB);
    const size_t szA = (size_t)batch * ldn * ldn, szL = (size_t)batch * ldn;
    size_t total = eigh_ws_bytes(batch, ld, true) + 8192;
    if (host) total += (2 * szA + szL) * 8 + 1024;
    IMCOM_TRY(ws_reserve(ctx, total));
    const double *A_d = A;
    double *lam_d = lam, *Q_d = Q;
    if (host) {
        double *t = (double *)ws_take(ctx, szA * 8);
        Q_d = (double *)ws_take(ctx, szA * 8);
        lam_d = (double *)ws_take(ctx, szL * 8);
        if (!t || !Q_d || !lam_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(t, A, szA * 8, hipMemcpyHostToDevice, ctx->stream));
        A_d = t;
    }
    IMCOM_HIP_CHECK(hipMemsetAsync(Q_d, 0, szA * 8, ctx->stream));
    IMCOM_HIP_CHECK(hipMemsetAsync(lam_d, 0, szL * 8, ctx->stream));
    IMCOM_TRY(eigh_device(ctx, batch, n, ld, A_d, ldn, (long)ldn * ldn, lam_d, ldn, Q_d, ldn, (long)ldn * ldn, nullptr));
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(lam, lam_d, szL * 8, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(Q, Q_d, szA * 8, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}

// The band basis (band.hip) whenever its N x 4 panel fits the LDS (N <= 4.8k); IMCOM_EIGEN_BASIS=tridiagonal keeps round 3's first form
// (A/B runs and a cross-check: the two bases share nothing but the block-reflector GEMMs).
static bool eigen_uses_band(int np)
{
    static const bool tri = getenv("IMCOM_EIGEN_BASIS") && !strcmp(getenv("IMCOM_EIGEN_BASIS"), "tridiagonal");
    return !tri && band_basis_fits(np);
}

// Householder reduction to band form alone (tests, diagnostics): band [batch][BAND_BW + 1][ldn] with band[t][i] = B[i + t][i], the
// reflectors V [batch][ldn][ldn] (row r = v_r) and tau [batch][ldn]; A = Q B Q^T with Q = H_0 H_1 ...; ldn a multiple of 128.
extern "C" int imcom_band_reduce(imcom_ctx *ctx, int batch, const int *n, int ldn, const double *A, double *band, double *V, double *tau, int memspace)
{
    IMCOM_TRY(ctx_ok(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && A && band && V && tau && ldn >= NB && ldn % NB == 0 && band_basis_fits(ldn), "bad arguments (ldn a multiple of 128, <= 4.8k)");
    for (int s = 0; s < batch; s++) IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn", s, n[s]);
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t szA = (size_t)batch * ldn * ldn * 8, szB = (size_t)batch * (BAND_BW + 1) * ldn * 8, szT = (size_t)batch * ldn * 8;
    IMCOM_TRY(ws_reserve(ctx, band_basis_ws_bytes(batch, ldn, NB) + (host ? szA : 0) + 65536));
    const double *A_d = A;
    if (host) {
        double *t = (double *)ws_take(ctx, szA);
        if (!t) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(t, A, szA, hipMemcpyHostToDevice, ctx->stream));
        A_d = t;
    }
    TrdBasis tb;
    IMCOM_TRY(band_basis_device(ctx, batch, n, ldn, NB, A_d, ldn, (long)ldn * ldn, &tb));
    const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    IMCOM_HIP_CHECK(hipMemcpyAsync(band, tb.band, szB, kind, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(V, tb.Vall, szA, kind, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(tau, tb.tauvec, szT, kind, ctx->stream));
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

static size_t eigen_fallback_bytes(int nf, int np, int mp, int m)
{
    size_t t = 0;
    auto add = [&](size_t b) { t = align_up(t, 256) + b; };
    add((size_t)nf * np * np * 8);       // A of the stamps, gathered
    add((size_t)nf * np * 8);            // lam
    add((size_t)nf * np * np * 8);       // Q
    for (int q = 0; q < 3; q++) add((size_t)nf * np * mp * 8);  // b^T / x^T [np][mp], P [mp][np], S [mp][np]
    add((size_t)nf * m * 8 * 3);         // kappa, Sigma, UC of lakernel1 (float64)
    add((size_t)nf * 4);
    return t + eigh_ws_bytes(nf, np, true) + 16384;
}

// The stamps `idx` of the batch through the eigendecomposition (see the head of the file).  Their x = T^T [np][mp] (float64) goes
// to Xout + idx * np * mp (the layout every stamp's result has before the final store), their maps to the stamps' own slots.  The
// workspace is taken from ctx->ws_used on and handed back; as many stamps at a time as it holds.
static int eigen_fallback(imcom_ctx *ctx, const std::vector<int> &idx, const int *n, const int *n_dev, int ldn, int m, int np, int mp,
                          const double *A_d, const double *Bt_res, const double *B_ref, const double *C, const double *kappaC, int nv,
                          double ucmin, double smax, int nbis, const double *par, int batch, double *Xout, float *UC_d, float *Sig_d,
                          float *kap_d, size_t limit)
{
    hipStream_t st = ctx->stream;
    limit = std::min(limit, ctx->ws_bytes);  // the end of the caller's share of the workspace
    const size_t mark = ctx->ws_used, room = limit - std::min(limit, align_up(mark, 256));
    int cap = (int)idx.size();
    if (const char *e = getenv("IMCOM_EIGEN_FALLBACK_CAP")) cap = std::max(1, std::min(cap, atoi(e)));  // (tests: several rounds of the loop below)
    while (cap > 1 && eigen_fallback_bytes(cap, np, mp, m) > room) cap = (cap + 1) / 2;
    if (eigen_fallback_bytes(cap, np, mp, m) > room) { set_error("internal: no workspace for the eigendecomposition of a stamp"); return IMCOM_ERR_NOMEM; }
    const size_t mat = (size_t)np * np, big = (size_t)np * mp;
    for (size_t f0 = 0; f0 < idx.size(); f0 += cap) {
        const int nf = (int)std::min((size_t)cap, idx.size() - f0);
        ctx->ws_used = mark;
        double *Ag = (double *)ws_take(ctx, nf * mat * 8), *lam = (double *)ws_take(ctx, (size_t)nf * np * 8), *Q = (double *)ws_take(ctx, nf * mat * 8);
        double *X = (double *)ws_take(ctx, nf * big * 8), *P = (double *)ws_take(ctx, nf * big * 8), *S = (double *)ws_take(ctx, nf * big * 8);
        double *pix = (double *)ws_take(ctx, (size_t)nf * m * 8 * 3);
        int *map = (int *)ws_take(ctx, (size_t)nf * 4);
        if (!Ag || !lam || !Q || !X || !P || !S || !pix || !map) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        std::vector<int> nsub(nf);
        IMCOM_HIP_CHECK(hipMemsetAsync(Ag, 0, nf * mat * 8, st));
        IMCOM_HIP_CHECK(hipMemsetAsync(Q, 0, nf * mat * 8, st));
        IMCOM_HIP_CHECK(hipMemsetAsync(lam, 0, (size_t)nf * np * 8, st));
        for (int f = 0; f < nf; f++) {
            const int s = idx[f0 + f];
            nsub[f] = n[s];
            IMCOM_HIP_CHECK(hipMemcpy2DAsync(Ag + f * mat, (size_t)np * 8, A_d + (size_t)s * ldn * ldn, (size_t)ldn * 8, (size_t)n[s] * 8, (size_t)n[s],
                                             hipMemcpyDeviceToDevice, st));
            if (Bt_res) IMCOM_HIP_CHECK(hipMemcpyAsync(X + f * big, Bt_res + (size_t)s * big, big * 8, hipMemcpyDeviceToDevice, st));
            else {
                hipLaunchKernelGGL(tri_pack_kernel, dim3(mp / 32, np / 32, 1), dim3(256), 0, st, B_ref + (size_t)s * m * ldn, (long)ldn, m, n_dev + s,
                                   X + f * big, np, mp);
                IMCOM_TRY(check_launch("tri_pack_kernel"));
            }
        }
        IMCOM_TRY(upload(ctx, map, &idx[f0], (size_t)nf));
        IMCOM_TRY(eigh_device(ctx, nf, nsub.data(), np, Ag, np, (long)mat, lam, np, Q, np, (long)mat, nullptr));
        {   // P [mp][np] = b Q  (b^T = X is input-pixel-major: the k-major A operand)
            ProfScope ps(ctx, "eigen_gemm");
            IMCOM_TRY(launch_gemm(ctx, true, true, mp, np, np, nf, X, mp, (long)big, Q, np, (long)mat, P, np, (long)big, 1.0, 0.0));
        }
        {
            ProfScope ps(ctx, "lakernel1");
            if (nv == 1) {
                hipLaunchKernelGGL(eigen_single_kernel, dim3(mp / 4, nf), dim3(256), 0, st, lam, P, mp, np, m, map, n_dev, par + batch, par, S, UC_d, Sig_d, kap_d);
                IMCOM_TRY(check_launch("eigen_single_kernel"));
            } else {
                IMCOM_HIP_CHECK(hipMemsetAsync(S, 0, nf * big * 8, st));
                for (int f = 0; f < nf; f++) {
                    const int s = idx[f0 + f];
                    double *k64 = pix + (size_t)f * m * 3, *S64 = k64 + m, *U64 = S64 + m;
                    IMCOM_TRY(launch_lakernel1(ctx, lam + (size_t)f * np, P + f * big, m, n[s], np, C[s], ucmin, kappaC[0] * C[s], kappaC[nv - 1] * C[s], nbis,
                                               k64, S64, U64, S + f * big, np, smax));
                    hipLaunchKernelGGL(eigen_multi_store_kernel, dim3((m + 255) / 256), dim3(256), 0, st, k64, S64, U64, m, C[s], UC_d + (size_t)s * m,
                                       Sig_d + (size_t)s * m, kap_d + (size_t)s * m);
                }
                IMCOM_TRY(check_launch("eigen_multi_store_kernel"));
            }
        }
        {   // x [np][mp] = Q S^T
            ProfScope ps(ctx, "eigen_gemm");
            IMCOM_TRY(launch_gemm(ctx, false, false, np, mp, np, nf, Q, np, (long)mat, S, np, (long)big, X, mp, (long)big, 1.0, 0.0));
        }
        for (int f = 0; f < nf; f++)
            IMCOM_HIP_CHECK(hipMemcpyAsync(Xout + (size_t)idx[f0 + f] * big, X + f * big, big * 8, hipMemcpyDeviceToDevice, st));
    }
    ctx->ws_used = mark;
    return IMCOM_OK;
}

// Shared body of the two entries.  Bt_res: -B/2 in the resident layout [np][mp] (then Tt_res [np][mp] float32 is the output);
// else B_ref [m][ldn] and T_ref [m][ldn] in the reference's layout.
//
// The work of a (sub-)batch comes in two parts: eigen_enqueue queues everything up to x = Q y on ctx->stream without waiting for
// the device; eigen_finish reads the positive-definiteness flags back, sends the stamps that failed through the eigenbasis route
// and stores T.  Between the two, other sub-batches can be queued on other streams (solve_eigen_core).
struct EigenJob {
    int batch = 0, nmax = 0;
    const int *n = nullptr;
    const double *A_d = nullptr, *Bt_res = nullptr, *B_ref = nullptr, *C = nullptr;
    float *Tt_res = nullptr, *T_ref = nullptr, *UC_d = nullptr, *Sig_d = nullptr, *kap_d = nullptr;
    int *info = nullptr;
    // state between the two parts
    double *Cb = nullptr, *par = nullptr;
    int *n_early = nullptr, *flag = nullptr;
    size_t keep = 0, limit = 0;
    hipStream_t st = nullptr;
    int *flag_h = nullptr;  // page-locked, in ctx->flag_pin
};

// columns per launch of band_solve_kernel: a quarter of the padded output pixels, in whole wavefronts -- for the batches whose
// workspace matters; a small batch takes all columns at once (four launches of a few hundred one-wave workgroups each are latency,
// 6 -> 13 ms per 32 cfg-3 stamps, and 64 stamps' factors are 14 GB)
static int band_solve_chunk(int mp, int batch)
{
    const int parts = batch <= 64 ? 1 : BAND_BW;
    return std::max(64, ((mp + parts - 1) / parts + 63) / 64 * 64);
}

static int eigen_enqueue(imcom_ctx *ctx, EigenJob &j, int ldn, int m, int np, int mp, const double *kappaC, int nv, double ucmin, double smax, int nbis,
                         bool allow_overlap)
{
    const int batch = j.batch, nmax = j.nmax;
    const int *n = j.n;
    const size_t big = (size_t)batch * np * mp * 8, szM = (size_t)batch * m;
    TrdBasis tb;
    const bool banded = eigen_uses_band(np);
    // buffers that outlive everything first; from `keep` on the workspace belongs to the basis (whose own scratch is handed back
    // after the reduction) and, once the band path has finished, to the eigenbasis route of the stamps that failed the check
    double *Cb = j.Cb = (double *)ws_take(ctx, big);
    double *kpix = (double *)ws_take(ctx, szM * 8), *par = j.par = (double *)ws_take(ctx, (size_t)batch * 8 * 5);
    int *n_early = j.n_early = (int *)ws_take(ctx, (size_t)batch * 4), *flag = j.flag = (int *)ws_take(ctx, (size_t)batch * 4);
    j.keep = ctx->ws_used;
    // the factors between the two sweeps of the final solve: [np][mp] in the tridiagonal basis; in the band basis BAND_BW of them per
    // CHUNK of mp / BAND_BW columns (the same bytes), taken once the reduction has handed its scratch back -- over it, not beside it
    double *Lb = banded ? nullptr : (double *)ws_take(ctx, big);
    if (!Cb || (!banded && !Lb) || !kpix || !par || !n_early || !flag) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    std::vector<double> ph(4 * (size_t)batch);
    for (int s = 0; s < batch; s++) {
        ph[s] = j.C[s];
        ph[batch + s] = kappaC[0] * j.C[s];           // single kappa / kCmin C (lakernel.py:166, 213)
        ph[2 * batch + s] = kappaC[nv - 1] * j.C[s];  // kCmax C (214)
    }
    IMCOM_TRY(upload(ctx, par, ph.data(), ph.size()));
    IMCOM_TRY(upload(ctx, n_early, n, (size_t)batch));
    hipStream_t st = j.st = ctx->stream;
    if (j.Bt_res) IMCOM_HIP_CHECK(hipMemcpyAsync(Cb, j.Bt_res, big, hipMemcpyDeviceToDevice, st));
    else {
        hipLaunchKernelGGL(tri_pack_kernel, dim3(mp / 32, np / 32, batch), dim3(256), 0, st, j.B_ref, (long)ldn, m, n_early, Cb, np, mp);
        IMCOM_TRY(check_launch("tri_pack_kernel"));
    }
    if (banded) {
        // c = Qh^T b panel by panel ON THE SECOND STREAM while the reduction goes on: a panel of 128 reflectors is final long
        // before the matrix is reduced, its GEMMs are matrix-pipe work, the reduction's passes are memory work
        while (ctx->sync_events.size() < 2) {
            hipEvent_t e;
            IMCOM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->sync_events.push_back(e);
        }
        hipEvent_t ev_main = ctx->sync_events[0], ev_aux = ctx->sync_events[1];
        // (the second queue only where the overlap below will use it: see `overlap`)
        const char *ov_ = getenv("IMCOM_EIGEN_OVERLAP");
        if (allow_overlap && nmax > 0 && (ov_ ? atoi(ov_) != 0 : batch <= 128)) IMCOM_TRY(ensure_aux(ctx));
        hipStream_t aux = ctx->aux_stream;
        struct AuxDrain {  // whatever path leaves this scope, nothing may still run on the second stream (the workspace is reused)
            hipStream_t s;
            bool armed = true;
            ~AuxDrain() { if (armed && s) hipStreamSynchronize(s); }
        } drain{aux};
        // Worth it while the reduction's one-workgroup-per-stamp step leaves most CUs idle: the products' tiles monopolise a CU
        // (registers, LDS) and the reduction's dependent chain queues behind them.  cfg-3, with / without: batch 32 251 / 265 ms,
        // 64 418 / 430, 128 762 / 772, 256 1485 / 1458.  IMCOM_EIGEN_OVERLAP=0 / 1 forces it.  (Not beside other sub-batches: the
        // second stream is one, and the sub-batches' own phases already fill each other's gaps.)
        const char *ov = getenv("IMCOM_EIGEN_OVERLAP");
        const bool overlap = allow_overlap && nmax > 0 && (ov ? atoi(ov) != 0 : batch <= 128);
        auto on_panel = [&](int p) -> int {
            IMCOM_HIP_CHECK(hipEventRecord(ev_main, st));
            IMCOM_HIP_CHECK(hipStreamWaitEvent(aux, ev_main, 0));
            ctx->stream = aux;
            const int rc = trd_panel_step(ctx, tb, batch, p, Cb, mp);
            ctx->stream = st;
            return rc;
        };
        const int rc = band_basis_device(ctx, batch, n, np, mp, j.A_d, ldn, (long)ldn * ldn, &tb, overlap ? std::function<int(int)>(on_panel) : nullptr);
        ctx->stream = st;
        IMCOM_TRY(rc);
        if (overlap) {
            IMCOM_HIP_CHECK(hipEventRecord(ev_aux, aux));
            IMCOM_HIP_CHECK(hipStreamWaitEvent(st, ev_aux, 0));  // join: c is complete when the main stream goes on
        }
        drain.armed = false;
        if (overlap) IMCOM_TRY(trd_pair_factors(ctx, &tb, batch));  // (the panels' own factors were made on the second stream)
        else if (nmax > 0) IMCOM_TRY(trd_apply_q(ctx, tb, batch, Cb, mp, true));  // c = Q^T b after the reduction
    } else {
        IMCOM_TRY(trd_basis_device(ctx, batch, n, np, mp, j.A_d, ldn, (long)ldn * ldn, &tb));
        if (nmax > 0) IMCOM_TRY(trd_apply_q(ctx, tb, batch, Cb, mp, true));  // c = Qh^T b
    }
    // is A + kappa_min I positive definite?  (par + batch = the lowest kappa of the call; the answer is read in eigen_finish, after
    // everything else has been queued: the host's wait hides behind the search)
    if (nmax > 0) {
        if (banded) hipLaunchKernelGGL(band_pd_kernel<BAND_BW>, dim3((batch + 63) / 64), dim3(64), 0, st, tb.band, np, batch, n_early, par + batch, flag, par + 4 * batch);
        else hipLaunchKernelGGL(tri_pd_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, tb.dvec, tb.evec, np, batch, n_early, par + batch, flag, par + 4 * batch);
        IMCOM_TRY(check_launch("pd_check"));
    }
    {
        ProfScope ps(ctx, "lakernel1");
        if (banded) {
            if (nv > 1) {
                hipLaunchKernelGGL(band_search_kernel<BAND_BW>, dim3((m + 63) / 64, batch), dim3(64), 0, st, tb.band, Cb, np, mp, m, n_early, par, par + batch,
                                   par + 2 * batch, ucmin, smax, nbis, kpix);
                IMCOM_TRY(check_launch("band_search_kernel"));
            }
            const int mc = band_solve_chunk(mp, batch);
            const size_t mark_l = ctx->ws_used;
            Lb = (double *)ws_take(ctx, (size_t)batch * BAND_BW * np * mc * 8);
            if (!Lb) { set_error("internal: workspace (band factors)"); return IMCOM_ERR_NOMEM; }
            for (int a0 = 0; a0 < m; a0 += mc)
                hipLaunchKernelGGL(band_solve_kernel<BAND_BW>, dim3((mc + 63) / 64, batch), dim3(64), 0, st, tb.band, Cb, Lb, np, mp, m, n_early, par, par + batch,
                                   nv > 1 ? kpix : nullptr, j.UC_d, j.Sig_d, j.kap_d, a0, mc);
            IMCOM_TRY(check_launch("band_solve_kernel"));
            ctx->ws_used = mark_l;  // (same stream: whatever takes this memory next runs behind the launches)
        } else {
            if (nv > 1) {
                hipLaunchKernelGGL(tri_search_kernel, dim3((m + 63) / 64, batch), dim3(64), 0, st, tb.dvec, tb.evec, Cb, np, mp, m, n_early, par,
                                   par + batch, par + 2 * batch, ucmin, smax, nbis, kpix);
                IMCOM_TRY(check_launch("tri_search_kernel"));
            }
            hipLaunchKernelGGL(tri_solve_kernel, dim3((m + 63) / 64, batch), dim3(64), 0, st, tb.dvec, tb.evec, Cb, Lb, np, mp, m, n_early, par,
                               par + batch, nv > 1 ? kpix : nullptr, j.UC_d, j.Sig_d, j.kap_d);
            IMCOM_TRY(check_launch("tri_solve_kernel"));
        }
    }
    if (nmax > 0) {
        IMCOM_TRY(trd_apply_q(ctx, tb, batch, Cb, mp, false));  // x = Qh y
        IMCOM_HIP_CHECK(hipMemcpyAsync(j.flag_h, flag, (size_t)batch * 4, hipMemcpyDeviceToHost, st));
    }
    return IMCOM_OK;
}

static int eigen_finish(imcom_ctx *ctx, EigenJob &j, int ldn, int m, int np, int mp, const double *kappaC, int nv, double ucmin, double smax, int nbis)
{
    const int batch = j.batch;
    hipStream_t st = j.st;
    std::vector<int> flagged;
    if (j.nmax > 0) {
        IMCOM_HIP_CHECK(hipStreamSynchronize(st));
        for (int s = 0; s < batch; s++)
            if (j.flag_h[s]) { flagged.push_back(s); j.info[s] = 1; }
    }
    if (!flagged.empty()) {
        // the band path's buffers are dead now (same stream: everything queued so far runs before whatever reuses them)
        ctx->ws_used = j.keep;
        IMCOM_TRY(eigen_fallback(ctx, flagged, j.n, j.n_early, ldn, m, np, mp, j.A_d, j.Bt_res, j.B_ref, j.C, kappaC, nv, ucmin, smax, nbis, j.par, batch, j.Cb,
                                 j.UC_d, j.Sig_d, j.kap_d, j.limit));
    }
    if (j.Tt_res) {
        hipLaunchKernelGGL(tri_store_resident_kernel, dim3((mp + 255) / 256, np, batch), dim3(256), 0, st, j.Cb, np, mp, m, j.n_early, j.Tt_res, np, mp);
        IMCOM_TRY(check_launch("tri_store_resident_kernel"));
    } else if (ldn > 0) {
        hipLaunchKernelGGL(tri_store_ref_kernel, dim3(mp / 32, (unsigned)((ldn + 31) / 32), batch), dim3(256), 0, st, j.Cb, np, mp, m, j.n_early, j.T_ref, (long)ldn);
        IMCOM_TRY(check_launch("tri_store_ref_kernel"));
    }
    return IMCOM_OK;
}

static size_t solve_eigen_ws(int batch, int np, int mp, int m);

// Sub-batches of one call on streams of their own (IMCOM_EIGEN_SPLIT = count).  The reduction alternates a bandwidth-bound pass
// over the trailing matrix (symv4) with a latency chain that keeps ONE workgroup per stamp busy (band_step: a seventh of the
// reduction at batch 32) and small launches around them; sub-batches in flight put one's latency chain and launch gaps beside
// another's memory pass.  Measured on the final kernels (cfg-3, ms per stamp, one / two sub-batches; profiles/r04_negative_results.txt
// item 5): batch 16: 9.79 / 9.80, 32: 7.40-7.50 / 7.14-7.15, 64: 6.29 / 5.97-6.01, 128: 5.82 / 5.63, 256: 5.48 / 5.32; three or four
// are slower (7.30 at 32, 5.50 at 256).  Two halves beat one batch WITH its reflector products on the low-priority queue beside the
// reduction (which the halves do not use).  Default: two sub-batches from 24 stamps on, else one.
// IMCOM_SPLIT_CUS=1 confines each sub-batch's stream to a share of the CUs of its own (hipExtStreamCreateWithCUMask): four quarters
// of 64 CUs reach 5.20-5.23 at batch 256 (-2 %), equal at 128, slower at 32 / 64 -- and any share that is not a whole number of
// CUs per XCD (three, five, six parts) more than doubles the time.  Not the default.
static int eigen_split(int batch, int np)
{
    const char *env = getenv("IMCOM_EIGEN_SPLIT");  // (read at every call: bench.py takes the per-launch timings of symv4 on one stream)
    const int forced = env ? atoi(env) : 0;
    int k = forced > 0 ? forced : (batch >= 24 ? 2 : 1);
    if (!eigen_uses_band(np)) k = 1;
    return std::max(1, std::min(std::min(k, batch), 8));
}

static int solve_eigen_core(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, int np, int mp, const double *A_d, const double *Bt_res,
                            const double *B_ref, const double *C, const double *kappaC, int nv, double ucmin, double smax, int nbis,
                            float *Tt_res, float *T_ref, float *UC_d, float *Sig_d, float *kap_d, int nmax, int *info)
{
    const int nsub = nmax > 0 ? eigen_split(batch, np) : 1;
    hipStream_t main = ctx->stream;
    if (ctx->flag_pin_count < (size_t)batch) {
        if (ctx->flag_pin) { IMCOM_HIP_CHECK(hipStreamSynchronize(main)); IMCOM_HIP_CHECK(hipHostFree(ctx->flag_pin)); ctx->flag_pin = nullptr; ctx->flag_pin_count = 0; }
        IMCOM_HIP_CHECK(hipHostMalloc((void **)&ctx->flag_pin, (size_t)std::max(batch, 256) * 4, hipHostMallocDefault));
        ctx->flag_pin_count = (size_t)std::max(batch, 256);
    }
    const size_t base = ctx->ws_used;
    std::vector<EigenJob> jobs(nsub);
    for (int q = 0; q < nsub; q++) {
        EigenJob &j = jobs[q];
        const int s0 = (int)((long)batch * q / nsub), s1 = (int)((long)batch * (q + 1) / nsub);
        j.batch = s1 - s0;
        j.n = n + s0;
        j.nmax = 0;
        for (int s = s0; s < s1; s++) j.nmax = std::max(j.nmax, n[s]);
        j.A_d = A_d ? A_d + (size_t)s0 * ldn * ldn : nullptr;
        j.Bt_res = Bt_res ? Bt_res + (size_t)s0 * np * mp : nullptr;
        j.B_ref = B_ref ? B_ref + (size_t)s0 * m * ldn : nullptr;
        j.C = C + s0;
        j.Tt_res = Tt_res ? Tt_res + (size_t)s0 * np * mp : nullptr;
        j.T_ref = T_ref ? T_ref + (size_t)s0 * m * ldn : nullptr;
        j.UC_d = UC_d + (size_t)s0 * m;
        j.Sig_d = Sig_d + (size_t)s0 * m;
        j.kap_d = kap_d + (size_t)s0 * m;
        j.info = info + s0;
        j.flag_h = ctx->flag_pin + s0;
    }
    if (nsub == 1) {
        jobs[0].limit = ctx->ws_bytes;
        IMCOM_TRY(eigen_enqueue(ctx, jobs[0], ldn, m, np, mp, kappaC, nv, ucmin, smax, nbis, true));
        return eigen_finish(ctx, jobs[0], ldn, m, np, mp, kappaC, nv, ucmin, smax, nbis);
    }
    // streams and events of the sub-batches (the first one stays on the caller's stream)
    while ((int)ctx->sub_streams.size() < nsub - 1) {
        hipStream_t s_;
        IMCOM_HIP_CHECK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
        ctx->sub_streams.push_back(s_);
    }
    while ((int)ctx->sync_events.size() < 3 + nsub) {
        hipEvent_t e;
        IMCOM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_events.push_back(e);
    }
    // IMCOM_SPLIT_CUS=1 (A/B runs): every sub-batch on a stream confined to a share of the CUs of its own, so that one's
    // one-workgroup-per-stamp step finds free CUs while another's memory pass runs
    const char *pc = getenv("IMCOM_SPLIT_CUS");
    const bool parted = pc && atoi(pc) != 0;
    if (parted && (int)ctx->part_streams.size() != nsub) {
        for (auto s_ : ctx->part_streams) { hipStreamSynchronize(s_); hipStreamDestroy(s_); }
        ctx->part_streams.clear();
        for (int q = 0; q < nsub; q++) {
            std::vector<uint32_t> mask((ctx->cu_count + 31) / 32, 0u);
            for (int i = ctx->cu_count * q / nsub; i < ctx->cu_count * (q + 1) / nsub; i++) mask[i / 32] |= 1u << (i % 32);
            hipStream_t s_;
            IMCOM_HIP_CHECK(hipExtStreamCreateWithCUMask(&s_, (uint32_t)mask.size(), mask.data()));
            ctx->part_streams.push_back(s_);
        }
    }
    auto stream_of = [&](int q) { return parted ? ctx->part_streams[q] : (q == 0 ? main : ctx->sub_streams[q - 1]); };
    // the small host arrays of all sub-batches go through the pinned ring: it must not wrap while copies queued on another stream
    // have not run yet
    IMCOM_HIP_CHECK(hipStreamSynchronize(main));
    ctx->pin_used = 0;
    hipEvent_t fork = ctx->sync_events[2];
    IMCOM_HIP_CHECK(hipEventRecord(fork, main));
    struct Restore {  // whatever path leaves this scope: the caller's stream is back and nothing runs on the others
        imcom_ctx *c;
        hipStream_t m;
        int k;
        ~Restore()
        {
            c->stream = m;
            c->ws_limit = 0;
            for (int q = 0; q + 1 < k; q++) hipStreamSynchronize(c->sub_streams[q]);
            for (auto s_ : c->part_streams) hipStreamSynchronize(s_);
        }
    } restore{ctx, main, nsub};
    size_t at = base;
    int rc = IMCOM_OK;
    for (int q = 0; q < nsub && rc == IMCOM_OK; q++) {
        EigenJob &j = jobs[q];
        ctx->stream = stream_of(q);
        if (ctx->stream != main) IMCOM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, fork, 0));
        ctx->ws_used = at;  // every sub-batch in a share of its own: nothing one hands back is reused by another while it runs
        at = align_up(at, 256) + solve_eigen_ws(j.batch, np, mp, m);
        j.limit = at;
        ctx->ws_limit = j.limit;  // a sub-batch that needs more than its share fails (IMCOM_ERR_NOMEM) instead of writing into the next one's
        rc = eigen_enqueue(ctx, j, ldn, m, np, mp, kappaC, nv, ucmin, smax, nbis, false);
    }
    for (int q = 0; q < nsub && rc == IMCOM_OK; q++) {
        ctx->stream = stream_of(q);
        ctx->ws_limit = jobs[q].limit;
        rc = eigen_finish(ctx, jobs[q], ldn, m, np, mp, kappaC, nv, ucmin, smax, nbis);
        if (rc == IMCOM_OK && ctx->stream != main) {  // join: the caller's stream goes on when the sub-batch's T has been stored
            IMCOM_HIP_CHECK(hipEventRecord(ctx->sync_events[3 + q], ctx->stream));
            IMCOM_HIP_CHECK(hipStreamWaitEvent(main, ctx->sync_events[3 + q], 0));
        }
    }
    ctx->stream = main;
    ctx->ws_used = at;
    return rc;
}

static size_t solve_eigen_ws(int batch, int np, int mp, int m)
{
    const bool banded = eigen_uses_band(np);
    const size_t big = (size_t)batch * np * mp * 8;
    // band basis: what the reduction keeps + the larger of its scratch and the chunk of band factors that later lies over it
    const size_t keep_b = banded ? band_basis_keep_bytes(batch, np, mp) : 0;
    const size_t basis = banded ? keep_b + std::max(band_basis_ws_bytes(batch, np, mp) - keep_b, (size_t)batch * BAND_BW * np * band_solve_chunk(mp, batch) * 8 + 512)
                                : trd_basis_ws_bytes(batch, np, mp) + big;
    // what stays (c -> y -> x, kappa per pixel, parameters) + the larger of the basis route and ONE stamp's eigendecomposition
    return big + (size_t)batch * m * 8 + (size_t)batch * 64 + std::max(basis, eigen_fallback_bytes(1, np, mp, m)) + 65536;
}

// all sub-batches of a call (each in a share of its own)
static size_t solve_eigen_ws_total(int batch, int np, int mp, int m)
{
    const int nsub = eigen_split(batch, np);
    size_t t = 0;
    for (int q = 0; q < nsub; q++) t = align_up(t, 256) + solve_eigen_ws((int)((long)batch * (q + 1) / nsub) - (int)((long)batch * q / nsub), np, mp, m);
    return t + 4096;
}

// Device workspace imcom_solve_eigen_resident takes for a batch (what a planner adds to its own buffers; no device call is made)
extern "C" int imcom_solve_eigen_workspace(int batch, int ldn, int ldm, int m, size_t *bytes)
{
    IMCOM_REQUIRE(batch >= 1 && ldn >= NB && ldn % NB == 0 && ldm >= NB && ldm % NB == 0 && m >= 1 && m <= ldm && bytes, "bad sizes (ldn, ldm multiples of 128)");
    *bytes = solve_eigen_ws_total(batch, ldn, ldm, m);
    return IMCOM_OK;
}

extern "C" int imcom_solve_eigen(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A, const double *mBhalf,
                                 const double *C, const double *kappaC, int nv, double ucmin, double smax, int nbis, float *T,
                                 float *UC, float *Sigma, float *kappa, int *info, int memspace)
{
    IMCOM_TRY(ctx_ok(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && C && kappaC && UC && Sigma && kappa && info, "null pointer / empty batch");
    IMCOM_REQUIRE(m >= 1 && nv >= 1 && ldn >= 0 && nbis >= 0, "bad sizes");
    int nmax = 0;
    for (int s = 0; s < batch; s++) {
        IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn=%d", s, n[s], ldn);
        nmax = std::max(nmax, n[s]);
        info[s] = 0;
    }
    IMCOM_REQUIRE(nmax == 0 || (A && mBhalf && T), "null matrix pointer");
    const bool host = memspace == IMCOM_MEM_HOST;
    const int np = (int)align_up((size_t)std::max(nmax, 1), NB), mp = (int)align_up((size_t)m, NB);
    const size_t szA = (size_t)batch * ldn * ldn, szB = (size_t)batch * m * ldn, szM = (size_t)batch * m;
    size_t total = solve_eigen_ws_total(batch, np, mp, m);
    if (host) total += szA * 8 + szB * 8 + szB * 4 + szM * 12 + 4096;
    IMCOM_TRY(ws_reserve(ctx, total));
    const double *A_d = A, *B_d = mBhalf;
    float *T_d = T, *UC_d = UC, *Sig_d = Sigma, *kap_d = kappa;
    if (host) {
        double *ta = (double *)ws_take(ctx, szA * 8), *tb = (double *)ws_take(ctx, szB * 8);
        T_d = (float *)ws_take(ctx, szB * 4);
        UC_d = (float *)ws_take(ctx, szM * 12);
        if (!ta || !tb || !T_d || !UC_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        Sig_d = UC_d + szM;
        kap_d = Sig_d + szM;
        if (szA) IMCOM_HIP_CHECK(hipMemcpyAsync(ta, A, szA * 8, hipMemcpyHostToDevice, ctx->stream));
        if (szB) IMCOM_HIP_CHECK(hipMemcpyAsync(tb, mBhalf, szB * 8, hipMemcpyHostToDevice, ctx->stream));
        A_d = ta;
        B_d = tb;
    }
    IMCOM_TRY(solve_eigen_core(ctx, batch, n, ldn, m, np, mp, A_d, nullptr, B_d, C, kappaC, nv, ucmin, smax, nbis, nullptr, T_d, UC_d, Sig_d,
                               kap_d, nmax, info));
    if (host) {
        if (szB) IMCOM_HIP_CHECK(hipMemcpyAsync(T, T_d, szB * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(UC, UC_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(Sigma, Sig_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(kappa, kap_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}

extern "C" int imcom_solve_eigen_resident(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, int ldm, const double *A,
                                          const double *Bt, const double *C, const double *kappaC, int nv, double ucmin, double smax,
                                          int nbis, float *Tt, float *UC, float *Sigma, float *kappa, int *info)
{
    IMCOM_TRY(ctx_ok(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && C && kappaC && A && Bt && Tt && UC && Sigma && kappa && info, "null pointer / empty batch");
    IMCOM_REQUIRE(m >= 1 && nv >= 1 && nbis >= 0 && ldn >= NB && ldn % NB == 0 && ldm % NB == 0 && ldm >= m, "bad sizes (ldn, ldm multiples of 128)");
    int nmax = 0;
    for (int s = 0; s < batch; s++) {
        IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn=%d", s, n[s], ldn);
        nmax = std::max(nmax, n[s]);
        info[s] = 0;
    }
    IMCOM_TRY(ws_reserve(ctx, solve_eigen_ws_total(batch, ldn, ldm, m)));
    return solve_eigen_core(ctx, batch, n, ldn, m, ldn, ldm, A, Bt, nullptr, C, kappaC, nv, ucmin, smax, nbis, Tt, nullptr, UC, Sigma, kappa, nmax, info);
}
