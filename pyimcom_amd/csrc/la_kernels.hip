// la_kernels.hip -- the non-GEMM kernels of the LA path: diagonal-block factorisation + inverse,
// layout packing, the single-/multi-kappa reductions (U/C, Sigma, kappa), lakernel1, the reduced-space
// kappa search, and the coaddition epilogue.
//
// Reference lines restated (src/pyimcom/): lakernel.py:281-394 (CholKernel), routine.py:341-588
// (lakernel1, lsolve_sps, build_reduced_T_wrap), coadd.py:1222-1292 (trapezoid), 1320-1354
// (_perform_coaddition).
#include "common.h"
#include "launchers.h"

namespace imcom {

// ------------------------------------------------------------------------------------------------
// diagonal of A + increments, applied the way the reference does: a sequence of in-place adds
// (lakernel.py:298, 356, 268/277), so the rounding matches; used by chol_update via `shift`.
__global__ void diag_shift_kernel(const double *__restrict__ A, int ldn, const double *__restrict__ inc,
                                  const int *__restrict__ ninc, double *__restrict__ dshift, int batch)
{
    // dshift[s][i] = ((A_ii + inc0) + inc1 ...) - A_ii is NOT exact in general; we store the full
    // shifted diagonal instead and chol_update replaces A_ii by it.
    const int s = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ldn || s >= batch) return;
    double a = A[(long)s * ldn * ldn + (long)i * ldn + i];
    const int n = ninc[s];
    for (int t = 0; t < n; t++) a += inc[s * MAX_INC + t];
    dshift[(long)s * ldn + i] = a;
}

// ------------------------------------------------------------------------------------------------
// layout packing for the host LA seam
__global__ void pack_A_kernel(const double *__restrict__ A, long lda, const int *__restrict__ n,
                              double *__restrict__ Ap, int ldp)
{
    const int s = blockIdx.z;
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldp) return;
    const int ns = n[s];
    double v = (i == j) ? 1.0 : 0.0;
    if (i < ns && j < ns) v = A[(long)s * lda * lda + (long)i * lda + j];
    Ap[(long)s * ldp * ldp + (long)i * ldp + j] = v;
}

// mBhalf[s][a][i] ([m][ldb]) -> Bt[s][i][a] ([ldp][ldm]), zero padded
__global__ __launch_bounds__(256) void pack_Bt_kernel(const double *__restrict__ B, long ldb, int m,
                                                      const int *__restrict__ n, double *__restrict__ Bt,
                                                      int ldp, int ldm)
{
    __shared__ double tile[32][33];
    const int s = blockIdx.z;
    const int i0 = blockIdx.y * 32, a0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int ns = n[s];
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, i = i0 + tx;
        tile[r][tx] = (a < m && i < ns) ? B[(long)s * m * ldb + (long)a * ldb + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, a = a0 + tx;
        if (i < ldp && a < ldm) Bt[(long)s * ldp * ldm + (long)i * ldm + a] = tile[tx][r];
    }
}

// Tt[s][i][a] float32 ([ldp][ldm]) -> T[s][a][i] float32 ([m][ldt]); columns i >= n[s] are zeroed
__global__ __launch_bounds__(256) void unpack_T_kernel(const float *__restrict__ Tt, int ldp, int ldm,
                                                       const int *__restrict__ n, int m,
                                                       float *__restrict__ T, long ldt)
{
    __shared__ float tile[32][33];
    const int s = blockIdx.z;
    const int i0 = blockIdx.y * 32, a0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ns = n[s];
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, a = a0 + tx;
        tile[r][tx] = (i < ns && i < ldp && a < ldm) ? Tt[(long)s * ldp * ldm + (long)i * ldm + a] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, i = i0 + tx;
        if (a < m && i < ldt) T[(long)s * m * ldt + (long)a * ldt + i] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// single kappa (lakernel.py:309-317): D_a = sum_i B_ai T_ai, N_a = sum_i T_ai^2 in float64 from the
// float64 solution X; Tt = float32(X); kappa_a = kappa; Sigma_a = N_a; UC_a = 1 - (kappa N_a + D_a)/C.
// Workgroup = 64 output pixels x 4 row groups.
__global__ __launch_bounds__(256) void finalize_single_kernel(const double *__restrict__ X,
                                                              const double *__restrict__ Bt, int ldn, int ldm,
                                                              int m, const int *__restrict__ n,
                                                              const double *__restrict__ kap,
                                                              const double *__restrict__ Cs,
                                                              float *__restrict__ Tt, float *__restrict__ UC,
                                                              float *__restrict__ Sigma,
                                                              float *__restrict__ kappa, const int *__restrict__ act)
{
    __shared__ double red[2][4][64];
    const int s = blockIdx.y, a = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    if (act && act[s] == 0) return;  // (uniform over the workgroup) not a stamp of this attempt
    const int ns = n[s];
    const long base = (long)s * ldn * ldm;
    double D = 0.0, N = 0.0;
    if (a < ldm) {
        for (int i = rg; i < ns; i += 4) {
            const double x = X[base + (long)i * ldm + a];
            const double b = Bt[base + (long)i * ldm + a];
            Tt[base + (long)i * ldm + a] = (float)x;
            D += b * x;
            N += x * x;
        }
        // rows beyond n[s] hold no pixels: keep Tt defined for the epilogue
        for (int i = ns + rg; i < ldn; i += 4) Tt[base + (long)i * ldm + a] = 0.0f;
    }
    red[0][rg][threadIdx.x & 63] = D;
    red[1][rg][threadIdx.x & 63] = N;
    __syncthreads();
    if (rg == 0 && a < m) {
        const int c = threadIdx.x & 63;
        D = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        N = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        const double k = kap[s], C = Cs[s];
        if (ns == 0) {  // lakernel.py:110-119
            UC[(long)s * m + a] = 1.0f; Sigma[(long)s * m + a] = 0.0f; kappa[(long)s * m + a] = 1.0f;
        } else {
            kappa[(long)s * m + a] = (float)k;
            Sigma[(long)s * m + a] = (float)N;
            UC[(long)s * m + a] = (float)(1.0 - (k * N + D) / C);
        }
    }
}

// The same maps when the triangular solves have already left T (float32) and the per-block-row column sums behind
// (gemm_f64.hip, tile_col_sumsq): D_a = sum over the forward partials, N_a = sum over the backward ones, in a fixed order.
// Rows of Tt beyond the stamp's last 128-block (shorter stamps of a ragged batch) are zeroed here.
__global__ __launch_bounds__(256) void finalize_fused_kernel(const double *__restrict__ Dpart, const double *__restrict__ Npart,
                                                             int ldn, int ldm, int m, const int *__restrict__ n,
                                                             const int *__restrict__ nblk, const double *__restrict__ kap,
                                                             const double *__restrict__ Cs, float *__restrict__ Tt,
                                                             float *__restrict__ UC, float *__restrict__ Sigma,
                                                             float *__restrict__ kappa, const int *__restrict__ act)
{
    const int s = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    if (a >= ldm || (act && act[s] == 0)) return;  // act: the stamps this attempt has solved (the others keep what they have)
    const int ns = n[s], nb = nblk[s], np = 2 * (ldn / NB);
    for (long i = (long)nb * NB; i < ldn; i++) Tt[((long)s * ldn + i) * ldm + a] = 0.0f;
    if (a >= m) return;
    double D = 0.0, N = 0.0;
    for (int p = 0; p < 2 * nb; p++) {
        D += Dpart[((long)s * np + p) * ldm + a];
        N += Npart[((long)s * np + p) * ldm + a];
    }
    const long o = (long)s * m + a;
    if (ns == 0) { UC[o] = 1.0f; Sigma[o] = 0.0f; kappa[o] = 1.0f; return; }  // lakernel.py:110-119
    const double k = kap[s], C = Cs[s];
    kappa[o] = (float)k;
    Sigma[o] = (float)N;
    UC[o] = (float)(1.0 - (k * N + D) / C);
}

// ------------------------------------------------------------------------------------------------
// routine.py:433-484 on tiny nv x nv systems held in thread-private arrays
constexpr int MAXNV = 8;

__device__ inline void lsolve_sps_dev(int N, double *A, double *x, const double *b)
{
    double p1[MAXNV];
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < i; j++) {
            double s = 0.0;
            for (int k = 0; k < j; k++) s += A[i * N + k] * A[j * N + k];
            A[i * N + j] = (A[i * N + j] - s) / A[j * N + j];
        }
        double s = 0.0;
        for (int k = 0; k < i; k++) s += A[i * N + k] * A[i * N + k];
        A[i * N + i] = sqrt(A[i * N + i] - s);
    }
    for (int i = 0; i < N; i++) {
        double s = 0.0;
        for (int j = 0; j < i; j++) s += A[i * N + j] * p1[j];
        p1[i] = (b[i] - s) / A[i * N + i];
    }
    for (int i = N - 1; i >= 0; i--) {
        double s = 0.0;
        for (int j = i + 1; j < N; j++) s += A[j * N + i] * x[j];
        x[i] = (p1[i] - s) / A[i * N + i];
    }
}

// routine.py:546-588 for one output pixel
__device__ inline void reduced_T_pixel(const double *Na, const double *Da, const double *Ea,
                                       const double *kappa, int nv, double ucmin, double smax,
                                       double *okappa, double *oS, double *oUC, double *w)
{
    double M[MAXNV * MAXNV];
    int iv = nv - 1;
    double UC = ucmin * 10, S = smax / 10;
    while (iv > 0 && ucmin < UC && smax > S) {
        iv--;
        S = Na[iv * (nv + 1)];
        UC = 1.0 - 2.0 * Da[iv] + Ea[iv * (nv + 1)];
    }
    double kmid = sqrt(kappa[iv] * kappa[iv + 1]);
    double factor = pow(kappa[iv + 1] / kappa[iv], 0.25);
    for (int it = 0; it < 12; it++) {
        for (int r = 0; r < nv; r++)
            for (int c = 0; c <= r; c++) M[r * nv + c] = Ea[r + nv * c] + kmid * Na[r + nv * c];
        lsolve_sps_dev(nv, M, w, Da);
        S = 0.0;
        for (int r = 0; r < nv; r++) {
            double s = 0.0;
            for (int c = 0; c < nv; c++) s += Na[r + nv * c] * w[c];
            S += s * w[r];
        }
        UC = 1.0 - kmid * S;
        for (int r = 0; r < nv; r++) UC -= Da[r] * w[r];
        kmid *= (ucmin < UC && smax > S) ? 1.0 / factor : factor;
        factor = sqrt(factor);
    }
    *okappa = kmid;
    *oS = S;
    *oUC = UC;
}

__global__ void build_reduced_T_kernel(const double *__restrict__ Nflat, const double *__restrict__ Dflat,
                                       const double *__restrict__ Eflat, const double *__restrict__ kappa,
                                       int nv, long m, double ucmin, double smax, double *__restrict__ ok,
                                       double *__restrict__ oS, double *__restrict__ oU,
                                       double *__restrict__ ow)
{
    const long a = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (a >= m) return;
    double Na[MAXNV * MAXNV], Ea[MAXNV * MAXNV], Da[MAXNV], kp[MAXNV], w[MAXNV];
    const int nv2 = nv * nv;
    for (int t = 0; t < nv2; t++) { Na[t] = Nflat[a * nv2 + t]; Ea[t] = Eflat[a * nv2 + t]; }
    for (int t = 0; t < nv; t++) { Da[t] = Dflat[a * nv + t]; kp[t] = kappa[t]; w[t] = 0.0; }
    double k, S, U;
    reduced_T_pixel(Na, Da, Ea, kp, nv, ucmin, smax, &k, &S, &U, w);
    ok[a] = k; oS[a] = S; oU[a] = U;
    for (int t = 0; t < nv; t++) ow[a * nv + t] = w[t];
}

// multi-kappa Cholesky (lakernel.py:361-393).  Xs = nv solutions, node p at Xs + p*node_stride.
// Dp[a][p] = sum_i B_ai X_p,ai ; Npq[a][p][q] = sum_i X_p,ai X_q,ai   (float64)
__global__ __launch_bounds__(256) void multi_reduce_kernel(const double *__restrict__ Xs, long node_stride,
                                                           const double *__restrict__ Bt, int ldn, int ldm,
                                                           int m, const int *__restrict__ n, int nv,
                                                           double *__restrict__ Dp, double *__restrict__ Npq)
{
    __shared__ double red[4][64];
    const int s = blockIdx.y, c = threadIdx.x & 63, a = blockIdx.x * 64 + c, rg = threadIdx.x >> 6;
    const int ns = n[s];
    const long base = (long)s * ldn * ldm;
    double d[MAXNV], nn[MAXNV * (MAXNV + 1) / 2];
    for (int p = 0; p < MAXNV; p++) d[p] = 0.0;
    for (int t = 0; t < MAXNV * (MAXNV + 1) / 2; t++) nn[t] = 0.0;
    if (a < ldm)
        for (int i = rg; i < ns; i += 4) {
            const long off = base + (long)i * ldm + a;
            const double b = Bt[off];
            double xv[MAXNV];
            for (int p = 0; p < nv; p++) xv[p] = Xs[p * node_stride + off];
            int t = 0;
            for (int p = 0; p < nv; p++) {
                d[p] += b * xv[p];
                for (int q = 0; q <= p; q++) nn[t++] += xv[p] * xv[q];
            }
        }
    // reduce the 4 row groups value by value through LDS
    const int nd = nv, ntri = nv * (nv + 1) / 2;
    for (int t = 0; t < nd + ntri; t++) {
        red[rg][c] = (t < nd) ? d[t] : nn[t - nd];
        __syncthreads();
        if (rg == 0 && a < m) {
            const double v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
            if (t < nd) Dp[((long)s * m + a) * nv + t] = v;
            else {
                // triangular index -> (p,q)
                int tt = t - nd, p = 0;
                while ((p + 1) * (p + 2) / 2 <= tt) p++;
                const int q = tt - p * (p + 1) / 2;
                Npq[(((long)s * m + a) * nv + p) * nv + q] = v;
                Npq[(((long)s * m + a) * nv + q) * nv + p] = v;
            }
        }
        __syncthreads();
    }
}

// per pixel: E from D,N (lakernel.py:364-368), reduced-space search, outputs (390-392) and weights
__global__ void multi_search_kernel(const double *__restrict__ Dp, const double *__restrict__ Npq, int m,
                                    const int *__restrict__ n, int nv, const double *__restrict__ kappaC,
                                    const double *__restrict__ Cs, double ucmin, double smax,
                                    float *__restrict__ UC, float *__restrict__ Sigma,
                                    float *__restrict__ kappa, double *__restrict__ W)
{
    const int s = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    const long pa = (long)s * m + a;
    if (n[s] == 0) {
        UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f;
        for (int p = 0; p < nv; p++) W[pa * nv + p] = 0.0;
        return;
    }
    const double C = Cs[s];
    double Na[MAXNV * MAXNV], Ea[MAXNV * MAXNV], Da[MAXNV], kp[MAXNV], w[MAXNV], kar[MAXNV];
    for (int p = 0; p < nv; p++) { kp[p] = kappaC[p]; kar[p] = kappaC[p] * C; w[p] = 0.0; }
    for (int t = 0; t < nv * nv; t++) Na[t] = Npq[pa * nv * nv + t];
    for (int p = 0; p < nv; p++) {
        const double dpp = Dp[pa * nv + p];
        for (int q = 0; q < p; q++) {
            const double e = Dp[pa * nv + q] - kar[p] * Na[p * nv + q];
            Ea[q * nv + p] = e / C;
            Ea[p * nv + q] = e / C;
        }
        Ea[p * nv + p] = (dpp - kar[p] * Na[p * nv + p]) / C;
        Da[p] = dpp / C;
    }
    double k, S, U;
    reduced_T_pixel(Na, Da, Ea, kp, nv, ucmin, smax, &k, &S, &U, w);
    kappa[pa] = (float)(k * C);
    Sigma[pa] = (float)S;
    UC[pa] = (float)U;
    for (int p = 0; p < nv; p++) W[pa * nv + p] = w[p];
}

// Tt[i][a] = float32( sum_p w[a][p] X_p[i][a] )   (lakernel.py:393)
__global__ __launch_bounds__(256) void multi_combine_kernel(const double *__restrict__ Xs, long node_stride,
                                                            int ldn, int ldm, int m,
                                                            const int *__restrict__ n, int nv,
                                                            const double *__restrict__ W,
                                                            float *__restrict__ Tt)
{
    const int s = blockIdx.y, a = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    if (a >= ldm) return;
    const int ns = n[s];
    const long base = (long)s * ldn * ldm;
    double w[MAXNV];
    for (int p = 0; p < nv; p++) w[p] = (a < m) ? W[((long)s * m + a) * nv + p] : 0.0;
    for (int i = rg; i < ldn; i += 4) {
        const long off = base + (long)i * ldm + a;
        double t = 0.0;
        if (i < ns)
            for (int p = 0; p < nv; p++) t += Xs[p * node_stride + off] * w[p];
        Tt[off] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------------
// routine.py:341-430 lakernel1: one wave per output pixel, lanes stride over the n eigen-components,
// DPP/shuffle tree for the two sums of every bisection step.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void lakernel1_kernel(const double *__restrict__ lam,
                                                        const double *__restrict__ mPhalf, long m, long n,
                                                        long ldp, double C, double targetleak, double kCmin,
                                                        double kCmax, int nbis, double *__restrict__ kappa,
                                                        double *__restrict__ Sigma, double *__restrict__ UC,
                                                        double *__restrict__ T, long ldt, double smax)
{
    const long a = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (a >= m) return;
    const double *p = mPhalf + a * ldp;
    double factor = sqrt(kCmax / kCmin);
    double kap = sqrt(kCmax * kCmin);
    for (int it = 0; it <= nbis; it++) {
        double s1 = 0.0, s2 = 0.0;
        const bool last = (it == nbis);
        for (long i = lane; i < n; i += 64) {
            const double l = lam[i];
            const double v = p[i] / (l + kap);
            if (last) T[a * ldt + i] = v;
            s2 += v * v;
            s1 += (l + 2.0 * kap) * v * v;
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const double udc = 1.0 - s1 / C;
        if (last) {
            if (lane == 0) { Sigma[a] = s2; kappa[a] = kap; UC[a] = udc; }
        } else {
            factor = sqrt(factor);
            kap *= (udc > targetleak && s2 < smax) ? 1.0 / factor : factor;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// coadd.py:1222-1292 trapezoid weights: s_k = k/(2f+1) - sin(2 pi k/(2f+1))/(2 pi), k = 1..2f,
// applied to the 2f outermost rows (B then T) and columns (L then R), one in-place multiply each.
__device__ __forceinline__ double taper_1d(int pos, int len, int fade)
{
    const int fk2 = 2 * fade;
    double f = 1.0;
    const double two_pi = 2.0 * 3.14159265358979323846;
    if (pos < fk2) {
        double s = (double)(pos + 1) / (fk2 + 1);
        s -= sin(two_pi * s) / two_pi;
        f *= s;
    }
    if (pos >= len - fk2) {
        double s = (double)(len - pos) / (fk2 + 1);
        s -= sin(two_pi * s) / two_pi;
        f *= s;
    }
    return f;
}

__device__ __forceinline__ float taper_f32(float v, int iy, int ix, int n2f, int fade)
{
    // numpy: float32 array *= float64 factors -> product in float64, rounded to float32, once for the
    // row pass(es) and once for the column pass(es)
    const int fk2 = 2 * fade;
    if (iy < fk2) { double s = (double)(iy + 1) / (fk2 + 1); s -= sin(6.283185307179586 * s) / 6.283185307179586; v = (float)((double)v * s); }
    if (iy >= n2f - fk2) { double s = (double)(n2f - iy) / (fk2 + 1); s -= sin(6.283185307179586 * s) / 6.283185307179586; v = (float)((double)v * s); }
    if (ix < fk2) { double s = (double)(ix + 1) / (fk2 + 1); s -= sin(6.283185307179586 * s) / 6.283185307179586; v = (float)((double)v * s); }
    if (ix >= n2f - fk2) { double s = (double)(n2f - ix) / (fk2 + 1); s -= sin(6.283185307179586 * s) / 6.283185307179586; v = (float)((double)v * s); }
    return v;
}

__global__ void trapezoid_f32_kernel(float *__restrict__ maps, long nmaps, int n2f, int fade)
{
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long m = (long)n2f * n2f;
    if (t >= nmaps * m) return;
    const int a = (int)(t % m);
    maps[t] = taper_f32(maps[t], a / n2f, a % n2f, n2f, fade);
}

// coadd.py:1104-1107: np.maximum(map, 1e-32) after the Iterative kernel (NaN propagates, as in numpy)
__global__ void clamp_min_f32_kernel(float *__restrict__ maps, long count, float lo)
{
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= count) return;
    const float v = maps[t];
    if (v < lo) maps[t] = lo;
}

// coadd.py:1320-1354.  Workgroup = 64 output pixels x 4 row groups over the input pixels.
//   acc layout in LDS: [n_expo + n_inframe][256]
constexpr int EPI_MAXF = 4;  // input frames (layers) accumulated in registers

// the four sequential float32 roundings of the fade taper (row pass lower / upper edge, column pass left / right
// edge; numpy multiplies the float32 array by float64 factors one pass at a time, coadd.py:1222-1292)
__device__ __forceinline__ void taper_factors(int iy, int ix, int n2f, int fade, double (&s)[4])
{
    const int fk2 = 2 * fade;
    auto edge = [&](int k) { double t = (double)k / (fk2 + 1); return t - sin(6.283185307179586 * t) / 6.283185307179586; };
    s[0] = iy < fk2 ? edge(iy + 1) : 1.0;
    s[1] = iy >= n2f - fk2 ? edge(n2f - iy) : 1.0;
    s[2] = ix < fk2 ? edge(ix + 1) : 1.0;
    s[3] = ix >= n2f - fk2 ? edge(n2f - ix) : 1.0;
}

// One thread owns CPT consecutive output pixels (CPT = 4: one 16-byte load per input pixel row; CPT = 1 when many
// exposures would not leave room in LDS), four row groups per block.
// The per-exposure sums ride in registers while the exposure index of the rows stays the same (pixels are ordered
// InStamp by InStamp, exposure-major inside) and are flushed to LDS when it changes.
#ifndef IMCOM_EPI_U
#define IMCOM_EPI_U 4
#endif
constexpr int EPI_U = IMCOM_EPI_U;  // rows in flight per thread

template <int CPT>
__global__ __launch_bounds__(256) void coadd_epilogue_kernel(float *__restrict__ Tt, int ldn, int ldm, int m,
                                                             int n2f, int fade, const int *__restrict__ n,
                                                             const float *__restrict__ indata, int n_inframe, int f0, int nf,
                                                             const int *__restrict__ expo, int n_expo,
                                                             float *__restrict__ outimage,
                                                             double *__restrict__ Tsum_image_part,
                                                             double *__restrict__ Tsum_inpix,
                                                             double *__restrict__ Neff)
{
    // frames f0 .. f0+nf-1 of the n_inframe input layers; the call with f0 == 0 also tapers T in place and writes
    // the weight sums (later calls find T already tapered)
    extern __shared__ double accs[];  // [max(n_expo, nf)][256][CPT]: the per-exposure sums; at the end, when those have been read, the frames' sums
    const bool first = f0 == 0;
    const int s = blockIdx.y, c = threadIdx.x & 63, a0 = (blockIdx.x * 64 + c) * CPT, rg = threadIdx.x >> 6;
    const int ns = n[s];
    for (int t = 0; t < n_expo; t++)
#pragma unroll
        for (int q = 0; q < CPT; q++) accs[(t * 256 + threadIdx.x) * CPT + q] = 0.0;
    const long base = (long)s * ldn * ldm;
    double tf[CPT][4];
#pragma unroll
    for (int q = 0; q < CPT; q++) {
        const int a = min(a0 + q, m - 1);
        taper_factors(a / n2f, a % n2f, n2f, fade > 0 ? fade : 0, tf[q]);
    }
    const bool live = a0 < m;  // ldm is a multiple of 128: the 16-byte access stays inside the row
    double racc[CPT], oacc[EPI_MAXF][CPT];
#pragma unroll
    for (int q = 0; q < CPT; q++) racc[q] = 0.0;
#pragma unroll
    for (int f = 0; f < EPI_MAXF; f++)
#pragma unroll
        for (int q = 0; q < CPT; q++) oacc[f][q] = 0.0;
    int cur = -1;
    const int *ex = expo + (long)s * ldn;
    if (live) {
        // the rows are independent loads: EPI_U of them (and their exposure / pixel values) are fetched before any is
        // consumed, so that a wave has several 1 KB requests in flight; the sums run in the same order as before
        for (int i0 = rg; i0 < ns; i0 += 4 * EPI_U) {
            float tl[EPI_U][CPT], xl[EPI_U][EPI_MAXF];
            int el[EPI_U];
#pragma unroll
            for (int u = 0; u < EPI_U; u++) {
                const int i = min(i0 + 4 * u, ns - 1);
                if constexpr (CPT == 4) {
                    const float4 tv = *(const float4 *)(Tt + base + (long)i * ldm + a0);
                    tl[u][0] = tv.x; tl[u][1] = tv.y; tl[u][2] = tv.z; tl[u][3] = tv.w;
                } else tl[u][0] = Tt[base + (long)i * ldm + a0];
                el[u] = ex[i];
#pragma unroll
                for (int f = 0; f < EPI_MAXF; f++) xl[u][f] = f < nf ? indata[((long)s * n_inframe + f0 + f) * ldn + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < EPI_U; u++) {
                const int i = i0 + 4 * u;
                if (i >= ns) break;
                float t4[CPT];
#pragma unroll
                for (int q = 0; q < CPT; q++) t4[q] = tl[u][q];
                if (fade > 0 && first) {
#pragma unroll
                    for (int q = 0; q < CPT; q++)
#pragma unroll
                        for (int k = 0; k < 4; k++) t4[q] = (float)((double)t4[q] * tf[q][k]);
                    if constexpr (CPT == 4) *(float4 *)(Tt + base + (long)i * ldm + a0) = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    else Tt[base + (long)i * ldm + a0] = t4[0];
                }
                const int e = el[u];
                if (e != cur) {
                    if (cur >= 0)
#pragma unroll
                        for (int q = 0; q < CPT; q++) accs[(cur * 256 + threadIdx.x) * CPT + q] += racc[q];
#pragma unroll
                    for (int q = 0; q < CPT; q++) racc[q] = 0.0;
                    cur = e;
                }
#pragma unroll
                for (int q = 0; q < CPT; q++) racc[q] += (double)t4[q];
#pragma unroll
                for (int f = 0; f < EPI_MAXF; f++) {
                    if (f < nf) {
                        const double x = (double)xl[u][f];
#pragma unroll
                        for (int q = 0; q < CPT; q++) oacc[f][q] += (double)t4[q] * x;
                    }
                }
            }
        }
        if (cur >= 0)
#pragma unroll
            for (int q = 0; q < CPT; q++) accs[(cur * 256 + threadIdx.x) * CPT + q] += racc[q];
    }
    __syncthreads();
    auto tot4 = [&](int t, int q) {
        return accs[(t * 256 + c) * CPT + q] + accs[(t * 256 + 64 + c) * CPT + q] + accs[(t * 256 + 128 + c) * CPT + q] + accs[(t * 256 + 192 + c) * CPT + q];
    };
    if (rg == 0 && live && first) {
        for (int q = 0; q < CPT; q++) {
            const int a = a0 + q;
            if (a >= m) break;
            {
                double tot = 0.0, sabs = 0.0, sq = 0.0;
                for (int e = 0; e < n_expo; e++) {
                    const double v = tot4(e, q);
                    accs[(e * 256 + c) * CPT + q] = v;  // only this thread reads it again
                    tot += v;
                    sabs += fabs(v);
                    Tsum_image_part[((long)s * m + a) * n_expo + e] = v;
                }
                for (int e = 0; e < n_expo; e++) { const double t = accs[(e * 256 + c) * CPT + q] / sabs; sq += t * t; }
                double neff = 1.0 / sq;
                if (fade > 0) { neff *= taper_1d(a / n2f, n2f, fade); neff *= taper_1d(a % n2f, n2f, fade); }
                Tsum_inpix[(long)s * m + a] = tot;
                Neff[(long)s * m + a] = neff;
            }
        }
    }
    // the frames' sums of the four row groups meet in the slots the exposures no longer need (the LDS request is what decides how
    // many workgroups a CU holds -- three instead of two at six exposures -- and this kernel's rate follows them: tools/ab_epi_occ.sh)
    __syncthreads();
    if (live)
#pragma unroll
        for (int f = 0; f < EPI_MAXF; f++)
            if (f < nf)
#pragma unroll
                for (int q = 0; q < CPT; q++) accs[(f * 256 + threadIdx.x) * CPT + q] = oacc[f][q];
    __syncthreads();
    if (rg == 0 && live)
        for (int q = 0; q < CPT; q++) {
            const int a = a0 + q;
            if (a >= m) break;
            for (int f = 0; f < nf; f++) outimage[((long)s * n_inframe + f0 + f) * m + a] = (float)tot4(f, q);
        }
}

// The tail of coadd_epilogue_kernel (coadd.py:1329-1354) from the per-block-row sums the backward launches left (tile_coadd_partials,
// gemm_f64.hip): pieces added in the order block row 0 .. nblk - 1, lower wave row first.  fade == 0 only (nothing to taper).
__global__ __launch_bounds__(256) void coadd_from_partials_kernel(const double *__restrict__ Epart, int ldn, int ldm, int m, const int *__restrict__ n,
                                                                  const int *__restrict__ nblk, int n_inframe, int n_expo,
                                                                  float *__restrict__ outimage, double *__restrict__ Tsum_image_part,
                                                                  double *__restrict__ Tsum_inpix, double *__restrict__ Neff)
{
    const int s = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    if (a >= m) return;
    const int nacc = n_expo + n_inframe, np = 2 * nblk[s];
    const double *base = Epart + (long)s * 2 * (ldn / NB) * nacc * ldm + a;
    auto total = [&](int q) {
        double v = 0.0;
        for (int kk = 0; kk < np; kk++) v += base[((long)kk * nacc + q) * ldm];
        return v;
    };
    double tot = 0.0, sabs = 0.0, sq = 0.0;
    double *ti = Tsum_image_part + ((long)s * m + a) * n_expo;
    for (int e = 0; e < n_expo; e++) {
        const double v = total(e);
        ti[e] = v;
        tot += v;
        sabs += fabs(v);
    }
    for (int e = 0; e < n_expo; e++) { const double t = ti[e] / sabs; sq += t * t; }
    Tsum_inpix[(long)s * m + a] = tot;
    Neff[(long)s * m + a] = 1.0 / sq;
    for (int f = 0; f < n_inframe; f++) outimage[((long)s * n_inframe + f) * m + a] = (float)total(n_expo + f);
}

__global__ __launch_bounds__(256) void tsum_stamp_kernel(const double *__restrict__ Tsum_image, int m, int n_expo,
                                                         int n2, double *__restrict__ Tsum_stamp)
{
    __shared__ double red[256];
    const int s = blockIdx.y, e = blockIdx.x;
    double v = 0.0;
    for (int a = threadIdx.x; a < m; a += 256) v += Tsum_image[((long)s * m + a) * n_expo + e];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) Tsum_stamp[(long)s * n_expo + e] = red[0] / ((double)n2 * n2);
}

// ------------------------------------------------------------------------------------------------
// launchers
int launch_diag_shift(imcom_ctx *ctx, const double *A, int ldn, const double *inc, const int *ninc, double *dshift, int batch)
{
    hipLaunchKernelGGL(diag_shift_kernel, dim3((ldn + 255) / 256, batch), dim3(256), 0, ctx->stream, A, ldn, inc, ninc, dshift, batch);
    return check_launch("diag_shift_kernel");
}

int launch_pack_A(imcom_ctx *ctx, const double *A, long lda, const int *n, double *Ap, int ldp, int batch)
{
    hipLaunchKernelGGL(pack_A_kernel, dim3((ldp + 255) / 256, ldp, batch), dim3(256), 0, ctx->stream, A, lda, n, Ap, ldp);
    return check_launch("pack_A_kernel");
}

int launch_pack_Bt(imcom_ctx *ctx, const double *B, long ldb, int m, const int *n, double *Bt, int ldp, int ldm, int batch)
{
    hipLaunchKernelGGL(pack_Bt_kernel, dim3((ldm + 31) / 32, (ldp + 31) / 32, batch), dim3(256), 0, ctx->stream, B, ldb, m, n, Bt, ldp, ldm);
    return check_launch("pack_Bt_kernel");
}

int launch_unpack_T(imcom_ctx *ctx, const float *Tt, int ldp, int ldm, const int *n, int m, float *T, long ldt, int batch)
{
    if (ldt <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(unpack_T_kernel, dim3((m + 31) / 32, (unsigned)((ldt + 31) / 32), batch), dim3(256), 0, ctx->stream, Tt, ldp, ldm, n, m, T, ldt);
    return check_launch("unpack_T_kernel");
}

int launch_finalize_single(imcom_ctx *ctx, const double *X, const double *Bt, int ldn, int ldm, int m, const int *n,
                           const double *kap, const double *Cs, float *Tt, float *UC, float *Sigma, float *kappa, int batch, const int *act)
{
    hipLaunchKernelGGL(finalize_single_kernel, dim3(ldm / 64, batch), dim3(256), 0, ctx->stream, X, Bt, ldn, ldm, m, n, kap, Cs, Tt, UC, Sigma, kappa, act);
    return check_launch("finalize_single_kernel");
}

int launch_finalize_fused(imcom_ctx *ctx, const double *Dpart, const double *Npart, int ldn, int ldm, int m, const int *n,
                          const int *nblk, const double *kap, const double *Cs, float *Tt, float *UC, float *Sigma, float *kappa, int batch, const int *act)
{
    hipLaunchKernelGGL(finalize_fused_kernel, dim3((ldm + 255) / 256, batch), dim3(256), 0, ctx->stream, Dpart, Npart, ldn, ldm, m, n, nblk, kap, Cs,
                       Tt, UC, Sigma, kappa, act);
    return check_launch("finalize_fused_kernel");
}

// Which stamps of an attempt go on to the solves: act[s] = fac[s] && the factorisation just queued did not fail; nblk_sol[s] = nblk[s]
// for those, 0 for the rest (a launch's tiles of a stamp with nblk = 0 return at once: the stamp keeps what an earlier attempt wrote).
__global__ void solve_mask_kernel(const int *__restrict__ nblk, const int *__restrict__ fac, const int *__restrict__ fail, int *__restrict__ nblk_sol,
                                  int *__restrict__ act, int batch)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    const int a = fac[s] != 0 && fail[s] == 0;
    act[s] = a;
    nblk_sol[s] = a ? nblk[s] : 0;
}

int launch_solve_mask(imcom_ctx *ctx, const int *nblk, const int *fac, const int *fail, int *nblk_sol, int *act, int batch)
{
    hipLaunchKernelGGL(solve_mask_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx->stream, nblk, fac, fail, nblk_sol, act, batch);
    return check_launch("solve_mask_kernel");
}

// ------------------------------------------------------------------------------------------------
// Small kernels of the smallest-eigenvalue iteration (api.hip: lambda_min_subspace)
// X [batch][ldn][P]: pseudo-random start block, zero on the rows beyond n[s] and for stamps that are not wanted
__global__ void lmin_init_kernel(double *__restrict__ X, int ldn, int P, const int *__restrict__ n, const int *__restrict__ want)
{
    const int s = blockIdx.y;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= (long)ldn * P) return;
    const int i = (int)(e / P);
    double v = 0.0;
    if (want[s] != 0 && i < n[s]) {
        // (the same start block for every stamp: what a stamp's iteration does must not depend on its place in the batch)
        unsigned long long x = (unsigned long long)e * 6364136223846793005ULL + 1442695040888963407ULL;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32; x *= 0x94D049BB133111EBULL; x ^= x >> 29;
        v = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    X[(long)s * ldn * P + e] = v;
}

// dmax[s] = max_i |A_ii| over the stamp's own rows (the scale of the first trial shift when kappa = 0)
__global__ __launch_bounds__(256) void diag_max_kernel(const double *__restrict__ A, int ldn, const int *__restrict__ n, double *__restrict__ dmax)
{
    __shared__ double red[256];
    const int s = blockIdx.x;
    double v = 0.0;
    for (int i = threadIdx.x; i < n[s]; i += 256) v = fmax(v, fabs(A[(long)s * ldn * ldn + (long)i * ldn + i]));
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) dmax[s] = red[0];
}

// G[s] += I on the stamps that are not wanted (their Gram matrix is zero: the Cholesky of the block orthogonalisation must not fail on them)
__global__ void gram_guard_kernel(double *__restrict__ G, int P, const int *__restrict__ want)
{
    const int s = blockIdx.x, c = threadIdx.x;
    if (c < P && want[s] == 0) G[(long)s * P * P + (long)c * P + c] = 1.0;
}

// Residuals of the two lowest Ritz pairs of the iteration's Rayleigh-Ritz step: with y_k = X q_k (X orthonormal [ldn][128], q_k the k-th
// eigenvector of H = X^T A X, eigenvectors in the COLUMNS of Qh) and Z = A X, r_k = Z q_k - theta_k y_k.  part[s][g][k] = the share of
// |r_k|^2 of the rows workgroup g took (added up by the host in the order of g: the same sums on every run).
constexpr int LMIN_RG = LMIN_RESID_GROUPS;
__global__ __launch_bounds__(256) void ritz_residual_kernel(const double *__restrict__ X, const double *__restrict__ Z, const double *__restrict__ Qh,
                                                            const double *__restrict__ lam, int ldn, const int *__restrict__ n, const int *__restrict__ want,
                                                            double *__restrict__ part)
{
    constexpr int P = 128;
    __shared__ double red[4][2];
    const int s = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc0 = 0.0, acc1 = 0.0;
    if (want[s] != 0) {
        const double *q = Qh + (long)s * P * P;
        const double q0a = q[(long)lane * P], q0b = q[(long)(lane + 64) * P], q1a = q[(long)lane * P + 1], q1b = q[(long)(lane + 64) * P + 1];
        const double th0 = lam[(long)s * P], th1 = lam[(long)s * P + 1];
        const double *Xs = X + (long)s * ldn * P, *Zs = Z + (long)s * ldn * P;
        for (int i = g * 4 + wave; i < n[s]; i += LMIN_RG * 4) {
            const double xa = Xs[(long)i * P + lane], xb = Xs[(long)i * P + lane + 64], za = Zs[(long)i * P + lane], zb = Zs[(long)i * P + lane + 64];
            double d0 = (za - th0 * xa) * q0a + (zb - th0 * xb) * q0b, d1 = (za - th1 * xa) * q1a + (zb - th1 * xb) * q1b;
            for (int o = 32; o > 0; o >>= 1) {
                d0 += __shfl_xor(d0, o);
                d1 += __shfl_xor(d1, o);
            }
            acc0 += d0 * d0;
            acc1 += d1 * d1;
        }
    }
    if (lane == 0) { red[wave][0] = acc0; red[wave][1] = acc1; }
    __syncthreads();
    if (threadIdx.x < 2) part[((long)s * LMIN_RG + g) * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int launch_ritz_residual(imcom_ctx *ctx, const double *X, const double *Z, const double *Qh, const double *lam, int ldn, int P, const int *n,
                         const int *want, double *part, int batch)
{
    IMCOM_REQUIRE(P == 128, "ritz_residual: blocks of 128 vectors");
    hipLaunchKernelGGL(ritz_residual_kernel, dim3(LMIN_RG, batch), dim3(256), 0, ctx->stream, X, Z, Qh, lam, ldn, n, want, part);
    return check_launch("ritz_residual_kernel");
}

int launch_lmin_init(imcom_ctx *ctx, double *X, int ldn, int P, const int *n, const int *want, int batch)
{
    hipLaunchKernelGGL(lmin_init_kernel, dim3((unsigned)(((long)ldn * P + 255) / 256), batch), dim3(256), 0, ctx->stream, X, ldn, P, n, want);
    return check_launch("lmin_init_kernel");
}

int launch_diag_max(imcom_ctx *ctx, const double *A, int ldn, const int *n, double *dmax, int batch)
{
    hipLaunchKernelGGL(diag_max_kernel, dim3(batch), dim3(256), 0, ctx->stream, A, ldn, n, dmax);
    return check_launch("diag_max_kernel");
}

int launch_gram_guard(imcom_ctx *ctx, double *G, int P, const int *want, int batch)
{
    hipLaunchKernelGGL(gram_guard_kernel, dim3(batch), dim3(P), 0, ctx->stream, G, P, want);
    return check_launch("gram_guard_kernel");
}

int launch_multi(imcom_ctx *ctx, const double *Xs, long node_stride, const double *Bt, int ldn, int ldm, int m, const int *n,
                 int nv, const double *kappaC_dev, const double *Cs, double ucmin, double smax, double *Dp, double *Npq,
                 double *W, float *Tt, float *UC, float *Sigma, float *kappa, int batch)
{
    IMCOM_REQUIRE(nv >= 2 && nv <= MAXNV, "multi-kappa: nv=%d outside [2,%d]", nv, MAXNV);
    hipLaunchKernelGGL(multi_reduce_kernel, dim3(ldm / 64, batch), dim3(256), 0, ctx->stream, Xs, node_stride, Bt, ldn, ldm, m, n, nv, Dp, Npq);
    IMCOM_TRY(check_launch("multi_reduce_kernel"));
    hipLaunchKernelGGL(multi_search_kernel, dim3((m + 127) / 128, batch), dim3(128), 0, ctx->stream, Dp, Npq, m, n, nv, kappaC_dev, Cs, ucmin, smax, UC, Sigma, kappa, W);
    IMCOM_TRY(check_launch("multi_search_kernel"));
    hipLaunchKernelGGL(multi_combine_kernel, dim3(ldm / 64, batch), dim3(256), 0, ctx->stream, Xs, node_stride, ldn, ldm, m, n, nv, W, Tt);
    return check_launch("multi_combine_kernel");
}

int launch_build_reduced_T(imcom_ctx *ctx, const double *Nf, const double *Df, const double *Ef, const double *kappa, int nv,
                           long m, double ucmin, double smax, double *ok, double *oS, double *oU, double *ow)
{
    IMCOM_REQUIRE(nv >= 2 && nv <= MAXNV, "build_reduced_T: nv=%d outside [2,%d]", nv, MAXNV);
    if (m <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(build_reduced_T_kernel, dim3((unsigned)((m + 127) / 128)), dim3(128), 0, ctx->stream, Nf, Df, Ef, kappa, nv, m, ucmin, smax, ok, oS, oU, ow);
    return check_launch("build_reduced_T_kernel");
}

int launch_lakernel1(imcom_ctx *ctx, const double *lam, const double *mPhalf, long m, long n, long ldp, double C,
                     double targetleak, double kCmin, double kCmax, int nbis, double *kappa, double *Sigma, double *UC,
                     double *T, long ldt, double smax)
{
    if (m <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(lakernel1_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, lam, mPhalf, m, n, ldp, C, targetleak, kCmin, kCmax, nbis, kappa, Sigma, UC, T, ldt, smax);
    return check_launch("lakernel1_kernel");
}

int launch_trapezoid_f32(imcom_ctx *ctx, float *maps, long nmaps, int n2f, int fade)
{
    const long tot = nmaps * n2f * n2f;
    if (tot <= 0 || fade <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(trapezoid_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, maps, nmaps, n2f, fade);
    return check_launch("trapezoid_f32_kernel");
}

int launch_clamp_min_f32(imcom_ctx *ctx, float *maps, long count, float lo)
{
    if (count <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(clamp_min_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, maps, count, lo);
    return check_launch("clamp_min_f32_kernel");
}

int launch_epilogue(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, int m, int ldm, int n2f, int fade, int n2, float *Tt,
                    const float *indata, int n_inframe, const int *expo, int n_expo, float *outimage, double *Tsum_image,
                    double *Tsum_stamp, double *Tsum_inpix, double *Neff)
{
    for (int f0 = 0; f0 < n_inframe; f0 += EPI_MAXF) {  // EPI_MAXF input layers per pass over T
        const int nf = n_inframe - f0 < EPI_MAXF ? n_inframe - f0 : EPI_MAXF;
        const int nslot = n_expo > nf ? n_expo : nf;
        const int cpt = (size_t)nslot * 256 * 4 * sizeof(double) <= 64 * 1024 + 2048 ? 4 : 1;  // at least 2 blocks per CU with 4
        size_t bytes = (size_t)nslot * 256 * cpt * sizeof(double);
        IMCOM_REQUIRE(bytes <= 128 * 1024, "epilogue: n_expo = %d too large", n_expo);
        if (const char *e = getenv("IMCOM_EPI_LDS_PAD")) bytes += (size_t)atoi(e);  // occupancy experiment (tools/ab_epi_occ.sh)
        if (cpt == 4) {
            IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)coadd_epilogue_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            hipLaunchKernelGGL(coadd_epilogue_kernel<4>, dim3((m + 255) / 256, batch), dim3(256), bytes, ctx->stream, Tt, ldn, ldm, m, n2f, fade,
                               n_dev, indata, n_inframe, f0, nf, expo, n_expo, outimage, Tsum_image, Tsum_inpix, Neff);
        } else {
            IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)coadd_epilogue_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            hipLaunchKernelGGL(coadd_epilogue_kernel<1>, dim3((m + 63) / 64, batch), dim3(256), bytes, ctx->stream, Tt, ldn, ldm, m, n2f, fade,
                               n_dev, indata, n_inframe, f0, nf, expo, n_expo, outimage, Tsum_image, Tsum_inpix, Neff);
        }
    }
    IMCOM_TRY(check_launch("coadd_epilogue_kernel"));
    hipLaunchKernelGGL(tsum_stamp_kernel, dim3(n_expo, batch), dim3(256), 0, ctx->stream, Tsum_image, m, n_expo, n2, Tsum_stamp);
    return check_launch("tsum_stamp_kernel");
}

int launch_coadd_from_partials(imcom_ctx *ctx, int batch, const int *n_dev, const int *nblk_dev, int ldn, int m, int ldm, int n2, const CoaddFuse &cf,
                               float *outimage, double *Tsum_image, double *Tsum_stamp, double *Tsum_inpix, double *Neff)
{
    hipLaunchKernelGGL(coadd_from_partials_kernel, dim3((m + 255) / 256, batch), dim3(256), 0, ctx->stream, cf.Epart, ldn, ldm, m, n_dev, nblk_dev, cf.n_inframe,
                       cf.n_expo, outimage, Tsum_image, Tsum_inpix, Neff);
    IMCOM_TRY(check_launch("coadd_from_partials_kernel"));
    hipLaunchKernelGGL(tsum_stamp_kernel, dim3(cf.n_expo, batch), dim3(256), 0, ctx->stream, Tsum_image, m, cf.n_expo, n2, Tsum_stamp);
    return check_launch("tsum_stamp_kernel");
}

}  // namespace imcom
