// eigen.hip -- the eigendecomposition LA path (reference src/pyimcom/lakernel.py:141-223 EigenKernel)
// and the public batched eigensolver entry point.
//
//   lam, Q = eigh(A)                     tridiag.hip (Householder tridiagonalisation + implicit QR)
//   P = (-B/2) Q                         fp64 MFMA GEMM
//   single kappa (154-172): Sigma_a = sum_i (P_ai/(lam_i+kappa))^2, UC_a = 1 - sum_i (lam_i+2 kappa) P_ai^2/(lam_i+kappa)^2 / C
//   multi kappa  (174-223): routine.lakernel1 per output pixel, then kappa *= C (line 222)
//   T = (P/(lam+kappa)) Q^T              fp64 MFMA GEMM, stored float32
#include "common.h"
#include "launchers.h"

namespace imcom {

// Bp[s][a][j] = B[s][a*ldb + j] for a < m, j < n[s]; zero elsewhere   ([mp][np] row-major)
__global__ void pad_B_kernel(const double *__restrict__ B, long ldb, int m, const int *__restrict__ n, double *__restrict__ Bp,
                             int mp, int np)
{
    const int s = blockIdx.z, a = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    double v = 0.0;
    if (a < m && j < n[s]) v = B[(long)s * m * ldb + (long)a * ldb + j];
    Bp[((long)s * mp + a) * np + j] = v;
}

// single kappa: one wave per output pixel
__global__ __launch_bounds__(256) void eigen_single_kernel(const double *__restrict__ lam, const double *__restrict__ P, int mp,
                                                           int np, int m, const int *__restrict__ n,
                                                           const double *__restrict__ kap, const double *__restrict__ Cs,
                                                           double *__restrict__ S, float *__restrict__ UC,
                                                           float *__restrict__ Sigma, float *__restrict__ kappa)
{
    const int s = blockIdx.y, a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a >= m) return;
    const int ns = n[s];
    const double k = kap[s], C = Cs[s];
    const double *l = lam + (long)s * np, *p = P + ((long)s * mp + a) * np;
    double *o = S + ((long)s * mp + a) * np;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < np; i += 64) {
        double v = 0.0;
        if (i < ns) {
            const double li = l[i];
            v = p[i] / (li + k);
            s2 += v * v;
            s1 += (li + 2.0 * k) * v * v;
        }
        o[i] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    if (lane == 0) {
        const long pa = (long)s * m + a;
        if (ns == 0) { UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f; }
        else { kappa[pa] = (float)k; Sigma[pa] = (float)s2; UC[pa] = (float)(1.0 - s1 / C); }
    }
}

// multi kappa post-processing: float32 stores as the reference does (lakernel.py:216-222)
__global__ void eigen_multi_store_kernel(const double *__restrict__ k64, const double *__restrict__ S64,
                                         const double *__restrict__ U64, int m, int ns, double C, float *__restrict__ UC,
                                         float *__restrict__ Sigma, float *__restrict__ kappa)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    if (ns == 0) { UC[a] = 1.0f; Sigma[a] = 0.0f; kappa[a] = 1.0f; return; }
    const float k32 = (float)k64[a];
    kappa[a] = (float)((double)k32 * C);
    Sigma[a] = (float)S64[a];
    UC[a] = (float)U64[a];
}

// T[s][a][i] (float32, [m][ldt]) = Tp[s][a][i] (float64, [mp][np]); columns i >= n[s] zero
__global__ void cast_T_kernel(const double *__restrict__ Tp, int mp, int np, int m, const int *__restrict__ n,
                              float *__restrict__ T, long ldt)
{
    const int s = blockIdx.z, a = blockIdx.y;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= ldt) return;
    float v = 0.0f;
    if (i < n[s]) v = (float)Tp[((long)s * mp + a) * np + i];
    T[(long)s * m * ldt + (long)a * ldt + i] = v;
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
    const size_t big = (size_t)batch * mp * np * 8;
    size_t total = eigh_ws_bytes(batch, np, true) + 4 * big + (size_t)batch * np * np * 8 + (size_t)batch * np * 8 + 3 * szM * 8 + 65536;
    if (host) total += szA * 8 + szB * 8 + szB * 4 + szM * 12;
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
    double *lam = (double *)ws_take(ctx, (size_t)batch * np * 8);
    double *Q = (double *)ws_take(ctx, (size_t)batch * np * np * 8);
    double *Bp = (double *)ws_take(ctx, big), *P = (double *)ws_take(ctx, big), *S = (double *)ws_take(ctx, big);
    double *Tp = (double *)ws_take(ctx, big);
    double *pix = (double *)ws_take(ctx, 3 * szM * 8);
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    double *kc = (double *)ws_take(ctx, (size_t)batch * 16);
    if (!lam || !Q || !Bp || !P || !S || !Tp || !pix || !n_dev || !kc) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    std::vector<double> kch(2 * (size_t)batch);
    for (int s = 0; s < batch; s++) { kch[s] = kappaC[0] * C[s]; kch[batch + s] = C[s]; }
    IMCOM_HIP_CHECK(hipMemcpyAsync(n_dev, n, (size_t)batch * 4, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(kc, kch.data(), kch.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    IMCOM_HIP_CHECK(hipMemsetAsync(Q, 0, (size_t)batch * np * np * 8, ctx->stream));
    IMCOM_HIP_CHECK(hipMemsetAsync(lam, 0, (size_t)batch * np * 8, ctx->stream));
    if (nmax > 0) {
        int sweeps = 0;
        IMCOM_TRY(eigh_device(ctx, batch, n, np, A_d, ldn, (long)ldn * ldn, lam, np, Q, np, (long)np * np, &sweeps));
    }
    hipLaunchKernelGGL(pad_B_kernel, dim3((np + 255) / 256, mp, batch), dim3(256), 0, ctx->stream, B_d, (long)ldn, m, n_dev, Bp, mp, np);
    IMCOM_TRY(check_launch("pad_B_kernel"));
    {   // P = Bp Q
        ProfScope ps(ctx, "eigen_gemm");
        IMCOM_TRY(launch_gemm(ctx, false, true, mp, np, np, batch, Bp, np, (long)mp * np, Q, np, (long)np * np, P, np, (long)mp * np, 1.0, 0.0));
    }
    if (nv == 1) {
        ProfScope ps(ctx, "lakernel1");
        hipLaunchKernelGGL(eigen_single_kernel, dim3((m + 3) / 4, batch), dim3(256), 0, ctx->stream, lam, P, mp, np, m, n_dev, kc, kc + batch, S,
                           UC_d, Sig_d, kap_d);
        IMCOM_TRY(check_launch("eigen_single_kernel"));
    } else {
        ProfScope ps(ctx, "lakernel1");
        IMCOM_HIP_CHECK(hipMemsetAsync(S, 0, big, ctx->stream));
        for (int s = 0; s < batch; s++) {
            double *k64 = pix + (size_t)s * m, *S64 = pix + szM + (size_t)s * m, *U64 = pix + 2 * szM + (size_t)s * m;
            if (n[s] > 0)
                IMCOM_TRY(launch_lakernel1(ctx, lam + (size_t)s * np, P + (size_t)s * mp * np, m, n[s], np, C[s], ucmin, kappaC[0] * C[s],
                                           kappaC[nv - 1] * C[s], nbis, k64, S64, U64, S + (size_t)s * mp * np, np, smax));
            hipLaunchKernelGGL(eigen_multi_store_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, k64, S64, U64, m, n[s], C[s],
                               UC_d + (size_t)s * m, Sig_d + (size_t)s * m, kap_d + (size_t)s * m);
        }
        IMCOM_TRY(check_launch("eigen_multi_store_kernel"));
    }
    {   // T = S Q^T
        ProfScope ps(ctx, "eigen_gemm");
        IMCOM_TRY(launch_gemm(ctx, false, false, mp, np, np, batch, S, np, (long)mp * np, Q, np, (long)np * np, Tp, np, (long)mp * np, 1.0, 0.0));
    }
    if (ldn > 0) {
        hipLaunchKernelGGL(cast_T_kernel, dim3((unsigned)((ldn + 255) / 256), m, batch), dim3(256), 0, ctx->stream, Tp, mp, np, m, n_dev, T_d, (long)ldn);
        IMCOM_TRY(check_launch("cast_T_kernel"));
    }
    if (host) {
        if (szB) IMCOM_HIP_CHECK(hipMemcpyAsync(T, T_d, szB * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(UC, UC_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(Sigma, Sig_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(kappa, kap_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}
