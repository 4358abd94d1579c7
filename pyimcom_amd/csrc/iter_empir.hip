// iter_empir.hip -- the two secondary LA kernels of the reference (src/pyimcom/lakernel.py):
//   IterKernel  533-744: per output pixel, conjugate gradients (397-442) on the sub-system of the input pixels
//                        within the acceptance radius rho_acc of that output pixel; float32 T
//   EmpirKernel 747-805: T_ai = max(rho_acc - dist_ai, 0) / sum_i(...), optional exact U/C, no linear solve
// Layout here is output-pixel-major ([m][n] rows, as the reference's T), one workgroup per output pixel.
#include <algorithm>

#include "common.h"
#include "launchers.h"

namespace imcom {

// iter_block.hip: the same solve for 4 x 4 patches of output pixels at a time (one MFMA product per CG step and patch)
size_t iter_block_select_bytes(int batch, int nblocks);
size_t iter_block_patch_bytes(int ups);
int iter_block_count(int m, int W);
int iter_block_umax();
int launch_iter_block(imcom_ctx *ctx, const double *A, long lda, long strideA, const double *diag, long ldd, const double *B, long ldb,
                      const double *oyx, const double *iy, const double *ix, long ldxy, const int *n, int m, int W, int batch,
                      double rho, double rtol, int maxiter, float *T, long ldt, void *ws, size_t budget, int *max_union, int *steps,
                      unsigned long long *stats, int *sym_used);

constexpr int CG_MAXSEL = 4096;            // input pixels inside one acceptance disc (LDS: 16 KB indices + 32 KB p)
constexpr int CG_SLOTS = CG_MAXSEL / 256;  // rows of the sub-system owned by one thread
constexpr int IT_MAXNV = 8;

// deterministic block sum over 256 threads (same order every run)
__device__ inline double block_sum4(double v, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// EmpirKernel weights (lakernel.py:763-771): T64 into the padded [mp][np] GEMM operand (may be null), T32 out.
__global__ __launch_bounds__(256) void empir_T_kernel(const double *__restrict__ oyx, const double *__restrict__ iy,
                                                      const double *__restrict__ ix, long ldxy, const int *__restrict__ n, int m,
                                                      double rho, double *__restrict__ Tp, int mp, int np,
                                                      float *__restrict__ T, long ldt)
{
    __shared__ double red[4];
    const int s = blockIdx.y, a = blockIdx.x, ns = n[s];
    const double oy = oyx[((long)s * 2 + 0) * m + a], ox = oyx[((long)s * 2 + 1) * m + a];
    const double *py = iy + s * ldxy, *px = ix + s * ldxy;
    double part = 0.0;
    for (int i = threadIdx.x; i < ns; i += 256) part += fmax(rho - hypot(oy - py[i], ox - px[i]), 0.0);
    const double tot = block_sum4(part, red);
    const int iend = max(Tp ? np : 0, (int)ldt);
    for (int i = threadIdx.x; i < iend; i += 256) {
        double v = 0.0;
        if (i < ns) v = fmax(rho - hypot(oy - py[i], ox - px[i]), 0.0) / tot;  // 0/0 = NaN as in the reference
        if (Tp && i < np) Tp[((long)s * mp + a) * np + i] = v;
        if (i < ldt) T[((long)s * m + a) * ldt + i] = (float)v;  // padding columns zero
    }
}

// D = sum B T, N = sum T^2, E = sum G T (G = T A);  kappa = kC * C, Sigma = N, UC = 1 + (E - 2 D) / C   (785-798)
__global__ __launch_bounds__(256) void empir_maps_kernel(const double *__restrict__ Tp, const double *__restrict__ G, int mp,
                                                         int np, const double *__restrict__ B, long ldb,
                                                         const int *__restrict__ n, int m, const double *__restrict__ kap,
                                                         const double *__restrict__ Cs, float *__restrict__ UC,
                                                         float *__restrict__ Sigma, float *__restrict__ kappa)
{
    __shared__ double red[4];
    const int s = blockIdx.y, a = blockIdx.x, ns = n[s];
    const double *t = Tp + ((long)s * mp + a) * np, *g = G + ((long)s * mp + a) * np;
    const double *b = B + ((long)s * m + a) * ldb;
    double d = 0.0, nn = 0.0, e = 0.0;
    for (int i = threadIdx.x; i < ns; i += 256) {
        const double ti = t[i];
        d += b[i] * ti;
        nn += ti * ti;
        e += g[i] * ti;
    }
    d = block_sum4(d, red);
    nn = block_sum4(nn, red);
    e = block_sum4(e, red);
    if (threadIdx.x == 0) {
        const long pa = (long)s * m + a;
        if (ns == 0) { UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f; return; }  // lakernel.py:110-119
        kappa[pa] = (float)kap[s];
        Sigma[pa] = (float)nn;
        UC[pa] = (float)(1.0 + (e - 2.0 * d) / Cs[s]);
    }
}

// ------------------------------------------------------------------------------------------------
// IterKernel._iterative_wrapper for one kappa node: one workgroup per (output pixel, stamp).
//   selection (ascending, as np.nonzero): hypot(oy - iy, ox - ix) < rho
//   CG from x = 0 on AA[sel][:, sel] x = b[sel], AA = A with the diagonal replaced by diag[] (the reference's
//   sequence of in-place adds, lakernel.py:632, 692), stop when |r| < rtol |b| or after maxiter steps.
// Row j of the sub-system belongs to thread (wave w = j & 3, lane (j >> 2) & 63), slot j >> 8: the matrix-vector
// product is formed row by row by whole waves (lanes over the gathered columns), the owner lane keeps the result.
__global__ __launch_bounds__(256) void iter_cg_kernel(const double *__restrict__ A, long lda, long strideA,
                                                      const double *__restrict__ diag, long ldd,
                                                      const double *__restrict__ B, long ldb,
                                                      const double *__restrict__ oyx, const double *__restrict__ iy,
                                                      const double *__restrict__ ix, long ldxy, const int *__restrict__ n, int m,
                                                      double rho, double rtol, int maxiter, float *__restrict__ T, long ldt,
                                                      int *__restrict__ status, int *__restrict__ steps)
{
    __shared__ int sel[CG_MAXSEL];
    __shared__ double pv[CG_MAXSEL];
    __shared__ double red[4];
    __shared__ int wcnt[4], total;
    const int s = blockIdx.y, a = blockIdx.x, ns = n[s];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double oy = oyx[((long)s * 2 + 0) * m + a], ox = oyx[((long)s * 2 + 1) * m + a];
    const double *py = iy + s * ldxy, *px = ix + s * ldxy;
    float *Trow = T + ((long)s * m + a) * ldt;
    for (int i = threadIdx.x; i < ldt; i += 256) Trow[i] = 0.0f;  // zero outside the disc and in the padding columns
    // ordered compaction of the accepted input pixels
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ns; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const bool in = i < ns && hypot(oy - py[i], ox - px[i]) < rho;
        const unsigned long long mask = __ballot(in);
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wave; w++) off += wcnt[w];
        off += __popcll(mask & ((1ull << lane) - 1ull));
        if (in && off < CG_MAXSEL) sel[off] = i;
        __syncthreads();
        if (threadIdx.x == 0) total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    const int nsel = total;
    if (nsel > CG_MAXSEL) {
        if (threadIdx.x == 0) atomicMax(status, nsel);
        return;
    }
    if (nsel == 0) {
        if (steps && threadIdx.x == 0) steps[(long)s * m + a] = 0;
        return;
    }
    const double *As = A + s * strideA, *dg = diag + s * ldd, *b = B + ((long)s * m + a) * ldb;
    double x[CG_SLOTS], r[CG_SLOTS], p[CG_SLOTS], q[CG_SLOTS];
    double bb = 0.0;
#pragma unroll
    for (int sl = 0; sl < CG_SLOTS; sl++) {
        const int j = 4 * (lane + 64 * sl) + wave;
        x[sl] = 0.0;
        q[sl] = 0.0;
        r[sl] = j < nsel ? b[sel[j]] : 0.0;
        p[sl] = r[sl];
        if (j < nsel) pv[j] = p[sl];
        bb += r[sl] * r[sl];
    }
    const double atol = sqrt(block_sum4(bb, red)) * rtol;
    double rho_prev = 0.0;
    const int nrows_w = (nsel - wave + 3) / 4;  // rows of this wave: j = wave + 4k, k < nrows_w
    int used = 0;
    for (int it = 0; it < maxiter; it++) {
        double rr = 0.0;
#pragma unroll
        for (int sl = 0; sl < CG_SLOTS; sl++) rr += r[sl] * r[sl];
        const double rho_cur = block_sum4(rr, red);
        if (sqrt(rho_cur) < atol) break;
        used++;
        if (it > 0) {
            const double beta = rho_cur / rho_prev;
#pragma unroll
            for (int sl = 0; sl < CG_SLOTS; sl++) {
                const int j = 4 * (lane + 64 * sl) + wave;
                p[sl] = p[sl] * beta + r[sl];
                if (j < nsel) pv[j] = p[sl];
            }
        }
        __syncthreads();
        // q = AA_sub p
#pragma unroll
        for (int sl = 0; sl < CG_SLOTS; sl++) {
            if (sl * 64 < nrows_w) {
                for (int kk = 0; kk < 64; kk++) {
                    const int k = sl * 64 + kk;
                    if (k >= nrows_w) break;
                    const int j = wave + 4 * k, gj = sel[j];
                    const double *row = As + (long)gj * lda;
                    double acc = 0.0;
                    for (int i = lane; i < nsel; i += 64) {
                        const int gi = sel[i];
                        acc += (gi == gj ? dg[gj] : row[gi]) * pv[i];
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                    if (lane == kk) q[sl] = acc;
                }
            }
        }
        double pq = 0.0;
#pragma unroll
        for (int sl = 0; sl < CG_SLOTS; sl++) {
            const int j = 4 * (lane + 64 * sl) + wave;
            if (j < nsel) pq += p[sl] * q[sl];
        }
        const double alpha = rho_cur / block_sum4(pq, red);
#pragma unroll
        for (int sl = 0; sl < CG_SLOTS; sl++) {
            x[sl] += alpha * p[sl];
            r[sl] -= alpha * q[sl];
        }
        rho_prev = rho_cur;
    }
#pragma unroll
    for (int sl = 0; sl < CG_SLOTS; sl++) {
        const int j = 4 * (lane + 64 * sl) + wave;
        if (j < nsel) Trow[sel[j]] = (float)x[sl];
    }
    if (steps && threadIdx.x == 0) steps[(long)s * m + a] = used;
}

// float32 node solution -> padded float64 GEMM operand [mp][np]
__global__ void iter_widen_kernel(const float *__restrict__ T, long ldt, int m, const int *__restrict__ n,
                                  double *__restrict__ Tp, int mp, int np)
{
    const int s = blockIdx.z, a = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    double v = 0.0;
    if (a < m && i < n[s]) v = (double)T[((long)s * m + a) * ldt + i];
    Tp[((long)s * mp + a) * np + i] = v;
}

// Node reductions of one output pixel (lakernel.py:636-646, 698-712): Dp[p] = sum B T_p, Npq = sum T_p T_q and
// Epq = sum (T_p A) T_q (exact) or Dp[q] - kappa_p Npq (approximate); everything from the float32 node solutions.
// nv == 1 writes the maps directly; nv > 1 fills the flat arrays of build_reduced_T_wrap (D and E divided by C).
__global__ __launch_bounds__(256) void iter_reduce_kernel(const float *__restrict__ Tn, long node_stride, long ldt,
                                                          const double *__restrict__ G, long g_node_stride, int mp, int np,
                                                          const double *__restrict__ B, long ldb,
                                                          const int *__restrict__ n, int m, int nv,
                                                          const double *__restrict__ kappaC, const double *__restrict__ Cs,
                                                          int exact, double *__restrict__ Nf, double *__restrict__ Df,
                                                          double *__restrict__ Ef, float *__restrict__ UC,
                                                          float *__restrict__ Sigma, float *__restrict__ kappa)
{
    __shared__ double red[4];
    const int s = blockIdx.y, a = blockIdx.x, ns = n[s];
    const double *b = B + ((long)s * m + a) * ldb;
    double d[IT_MAXNV], nn[IT_MAXNV][IT_MAXNV], ee[IT_MAXNV][IT_MAXNV];
#pragma unroll
    for (int p = 0; p < IT_MAXNV; p++) {
        d[p] = 0.0;
#pragma unroll
        for (int q = 0; q < IT_MAXNV; q++) { nn[p][q] = 0.0; ee[p][q] = 0.0; }
    }
    for (int i = threadIdx.x; i < ns; i += 256) {
        double t[IT_MAXNV];
#pragma unroll
        for (int p = 0; p < IT_MAXNV; p++)
            t[p] = p < nv ? (double)Tn[p * node_stride + ((long)s * m + a) * ldt + i] : 0.0;
        const double bi = b[i];
#pragma unroll
        for (int p = 0; p < IT_MAXNV; p++) {
            if (p < nv) {
                d[p] += bi * t[p];
                const double gp = exact ? G[p * g_node_stride + ((long)s * mp + a) * np + i] : 0.0;
#pragma unroll
                for (int q = 0; q <= p; q++) {
                    nn[p][q] += t[p] * t[q];
                    ee[p][q] += gp * t[q];
                }
            }
        }
    }
    const double C = Cs[s];
    const long pa = (long)s * m + a;
#pragma unroll
    for (int p = 0; p < IT_MAXNV; p++) {
        if (p < nv) {
            d[p] = block_sum4(d[p], red);
#pragma unroll
            for (int q = 0; q <= p; q++) {
                nn[p][q] = block_sum4(nn[p][q], red);
                if (exact) ee[p][q] = block_sum4(ee[p][q], red);
            }
        }
    }
    if (threadIdx.x != 0) return;
    if (ns == 0) {  // a stamp without input pixels inside a batch: lakernel.py:110-119
        UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f;
        if (nv > 1) {  // neutral node data: the reduced-space search of this pixel is overwritten by the combine kernel
            for (int p = 0; p < nv; p++) {
                Df[pa * nv + p] = 0.0;
                for (int q = 0; q < nv; q++) { Nf[(pa * nv + p) * nv + q] = 0.0; Ef[(pa * nv + p) * nv + q] = 0.0; }
            }
        }
        return;
    }
    if (nv == 1) {
        const double k = kappaC[0] * C;
        kappa[pa] = (float)k;
        Sigma[pa] = (float)nn[0][0];
        UC[pa] = exact ? (float)(1.0 + (ee[0][0] - 2.0 * d[0]) / C) : (float)(1.0 - (k * nn[0][0] + d[0]) / C);
        return;
    }
#pragma unroll
    for (int p = 0; p < IT_MAXNV; p++) {
        if (p < nv) {
            Df[pa * nv + p] = d[p] / C;
#pragma unroll
            for (int q = 0; q <= p; q++) {
                const double e = exact ? ee[p][q] : d[q] - (kappaC[p] * C) * nn[p][q];
                Nf[(pa * nv + p) * nv + q] = Nf[(pa * nv + q) * nv + p] = nn[p][q];
                Ef[(pa * nv + p) * nv + q] = Ef[(pa * nv + q) * nv + p] = e / C;
            }
        }
    }
}

// multi-kappa outputs (lakernel.py:735-741): kappa * C, Sigma, UC and T = sum_p w_p T_p
__global__ __launch_bounds__(256) void iter_combine_kernel(const float *__restrict__ Tn, long node_stride, long ldt, int m,
                                                           const int *__restrict__ n, int nv, const double *__restrict__ w,
                                                           const double *__restrict__ ok, const double *__restrict__ oS,
                                                           const double *__restrict__ oU, const double *__restrict__ Cs,
                                                           float *__restrict__ T, float *__restrict__ UC,
                                                           float *__restrict__ Sigma, float *__restrict__ kappa)
{
    const int s = blockIdx.y, a = blockIdx.x, ns = n[s];
    const long pa = (long)s * m + a;
    double wp[IT_MAXNV];
#pragma unroll
    for (int p = 0; p < IT_MAXNV; p++) wp[p] = p < nv ? w[pa * nv + p] : 0.0;
    for (int i = threadIdx.x; i < ns; i += 256) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < IT_MAXNV; p++)
            if (p < nv) acc += (double)Tn[p * node_stride + pa * ldt + i] * wp[p];
        T[pa * ldt + i] = (float)acc;
    }
    if (threadIdx.x == 0) {
        if (ns == 0) { UC[pa] = 1.0f; Sigma[pa] = 0.0f; kappa[pa] = 1.0f; return; }
        kappa[pa] = (float)(ok[pa] * Cs[s]);
        Sigma[pa] = (float)oS[pa];
        UC[pa] = (float)oU[pa];
    }
}

}  // namespace imcom

using namespace imcom;

static int ctx_ok2(imcom_ctx *ctx)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    return IMCOM_OK;
}

namespace {
struct Stage {  // host -> device staging of one array through the context workspace
    imcom_ctx *ctx;
    bool host;
    template <typename T>
    int in(const T *src, size_t count, const T **dst)
    {
        *dst = src;
        if (!host || !src || count == 0) return IMCOM_OK;
        T *d = (T *)ws_take(ctx, count * sizeof(T));
        if (!d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(d, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        *dst = d;
        return IMCOM_OK;
    }
    template <typename T>
    int out(T *user, size_t count, T **dev)
    {
        *dev = user;
        if (!host || count == 0) return IMCOM_OK;
        *dev = (T *)ws_take(ctx, count * sizeof(T));
        if (!*dev) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        return IMCOM_OK;
    }
    template <typename T>
    int back(T *user, const T *dev, size_t count)
    {
        if (!host || count == 0) return IMCOM_OK;
        IMCOM_HIP_CHECK(hipMemcpyAsync(user, dev, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        return IMCOM_OK;
    }
};
}  // namespace

extern "C" int imcom_solve_empir(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A, const double *mBhalf,
                                 const double *C, double kappaC0, const double *out_yx, const double *in_y, const double *in_x,
                                 double rho_acc, int no_qlt_ctrl, float *T, float *UC, float *Sigma, float *kappa, int memspace)
{
    IMCOM_TRY(ctx_ok2(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && out_yx && m >= 1 && ldn >= 0 && UC && Sigma && kappa, "null pointer / bad sizes");
    int nmax = 0;
    for (int s = 0; s < batch; s++) {
        IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn=%d", s, n[s], ldn);
        nmax = std::max(nmax, n[s]);
    }
    IMCOM_REQUIRE(nmax == 0 || (in_y && in_x && T), "null coordinate / T pointer");
    IMCOM_REQUIRE(no_qlt_ctrl || nmax == 0 || (A && mBhalf && C), "quality control needs A, -B/2 and C");
    const bool host = memspace == IMCOM_MEM_HOST, qc = !no_qlt_ctrl;
    const int np = (int)align_up((size_t)std::max(nmax, 1), NB), mp = (int)align_up((size_t)m, NB);
    const size_t szA = (size_t)batch * ldn * ldn, szB = (size_t)batch * m * ldn, szM = (size_t)batch * m;
    const size_t big = (size_t)batch * mp * np * 8;
    size_t total = 65536 + (qc ? 2 * big + (size_t)batch * np * np * 8 : 0) + (size_t)batch * 24;
    if (host) total += (qc ? (szA + szB) * 8 : 0) + szB * 4 + szM * 12 + szM * 16 + (size_t)batch * ldn * 16 + 4096;
    IMCOM_TRY(ws_reserve(ctx, total));
    Stage st{ctx, host};
    const double *A_d = nullptr, *B_d = nullptr, *yx_d, *iy_d, *ix_d;
    if (qc) { IMCOM_TRY(st.in(A, szA, &A_d)); IMCOM_TRY(st.in(mBhalf, szB, &B_d)); }
    IMCOM_TRY(st.in(out_yx, 2 * szM, &yx_d));
    IMCOM_TRY(st.in(in_y, (size_t)batch * ldn, &iy_d));
    IMCOM_TRY(st.in(in_x, (size_t)batch * ldn, &ix_d));
    float *T_d, *UC_d, *Sig_d, *kap_d;
    IMCOM_TRY(st.out(T, szB, &T_d));
    IMCOM_TRY(st.out(UC, szM, &UC_d));
    IMCOM_TRY(st.out(Sigma, szM, &Sig_d));
    IMCOM_TRY(st.out(kappa, szM, &kap_d));
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    double *kc = (double *)ws_take(ctx, (size_t)batch * 16);
    if (!n_dev || !kc) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    std::vector<double> kch(2 * (size_t)batch, 0.0);
    for (int s = 0; s < batch && qc; s++) { kch[s] = kappaC0 * C[s]; kch[batch + s] = C[s]; }
    IMCOM_HIP_CHECK(hipMemcpyAsync(n_dev, n, (size_t)batch * 4, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(kc, kch.data(), kch.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double *Tp = nullptr, *G = nullptr, *Ap = nullptr;
    if (qc && nmax > 0) {
        Tp = (double *)ws_take(ctx, big);
        G = (double *)ws_take(ctx, big);
        Ap = (double *)ws_take(ctx, (size_t)batch * np * np * 8);
        if (!Tp || !G || !Ap) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemsetAsync(Tp, 0, big, ctx->stream));
    }
    IMCOM_HIP_CHECK(hipMemsetAsync(UC_d, 0, szM * 4, ctx->stream));  // no quality control: maps stay zero (lakernel.py:123-125, 774-777)
    IMCOM_HIP_CHECK(hipMemsetAsync(Sig_d, 0, szM * 4, ctx->stream));
    IMCOM_HIP_CHECK(hipMemsetAsync(kap_d, 0, szM * 4, ctx->stream));
    if (nmax > 0) {
        ProfScope ps(ctx, "empir");
        hipLaunchKernelGGL(empir_T_kernel, dim3(m, batch), dim3(256), 0, ctx->stream, yx_d, iy_d, ix_d, (long)ldn, n_dev, m, rho_acc, Tp, mp, np,
                           T_d, (long)ldn);
        IMCOM_TRY(check_launch("empir_T_kernel"));
        if (qc) {
            IMCOM_TRY(launch_pack_A(ctx, A_d, ldn, n_dev, Ap, np, batch));
            IMCOM_TRY(launch_gemm(ctx, false, true, mp, np, np, batch, Tp, np, (long)mp * np, Ap, np, (long)np * np, G, np, (long)mp * np, 1.0, 0.0));
            hipLaunchKernelGGL(empir_maps_kernel, dim3(m, batch), dim3(256), 0, ctx->stream, Tp, G, mp, np, B_d, (long)ldn, n_dev, m, kc, kc + batch,
                               UC_d, Sig_d, kap_d);
            IMCOM_TRY(check_launch("empir_maps_kernel"));
        }
    }
    IMCOM_TRY(st.back(T, T_d, szB));
    IMCOM_TRY(st.back(UC, UC_d, szM));
    IMCOM_TRY(st.back(Sigma, Sig_d, szM));
    IMCOM_TRY(st.back(kappa, kap_d, szM));
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

extern "C" int imcom_solve_iter(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A, const double *mBhalf,
                                const double *C, const double *kappaC, int nv, double ucmin, double smax, const double *out_yx,
                                const double *in_y, const double *in_x, double rho_acc, double rtol, int maxiter, int exact_UC,
                                float *T, float *UC, float *Sigma, float *kappa, int memspace)
{
    IMCOM_TRY(ctx_ok2(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && C && kappaC && out_yx && UC && Sigma && kappa, "null pointer / empty batch");
    IMCOM_REQUIRE(m >= 1 && nv >= 1 && nv <= IT_MAXNV && ldn >= 0 && maxiter >= 0, "bad sizes (nv <= %d)", IT_MAXNV);
    int nmax = 0;
    for (int s = 0; s < batch; s++) {
        IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn=%d", s, n[s], ldn);
        nmax = std::max(nmax, n[s]);
    }
    IMCOM_REQUIRE(nmax == 0 || (A && mBhalf && T && in_y && in_x), "null matrix / coordinate pointer");
    const bool host = memspace == IMCOM_MEM_HOST, exact = exact_UC != 0;
    const int np = (int)align_up((size_t)std::max(nmax, 1), NB), mp = (int)align_up((size_t)m, NB);
    // Width of the output-pixel grid (pixels arrive row by row: yx_val.ravel(), lakernel.py:613-614): the first index whose y
    // differs from pixel 0's.  It only decides which 16 pixels share a workgroup of the blocked solver; any value is correct.
    static const bool per_pixel = getenv("IMCOM_ITER_PER_PIXEL") != nullptr;  // A/B and cross-check: the one-pixel-per-workgroup kernel
    int gridW = m;
    if (!per_pixel && nmax > 0) {
        std::vector<double> y0((size_t)std::min(m, 8192));
        if (host) std::copy(out_yx, out_yx + y0.size(), y0.begin());
        else {
            IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            IMCOM_HIP_CHECK(hipMemcpy(y0.data(), out_yx, y0.size() * 8, hipMemcpyDeviceToHost));
        }
        for (size_t a = 1; a < y0.size(); a++)
            if (y0[a] != y0[0]) { gridW = (int)a; break; }
    }
    const int nblocks = iter_block_count(m, gridW);
    const size_t szA = (size_t)batch * ldn * ldn, szB = (size_t)batch * m * ldn, szM = (size_t)batch * m;
    const size_t big = (size_t)batch * mp * np * 8;
    size_t total = 65536 + (size_t)nv * szB * 4 + (size_t)batch * ldn * 8 + (size_t)batch * (MAX_INC + 4) * 8 + szM * 8 * (4 + nv + 2 * nv * nv);
    if (exact) total += big * (1 + nv) + (size_t)batch * np * np * 8;
    if (host) total += (szA + szB) * 8 + szB * 4 + szM * 12 + szM * 16 + (size_t)batch * ldn * 16 + 4096;
    // The blocked solver keeps one dense sub-matrix per 4 x 4 patch (4 MB at the reference's default configuration, 64 patches per
    // stamp): the patches of a call go through it in groups that fit a fixed share of workspace instead of batch x 0.3 GB.
    const size_t blk_budget = per_pixel ? 0 : std::min<size_t>((size_t)8 << 30, (size_t)batch * nblocks * iter_block_patch_bytes(iter_block_umax()));
    if (!per_pixel) total += iter_block_select_bytes(batch, nblocks) + blk_budget + 8192;
    total += szM * 4 + 64;  // per-pixel step counts, statistics
    IMCOM_TRY(ws_reserve(ctx, total));
    Stage st{ctx, host};
    const double *A_d, *B_d, *yx_d, *iy_d, *ix_d;
    IMCOM_TRY(st.in(A, szA, &A_d));
    IMCOM_TRY(st.in(mBhalf, szB, &B_d));
    IMCOM_TRY(st.in(out_yx, 2 * szM, &yx_d));
    IMCOM_TRY(st.in(in_y, (size_t)batch * ldn, &iy_d));
    IMCOM_TRY(st.in(in_x, (size_t)batch * ldn, &ix_d));
    float *T_d, *UC_d, *Sig_d, *kap_d;
    IMCOM_TRY(st.out(T, szB, &T_d));
    IMCOM_TRY(st.out(UC, szM, &UC_d));
    IMCOM_TRY(st.out(Sigma, szM, &Sig_d));
    IMCOM_TRY(st.out(kappa, szM, &kap_d));
    float *Tn = nv == 1 ? T_d : (float *)ws_take(ctx, (size_t)nv * szB * 4);  // node solutions [nv][batch][m][ldn]
    double *dsh = (double *)ws_take(ctx, (size_t)batch * ldn * 8);
    double *inc = (double *)ws_take(ctx, (size_t)batch * MAX_INC * 8);
    int *ints = (int *)ws_take(ctx, (size_t)batch * 8 + 8);
    double *kc = (double *)ws_take(ctx, (size_t)(batch + nv) * 8);
    double *flat = (double *)ws_take(ctx, szM * 8 * (4 + nv + 2 * nv * nv));
    void *blkws = per_pixel ? nullptr : ws_take(ctx, iter_block_select_bytes(batch, nblocks) + blk_budget + 4096);
    int *steps_d = (int *)ws_take(ctx, szM * 4);
    unsigned long long *stats_d = (unsigned long long *)ws_take(ctx, 64);
    if (!Tn || !dsh || !inc || !ints || !kc || !flat || (!per_pixel && !blkws) || !steps_d || !stats_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_HIP_CHECK(hipMemsetAsync(stats_d, 0, 64, ctx->stream));
    IMCOM_HIP_CHECK(hipMemsetAsync(steps_d, 0, szM * 4, ctx->stream));
    ctx->iter_steps.clear();
    for (double &v : ctx->iter_stats) v = 0.0;
    int *n_dev = ints, *ninc = ints + batch, *status = ints + 2 * batch;
    double *Cs_d = kc, *kappaC_d = kc + batch;
    double *Nf = flat, *Df = Nf + szM * nv * nv, *Ef = Df + szM * nv, *ok = Ef + szM * nv * nv, *oS = ok + szM, *oU = oS + szM, *ow = oU + szM;
    IMCOM_HIP_CHECK(hipMemcpyAsync(n_dev, n, (size_t)batch * 4, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(Cs_d, C, (size_t)batch * 8, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(kappaC_d, kappaC, (size_t)nv * 8, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemsetAsync(status, 0, 4, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (nmax == 0) {  // lakernel.py:110-119
        std::vector<float> one(szM, 1.0f);
        IMCOM_HIP_CHECK(hipMemcpyAsync(UC_d, one.data(), szM * 4, hipMemcpyHostToDevice, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(kap_d, one.data(), szM * 4, hipMemcpyHostToDevice, ctx->stream));
        IMCOM_HIP_CHECK(hipMemsetAsync(Sig_d, 0, szM * 4, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } else {
        std::vector<double> inc_h((size_t)batch * MAX_INC, 0.0);
        std::vector<int> ninc_h(batch, 0);
        for (int p = 0; p < nv; p++) {
            // diagonal of AA after the reference's in-place adds: += kappa_0 (skipped when zero, 631-632) resp.
            // += kappa_j - kappa_{j-1} (692)
            for (int s = 0; s < batch; s++) {
                const double kp = kappaC[p] * C[s], kprev = p > 0 ? kappaC[p - 1] * C[s] : 0.0;
                if (nv == 1) { ninc_h[s] = kp != 0.0 ? 1 : 0; inc_h[(size_t)s * MAX_INC] = kp; }
                else { inc_h[(size_t)s * MAX_INC + p] = kp - kprev; ninc_h[s] = p + 1; }
            }
            IMCOM_HIP_CHECK(hipMemcpyAsync(inc, inc_h.data(), inc_h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            IMCOM_HIP_CHECK(hipMemcpyAsync(ninc, ninc_h.data(), (size_t)batch * 4, hipMemcpyHostToDevice, ctx->stream));
            IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            IMCOM_TRY(launch_diag_shift(ctx, A_d, ldn, inc, ninc, dsh, batch));
            bool blocked = false;
            if (!per_pixel) {  // 4 x 4 patches of output pixels per workgroup (iter_block.hip), unless a patch selects too many input pixels
                int mx = 0, sym_used = 0;
                IMCOM_HIP_CHECK(hipMemsetAsync(Tn + (size_t)p * szB, 0, szB * 4, ctx->stream));
                IMCOM_HIP_CHECK(hipMemsetAsync(stats_d, 0, 64, ctx->stream));  // (statistics and step counts: of the LAST node)
                IMCOM_TRY(launch_iter_block(ctx, A_d, (long)ldn, (long)ldn * ldn, dsh, (long)ldn, B_d, (long)ldn, yx_d, iy_d, ix_d, (long)ldn, n_dev, m, gridW,
                                            batch, rho_acc, rtol, maxiter, Tn + (size_t)p * szB, (long)ldn, blkws, blk_budget, &mx, steps_d, stats_d, &sym_used));
                blocked = mx <= iter_block_umax();  // else: some patch selects more input pixels than the blocked solver holds -- the per-pixel kernel redoes the node
                ctx->iter_stats[4] = (double)mx;
                ctx->iter_stats[7] = (double)sym_used;
            }
            if (!blocked) {
                ProfScope ps(ctx, "iter_cg");
                hipLaunchKernelGGL(iter_cg_kernel, dim3(m, batch), dim3(256), 0, ctx->stream, A_d, (long)ldn, (long)ldn * ldn, dsh, (long)ldn, B_d,
                                   (long)ldn, yx_d, iy_d, ix_d, (long)ldn, n_dev, m, rho_acc, rtol, maxiter, Tn + (size_t)p * szB, (long)ldn, status, steps_d);
                IMCOM_TRY(check_launch("iter_cg_kernel"));
            }
            ctx->iter_stats[5] = blocked ? 1.0 : 0.0;
        }
        int st_h = 0;
        unsigned long long stats_h[5] = {0, 0, 0, 0, 0};
        ctx->iter_steps.resize(szM);
        IMCOM_HIP_CHECK(hipMemcpyAsync(&st_h, status, 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(stats_h, stats_d, 40, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(ctx->iter_steps.data(), steps_d, szM * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 4; k++) ctx->iter_stats[k] = (double)stats_h[k];
        ctx->iter_stats[6] = (double)stats_h[4];
        IMCOM_REQUIRE(st_h == 0, "iterative kernel: %d input pixels inside one acceptance disc exceed the limit of %d", st_h, CG_MAXSEL);
        double *G = nullptr;
        if (exact) {
            double *Tp = (double *)ws_take(ctx, big), *Ap = (double *)ws_take(ctx, (size_t)batch * np * np * 8);
            G = (double *)ws_take(ctx, big * nv);
            if (!Tp || !Ap || !G) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
            IMCOM_TRY(launch_pack_A(ctx, A_d, ldn, n_dev, Ap, np, batch));
            for (int p = 0; p < nv; p++) {
                hipLaunchKernelGGL(iter_widen_kernel, dim3((np + 255) / 256, mp, batch), dim3(256), 0, ctx->stream, Tn + (size_t)p * szB, (long)ldn, m,
                                   n_dev, Tp, mp, np);
                IMCOM_TRY(launch_gemm(ctx, false, true, mp, np, np, batch, Tp, np, (long)mp * np, Ap, np, (long)np * np, G + (size_t)p * batch * mp * np, np,
                                      (long)mp * np, 1.0, 0.0));
            }
        }
        ProfScope ps_red(ctx, "finalize");
        hipLaunchKernelGGL(iter_reduce_kernel, dim3(m, batch), dim3(256), 0, ctx->stream, Tn, (long)szB, (long)ldn, G, (long)batch * mp * np, mp, np, B_d,
                           (long)ldn, n_dev, m, nv, kappaC_d, Cs_d, exact ? 1 : 0, Nf, Df, Ef, UC_d, Sig_d, kap_d);
        IMCOM_TRY(check_launch("iter_reduce_kernel"));
        if (nv > 1) {
            IMCOM_TRY(launch_build_reduced_T(ctx, Nf, Df, Ef, kappaC_d, nv, (long)szM, ucmin, smax, ok, oS, oU, ow));
            hipLaunchKernelGGL(iter_combine_kernel, dim3(m, batch), dim3(256), 0, ctx->stream, Tn, (long)szB, (long)ldn, m, n_dev, nv, ow, ok, oS, oU,
                               Cs_d, T_d, UC_d, Sig_d, kap_d);
            IMCOM_TRY(check_launch("iter_combine_kernel"));
        }
    }
    IMCOM_TRY(st.back(T, T_d, szB));
    IMCOM_TRY(st.back(UC, UC_d, szM));
    IMCOM_TRY(st.back(Sigma, Sig_d, szM));
    IMCOM_TRY(st.back(kappa, kap_d, szM));
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}


extern "C" int imcom_solve_iter_stats(imcom_ctx *ctx, double *stats, int *steps, long nsteps)
{
    IMCOM_TRY(ctx_ok2(ctx));
    IMCOM_REQUIRE(stats, "null pointer");
    for (int k = 0; k < 8; k++) stats[k] = ctx->iter_stats[k];
    if (steps) {
        IMCOM_REQUIRE(nsteps == (long)ctx->iter_steps.size(), "step counts of the last imcom_solve_iter call: %zu pixels, caller asks for %ld", ctx->iter_steps.size(), nsteps);
        std::copy(ctx->iter_steps.begin(), ctx->iter_steps.end(), steps);
    }
    return IMCOM_OK;
}
