// gemm_f64.hip -- fp64 MFMA tile engine and the blocked Cholesky / triangular-solve kernels.
//
// Replaces scipy.linalg.cholesky + cho_solve as used by lakernel.CholKernel
// (reference src/pyimcom/lakernel.py:262-279, 295-304, 355-358) for a BATCH of postage stamps.
//
// Design (gfx950): one 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each wave
// accumulates a 64x64 sub-tile as 4x4 v_mfma_f64_16x16x4_f64 tiles (128 accumulator VGPRs).  Operand
// tiles (128 x 16 doubles) are staged global -> registers -> LDS with the next tile's global loads in
// flight behind the current tile's 64 MFMAs per wave.  LDS images are padded so the per-lane
// ds_read_b64 fragment reads are bank-conflict-free:
//   row-major image  [128][16+2]  : lane (i = l&15, k = l>>4) reads word i*18 + k  -> 32 distinct banks
//   k-major image    [16][128+16] : lane reads word k*144 + i                      -> 32 distinct banks
// fp64 MFMA runs at the fp64 vector rate on gfx950, so the point of MFMA here is operand reuse
// (one 8-byte LDS read per lane feeds 2048 flops), not a higher peak.
#include "common.h"

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int LDS_RM = BK + 2;         // row-major image stride (doubles)
constexpr int LDS_KM = NB + 16;        // k-major image stride (doubles)
constexpr int TILE_WORDS = NB * LDS_RM;  // == BK * LDS_KM == 2304 doubles
static_assert(NB * LDS_RM == BK * LDS_KM, "operand images must have one size");

// acc[mi][ni] += sum_k Aop[m][k] * Bop[k][n] over K (multiple of 16), for the 128x128 tile whose
// operands start at Ag / Bg.  KMAJOR operand: element (r,k) at p[k*ld + r]; else at p[r*ld + k].
template <bool AKM, bool BKM>
__device__ __forceinline__ void mma_tile(f64x4 (&acc)[4][4], const double *__restrict__ Ag, long lda,
                                         const double *__restrict__ Bg, long ldb, int K, double *sA,
                                         double *sB)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    f64x2 ra[4], rb[4];
    const int nt = K / BK;

    auto gload = [&](int t) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = tid + 256 * q;
            if (AKM) ra[q] = *(const f64x2 *)(Ag + (long)(t * BK + (c >> 6)) * lda + (c & 63) * 2);
            else     ra[q] = *(const f64x2 *)(Ag + (long)(c >> 3) * lda + t * BK + (c & 7) * 2);
            if (BKM) rb[q] = *(const f64x2 *)(Bg + (long)(t * BK + (c >> 6)) * ldb + (c & 63) * 2);
            else     rb[q] = *(const f64x2 *)(Bg + (long)(c >> 3) * ldb + t * BK + (c & 7) * 2);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = tid + 256 * q;
            if (AKM) *(f64x2 *)(sA + (c >> 6) * LDS_KM + (c & 63) * 2) = ra[q];
            else     *(f64x2 *)(sA + (c >> 3) * LDS_RM + (c & 7) * 2) = ra[q];
            if (BKM) *(f64x2 *)(sB + (c >> 6) * LDS_KM + (c & 63) * 2) = rb[q];
            else     *(f64x2 *)(sB + (c >> 3) * LDS_RM + (c & 7) * 2) = rb[q];
        }
    };

    if (nt > 0) gload(0);
    for (int t = 0; t < nt; t++) {
        __syncthreads();  // every wave finished reading the previous images
        sstore();
        __syncthreads();
        if (t + 1 < nt) gload(t + 1);  // in flight behind the MFMAs below
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                a[i] = AKM ? sA[(kk * 4 + lk) * LDS_KM + wm * 64 + i * 16 + li]
                           : sA[(wm * 64 + i * 16 + li) * LDS_RM + kk * 4 + lk];
                b[i] = BKM ? sB[(kk * 4 + lk) * LDS_KM + wn * 64 + i * 16 + li]
                           : sB[(wn * 64 + i * 16 + li) * LDS_RM + kk * 4 + lk];
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void zero_acc(f64x4 (&acc)[4][4])
{
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
}

// C/D map of v_mfma_f64_16x16x4_f64: register r of lane l holds (row (l>>4) + 4r, col l&15).
#define IMCOM_FOR_ACC(ROW, COL, VAL, BODY)                                   \
    {                                                                        \
        const int lane__ = threadIdx.x & 63, wave__ = threadIdx.x >> 6;      \
        const int wm__ = wave__ >> 1, wn__ = wave__ & 1;                     \
        _Pragma("unroll") for (int i__ = 0; i__ < 4; i__++)                  \
        _Pragma("unroll") for (int j__ = 0; j__ < 4; j__++)                  \
        _Pragma("unroll") for (int r__ = 0; r__ < 4; r__++) {                \
            const int ROW = wm__ * 64 + i__ * 16 + (lane__ >> 4) + 4 * r__;  \
            const int COL = wn__ * 64 + j__ * 16 + (lane__ & 15);            \
            const double VAL = acc[i__][j__][r__];                           \
            BODY                                                             \
        }                                                                    \
    }

// ---------------------------------------------------------------------------------------------
// Left-looking blocked Cholesky of A + kappa*I, block column k:
//   P[i] = A[i,k] + kappa*delta - sum_{j<k} L[i,j] L[k,j]^T          (chol_update_kernel, i >= k)
//   L[k,k], Linv[k] from P[k]                                         (chol_diag.hip)
//   L[i,k] = P[i] Linv[k]^T                                           (chol_trsm_kernel, i > k)
// A is never written; L lives in its own buffer (the repair path and the multi-kappa path need A).
__global__ __launch_bounds__(256, 2) void chol_update_kernel(const double *__restrict__ A,
                                                             double *__restrict__ L, int ldn, int k,
                                                             const int *__restrict__ nblk,
                                                             const double *__restrict__ dshift)
{
    __shared__ double smem[2 * TILE_WORDS];
    const int s = blockIdx.y, i = k + blockIdx.x;
    if (i >= nblk[s]) return;
    const long sA = (long)ldn * ldn;
    double *Ls = L + s * sA;
    f64x4 acc[4][4];
    zero_acc(acc);
    mma_tile<false, false>(acc, Ls + (long)i * NB * ldn, ldn, Ls + (long)k * NB * ldn, ldn, k * NB, smem,
                           smem + TILE_WORDS);
    const double *As = A + s * sA + (long)i * NB * ldn + k * NB;
    double *Lo = Ls + (long)i * NB * ldn + k * NB;
    // dshift = diagonal of A with the kappa increments already applied (diag_shift_kernel)
    const double *dsh = dshift + (long)s * ldn + i * NB;
    IMCOM_FOR_ACC(row, col, v, {
        double a = As[(long)row * ldn + col];
        if (i == k && row == col) a = dsh[row];
        Lo[(long)row * ldn + col] = a - v;
    })
}

__global__ __launch_bounds__(256, 2) void chol_trsm_kernel(double *__restrict__ L,
                                                           const double *__restrict__ Dinv, int ldn,
                                                           int k, const int *__restrict__ nblk)
{
    __shared__ double smem[2 * TILE_WORDS];
    const int s = blockIdx.y, i = k + 1 + blockIdx.x;
    if (i >= nblk[s]) return;
    double *P = L + (long)s * ldn * ldn + (long)i * NB * ldn + k * NB;
    const double *Di = Dinv + ((long)s * (ldn / NB) + k) * NB * NB;
    f64x4 acc[4][4];
    zero_acc(acc);
    mma_tile<false, false>(acc, P, ldn, Di, NB, NB, smem, smem + TILE_WORDS);
    __syncthreads();  // all of P[i] has been read by every wave before it is overwritten
    IMCOM_FOR_ACC(row, col, v, { P[(long)row * ldn + col] = v; })
}

// ---------------------------------------------------------------------------------------------
// Blocked triangular solves with m right-hand sides, input-pixel-major: L Y = Bt, L^T X = Y.
//   forward  block row k:  R = Bt_k - L[k,0:k] Y[0:k]   then  Y_k = Linv[k]   R
//   backward block row k:  R = Y_k - L[k+1:,k]^T X[k+1:] then  X_k = Linv[k]^T R
// The update kernels carry all the 2 N^2 m flops of the solve (the "solve_gemm" family).
__global__ __launch_bounds__(256, 2) void solve_fwd_kernel(const double *__restrict__ L,
                                                           const double *__restrict__ Bt,
                                                           double *__restrict__ Y, int ldn, int ldm,
                                                           int k, const int *__restrict__ nblk)
{
    __shared__ double smem[2 * TILE_WORDS];
    const int s = blockIdx.y, c = blockIdx.x;
    if (k >= nblk[s]) return;
    const double *Ls = L + (long)s * ldn * ldn + (long)k * NB * ldn;
    double *Ys = Y + (long)s * ldn * ldm + c * NB;
    f64x4 acc[4][4];
    zero_acc(acc);
    mma_tile<false, true>(acc, Ls, ldn, Ys, ldm, k * NB, smem, smem + TILE_WORDS);
    const double *Bs = Bt + (long)s * ldn * ldm + (long)k * NB * ldm + c * NB;
    double *Yo = Ys + (long)k * NB * ldm;
    IMCOM_FOR_ACC(row, col, v, { Yo[(long)row * ldm + col] = Bs[(long)row * ldm + col] - v; })
}

__global__ __launch_bounds__(256, 2) void solve_bwd_kernel(const double *__restrict__ L,
                                                           double *__restrict__ Y, int ldn, int ldm,
                                                           int k, const int *__restrict__ nblk)
{
    __shared__ double smem[2 * TILE_WORDS];
    const int s = blockIdx.y, c = blockIdx.x;
    const int nb = nblk[s];
    if (k >= nb - 1) return;  // last block row: nothing to subtract
    const double *Lc = L + (long)s * ldn * ldn + (long)(k + 1) * NB * ldn + k * NB;  // L[k+1:, k]
    double *Ys = Y + (long)s * ldn * ldm + c * NB;
    f64x4 acc[4][4];
    zero_acc(acc);
    mma_tile<true, true>(acc, Lc, ldn, Ys + (long)(k + 1) * NB * ldm, ldm, (nb - 1 - k) * NB, smem,
                         smem + TILE_WORDS);
    double *Yo = Ys + (long)k * NB * ldm;
    IMCOM_FOR_ACC(row, col, v, { Yo[(long)row * ldm + col] -= v; })
}

// Y_k <- Linv[k] Y_k (TRANS=false) or Linv[k]^T Y_k (TRANS=true), in place.
template <bool TRANS>
__global__ __launch_bounds__(256, 2) void solve_dinv_kernel(const double *__restrict__ Dinv,
                                                            double *__restrict__ Y, int ldn, int ldm,
                                                            int k, const int *__restrict__ nblk)
{
    __shared__ double smem[2 * TILE_WORDS];
    const int s = blockIdx.y, c = blockIdx.x;
    if (k >= nblk[s]) return;
    const double *Di = Dinv + ((long)s * (ldn / NB) + k) * NB * NB;
    double *Yk = Y + (long)s * ldn * ldm + (long)k * NB * ldm + c * NB;
    f64x4 acc[4][4];
    zero_acc(acc);
    mma_tile<TRANS, true>(acc, Di, NB, Yk, ldm, NB, smem, smem + TILE_WORDS);
    __syncthreads();
    IMCOM_FOR_ACC(row, col, v, { Yk[(long)row * ldm + col] = v; })
}

// ---------------------------------------------------------------------------------------------
// Generic batched C = alpha * op(A) * op(B) + beta * C on 128-multiples (used by the eigen path:
// P = B Q, T = (P/(lam+kappa)) Q^T).  Element (r,k) of a K-major operand is p[k*ld + r].
template <bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const double *__restrict__ A, long lda,
                                                      long strideA, const double *__restrict__ B,
                                                      long ldb, long strideB, double *__restrict__ C,
                                                      long ldc, long strideC, int K, double alpha,
                                                      double beta)
{
    __shared__ double smem[2 * TILE_WORDS];
    const int s = blockIdx.z, tm = blockIdx.y, tn = blockIdx.x;
    const double *Ag = A + s * strideA + (AKM ? (long)tm * NB : (long)tm * NB * lda);
    const double *Bg = B + s * strideB + (BKM ? (long)tn * NB : (long)tn * NB * ldb);
    f64x4 acc[4][4];
    zero_acc(acc);
    mma_tile<AKM, BKM>(acc, Ag, lda, Bg, ldb, K, smem, smem + TILE_WORDS);
    double *Co = C + s * strideC + (long)tm * NB * ldc + (long)tn * NB;
    IMCOM_FOR_ACC(row, col, v, {
        double *p = Co + (long)row * ldc + col;
        *p = (beta == 0.0) ? alpha * v : alpha * v + beta * *p;
    })
}

// ---------------------------------------------------------------------------------------------
// host-side launchers

int launch_chol_update(imcom_ctx *ctx, const double *A, double *L, int ldn, int k, int nbmax, int batch,
                       const int *nblk, const double *dshift)
{
    dim3 grid(nbmax - k, batch);
    hipLaunchKernelGGL(chol_update_kernel, grid, dim3(256), 0, ctx->stream, A, L, ldn, k, nblk, dshift);
    return check_launch("chol_update_kernel");
}

int launch_chol_trsm(imcom_ctx *ctx, double *L, const double *Dinv, int ldn, int k, int nbmax, int batch,
                     const int *nblk)
{
    if (nbmax - k - 1 <= 0) return IMCOM_OK;
    dim3 grid(nbmax - k - 1, batch);
    hipLaunchKernelGGL(chol_trsm_kernel, grid, dim3(256), 0, ctx->stream, L, Dinv, ldn, k, nblk);
    return check_launch("chol_trsm_kernel");
}

int launch_solve_fwd(imcom_ctx *ctx, const double *L, const double *Bt, double *Y, int ldn, int ldm, int k,
                     int batch, const int *nblk)
{
    dim3 grid(ldm / NB, batch);
    hipLaunchKernelGGL(solve_fwd_kernel, grid, dim3(256), 0, ctx->stream, L, Bt, Y, ldn, ldm, k, nblk);
    return check_launch("solve_fwd_kernel");
}

int launch_solve_bwd(imcom_ctx *ctx, const double *L, double *Y, int ldn, int ldm, int k, int batch,
                     const int *nblk)
{
    dim3 grid(ldm / NB, batch);
    hipLaunchKernelGGL(solve_bwd_kernel, grid, dim3(256), 0, ctx->stream, L, Y, ldn, ldm, k, nblk);
    return check_launch("solve_bwd_kernel");
}

int launch_solve_dinv(imcom_ctx *ctx, const double *Dinv, double *Y, int ldn, int ldm, int k, int batch,
                      const int *nblk, bool trans)
{
    dim3 grid(ldm / NB, batch);
    if (trans)
        hipLaunchKernelGGL(solve_dinv_kernel<true>, grid, dim3(256), 0, ctx->stream, Dinv, Y, ldn, ldm, k, nblk);
    else
        hipLaunchKernelGGL(solve_dinv_kernel<false>, grid, dim3(256), 0, ctx->stream, Dinv, Y, ldn, ldm, k, nblk);
    return check_launch("solve_dinv_kernel");
}

int launch_gemm(imcom_ctx *ctx, bool akm, bool bkm, int M, int N, int K, int batch, const double *A, long lda,
                long strideA, const double *B, long ldb, long strideB, double *C, long ldc, long strideC,
                double alpha, double beta)
{
    IMCOM_REQUIRE(M % NB == 0 && N % NB == 0 && K % BK == 0, "launch_gemm: sizes must be padded (M=%d N=%d K=%d)", M, N, K);
    dim3 grid(N / NB, M / NB, batch);
#define IMCOM_GEMM_CASE(a, b)                                                                              \
    hipLaunchKernelGGL((gemm_kernel<a, b>), grid, dim3(256), 0, ctx->stream, A, lda, strideA, B, ldb, strideB, \
                       C, ldc, strideC, K, alpha, beta)
    if (akm && bkm) IMCOM_GEMM_CASE(true, true);
    else if (akm) IMCOM_GEMM_CASE(true, false);
    else if (bkm) IMCOM_GEMM_CASE(false, true);
    else IMCOM_GEMM_CASE(false, false);
#undef IMCOM_GEMM_CASE
    return check_launch("gemm_kernel");
}

}  // namespace imcom
