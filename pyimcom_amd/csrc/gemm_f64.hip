// gemm_f64.hip -- fp64 MFMA tile engine and the blocked Cholesky / triangular-solve kernels.
//
// Replaces scipy.linalg.cholesky + cho_solve as used by lakernel.CholKernel
// (reference src/pyimcom/lakernel.py:262-279, 295-304, 355-358) for a BATCH of postage stamps.
//
// Design (gfx950): one 512-thread workgroup (8 waves, 2x4) owns a 128x128 output tile; each wave accumulates a
// 64x32 sub-tile as 4x2 v_mfma_f64_16x16x4_f64 tiles (64 accumulator VGPRs, four waves per SIMD resident).
// Operand slices of k = 16 stream global -> LDS by LDS-DMA through a double buffer (mma_dma.h).
// fp64 MFMA runs at the fp64 vector rate on gfx950, so the point of MFMA here is operand reuse, not a higher peak.
#include <cstdlib>

#include "common.h"
#include "launchers.h"
#include "mma_dma.h"

namespace imcom {

constexpr int BK = DBK;  // K granularity of the tile engine (mma_dma.h)
// waves per SIMD a kernel is compiled for: two 8-wave workgroups per CU = 4 (the compiler keeps these kernels at 128 registers on
// its own; the attribute only has to allow it), or three 4-wave ones (64 x 64 per wave: 128 accumulator registers of 168)
constexpr int MMA_MINWAVES = MMA_WAVES == 4 ? 3 : 2;

__device__ __forceinline__ void zero_acc(f64x4 (&acc)[4][MMA_NJ])
{
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < MMA_NJ; j++) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
}

// C/D map of v_mfma_f64_16x16x4_f64: register r of lane l holds (row (l>>4) + 4r, col l&15).
#define IMCOM_FOR_ACC(ROW, COL, VAL, BODY)                                   \
    {                                                                        \
        const int lane__ = threadIdx.x & 63, wave__ = threadIdx.x >> 6;      \
        const int wm__ = wave__ / MMA_WN, wn__ = wave__ % MMA_WN;            \
        _Pragma("unroll") for (int i__ = 0; i__ < 4; i__++)                  \
        _Pragma("unroll") for (int j__ = 0; j__ < MMA_NJ; j__++)             \
        _Pragma("unroll") for (int r__ = 0; r__ < 4; r__++) {                \
            const int ROW = wm__ * 64 + i__ * 16 + (lane__ >> 4) + 4 * r__;  \
            const int COL = wn__ * (16 * MMA_NJ) + j__ * 16 + (lane__ & 15); \
            const double VAL = acc[i__][j__][r__];                           \
            BODY                                                             \
        }                                                                    \
    }

// accumulator map of the engine's TRI mode: the wave's row groups are 2 i + wm
#define IMCOM_FOR_ACC_TRI(ROW, COL, VAL, BODY)                               \
    {                                                                        \
        const int lane__ = threadIdx.x & 63, wave__ = threadIdx.x >> 6;      \
        const int wm__ = wave__ / MMA_WN, wn__ = wave__ % MMA_WN;            \
        _Pragma("unroll") for (int i__ = 0; i__ < 4; i__++)                  \
        _Pragma("unroll") for (int j__ = 0; j__ < MMA_NJ; j__++)             \
        _Pragma("unroll") for (int r__ = 0; r__ < 4; r__++) {                \
            const int ROW = (2 * i__ + wm__) * 16 + (lane__ >> 4) + 4 * r__; \
            const int COL = wn__ * (16 * MMA_NJ) + j__ * 16 + (lane__ & 15); \
            const double VAL = acc[i__][j__][r__];                           \
            BODY                                                             \
        }                                                                    \
    }

// Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own L2.  All column
// tiles of one stamp share that stamp's L panel, so they are placed on one XCD: XCD x takes stamps
// x, x+8, ...  Bijective when the batch is a multiple of 8 (else the plain mapping is used); placement
// only affects speed.
__device__ __forceinline__ void solve_tile_of_block(int &c, int &s)
{
    const int ntile = gridDim.x, batch = gridDim.y;
    c = blockIdx.x;
    s = blockIdx.y;
    if ((batch & 7) == 0) {
        const int b = blockIdx.y * ntile + blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        s = (slot / ntile) * 8 + xcd;
        c = slot % ntile;
    }
}

// ---------------------------------------------------------------------------------------------
// Left-looking blocked Cholesky of A + kappa*I, block column k:
//   P[i] = A[i,k] + kappa*delta - sum_{j<k} L[i,j] L[k,j]^T          (chol_update_kernel, i >= k)
//   L[k,k], Linv[k] from P[k]                                         (chol_diag.hip)
//   L[i,k] = P[i] Linv[k]^T                                           (chol_trsm_kernel, i > k)
// A is never written; L lives in its own buffer (the repair path and the multi-kappa path need A).
// Split-K for small batches (the kernel-class seam hands over ONE stamp per call: 18 workgroups per launch on 256 CUs).
// The K range of a tile is dealt to `nparts` workgroups that store their partial 128 x 128 products
// (partial[(stamp * gridDim.x + tile) * nparts + part][128][128], plain row-major); the tile's own kernel then adds them
// up in a fixed order instead of running the K loop, so the result does not depend on which part finished first.
__device__ __forceinline__ void store_partial(const f64x4 (&acc)[4][MMA_NJ], double *P)
{
    IMCOM_FOR_ACC(row, col, v, { P[row * NB + col] = v; })
}

__device__ __forceinline__ void sum_partials(f64x4 (&acc)[4][MMA_NJ], const double *P, int nparts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / MMA_WN, wn = wave % MMA_WN;
    for (int p = 0; p < nparts; p++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < MMA_NJ; j++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acc[i][j][r] += P[(long)p * NB * NB + (wm * 64 + i * 16 + (lane >> 4) + 4 * r) * NB + wn * (16 * MMA_NJ) + j * 16 + (lane & 15)];
}

// K blocks [k0, k1) of part `p` of `nparts` over `kb` blocks
__device__ __forceinline__ void part_range(int kb, int p, int nparts, int &k0, int &k1)
{
    k0 = (int)((long)p * kb / nparts);
    k1 = (int)((long)(p + 1) * kb / nparts);
}

__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void chol_partial_kernel(const double *__restrict__ L, int ldn, int k,
                                                              const int *__restrict__ nblk, int nparts, double *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    const int c = blockIdx.x, p = blockIdx.y, s = blockIdx.z, i = k + c;
    if (i >= nblk[s]) return;
    const double *Ls = L + (long)s * ldn * ldn;
    int k0, k1;
    part_range(k, p, nparts, k0, k1);
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    mma_tile_dma<false, false>(acc, Ls + (long)i * NB * ldn + k0 * NB, ldn, Ls + (long)k * NB * ldn + k0 * NB, ldn, (k1 - k0) * NB, smem);
    store_partial(acc, partial + (((long)s * gridDim.x + c) * nparts + p) * NB * NB);
}

__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void chol_update_kernel(const double *__restrict__ A,
                                                             double *__restrict__ L, int ldn, int k,
                                                             const int *__restrict__ nblk,
                                                             const double *__restrict__ dshift,
                                                             const double *__restrict__ partial, int nparts, int abatch)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    int s, c;
    solve_tile_of_block(c, s);  // the row tiles of a stamp share L[k,0:k]: one XCD
    const int i = k + c;
    if (i >= nblk[s]) return;
    const long sA = (long)ldn * ldn;
    double *Ls = L + s * sA;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    if (partial) sum_partials(acc, partial + ((long)s * gridDim.x + c) * nparts * NB * NB, nparts);
    else mma_tile_dma<false, false>(acc, Ls + (long)i * NB * ldn, ldn, Ls + (long)k * NB * ldn, ldn, k * NB, smem);
    const double *As = A + (s % abatch) * sA + (long)i * NB * ldn + k * NB;  // node-batched passes: stamp s of the pass reads A of stamp s mod abatch
    double *Lo = Ls + (long)i * NB * ldn + k * NB;
    // dshift = diagonal of A with the kappa increments already applied (diag_shift_kernel)
    const double *dsh = dshift + (long)s * ldn + i * NB;
    IMCOM_FOR_ACC(row, col, v, {
        double a = As[(long)row * ldn + col];
        if (i == k && row == col) a = dsh[row];
        Lo[(long)row * ldn + col] = a - v;
    })
}

__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void chol_trsm_kernel(double *__restrict__ L,
                                                           const double *__restrict__ Dinv, int ldn,
                                                           int k, const int *__restrict__ nblk)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    int s, c;
    solve_tile_of_block(c, s);
    const int i = k + 1 + c;
    if (i >= nblk[s]) return;
    double *P = L + (long)s * ldn * ldn + (long)i * NB * ldn + k * NB;
    const double *Di = Dinv + ((long)s * (ldn / NB) + k) * NB * NB;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    mma_tile_dma<false, false>(acc, P, ldn, Di, NB, NB, smem);
    __syncthreads();  // all of P[i] has been read by every wave before it is overwritten
    IMCOM_FOR_ACC(row, col, v, { P[(long)row * ldn + col] = v; })
}

// ---------------------------------------------------------------------------------------------
// Blocked triangular solves with m right-hand sides, input-pixel-major: L Y = Bt, L^T X = Y.
//   forward  block row k:  R = Bt_k - L[k,0:k] Y[0:k]   then  Y_k = Linv[k]   R
//   backward block row k:  R = Y_k - L[k+1:,k]^T X[k+1:] then  X_k = Linv[k]^T R
// The update kernels carry all the 2 N^2 m flops of the solve (the "solve_gemm" family).
// The diagonal block of a block row, applied by the workgroup that has just written the row's residual R (its own
// 128 x 128 tile of Y_k): Y_k = Linv[k] R (or Linv[k]^T R).  R comes back through the same L1 / L2 it was written to
// -- workgroup-scope visibility is all that is needed -- so the residual makes no extra trip to HBM and the row costs
// one launch instead of two.
template <bool TRANS>
__device__ __forceinline__ void solve_dinv_tile(f64x4 (&acc)[4][MMA_NJ], const double *__restrict__ Di, double *Yk, int ldm,
                                                double *smem)
{
    __syncthreads();  // R is written (the barrier waits for vmcnt(0)) and the LDS ring of the update is free
    zero_acc(acc);
    // Linv[k] is lower triangular (chol_diag.hip writes exact zeros above the diagonal), its transpose upper
    mma_tile_dma<TRANS, true, false, TRANS ? 2 : 1>(acc, Di, NB, Yk, ldm, NB, smem);
    __syncthreads();  // every wave has read all of R before it is overwritten
    IMCOM_FOR_ACC_TRI(row, col, v, { Yk[(long)row * ldm + col] = v; })
}

// partial products of one block row of the triangular solves (split-K, see chol_partial_kernel)
template <bool BWD>
__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void solve_partial_kernel(const double *__restrict__ L, const double *__restrict__ Y,
                                                               int ldn, int ldm, int k, const int *__restrict__ nblk,
                                                               const int *__restrict__ n, int nparts, double *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    const int c = blockIdx.x, p = blockIdx.y, s = blockIdx.z, nb = nblk[s];
    if (k >= nb) return;
    const double *Ys = Y + (long)s * ldn * ldm + c * NB;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    int k0, k1;
    if (!BWD) {
        part_range(k, p, nparts, k0, k1);
        const double *Ls = L + (long)s * ldn * ldn + (long)k * NB * ldn + k0 * NB;
        const int mrows = min(NB, n[s] - k * NB);
        if (mrows == NB) mma_tile_dma<false, true>(acc, Ls, ldn, Ys + (long)k0 * NB * ldm, ldm, (k1 - k0) * NB, smem);
        else mma_tile_dma<false, true, true>(acc, Ls, ldn, Ys + (long)k0 * NB * ldm, ldm, (k1 - k0) * NB, smem, mrows);
    } else {
        part_range(nb - 1 - k, p, nparts, k0, k1);
        const int kend = ((n[s] + DBK - 1) / DBK) * DBK - (k + 1) * NB;  // rows of L below n[s] are zero in this block column
        const int K = min(k1 * NB, kend) - k0 * NB;
        const double *Lc = L + (long)s * ldn * ldn + (long)(k + 1 + k0) * NB * ldn + k * NB;
        if (K > 0) mma_tile_dma<true, true>(acc, Lc, ldn, Ys + (long)(k + 1 + k0) * NB * ldm, ldm, K, smem);
    }
    store_partial(acc, partial + (((long)s * gridDim.x + c) * nparts + p) * NB * NB);
}

// Column sums of squares of the tile a workgroup has just finished (its accumulators, any row map): out[wm * stride + column].
// They let the single-kappa maps do without a pass over X and -B/2: D_a = sum_i B_ai T_ai = b^T (A + kappa)^-1 b = |L^-1 b|^2
// = sum_i Y_ia^2 (forward solve), N_a = sum_i X_ia^2 (backward solve); the two wave rows keep separate partial sums.
__device__ __forceinline__ void tile_col_sumsq(const f64x4 (&acc)[4][MMA_NJ], double *__restrict__ out, int stride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / MMA_WN, wn = wave % MMA_WN;
#pragma unroll
    for (int j = 0; j < MMA_NJ; j++) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) q += acc[i][j][r] * acc[i][j][r];
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 16) out[(long)wm * stride + wn * (16 * MMA_NJ) + j * 16 + lane] = q;
    }
}

// The coaddition's sums from the tile of T a workgroup has just finished (accumulators in the TRI row map): for every column a of
// the tile, sum over the tile's rows j of float(T[j][a]) w_q[j] -- w_q the indicator of exposure q, then the pixel values of input
// frame f -- per wave row, so that launch_coadd_from_partials can add the pieces in a fixed order and the pass of the stand-alone
// epilogue over T (20 MB per stamp) is not needed.  T enters as the float32 the reference holds (lakernel.py:96).
__device__ __forceinline__ void tile_coadd_partials(const f64x4 (&acc)[4][MMA_NJ], int s, int k, int c, int ldn, int ldm, int ns, const float *__restrict__ indata,
                                                    const int *__restrict__ expo, int n_inframe, int n_expo, double *__restrict__ Epart)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / MMA_WN, wn = wave % MMA_WN;
    const int nacc = n_expo + n_inframe;
    double *out = Epart + (((long)s * 2 * (ldn / NB) + 2 * k + wm) * nacc) * ldm + c * NB + wn * (16 * MMA_NJ) + (lane & 15);
    const int g0 = k * NB + wm * 16 + (lane >> 4);  // row of (i, r): g0 + 32 i + 4 r
    const int *ex = expo + (long)s * ldn;
    // (the engine keeps 64 accumulator registers per thread alive here and two workgroups per CU need the kernel below 128: the
    // passes re-read the rows' exposure / pixel value instead of holding them, and the loops over passes are not unrolled)
#pragma unroll 1
    for (int e0 = 0; e0 < n_expo; e0 += 2) {  // two exposures per pass over the accumulators
        double v0[MMA_NJ], v1[MMA_NJ];
#pragma unroll
        for (int j = 0; j < MMA_NJ; j++) v0[j] = v1[j] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int g = g0 + 32 * i + 4 * r;
                const int d = (g < ns ? ex[g] : -1) - e0;  // rows beyond the stamp's pixels: no exposure
#pragma unroll
                for (int j = 0; j < MMA_NJ; j++) {
                    const double t = (double)(float)acc[i][j][r];
                    v0[j] += d == 0 ? t : 0.0;
                    v1[j] += d == 1 ? t : 0.0;
                }
            }
#pragma unroll
        for (int j = 0; j < MMA_NJ; j++) {
            double t0 = v0[j], t1 = v1[j];
            t0 += __shfl_xor(t0, 16, 64); t1 += __shfl_xor(t1, 16, 64);
            t0 += __shfl_xor(t0, 32, 64); t1 += __shfl_xor(t1, 32, 64);
            if (lane < 16) {
                out[(long)e0 * ldm + j * 16] = t0;
                if (e0 + 1 < n_expo) out[(long)(e0 + 1) * ldm + j * 16] = t1;
            }
        }
    }
#pragma unroll 1
    for (int f = 0; f < n_inframe; f++) {
        double v[MMA_NJ];
#pragma unroll
        for (int j = 0; j < MMA_NJ; j++) v[j] = 0.0;
        const float *xin = indata + ((long)s * n_inframe + f) * ldn;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int g = g0 + 32 * i + 4 * r;
                const double x = g < ns ? (double)xin[g] : 0.0;
#pragma unroll
                for (int j = 0; j < MMA_NJ; j++) v[j] += (double)(float)acc[i][j][r] * x;
            }
#pragma unroll
        for (int j = 0; j < MMA_NJ; j++) {
            double t = v[j];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (lane < 16) out[(long)(n_expo + f) * ldm + j * 16] = t;
        }
    }
}

__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void solve_fwd_kernel(const double *__restrict__ L,
                                                           const double *__restrict__ Bt,
                                                           double *__restrict__ Y, int ldn, int ldm,
                                                           int k, const int *__restrict__ nblk, const int *__restrict__ n,
                                                           const double *__restrict__ Dinv,
                                                           const double *__restrict__ partial, int nparts, int bbatch,
                                                           double *__restrict__ Dpart)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    int s, c;
    solve_tile_of_block(c, s);
    if (k >= nblk[s]) return;
    const double *Ls = L + (long)s * ldn * ldn + (long)k * NB * ldn;
    double *Ys = Y + (long)s * ldn * ldm + c * NB;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    // rows of the last block row beyond n[s] are identity padding: their Y is Bt (zero), nothing has to be multiplied
    const int mrows = min(NB, n[s] - k * NB);
    if (partial) sum_partials(acc, partial + ((long)s * gridDim.x + c) * nparts * NB * NB, nparts);
    else if (mrows == NB) mma_tile_dma<false, true>(acc, Ls, ldn, Ys, ldm, k * NB, smem);
    else mma_tile_dma<false, true, true>(acc, Ls, ldn, Ys, ldm, k * NB, smem, mrows);
    const double *Bs = Bt + (long)(s % bbatch) * ldn * ldm + (long)k * NB * ldm + c * NB;  // node-batched passes share -B/2
    double *Yo = Ys + (long)k * NB * ldm;
    IMCOM_FOR_ACC(row, col, v, { Yo[(long)row * ldm + col] = Bs[(long)row * ldm + col] - v; })
    if (Dinv) {
        solve_dinv_tile<false>(acc, Dinv + ((long)s * (ldn / NB) + k) * NB * NB, Yo, ldm, smem);
        // partial D of this block row: Dpart[stamp][2 k + wm][ldm]
        if (Dpart) tile_col_sumsq(acc, Dpart + ((long)s * 2 * (ldn / NB) + 2 * k) * ldm + c * NB, ldm);
    }
}

// CO: the coaddition's sums taken from the finished tile of T (tile_coadd_partials; opt-in, IMCOM_EPILOGUE_FUSED): an instantiation of
// its own, bounded to four waves per SIMD, so that the default kernel's registers are exactly what they were without it
template <bool CO>
__global__ __launch_bounds__(MMA_THREADS, (CO && MMA_WAVES == 8) ? 4 : MMA_MINWAVES) void solve_bwd_kernel(const double *__restrict__ L,
                                                           double *__restrict__ Y, int ldn, int ldm,
                                                           int k, const int *__restrict__ nblk,
                                                           const int *__restrict__ n, const double *__restrict__ Dinv,
                                                           const double *__restrict__ partial, int nparts,
                                                           double *__restrict__ Npart, float *__restrict__ Tt,
                                                           const float *__restrict__ co_indata, const int *__restrict__ co_expo,
                                                           int co_n_inframe, int co_n_expo, double *__restrict__ co_Epart)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    int s, c;
    solve_tile_of_block(c, s);
    const int nb = nblk[s];
    if (k >= nb || (k == nb - 1 && !Dinv)) return;  // last block row: nothing to subtract
    double *Ys = Y + (long)s * ldn * ldm + c * NB;
    double *Yo = Ys + (long)k * NB * ldm;
    f64x4 acc[4][MMA_NJ];
    if (k < nb - 1) {
        const double *Lc = L + (long)s * ldn * ldn + (long)(k + 1) * NB * ldn + k * NB;  // L[k+1:, k]
        zero_acc(acc);
        // rows of L below n[s] are identity padding (zero in this block column): stop the k loop at n rounded to 8
        const int kend = ((n[s] + DBK - 1) / DBK) * DBK - (k + 1) * NB;
        if (partial) sum_partials(acc, partial + ((long)s * gridDim.x + c) * nparts * NB * NB, nparts);
        else mma_tile_dma<true, true>(acc, Lc, ldn, Ys + (long)(k + 1) * NB * ldm, ldm, kend, smem);
        IMCOM_FOR_ACC(row, col, v, { Yo[(long)row * ldm + col] -= v; })
    }
    if (Dinv) {
        solve_dinv_tile<true>(acc, Dinv + ((long)s * (ldn / NB) + k) * NB * NB, Yo, ldm, smem);
        if (Npart) {  // X_k is final: its float32 copy (T, input-pixel-major) and its share of N_a = sum_i T_ai^2
            tile_col_sumsq(acc, Npart + ((long)s * 2 * (ldn / NB) + 2 * k) * ldm + c * NB, ldm);
            float *To = Tt + (long)s * ldn * ldm + (long)k * NB * ldm + c * NB;
            IMCOM_FOR_ACC_TRI(row, col, v, { To[(long)row * ldm + col] = (float)v; })
            if constexpr (CO) tile_coadd_partials(acc, s, k, c, ldn, ldm, n[s], co_indata, co_expo, co_n_inframe, co_n_expo, co_Epart);
        }
    }
}

// Y_k <- Linv[k] Y_k (TRANS=false) or Linv[k]^T Y_k (TRANS=true), in place.
template <bool TRANS>
__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void solve_dinv_kernel(const double *__restrict__ Dinv,
                                                            double *__restrict__ Y, int ldn, int ldm,
                                                            int k, const int *__restrict__ nblk)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    int s, c;
    solve_tile_of_block(c, s);
    if (k >= nblk[s]) return;
    const double *Di = Dinv + ((long)s * (ldn / NB) + k) * NB * NB;
    double *Yk = Y + (long)s * ldn * ldm + (long)k * NB * ldm + c * NB;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    mma_tile_dma<TRANS, true>(acc, Di, NB, Yk, ldm, NB, smem);
    __syncthreads();
    IMCOM_FOR_ACC(row, col, v, { Yk[(long)row * ldm + col] = v; })
}

// ---------------------------------------------------------------------------------------------
// Generic batched C = alpha * op(A) * op(B) + beta * C on 128-multiples (used by the eigen path:
// P = B Q, T = (P/(lam+kappa)) Q^T).  Element (r,k) of a K-major operand is p[k*ld + r].
template <bool AKM, bool BKM>
__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void gemm_kernel(const double *__restrict__ A, long lda,
                                                      long strideA, const double *__restrict__ B,
                                                      long ldb, long strideB, double *__restrict__ C,
                                                      long ldc, long strideC, int K, double alpha,
                                                      double beta)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    const int s = blockIdx.z, tm = blockIdx.y, tn = blockIdx.x;
    const double *Ag = A + s * strideA + (AKM ? (long)tm * NB : (long)tm * NB * lda);
    const double *Bg = B + s * strideB + (BKM ? (long)tn * NB : (long)tn * NB * ldb);
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    mma_tile_dma<AKM, BKM>(acc, Ag, lda, Bg, ldb, K, smem);
    double *Co = C + s * strideC + (long)tm * NB * ldc + (long)tn * NB;
    IMCOM_FOR_ACC(row, col, v, {
        double *p = Co + (long)row * ldc + col;
        *p = (beta == 0.0) ? alpha * v : alpha * v + beta * *p;
    })
}

// Symmetric rank-2K update of the LOWER 128-tiles only: C[i][j] += alpha sum_k (V[k][i] W[k][j] + W[k][i] V[k][j]) for the tiles
// with tile column <= tile row (diagonal tiles whole).  Both products go through the same accumulators; the band reduction's
// trailing matrix is only ever read from its lower triangle, so half the tiles of a square update are enough.
__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void syr2k_lower_kernel(const double *__restrict__ V, long ldv, long strideV,
                                                                                const double *__restrict__ W, long ldw, long strideW,
                                                                                double *__restrict__ C, long ldc, long strideC, int K, double alpha)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    const int s = blockIdx.y, t = blockIdx.x;
    int tm = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
    while (tm * (tm + 1) / 2 > t) tm--;
    const int tn = t - tm * (tm + 1) / 2;
    const double *Vs = V + s * strideV, *Ws = W + s * strideW;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    mma_tile_dma<true, true>(acc, Vs + (long)tm * NB, ldv, Ws + (long)tn * NB, ldw, K, smem);
    mma_tile_dma<true, true>(acc, Ws + (long)tm * NB, ldw, Vs + (long)tn * NB, ldv, K, smem);
    double *Co = C + s * strideC + (long)tm * NB * ldc + (long)tn * NB;
    IMCOM_FOR_ACC(row, col, v, {
        double *p = Co + (long)row * ldc + col;
        *p += alpha * v;
    })
}

int launch_syr2k_lower(imcom_ctx *ctx, int N, int K, int batch, const double *V, long ldv, long strideV, const double *W, long ldw, long strideW,
                       double *C, long ldc, long strideC, double alpha)
{
    IMCOM_REQUIRE(N % NB == 0 && K % BK == 0 && N > 0, "launch_syr2k_lower: sizes must be padded (N=%d K=%d)", N, K);
    const int nt = N / NB;
    hipLaunchKernelGGL(syr2k_lower_kernel, dim3(nt * (nt + 1) / 2, batch), dim3(MMA_THREADS), 0, ctx->stream, V, ldv, strideV, W, ldw, strideW, C, ldc,
                       strideC, K, alpha);
    return check_launch("syr2k_lower_kernel");
}

// diagnostic: the k-major x k-major product with parts of the k loop taken out (mma_tile_dma's ABL); the result is not a product
template <int ABL>
__global__ __launch_bounds__(MMA_THREADS, MMA_MINWAVES) void gemm_abl_kernel(const double *__restrict__ A, long lda, long strideA,
                                                                             const double *__restrict__ B, long ldb, long strideB,
                                                                             double *__restrict__ C, long ldc, long strideC, int K)
{
    __shared__ __attribute__((aligned(16))) double smem[DMA_LDS_DOUBLES];
    const int s = blockIdx.z, tm = blockIdx.y, tn = blockIdx.x;
    f64x4 acc[4][MMA_NJ];
    zero_acc(acc);
    mma_tile_dma<true, true, false, 0, ABL>(acc, A + s * strideA + (long)tm * NB, lda, B + s * strideB + (long)tn * NB, ldb, K, smem);
    double *Co = C + s * strideC + (long)tm * NB * ldc + (long)tn * NB;
    IMCOM_FOR_ACC(row, col, v, { Co[(long)row * ldc + col] = v; })
}

int launch_gemm_abl(imcom_ctx *ctx, int abl, int M, int N, int K, int batch, const double *A, const double *B, double *C)
{
    dim3 grid(N / NB, M / NB, batch);
#define IMCOM_ABL_CASE(a) hipLaunchKernelGGL(gemm_abl_kernel<a>, grid, dim3(MMA_THREADS), 0, ctx->stream, A, (long)M, (long)M * K, B, (long)N, (long)K * N, C, (long)N, (long)M * N, K)
    if (abl == 1) IMCOM_ABL_CASE(1);
    else if (abl == 2) IMCOM_ABL_CASE(2);
    else if (abl == 4) IMCOM_ABL_CASE(4);
    else IMCOM_ABL_CASE(3);
#undef IMCOM_ABL_CASE
    return check_launch("gemm_abl_kernel");
}

// ---------------------------------------------------------------------------------------------
// Rate probe (imcom_ctx_mfma_probe): the fp64 MFMA pipe of every SIMD kept busy by four waves with eight independent
// accumulators each, and nothing else -- no LDS, no memory, no barriers.  What it reaches (77.5 TFLOP/s measured, 98.6 % of the
// guide's 78.6) is the ceiling a kernel with operand traffic can be compared with.
template <int MODE>
__global__ __launch_bounds__(MMA_THREADS, 2) void mfma_probe_kernel(int iters, double seed, double *__restrict__ sink)
{
    // MODE 0: constant operands (the ceiling quoted in DESIGN.md).  MODE 1: four A and two B fragments with full pseudo-random
    // mantissas, different in every lane (the switching activity of real data; same instruction stream otherwise).
    // MODE 2: as 1, but the six fragments are re-read from LDS in front of every eight MFMAs (three 16-byte ds_reads per lane, as
    // the tile engine's k-quad does; no DMA, no barriers).
    __shared__ double frag[MODE == 2 ? 6 * 64 * MMA_WAVES : 1];
    f64x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f64x4{seed, 0.0, 0.0, (double)i};
    double a[4], b[2];
    if (MODE == 0) {
        a[0] = a[1] = a[2] = a[3] = 1.0 + seed * (double)(threadIdx.x & 3);
        b[0] = b[1] = 1.0 - seed;
    } else {
        unsigned long h = 0x9E3779B97F4A7C15ul * (threadIdx.x + 1 + 977ul * blockIdx.x);
        auto rnd = [&]() { h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ul; h ^= h >> 32; return 0.5 + (double)(h >> 11) * (1.0 / 9007199254740992.0); };
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = rnd() * 1e-3;
        b[0] = rnd() * 1e-3; b[1] = -rnd() * 1e-3;
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) frag[(i * MMA_WAVES * 64) + threadIdx.x] = a[i];
            frag[4 * MMA_WAVES * 64 + threadIdx.x] = b[0];
            frag[5 * MMA_WAVES * 64 + threadIdx.x] = b[1];
            __syncthreads();
        }
    }
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
            asm volatile("" ::: "memory");  // the reads stay inside the loop
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = frag[(i * MMA_WAVES * 64) + threadIdx.x];
            b[0] = frag[4 * MMA_WAVES * 64 + threadIdx.x];
            b[1] = frag[5 * MMA_WAVES * 64 + threadIdx.x];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i >> 1], b[i & 1], acc[i], 0, 0, 0);
    }
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) v += acc[i][0] + acc[i][3];
    if (v == 0.123456789) sink[0] = v;  // keeps the loop alive
}

int launch_mfma_probe(imcom_ctx *ctx, int nwg, int iters, double *sink, int *waves_per_wg)
{
    *waves_per_wg = MMA_WAVES;
    const char *e = getenv("IMCOM_MFMA_PROBE_MODE");  // diagnostic variants, see the kernel
    const int mode = e ? atoi(e) : 0;
    if (mode == 1) hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(nwg), dim3(MMA_THREADS), 0, ctx->stream, iters, 1e-9, sink);
    else if (mode == 2) hipLaunchKernelGGL(mfma_probe_kernel<2>, dim3(nwg), dim3(MMA_THREADS), 0, ctx->stream, iters, 1e-9, sink);
    else hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(nwg), dim3(MMA_THREADS), 0, ctx->stream, iters, 1e-9, sink);
    return check_launch("mfma_probe_kernel");
}

// ---------------------------------------------------------------------------------------------
// host-side launchers

int launch_chol_update(imcom_ctx *ctx, const double *A, double *L, int ldn, int k, int nbmax, int batch, int abatch,
                       const int *nblk, const double *dshift, double *partial, int nparts)
{
    dim3 grid(nbmax - k, batch);
    if (nparts > 1 && k >= nparts) {
        hipLaunchKernelGGL(chol_partial_kernel, dim3(nbmax - k, nparts, batch), dim3(MMA_THREADS), 0, ctx->stream, L, ldn, k, nblk, nparts, partial);
        IMCOM_TRY(check_launch("chol_partial_kernel"));
        hipLaunchKernelGGL(chol_update_kernel, grid, dim3(MMA_THREADS), 0, ctx->stream, A, L, ldn, k, nblk, dshift, partial, nparts, abatch);
    } else
        hipLaunchKernelGGL(chol_update_kernel, grid, dim3(MMA_THREADS), 0, ctx->stream, A, L, ldn, k, nblk, dshift, nullptr, 0, abatch);
    return check_launch("chol_update_kernel");
}

int launch_chol_trsm(imcom_ctx *ctx, double *L, const double *Dinv, int ldn, int k, int nbmax, int batch,
                     const int *nblk)
{
    if (nbmax - k - 1 <= 0) return IMCOM_OK;
    dim3 grid(nbmax - k - 1, batch);
    hipLaunchKernelGGL(chol_trsm_kernel, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Dinv, ldn, k, nblk);
    return check_launch("chol_trsm_kernel");
}

// nparts > 1: the K loop of the block row is dealt to nparts workgroups per tile first (split-K for small batches); kb = K blocks of the row
int launch_solve_fwd(imcom_ctx *ctx, const double *L, const double *Bt, double *Y, int ldn, int ldm, int k,
                     int batch, int bbatch, const int *nblk, const int *n, const double *Dinv, double *partial, int nparts, double *Dpart)
{
    dim3 grid(ldm / NB, batch);
    if (nparts > 1 && k >= nparts) {
        hipLaunchKernelGGL(solve_partial_kernel<false>, dim3(ldm / NB, nparts, batch), dim3(MMA_THREADS), 0, ctx->stream, L, Y, ldn, ldm, k, nblk, n, nparts, partial);
        IMCOM_TRY(check_launch("solve_partial_kernel"));
        hipLaunchKernelGGL(solve_fwd_kernel, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Bt, Y, ldn, ldm, k, nblk, n, Dinv, partial, nparts, bbatch, Dpart);
    } else
        hipLaunchKernelGGL(solve_fwd_kernel, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Bt, Y, ldn, ldm, k, nblk, n, Dinv, nullptr, 0, bbatch, Dpart);
    return check_launch("solve_fwd_kernel");
}

int launch_solve_bwd(imcom_ctx *ctx, const double *L, double *Y, int ldn, int ldm, int k, int nbmax, int batch,
                     const int *nblk, const int *n, const double *Dinv, double *partial, int nparts, double *Npart, float *Tt,
                     const CoaddFuse *cf)
{
    dim3 grid(ldm / NB, batch);
    const bool co = cf && cf->Epart && Npart && Dinv;  // (the sums are taken where the tile of T is final: the fused launches)
    const float *ci = co ? cf->indata : nullptr;
    const int *ce = co ? cf->expo : nullptr;
    const int cf_ = co ? cf->n_inframe : 0, cn = co ? cf->n_expo : 0;
    double *cp = co ? cf->Epart : nullptr;
    if (nparts > 1 && nbmax - 1 - k >= nparts) {
        hipLaunchKernelGGL(solve_partial_kernel<true>, dim3(ldm / NB, nparts, batch), dim3(MMA_THREADS), 0, ctx->stream, L, Y, ldn, ldm, k, nblk, n, nparts, partial);
        IMCOM_TRY(check_launch("solve_partial_kernel"));
        if (co) hipLaunchKernelGGL(solve_bwd_kernel<true>, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Y, ldn, ldm, k, nblk, n, Dinv, partial, nparts, Npart, Tt, ci, ce, cf_, cn, cp);
        else hipLaunchKernelGGL(solve_bwd_kernel<false>, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Y, ldn, ldm, k, nblk, n, Dinv, partial, nparts, Npart, Tt, ci, ce, cf_, cn, cp);
    } else if (co)
        hipLaunchKernelGGL(solve_bwd_kernel<true>, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Y, ldn, ldm, k, nblk, n, Dinv, nullptr, 0, Npart, Tt, ci, ce, cf_, cn, cp);
    else
        hipLaunchKernelGGL(solve_bwd_kernel<false>, grid, dim3(MMA_THREADS), 0, ctx->stream, L, Y, ldn, ldm, k, nblk, n, Dinv, nullptr, 0, Npart, Tt, ci, ce, cf_, cn, cp);
    return check_launch("solve_bwd_kernel");
}

int launch_solve_dinv(imcom_ctx *ctx, const double *Dinv, double *Y, int ldn, int ldm, int k, int batch,
                      const int *nblk, bool trans)
{
    dim3 grid(ldm / NB, batch);
    if (trans)
        hipLaunchKernelGGL(solve_dinv_kernel<true>, grid, dim3(MMA_THREADS), 0, ctx->stream, Dinv, Y, ldn, ldm, k, nblk);
    else
        hipLaunchKernelGGL(solve_dinv_kernel<false>, grid, dim3(MMA_THREADS), 0, ctx->stream, Dinv, Y, ldn, ldm, k, nblk);
    return check_launch("solve_dinv_kernel");
}

int launch_gemm(imcom_ctx *ctx, bool akm, bool bkm, int M, int N, int K, int batch, const double *A, long lda,
                long strideA, const double *B, long ldb, long strideB, double *C, long ldc, long strideC,
                double alpha, double beta)
{
    IMCOM_REQUIRE(M % NB == 0 && N % NB == 0 && K % BK == 0, "launch_gemm: sizes must be padded (M=%d N=%d K=%d)", M, N, K);
    dim3 grid(N / NB, M / NB, batch);
#define IMCOM_GEMM_CASE(a, b)                                                                              \
    hipLaunchKernelGGL((gemm_kernel<a, b>), grid, dim3(MMA_THREADS), 0, ctx->stream, A, lda, strideA, B, ldb, strideB, \
                       C, ldc, strideC, K, alpha, beta)
    if (akm && bkm) IMCOM_GEMM_CASE(true, true);
    else if (akm) IMCOM_GEMM_CASE(true, false);
    else if (bkm) IMCOM_GEMM_CASE(false, true);
    else IMCOM_GEMM_CASE(false, false);
#undef IMCOM_GEMM_CASE
    return check_launch("gemm_kernel");
}

// operands with full mantissas (a product of zeros draws half the power of one of real data and runs at a higher clock)
__global__ void probe_fill_kernel(double *__restrict__ p, long count, unsigned seed)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned long long x = (unsigned long long)i * 6364136223846793005ULL + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    p[i] = ((double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3;  // (-5e-4, 5e-4): no overflow over K <= 1e6
}

int launch_probe_fill(imcom_ctx *ctx, double *p, long count, unsigned seed)
{
    hipLaunchKernelGGL(probe_fill_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, p, count, seed);
    return check_launch("probe_fill_kernel");
}


}  // namespace imcom
