// lmin_skinny.hip -- the Cholesky repair's smallest-eigenvalue iteration on blocks of 16 vectors.
//
// The reference's repair needs w[0] of eigh(A) and nothing else (reference src/pyimcom/lakernel.py:262-279); api.hip
// lambda_min_subspace finds it by inverse subspace iteration on a Cholesky factor of A + sigma I.  With the shift sigma a few
// per cent above |lambda_min| -- which is what the block-constant hint of a production block delivers -- the convergence per step is
// governed by (lambda_1 + sigma) / (lambda_{P+1} + sigma), and on a production stamp (configs/paper4: N = 6.2k, lambda_1 = -1.95e-6,
// lambda_17 = -1.2e-6, lambda_129 = -3.0e-7) a block of 16 vectors needs 7-9 steps where one of 128 needs 5-7.  The 128-column solves
// are bound by the matrix pipe (0.125 ms per stamp and step on the tile engine); 16 columns are one MFMA column group, the
// products cost nothing and a sweep is one pass over the factor: 154 MB per stamp, bound by HBM.  So the iteration runs on
// [ldn][16] blocks with the kernels of this file.  A pass of more than 128 stamps: ONE workgroup per stamp streams the stamp's
// factor (skinny_solve_kernel); fewer stamps: two short launches per block row, the sums dealt to many workgroups (skinny_part /
// skinny_fin).  In both the rows of L go from global memory straight into MFMA A-fragments (every lane 32 contiguous bytes; a wave
// 16 rows x 1 KB per 128 columns), the block of vectors is the B operand.  No LDS staging of L: nothing of it is used twice.
//
//   skinny_solve_kernel   Y = (L L^T)^-1 X, both sweeps, in place in Y (block rows of 128 with the inverted diagonal blocks)
//   skinny_part / _fin    the same, a block row per pair of launches (few stamps)
//   skinny_ax_kernel      Z = A X (a workgroup per block row: no dependence)
//   skinny_orth_kernel    X <- X R^-1 with R^T R = X^T X (one CholQR pass; two of them orthonormalise)
//   skinny_rr_kernel      H = X^T Z, its eigenvalues (Jacobi), the residuals of the two lowest Ritz pairs
#include "common.h"
#include "launchers.h"
#include "mma_dma.h"

namespace imcom {

constexpr int SP = LMIN_SKINNY_P;  // 16 columns
constexpr int SK_THREADS = 512, SK_WAVES = 8;

#define SK_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

typedef double f64x4v __attribute__((ext_vector_type(4)));

// acc(16 x 16) += M[rows r0 .. r0+15][k0 .. k0+16 NCH-1] * V[k0 .. k0+16 NCH-1][0..15]
// M row-major with leading dimension ld, V row-major [.][16].  A-fragment of MFMA j of chunk u: lane (r = l & 15, q = l >> 4) holds
// M[r0 + r][k0 + 16 u + 4 q + j] -- the k of a chunk are dealt so that a lane's four values are contiguous (one 32-byte load); the
// B-fragment holds V[k0 + 16 u + 4 q + j][l & 15], the same permutation of k.  All loads are issued before the first product.
template <int NCH>
__device__ __forceinline__ void sk_rows_times_block(f64x4 &acc, const double *__restrict__ Mrow, const double *V)
{
    // Mrow = &M[r0 + r][k0 + 4 q], V = &V[k0 + 4 q][l & 15]
    f64x4v a[NCH];
    double b[NCH][4];
#pragma unroll
    for (int u = 0; u < NCH; u++) a[u] = *(const f64x4v *)(Mrow + 16 * u);
#pragma unroll
    for (int u = 0; u < NCH; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) b[u][j] = V[(long)(16 * u + j) * SP];
#pragma unroll
    for (int u = 0; u < NCH; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc = SK_MFMA(a[u][j], b[u][j], acc);
}

// The transposed product of the backward sweep, D[v][c] += sum_k Z[k][v] M[k][c0 + c] over 32 rows k0 .. k0+31 of M and 32 of its
// columns: the block of vectors is the A operand here (row v = vector), the matrix the B operand, and a lane loads TWO neighbouring
// columns (16 bytes; the 16 lanes of a k: 256 contiguous bytes) -- acc0 takes the even columns c0 + 2 c, acc1 the odd ones.
__device__ __forceinline__ void sk_block_times_cols(f64x4 &acc0, f64x4 &acc1, const double *Zp, const double *__restrict__ Mp, long ld)
{
    // Zp = &Z[k0 + 4 q][l & 15], Mp = &M[k0 + 4 q][c0 + 2 (l & 15)]
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    f64x2v m[2][4];
    double z[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) m[u][j] = *(const f64x2v *)(Mp + (long)(16 * u + j) * ld);
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) z[u][j] = Zp[(long)(16 * u + j) * SP];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc0 = SK_MFMA(z[u][j], m[u][j][0], acc0);
            acc1 = SK_MFMA(z[u][j], m[u][j][1], acc1);
        }
}

// Y = (L L^T)^-1 X for the stamps with nblk[s] > 0.  L: the lower factor in the stamp's [ldn][ldn] array, Dinv: the inverted diagonal
// blocks [ldn / 128][128][128] (lower triangular, exact zeros above the diagonal).  X may be Y.
// One workgroup of SIXTEEN waves per stamp (four per SIMD: one wave's loads are in flight under another's products -- with eight
// waves the kernel streamed 3.6 TB/s, the stamps of a pass being fewer than the CUs).  Forward, block row I: wave (g, h) adds up rows
// 16 g .. 16 g + 15 over the 128-column blocks kb = h, h + 2, ...; the two partial sums meet in LDS.  Backward, block column I: wave
// (cg, h) takes columns 32 cg .. 32 cg + 31 over the row blocks kb = I + 1 + h, + 4, ...
constexpr int SKS_THREADS = 1024;
__global__ __launch_bounds__(SKS_THREADS) void skinny_solve_kernel(const double *__restrict__ L, const double *__restrict__ Dinv, const double *X, double *Y, int ldn,
                                                                   const int *__restrict__ nblk)
{
    __shared__ double Rp[5][NB * SP];  // partial sums [<= 4][row][vector]; Rp[4]: the block row's right-hand side
    double *Rs = Rp[4];
    const int s = blockIdx.x, nb = nblk[s];
    if (nb <= 0) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const double *Ls = L + (long)s * ldn * ldn, *Ds = Dinv + (long)s * (ldn / NB) * NB * NB;
    const double *Xs = X + (long)s * ldn * SP;
    double *Ys = Y + (long)s * ldn * SP;
    // forward: Y_I = Linv_I (X_I - sum_{K < I} L_IK Y_K)
    {
        const int g = w & 7, h = w >> 3;
        for (int I = 0; I < nb; I++) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            const double *Lrow = Ls + ((long)I * NB + 16 * g + r) * ldn + 4 * q;
            for (int kb = h; kb < I; kb += 2) {
                sk_rows_times_block<4>(acc, Lrow + (long)kb * NB, Ys + ((long)kb * NB + 4 * q) * SP + r);
                sk_rows_times_block<4>(acc, Lrow + (long)kb * NB + 64, Ys + ((long)kb * NB + 64 + 4 * q) * SP + r);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) Rp[h][(16 * g + q + 4 * t) * SP + r] = acc[t];
            __syncthreads();
            for (int e = threadIdx.x; e < NB * SP; e += SKS_THREADS) Rs[e] = Xs[(long)I * NB * SP + e] - (Rp[0][e] + Rp[1][e]);
            __syncthreads();
            if (h == 0) {
                f64x4 y = {0.0, 0.0, 0.0, 0.0};
                const double *Drow = Ds + (long)I * NB * NB + (long)(16 * g + r) * NB + 4 * q;
                for (int u = 0; u <= g; u++) {  // (lower triangular: chunks up to the diagonal one)
                    const f64x4v a = *(const f64x4v *)(Drow + 16 * u);
#pragma unroll
                    for (int j = 0; j < 4; j++) y = SK_MFMA(a[j], Rs[(16 * u + 4 * q + j) * SP + r], y);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) Ys[((long)I * NB + 16 * g + q + 4 * t) * SP + r] = y[t];
            }
            __syncthreads();  // Y_I is out for every wave of this workgroup (same CU: one L1), the LDS buffers are free
        }
    }
    // backward, in place: Z_I = Linv_I^T (Y_I - sum_{K > I} L_KI^T Z_K)
    {
        const int cg = w & 3, h = w >> 2, g = w & 7;
        for (int I = nb - 1; I >= 0; I--) {
            f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            const double *Lp = Ls + (long)(4 * q) * ldn + (long)I * NB + 32 * cg + 2 * r;
            const double *Zp = Ys + (long)(4 * q) * SP + r;
            for (int kb = I + 1 + h; kb < nb; kb += 4)
#pragma unroll 1
                for (int piece = 0; piece < 4; piece++) {
                    const long k0 = (long)kb * NB + 32 * piece;
                    sk_block_times_cols(acc0, acc1, Zp + k0 * SP, Lp + k0 * ldn, ldn);
                }
            // D[v][c]: register t of lane (r, q) holds vector v = q + 4 t, columns 32 cg + 2 r (+ 1)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                Rp[h][(32 * cg + 2 * r) * SP + q + 4 * t] = acc0[t];
                Rp[h][(32 * cg + 2 * r + 1) * SP + q + 4 * t] = acc1[t];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < NB * SP; e += SKS_THREADS) Rs[e] = Ys[(long)I * NB * SP + e] - ((Rp[0][e] + Rp[1][e]) + (Rp[2][e] + Rp[3][e]));
            __syncthreads();
            if (w < 8) {
                f64x4 z = {0.0, 0.0, 0.0, 0.0};
                const double *Dcol = Ds + (long)I * NB * NB + 16 * g + r + (long)(4 * q) * NB;
                for (int u = g; u < 8; u++) {  // (Linv^T is upper triangular: chunks from the diagonal one on)
#pragma unroll
                    for (int j = 0; j < 4; j++) z = SK_MFMA(Dcol[(long)(16 * u + j) * NB], Rs[(16 * u + 4 * q + j) * SP + r], z);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) Ys[((long)I * NB + 16 * g + q + 4 * t) * SP + r] = z[t];
            }
            __syncthreads();
        }
    }
}

// ---- few stamps (the kernel-class seam hands over one, a 2 x 2 group four): a workgroup per stamp would stream a production stamp's
// factor at one CU's rate (12 ms per step).  Here a block row is TWO launches: the sum over the blocks it depends on dealt to up to
// SK_PARTS workgroups per stamp (each leaves its 128 x 16 partial sum), then one workgroup per stamp that adds them in a fixed order,
// applies the inverted diagonal block and writes the block row.  196 short launches per solve of a production stamp -- as many as
// the right-looking 128-column form, but of 5-8 us instead of 30 (whose K = 128 tile products are latency-bound on the two-stage ring).
constexpr int SK_PARTS = 48;

__device__ __forceinline__ void sk_part_range(int nk, int p, int np, int &k0, int &k1)
{
    k0 = (int)((long)p * nk / np);
    k1 = (int)((long)(p + 1) * nk / np);
}

// partial[(s * SK_PARTS + p)][row][vector]: forward (BWD = false) rows of block row I over blocks [0, I); backward columns of block column I
// over the row blocks (I, nb)
template <bool BWD>
__global__ __launch_bounds__(BWD ? 256 : 512) void skinny_part_kernel(const double *__restrict__ L, const double *Y, double *__restrict__ partial, int ldn, int I,
                                                                        const int *__restrict__ nblk)
{
    const int s = blockIdx.y, p = blockIdx.x, nb = nblk[s];
    if (I >= nb) return;
    const int nk = BWD ? nb - 1 - I : I, np = nk < (int)gridDim.x ? nk : (int)gridDim.x;
    if (p >= np) return;
    int k0, k1;
    sk_part_range(nk, p, np, k0, k1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const double *Ls = L + (long)s * ldn * ldn, *Ys = Y + (long)s * ldn * SP;
    double *P = partial + ((long)s * SK_PARTS + p) * NB * SP;
    if (!BWD) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const double *Lrow = Ls + ((long)I * NB + 16 * w + r) * ldn + 4 * q;
        for (int kb = k0; kb < k1; kb++) sk_rows_times_block<8>(acc, Lrow + (long)kb * NB, Ys + ((long)kb * NB + 4 * q) * SP + r);
#pragma unroll
        for (int t = 0; t < 4; t++) P[(16 * w + q + 4 * t) * SP + r] = acc[t];
    } else {
        f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double *Lp = Ls + (long)(4 * q) * ldn + (long)I * NB + 32 * w + 2 * r;
        const double *Zp = Ys + (long)(4 * q) * SP + r;
        for (int kb = I + 1 + k0; kb < I + 1 + k1; kb++)
#pragma unroll
            for (int piece = 0; piece < 4; piece++) {
                const long k = (long)kb * NB + 32 * piece;
                sk_block_times_cols(acc0, acc1, Zp + k * SP, Lp + k * ldn, ldn);
            }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            P[(32 * w + 2 * r) * SP + q + 4 * t] = acc0[t];
            P[(32 * w + 2 * r + 1) * SP + q + 4 * t] = acc1[t];
        }
    }
}

// block row I of the sweep from its partial sums: Y_I = Linv_I (X_I - sum) forward, Y_I <- Linv_I^T (Y_I - sum) backward
template <bool BWD>
__global__ __launch_bounds__(SK_THREADS, 2) void skinny_fin_kernel(const double *__restrict__ Dinv, const double *X, double *Y, const double *__restrict__ partial,
                                                                   int ldn, int I, int nparts, const int *__restrict__ nblk)
{
    __shared__ double Rs[NB * SP];
    const int s = blockIdx.x, nb = nblk[s];
    if (I >= nb) return;
    const int nk = BWD ? nb - 1 - I : I, np = nk < nparts ? nk : nparts;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const double *Ds = Dinv + ((long)s * (ldn / NB) + I) * NB * NB;
    const double *Xi = (BWD ? Y : X) + (long)s * ldn * SP + (long)I * NB * SP;
    double *Yi = Y + (long)s * ldn * SP + (long)I * NB * SP;
    const double *P = partial + (long)s * SK_PARTS * NB * SP;
    // the inverted diagonal block's fragments first: their trip to HBM passes behind the sums (all eight chunks: exact zeros beyond the diagonal)
    double d[8][4];
    if (!BWD) {
        const double *Drow = Ds + (long)(16 * g + r) * NB + 4 * q;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const f64x4v a = *(const f64x4v *)(Drow + 16 * u);
#pragma unroll
            for (int j = 0; j < 4; j++) d[u][j] = a[j];
        }
    } else {
        const double *Dcol = Ds + 16 * g + r + (long)(4 * q) * NB;
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) d[u][j] = Dcol[(long)(16 * u + j) * NB];
    }
    // the partial sums in a fixed order, eight loads per element in flight (one after the other they cost an L2 round trip each: 30 us)
    {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < np; p0 += 8) {
            double t[8][4];
#pragma unroll
            for (int pp = 0; pp < 8; pp++)
#pragma unroll
                for (int c = 0; c < 4; c++) t[pp][c] = p0 + pp < np ? P[(long)(p0 + pp) * NB * SP + threadIdx.x + c * SK_THREADS] : 0.0;
#pragma unroll
            for (int pp = 0; pp < 8; pp++)
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] += t[pp][c];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) { const int e = threadIdx.x + c * SK_THREADS; Rs[e] = Xi[e] - v[c]; }
    }
    __syncthreads();
    f64x4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) y = SK_MFMA(d[u][j], Rs[(16 * u + 4 * q + j) * SP + r], y);
#pragma unroll
    for (int t = 0; t < 4; t++) Yi[(16 * g + q + 4 * t) * SP + r] = y[t];
}

// Z = A X, block row blockIdx.x of stamp blockIdx.y (A: the full symmetric matrix, identity-padded)
__global__ __launch_bounds__(SK_THREADS, 2) void skinny_ax_kernel(const double *__restrict__ A, const double *__restrict__ X, double *__restrict__ Z, int ldn,
                                                                  const int *__restrict__ nblk)
{
    const int s = blockIdx.y, I = blockIdx.x, nb = nblk[s];
    if (I >= nb) return;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const double *Arow = A + (long)s * ldn * ldn + ((long)I * NB + 16 * g + r) * ldn + 4 * q;
    const double *Xs = X + (long)s * ldn * SP;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < nb; kb++) sk_rows_times_block<8>(acc, Arow + (long)kb * NB, Xs + ((long)kb * NB + 4 * q) * SP + r);
    double *Zs = Z + (long)s * ldn * SP;
#pragma unroll
    for (int t = 0; t < 4; t++) Zs[((long)I * NB + 16 * g + q + 4 * t) * SP + r] = acc[t];
}

// sum over the rows [0, rows) of P[i][a] Q[i][c] -> G[a][c] (16 x 16) in LDS, all threads of the workgroup; P, Q row-major [.][16]
__device__ __forceinline__ void sk_gram(const double *__restrict__ P, const double *__restrict__ Q, int rows, double *Gpart /* [SK_WAVES][256] */, double *G /* [256] */)
{
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    // A(row a, k = i) = P[i][a], B(k = i, col c) = Q[i][c]: lane (l & 15, q) reads [i0 + 4 q' ...][l & 15] of either
    for (int i0 = 16 * g; i0 < rows; i0 += 16 * SK_WAVES) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long i = i0 + 4 * j + q;  // (any dealing of the 16 rows to the 4 x 4 (j, q) slots: the same for both operands)
            acc = SK_MFMA(P[i * SP + r], Q[i * SP + r], acc);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) Gpart[g * 256 + (q + 4 * t) * SP + r] = acc[t];
    __syncthreads();
    if (threadIdx.x < 256) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < SK_WAVES; w++) v += Gpart[w * 256 + threadIdx.x];  // fixed order
        G[threadIdx.x] = v;
    }
    __syncthreads();
}

// one CholQR pass: dst = src Linv^T with L L^T = src^T src.  fail[s] is set (never cleared) where the Gram matrix is not positive definite.
__global__ __launch_bounds__(SK_THREADS, 2) void skinny_orth_kernel(const double *__restrict__ src, double *__restrict__ dst, int ldn, const int *__restrict__ nblk,
                                                                    int *__restrict__ fail)
{
    __shared__ double Gpart[SK_WAVES * 256], G[256], Li[256];
    const int s = blockIdx.x, nb = nblk[s];
    if (nb <= 0) return;
    const int rows = nb * NB;
    const double *S = src + (long)s * ldn * SP;
    double *D = dst + (long)s * ldn * SP;
    sk_gram(S, S, rows, Gpart, G);
    if (threadIdx.x == 0) {
        // Cholesky factor of G (lower, in place), then its inverse Li (lower): 16 x 16, one lane
        bool ok = true;
        for (int j = 0; j < SP && ok; j++) {
            double d = G[j * SP + j];
            for (int k = 0; k < j; k++) d -= G[j * SP + k] * G[j * SP + k];
            if (!(d > 0.0) || !(d < 1e300)) { ok = false; break; }
            d = sqrt(d);
            G[j * SP + j] = d;
            for (int i = j + 1; i < SP; i++) {
                double v = G[i * SP + j];
                for (int k = 0; k < j; k++) v -= G[i * SP + k] * G[j * SP + k];
                G[i * SP + j] = v / d;
            }
        }
        for (int e = 0; e < 256; e++) Li[e] = 0.0;
        if (ok) {
            for (int c = 0; c < SP; c++) {  // column c of the inverse: forward substitution of e_c
                Li[c * SP + c] = 1.0 / G[c * SP + c];
                for (int i = c + 1; i < SP; i++) {
                    double v = 0.0;
                    for (int k = c; k < i; k++) v -= G[i * SP + k] * Li[k * SP + c];
                    Li[i * SP + c] = v / G[i * SP + i];
                }
            }
        } else {
            for (int c = 0; c < SP; c++) Li[c * SP + c] = 1.0;  // (the block goes on unchanged; the caller reads the flag)
            atomicExch(fail + s, 1);
        }
    }
    __syncthreads();
    // dst[i][c] = sum_j src[i][j] Li[c][j]: A(row i, k = j) = src[i][j] (a lane's four k contiguous), B(k = j, col c) = Li[c][j]
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    double b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = Li[r * SP + 4 * q + j];
    for (int i0 = 16 * g; i0 < rows; i0 += 16 * SK_WAVES) {
        const f64x4v a = *(const f64x4v *)(S + (long)(i0 + r) * SP + 4 * q);
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; j++) acc = SK_MFMA(a[j], b[j], acc);
#pragma unroll
        for (int t = 0; t < 4; t++) D[(long)(i0 + q + 4 * t) * SP + r] = acc[t];
    }
}

// Rayleigh-Ritz on the block: H = X^T Z (symmetrised), lam[s][0..15] its eigenvalues in ascending order (cyclic Jacobi in LDS:
// a 16 x 16 matrix), and the squared residual norms |Z y - theta X y|^2 of the two lowest Ritz pairs -> part[s][0][0..1] (the other
// groups of part[s] zero: the layout of launch_ritz_residual).
__global__ __launch_bounds__(SK_THREADS, 2) void skinny_rr_kernel(const double *__restrict__ X, const double *__restrict__ Z, int ldn, const int *__restrict__ nblk,
                                                                  double *__restrict__ lam, double *__restrict__ part, int ngroups)
{
    __shared__ double Gpart[SK_WAVES * 256], H[256], V[256], red[SK_WAVES][2], rsum[2 * SP], rot[8][2];
    __shared__ double th[2];
    __shared__ int order[SP], rpq[8][2];
    const int s = blockIdx.x, nb = nblk[s];
    if (nb <= 0) return;
    const int rows = nb * NB;
    const double *Xs = X + (long)s * ldn * SP, *Zs = Z + (long)s * ldn * SP;
    sk_gram(Xs, Zs, rows, Gpart, H);
    // symmetrise; V = I
    double hv = 0.0;
    const int hi = (threadIdx.x >> 4) & 15, hj = threadIdx.x & 15;
    if (threadIdx.x < 256) hv = 0.5 * (H[hi * SP + hj] + H[hj * SP + hi]);
    __syncthreads();
    if (threadIdx.x < 256) { H[threadIdx.x] = hv; V[threadIdx.x] = hi == hj ? 1.0 : 0.0; }
    __syncthreads();
    // cyclic Jacobi, the 120 pairs of a sweep as 15 rounds of 8 disjoint pairs (round-robin pairing), 128 threads per phase
    for (int sweep = 0; sweep < 30; sweep++) {
        if (threadIdx.x < SP) {
            const int i = threadIdx.x;
            double off = 0.0;
            for (int j = 0; j < SP; j++) off += j == i ? 0.0 : H[i * SP + j] * H[i * SP + j];
            rsum[i] = off;
            rsum[SP + i] = H[i * SP + i] * H[i * SP + i];
        }
        __syncthreads();
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < SP; i++) { off += rsum[i]; dg += rsum[SP + i]; }  // (every thread the same sums: a uniform decision)
        __syncthreads();
        if (!(off > 1e-34 * dg)) break;  // (also leaves on NaN)
        for (int rd = 0; rd < SP - 1; rd++) {
            if (threadIdx.x < 8) {
                const int pi = threadIdx.x;
                int p = pi == 0 ? SP - 1 : (rd + pi) % (SP - 1), qq = pi == 0 ? rd : (rd - pi + (SP - 1)) % (SP - 1);
                if (p > qq) { const int t_ = p; p = qq; qq = t_; }
                const double apq = H[p * SP + qq];
                double c = 1.0, sn = 0.0;
                if (apq != 0.0) {
                    const double tau = (H[qq * SP + qq] - H[p * SP + p]) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    sn = t * c;
                }
                rot[pi][0] = c; rot[pi][1] = sn; rpq[pi][0] = p; rpq[pi][1] = qq;
            }
            __syncthreads();
            if (threadIdx.x < 128) {  // columns: H <- H J, V <- V J
                const int pi = threadIdx.x >> 4, k = threadIdx.x & 15, p = rpq[pi][0], qq = rpq[pi][1];
                const double c = rot[pi][0], sn = rot[pi][1];
                const double hkp = H[k * SP + p], hkq = H[k * SP + qq], vkp = V[k * SP + p], vkq = V[k * SP + qq];
                H[k * SP + p] = c * hkp - sn * hkq;
                H[k * SP + qq] = sn * hkp + c * hkq;
                V[k * SP + p] = c * vkp - sn * vkq;
                V[k * SP + qq] = sn * vkp + c * vkq;
            }
            __syncthreads();
            if (threadIdx.x < 128) {  // rows: H <- J^T H
                const int pi = threadIdx.x >> 4, k = threadIdx.x & 15, p = rpq[pi][0], qq = rpq[pi][1];
                const double c = rot[pi][0], sn = rot[pi][1];
                const double hpk = H[p * SP + k], hqk = H[qq * SP + k];
                H[p * SP + k] = c * hpk - sn * hqk;
                H[qq * SP + k] = sn * hpk + c * hqk;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < SP; i++) order[i] = i;
        for (int i = 1; i < SP; i++) {  // ascending
            const int o = order[i];
            int j = i - 1;
            while (j >= 0 && H[order[j] * SP + order[j]] > H[o * SP + o]) { order[j + 1] = order[j]; j--; }
            order[j + 1] = o;
        }
        for (int i = 0; i < SP; i++) lam[(long)s * SP + i] = H[order[i] * SP + order[i]];
        th[0] = H[order[0] * SP + order[0]];
        th[1] = H[order[1] * SP + order[1]];
    }
    __syncthreads();
    // residuals of the two lowest pairs: every thread takes rows t, t + 512, ...
    double y0[SP], y1[SP];
#pragma unroll
    for (int c = 0; c < SP; c++) { y0[c] = V[c * SP + order[0]]; y1[c] = V[c * SP + order[1]]; }
    const double t0 = th[0], t1 = th[1];
    double a0 = 0.0, a1 = 0.0;
    for (int i = threadIdx.x; i < rows; i += SK_THREADS) {
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int c = 0; c < SP; c++) {
            const double x = Xs[(long)i * SP + c], z = Zs[(long)i * SP + c];
            d0 += (z - t0 * x) * y0[c];
            d1 += (z - t1 * x) * y1[c];
        }
        a0 += d0 * d0;
        a1 += d1 * d1;
    }
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a0; red[threadIdx.x >> 6][1] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = 0.0;
        for (int w = 0; w < SK_WAVES; w++) v += red[w][threadIdx.x];
        part[(long)s * ngroups * 2 + threadIdx.x] = v;
    }
    for (int e = 2 + threadIdx.x; e < ngroups * 2; e += SK_THREADS) part[(long)s * ngroups * 2 + e] = 0.0;
}

int launch_skinny_solve(imcom_ctx *ctx, const double *L, const double *Dinv, const double *X, double *Y, int ldn, const int *nblk, int batch)
{
    hipLaunchKernelGGL(skinny_solve_kernel, dim3(batch), dim3(SKS_THREADS), 0, ctx->stream, L, Dinv, X, Y, ldn, nblk);
    return check_launch("skinny_solve_kernel");
}

// few stamps: Y = (L L^T)^-1 X as a pair of launches per block row and sweep (nbmax: the largest nblk; partial: skinny_few_partial_doubles(batch) doubles)
int launch_skinny_solve_few(imcom_ctx *ctx, const double *L, const double *Dinv, const double *X, double *Y, int ldn, const int *nblk, int nbmax, int batch, double *partial)
{
    const int parts = nbmax - 1 < SK_PARTS ? nbmax - 1 : SK_PARTS;
    for (int I = 0; I < nbmax; I++) {
        if (I > 0) hipLaunchKernelGGL(skinny_part_kernel<false>, dim3(I < parts ? I : parts, batch), dim3(512), 0, ctx->stream, L, Y, partial, ldn, I, nblk);
        hipLaunchKernelGGL(skinny_fin_kernel<false>, dim3(batch), dim3(SK_THREADS), 0, ctx->stream, Dinv, X, Y, partial, ldn, I, parts, nblk);
    }
    IMCOM_TRY(check_launch("skinny_fin_kernel (forward)"));
    for (int I = nbmax - 1; I >= 0; I--) {
        // (a stamp with fewer blocks than nbmax: its kernels return for I >= nblk[s], and its own last block row has nothing below it)
        const int nk = nbmax - 1 - I;
        if (nk > 0) hipLaunchKernelGGL(skinny_part_kernel<true>, dim3(nk < parts ? nk : parts, batch), dim3(256), 0, ctx->stream, L, Y, partial, ldn, I, nblk);
        hipLaunchKernelGGL(skinny_fin_kernel<true>, dim3(batch), dim3(SK_THREADS), 0, ctx->stream, Dinv, Y, Y, partial, ldn, I, parts, nblk);
    }
    return check_launch("skinny_fin_kernel (backward)");
}

size_t skinny_few_partial_doubles(int batch) { return (size_t)batch * SK_PARTS * NB * SP; }

int launch_skinny_ax(imcom_ctx *ctx, const double *A, const double *X, double *Z, int ldn, const int *nblk, int nbmax, int batch)
{
    if (nbmax <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(skinny_ax_kernel, dim3(nbmax, batch), dim3(SK_THREADS), 0, ctx->stream, A, X, Z, ldn, nblk);
    return check_launch("skinny_ax_kernel");
}

int launch_skinny_orth(imcom_ctx *ctx, const double *src, double *dst, int ldn, const int *nblk, int *fail, int batch)
{
    hipLaunchKernelGGL(skinny_orth_kernel, dim3(batch), dim3(SK_THREADS), 0, ctx->stream, src, dst, ldn, nblk, fail);
    return check_launch("skinny_orth_kernel");
}

int launch_skinny_rr(imcom_ctx *ctx, const double *X, const double *Z, int ldn, const int *nblk, double *lam, double *part, int ngroups, int batch)
{
    hipLaunchKernelGGL(skinny_rr_kernel, dim3(batch), dim3(SK_THREADS), 0, ctx->stream, X, Z, ldn, nblk, lam, part, ngroups);
    return check_launch("skinny_rr_kernel");
}

}  // namespace imcom
