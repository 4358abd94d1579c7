// lmin_skinny.hip -- the Cholesky repair's smallest-eigenvalue iteration on blocks of 16 vectors.
//
// The reference's repair needs w[0] of eigh(A) and nothing else (reference src/pyimcom/lakernel.py:262-279); api.hip
// lambda_min_subspace finds it by inverse subspace iteration on a Cholesky factor of A + sigma I.  With the shift sigma a few
// per cent above |lambda_min| -- which is what the block-constant hint of a production block delivers -- the convergence per step is
// governed by (lambda_1 + sigma) / (lambda_{P+1} + sigma), and on a production stamp (configs/paper4: N = 6.2k, lambda_1 = -1.95e-6,
// lambda_17 = -1.2e-6, lambda_129 = -3.0e-7) a block of 16 vectors needs 7-9 steps where one of 128 needs 5-7.  The 128-column solves
// are bound by the matrix pipe (0.125 ms per stamp and step on the tile engine); 16 columns are one MFMA column group, the
// products cost nothing and a sweep is one pass over the factor: 154 MB per stamp, bound by HBM.  So a pass of >= 16 stamps runs the
// iteration on [ldn][16] blocks with the kernels of this file: ONE workgroup per stamp streams the stamp's factor, the rows of L go
// from global memory straight into MFMA A-fragments (every lane 32 contiguous bytes; a wave 16 rows x 1 KB per 128 columns), the
// block of vectors is the B operand.  No LDS staging of L: nothing of it is used twice.
//
//   skinny_solve_kernel   Y = (L L^T)^-1 X, both sweeps, in place in Y (block rows of 128 with the inverted diagonal blocks)
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

// The pieces of the sweeps' K loops, software-pipelined by hand: a wave keeps TWO pieces' loads in flight while it multiplies a third
// (left to itself the compiler interleaves loads and products and keeps 7-9 loads outstanding, draining them at the end of every
// iteration: 3.4 TB/s over the forward sweep).  The scheduling barriers pin "all loads of a piece, then all products of another";
// the waits the compiler inserts are then counted ones (the loads return in order).
#define SK_SB() __builtin_amdgcn_sched_barrier(0)

// forward piece: 16 columns k0 .. k0+15 of the wave's 16 rows (one 32-byte load per lane) and the 16 rows of the block of vectors
struct SkFwdPiece {
    f64x4v a[1];
    double b[1][4];
};
__device__ __forceinline__ void sk_fwd_issue(SkFwdPiece &P, const double *__restrict__ Mrow, const double *V)
{
    // Mrow = &M[r0 + r][k0 + 4 q], V = &V[k0 + 4 q][l & 15]
#pragma unroll
    for (int u = 0; u < 1; u++) P.a[u] = *(const f64x4v *)(Mrow + 16 * u);
#pragma unroll
    for (int u = 0; u < 1; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) P.b[u][j] = V[(long)(16 * u + j) * SP];
}
__device__ __forceinline__ void sk_fwd_compute(f64x4 &acc, const SkFwdPiece &P)
{
#pragma unroll
    for (int u = 0; u < 1; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc = SK_MFMA(P.a[u][j], P.b[u][j], acc);
}

// backward piece: D[v][c] += sum_k Z[k][v] M[k][c0 + c] over 16 rows k0 .. k0+15 of M and 32 of its columns -- the block of vectors
// is the A operand here (row v = vector), the matrix the B operand, and a lane loads TWO neighbouring columns (16 bytes; the 16
// lanes of a k: 256 contiguous bytes): acc0 takes the even columns c0 + 2 c, acc1 the odd ones.
typedef double f64x2v __attribute__((ext_vector_type(2)));
struct SkBwdPiece {
    f64x2v m[4];
    double z[4];
};
__device__ __forceinline__ void sk_bwd_issue(SkBwdPiece &P, const double *Zp, const double *__restrict__ Mp, long ld)
{
    // Zp = &Z[k0 + 4 q][l & 15], Mp = &M[k0 + 4 q][c0 + 2 (l & 15)]
#pragma unroll
    for (int j = 0; j < 4; j++) P.m[j] = *(const f64x2v *)(Mp + (long)j * ld);
#pragma unroll
    for (int j = 0; j < 4; j++) P.z[j] = Zp[(long)j * SP];
}
__device__ __forceinline__ void sk_bwd_compute(f64x4 &acc0, f64x4 &acc1, const SkBwdPiece &P)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        acc0 = SK_MFMA(P.z[j], P.m[j][0], acc0);
        acc1 = SK_MFMA(P.z[j], P.m[j][1], acc1);
    }
}

// Y = (L L^T)^-1 X for the stamps with nblk[s] > 0.  L: the lower factor in the stamp's [ldn][ldn] array, Dinv: the inverted diagonal
// blocks [ldn / 128][128][128] (lower triangular, exact zeros above the diagonal).  X may be Y.
// One workgroup of SIXTEEN waves per stamp (four per SIMD; the stamps of a pass are fewer than the CUs).  Forward, block row I: wave
// (g, h) adds up rows 16 g .. 16 g + 15 over the 128-column blocks kb = h, h + 2, ...; the two partial sums meet in LDS.  Backward,
// block column I: wave (cg, h) takes columns 32 cg .. 32 cg + 31 over the row blocks kb = I + 1 + h, + 4, ...
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
            const double *Vp = Ys + (long)(4 * q) * SP + r;
            const int np = I > h ? 8 * ((I - h + 1) / 2) : 0;  // pieces of 16 columns: blocks kb = h, h + 2, ... < I (a multiple of 4 pieces)
            if (np > 0) {
                auto k0 = [&](int p) -> long { p = p < np ? p : np - 1; return (long)(h + 2 * (p >> 3)) * NB + 16 * (p & 7); };  // (beyond the end: the last piece again, not used)
                SkFwdPiece buf[4];  // three pieces in flight under the products of the fourth
#pragma unroll
                for (int c = 0; c < 3; c++) sk_fwd_issue(buf[c], Lrow + k0(c), Vp + k0(c) * SP);
                for (int p = 0; p < np; p += 4) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        sk_fwd_issue(buf[(c + 3) & 3], Lrow + k0(p + c + 3), Vp + k0(p + c + 3) * SP);
                        SK_SB();
                        sk_fwd_compute(acc, buf[c]);
                        SK_SB();
                    }
                }
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
            const int nkb = nb - 1 - I - h, np = nkb > 0 ? 8 * ((nkb + 3) / 4) : 0;  // pieces of 16 rows: blocks kb = I + 1 + h, + 4, ... < nb
            if (np > 0) {
                auto k0 = [&](int p) -> long { p = p < np ? p : np - 1; return (long)(I + 1 + h + 4 * (p >> 3)) * NB + 16 * (p & 7); };
                SkBwdPiece buf[3];  // two pieces in flight under the products of the third
#pragma unroll
                for (int c = 0; c < 2; c++) sk_bwd_issue(buf[c], Zp + k0(c) * SP, Lp + k0(c) * ldn, ldn);
                for (int p = 0; p < np; p += 3) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        sk_bwd_issue(buf[(c + 2) % 3], Zp + k0(p + c + 2) * SP, Lp + k0(p + c + 2) * ldn, ldn);  // (always: the buffers carry no branch)
                        SK_SB();
                        if (p + c < np) sk_bwd_compute(acc0, acc1, buf[c]);
                        SK_SB();
                    }
                }
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
