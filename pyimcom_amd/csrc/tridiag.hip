// tridiag.hip -- batched symmetric eigensolver: blocked Householder tridiagonalisation, implicit-shift QR on
// the tridiagonal with logged Givens rotations, and a register-window wavefront kernel that applies the
// logged rotations to the accumulated reflectors.  Replaces numpy.linalg.eigh at reference
// src/pyimcom/lakernel.py:162, 201 (EigenKernel) and 266 (Cholesky repair).
//
//   A = Qh T Qh^T      Qh = H_0 H_1 ... H_{n-3},  H_j = I - tau_j v_j v_j^T          (trd_* kernels + GEMM)
//   T = G Lambda G^T   G = product of the logged rotations                          (tql_chunk_kernel)
//   eigenvectors       rows of X = (Qh G)^T, built as X = Qh^T then rotated in place  (orgtr GEMMs + rot_apply_kernel)
//
// Layouts (per stamp, all [ld][ld] with ld a multiple of 128, zero padded beyond n):
//   At    working copy of A; after panel p the trailing block [pe:, pe:] holds the two-sided update
//   Vall  Vall[j][r] = component r of reflector j (v_j[j+1] = 1, zero above) -- a panel of 128 reflectors is a
//         k-major GEMM operand as it lies
//   X     X[k][i] = component i of eigenvector k
// The panel algorithm is the classic one (w_j = tau (A v - V W^T v - W V^T v), w += -1/2 tau (w.v) v, trailing
// update A -= V W^T + W V^T); what is specific here is the batching: every kernel runs one column step for the
// whole batch of stamps, the matrix-vector product is the only pass over the trailing matrix (HBM bound,
// 8 n^2 bytes per column), and ragged n is handled by per-stamp guards.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "launchers.h"

namespace imcom {

constexpr int TP = NB;          // reflectors per block reflector when Qh is formed
constexpr int TPL = 64;         // reflectors per tridiagonalisation panel (the V / W corrections read 2 k values per row)
constexpr int QRS = 16;         // QR sweeps per chunk = depth of the rotation wavefront
constexpr int ROTPAD = 4 * QRS; // identity margin of the rotation log on both sides
constexpr int QR_RING = 4;      // rotation logs in flight between the QR chain and the rotation kernels
constexpr int LARFT_LDS = TP * (TP + 1) * 8;  // trd_larft_kernel: T with padded rows

// ascending rank of every eigenvalue (ties by index), lam[rank] = value
__global__ void eig_rank_kernel(const double *__restrict__ lam_raw, int ld, const int *__restrict__ n,
                                   int *__restrict__ rank, double *__restrict__ lam, long ldlam)
{
    const int s = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = n[s];
    if (a >= ns) return;
    const double *l = lam_raw + (long)s * ld;
    const double mine = l[a];
    int r = 0;
    for (int b = 0; b < ns; b++) {
        const double o = l[b];
        r += (o < mine || (o == mine && b < a)) ? 1 : 0;
    }
    rank[(long)s * ld + a] = r;
    lam[s * ldlam + r] = mine;
}

// Q[i][rank[a]] = Vt[a][i] (eigenvectors in columns, ascending); Q is [ldq][ldq], zero outside n x n
__global__ __launch_bounds__(256) void eig_scatter_kernel(const double *__restrict__ Vt, int ld, const int *__restrict__ rank,
                                                             const int *__restrict__ n, double *__restrict__ Q, long ldq,
                                                             long strideQ)
{
    __shared__ double tile[32][33];
    __shared__ int rk[32];
    const int s = blockIdx.z, a0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ns = n[s];
    if (threadIdx.x < 32) rk[threadIdx.x] = (a0 + threadIdx.x < ns) ? rank[(long)s * ld + a0 + threadIdx.x] : -1;
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, i = i0 + tx;
        tile[r][tx] = (a < ns && i < ns) ? Vt[(long)s * ld * ld + (long)a * ld + i] : 0.0;
    }
    __syncthreads();
    // thread (ty, tx): row i = i0 + ty.., source column a = a0 + tx -> writes Q[i][rank[a]] (scattered within the row)
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r;
        if (i < ns && rk[tx] >= 0) Q[s * strideQ + (long)i * ldq + rk[tx]] = tile[tx][r];
    }
}

// Shared tail of both eigensolvers: ascending order + eigenvectors scattered into the columns of Q (Q may be
// null: eigenvalues only).  Vt rows are eigenvectors, lam_raw[s*ld + a] their eigenvalues.
int launch_eig_sort_scatter(imcom_ctx *ctx, const double *Vt, int ld, const double *lam_raw, const int *n_dev, int *rank,
                            double *lam, long ldlam, double *Q, long ldq, long strideQ, int batch)
{
    hipLaunchKernelGGL(eig_rank_kernel, dim3((ld + 255) / 256, batch), dim3(256), 0, ctx->stream, lam_raw, ld, n_dev, rank, lam, ldlam);
    if (Q) hipLaunchKernelGGL(eig_scatter_kernel, dim3(ld / 32, ld / 32, batch), dim3(256), 0, ctx->stream, Vt, ld, rank, n_dev, Q, ldq, strideQ);
    return check_launch("eig sort/scatter");
}


__device__ inline double block_sum_256(double v, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// At = A on the leading n x n (zero elsewhere), X = I
__global__ void trd_init_kernel(const double *__restrict__ A, long lda, long strideA, const int *__restrict__ n,
                                double *__restrict__ At, double *__restrict__ X, int ld)
{
    const int s = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const int ns = n[s];
    const long o = (long)s * ld * ld + (long)i * ld + j;
    At[o] = (i < ns && j < ns) ? A[s * strideA + (long)i * lda + j] : 0.0;
    if (X) X[o] = (i == j) ? 1.0 : 0.0;
}

// Column step, part 1.  (a) finish the previous reflector's w:  W[k-1] = w' + alpha v, alpha = -1/2 tau (w'.v);
// (b) current column of the implicitly updated matrix: u = A[:, j] - V W[j,:]^T - W V[j,:]^T on rows >= j,
// d_j = u_j, partial sums of u_r^2 (r >= j+2) for the Householder norm.
__global__ __launch_bounds__(256) void trd_column_kernel(const double *__restrict__ At, double *__restrict__ Vall,
                                                         double *__restrict__ Wp, const double *__restrict__ wprime,
                                                         double *__restrict__ ubuf, double *__restrict__ dvec,
                                                         const double *__restrict__ hd, const double *__restrict__ pdot,
                                                         double *__restrict__ pnorm, const int *__restrict__ n, int ld, int j,
                                                         int ps, int npart, int column_part)
{
    __shared__ double bw[TP], bv[TP], red[4];
    const int s = blockIdx.y, ns = n[s];
    if (j >= ns) return;
    const int k = j - ps;
    const long so = (long)s * ld * ld;
    const double *V = Vall + so + (long)ps * ld;
    double *W = Wp + (long)s * TPL * ld;
    const double *wp = wprime + (long)s * ld;
    const int r = blockIdx.x * 256 + threadIdx.x;
    double alpha = 0.0;
    if (k > 0) {
        double dsum = 0.0;
        for (int q = 0; q < npart; q++) dsum += pdot[s * npart + q];
        alpha = -0.5 * hd[s * 4] * dsum;
    }
    if ((int)threadIdx.x < k) {
        const int c = threadIdx.x;
        const double vj = V[(long)c * ld + j];
        bv[c] = vj;
        bw[c] = (c == k - 1) ? wp[j] + alpha * vj : W[(long)c * ld + j];
    }
    __syncthreads();
    const bool act = r >= j && r < ns;
    double u = 0.0;
    if (act) {
        double wlast = 0.0, vlast = 0.0;
        if (k > 0) {
            vlast = V[(long)(k - 1) * ld + r];
            wlast = wp[r] + alpha * vlast;
            W[(long)(k - 1) * ld + r] = wlast;
        }
        if (column_part) {
            u = At[so + (long)j * ld + r];  // row j read as column j (the trailing matrix is symmetric)
            for (int c = 0; c < k - 1; c++) u -= V[(long)c * ld + r] * bw[c] + W[(long)c * ld + r] * bv[c];
            if (k > 0) u -= vlast * bw[k - 1] + wlast * bv[k - 1];
            ubuf[(long)s * ld + r] = u;
            if (r == j) dvec[(long)s * ld + j] = u;
        }
    }
    if (column_part) {
        const double sq = block_sum_256((act && r >= j + 2) ? u * u : 0.0, red);
        if (threadIdx.x == 0) pnorm[s * npart + blockIdx.x] = sq;
    }
}

struct HouseScalars {
    double beta, tau, scale;
};

__device__ inline HouseScalars house_scalars(const double *pnorm, int npart, double a0)
{
    double xn2 = 0.0;
    for (int q = 0; q < npart; q++) xn2 += pnorm[q];
    HouseScalars h;
    if (xn2 == 0.0) { h.beta = a0; h.tau = 0.0; h.scale = 0.0; }
    else {
        h.beta = -copysign(sqrt(a0 * a0 + xn2), a0);
        h.tau = (h.beta - a0) / h.beta;
        h.scale = 1.0 / (a0 - h.beta);
    }
    return h;
}

// Column step, part 2.  v = e_{j+1} + scale * u_{>= j+2}.  Strip blocks: the one pass over the trailing matrix,
// reading only the lower triangle (plus diagonal blocks): a strip of 32 rows r accumulates its row sums over the
// columns c < rb+32 and, from the same loads, the transposed contributions sum_r A[r][c] u_r to the rows c < rb,
// which go to part[strip][c] and are added up by trd_w_kernel.  p = A[:, j+1] + scale * (row sums + partials).
// Dot blocks: (V^T v)_c and (W^T v)_c for the panel's earlier reflectors.
__global__ __launch_bounds__(256) void trd_symv_kernel(const double *__restrict__ At, const double *__restrict__ Vall,
                                                       const double *__restrict__ Wp, const double *__restrict__ ubuf,
                                                       const double *__restrict__ pnorm, double *__restrict__ pvec,
                                                       double *__restrict__ part, double *__restrict__ hd,
                                                       double *__restrict__ evec, double *__restrict__ tauvec,
                                                       double *__restrict__ wv, const int *__restrict__ n, int ld, int j, int ps,
                                                       int npart, int nrowtiles)
{
    __shared__ double y2s[2][4][128];
    const int s = blockIdx.y, ns = n[s];
    if (j + 1 >= ns) return;
    const double *u = ubuf + (long)s * ld;
    const HouseScalars h = house_scalars(pnorm + s * npart, npart, u[j + 1]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hd[s * 4] = h.tau;
        hd[s * 4 + 1] = h.scale;
        evec[(long)s * ld + j] = h.beta;
        tauvec[(long)s * ld + j] = h.tau;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long so = (long)s * ld * ld;
    if ((int)blockIdx.x < nrowtiles) {
        // longest strips (bottom of the matrix) first: they bound the launch's critical path
        const int strip = ((j + 1) >> 5) + (nrowtiles - 1 - (int)blockIdx.x), rb = strip * 32;
        if (rb >= ns) return;
        const int r0 = rb + wave * 8;  // this wave's 8 rows; all four waves walk the same column chunks
        double ur[8];                  // u-tilde of those rows (wave-uniform)
#pragma unroll
        for (int i = 0; i < 8; i++) ur[i] = (r0 + i >= j + 2 && r0 + i < ns) ? u[r0 + i] : 0.0;
        const double *A = At + so + (long)r0 * ld;
        double *part_s = part + ((long)s * (ld / 32) + strip) * ld;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int cend = rb + 32;
        int buf = 0;
        // chunks of 128 columns, the next chunk's loads in flight while this one is reduced (a strip at the bottom of
        // the matrix walks ~N/128 chunks one after the other: without the prefetch each costs a full memory latency)
        double2 an[8], un;
        auto fetch = [&](int c) {
            const int cc = c + 2 * lane;
            un = *(const double2 *)(u + cc);
            if (cc < j + 2 || cc >= cend) un.x = 0.0;
            if (cc + 1 < j + 2 || cc + 1 >= cend) un.y = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) an[i] = *(const double2 *)(A + (long)i * ld + cc);
        };
        const int cfirst = (j + 1) & ~127;
        fetch(cfirst);
        for (int c = cfirst; c < cend; c += 128, buf ^= 1) {
            double2 a[8];
            const double2 uu = un;
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = an[i];
            if (c + 128 < cend) fetch(c + 128);
            double y2x = 0.0, y2y = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i] += a[i].x * uu.x + a[i].y * uu.y;
                y2x += a[i].x * ur[i];
                y2y += a[i].y * ur[i];
            }
            if (c < rb) {  // columns strictly left of the diagonal block receive the transposed contributions
                *(double2 *)&y2s[buf][wave][2 * lane] = make_double2(y2x, y2y);
                __syncthreads();
                if (threadIdx.x < 128) {
                    const int t = threadIdx.x;
                    if (c + t < rb) part_s[c + t] = (y2s[buf][0][t] + y2s[buf][1][t]) + (y2s[buf][2][t] + y2s[buf][3][t]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r = r0 + i;
                if (r >= j + 1 && r < ns) pvec[(long)s * ld + r] = A[(long)i * ld + (j + 1)] + h.scale * acc[i];
            }
        }
    } else {
        const int wid = (blockIdx.x - nrowtiles) * 4 + wave, c = wid >> 1, which = wid & 1;
        if (c >= j - ps) return;
        const double *vec = (which ? Wp + (long)s * TPL * ld : Vall + so + (long)ps * ld) + (long)c * ld;
        double acc = 0.0;
        for (int r = j + 2 + lane; r < ns; r += 64) acc += vec[r] * u[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) wv[((long)s * 2 + which) * TP + c] = vec[j + 1] + h.scale * acc;
    }
}

// Column step, part 3.  p completed with the strips' partials; w' = tau (p - V (W^T v) - W (V^T v)) on rows >= j+1, v stored as reflector j, partial w'.v
__global__ __launch_bounds__(256) void trd_w_kernel(double *__restrict__ Vall, const double *__restrict__ Wp,
                                                    const double *__restrict__ ubuf, const double *__restrict__ pvec,
                                                    const double *__restrict__ hd, const double *__restrict__ wv,
                                                    const double *__restrict__ part, double *__restrict__ wprime,
                                                    double *__restrict__ pdot, const int *__restrict__ n, int ld, int j, int ps,
                                                    int npart)
{
    __shared__ double swv[TP], svv[TP], red[4];
    const int s = blockIdx.y, ns = n[s];
    if (j + 1 >= ns) return;
    const int k = j - ps;
    const double tau = hd[s * 4], scale = hd[s * 4 + 1];
    if ((int)threadIdx.x < k) {
        svv[threadIdx.x] = wv[((long)s * 2 + 0) * TP + threadIdx.x];
        swv[threadIdx.x] = wv[((long)s * 2 + 1) * TP + threadIdx.x];
    }
    __syncthreads();
    const long so = (long)s * ld * ld;
    double *V = Vall + so + (long)ps * ld;
    const double *W = Wp + (long)s * TPL * ld;
    const int r = blockIdx.x * 256 + threadIdx.x;
    double prod = 0.0;
    if (r >= j + 1 && r < ns) {
        const double v = (r == j + 1) ? 1.0 : ubuf[(long)s * ld + r] * scale;
        // transposed half of the symmetric product, one partial per strip below this row's strip.  Up to N/32 partials and 2 x 63
        // panel entries per thread: eight independent loads / accumulators per round (one dependent chain made this kernel a
        // chain of ~150 memory latencies, 26 us per column at N = 2.9k)
        double tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const double *pp = part + ((long)s * (ld / 32)) * ld + r;
        int strip = (r >> 5) + 1;
        const int slast = (ns - 1) >> 5;
        for (; strip + 7 <= slast; strip += 8) {
#pragma unroll
            for (int q = 0; q < 8; q++) tq[q] += pp[(long)(strip + q) * ld];
        }
        for (; strip <= slast; strip++) tq[0] += pp[(long)strip * ld];
        const double tr = ((tq[0] + tq[1]) + (tq[2] + tq[3])) + ((tq[4] + tq[5]) + (tq[6] + tq[7]));
        double aq[4] = {0, 0, 0, 0};
        int c = 0;
        for (; c + 3 < k; c += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) aq[q] += V[(long)(c + q) * ld + r] * swv[c + q] + W[(long)(c + q) * ld + r] * svv[c + q];
        }
        for (; c < k; c++) aq[0] += V[(long)c * ld + r] * swv[c] + W[(long)c * ld + r] * svv[c];
        const double acc = pvec[(long)s * ld + r] + scale * tr - ((aq[0] + aq[1]) + (aq[2] + aq[3]));
        const double w = tau * acc;
        wprime[(long)s * ld + r] = w;
        V[(long)k * ld + r] = v;
        prod = w * v;
    }
    const double t = block_sum_256(prod, red);
    if (threadIdx.x == 0) pdot[s * npart + blockIdx.x] = t;
}

static bool larft_serial()
{
    static const bool v = getenv("IMCOM_LARFT") && !strcmp(getenv("IMCOM_LARFT"), "serial");
    return v;
}

// Triangular factor of a panel's block reflector H_ps ... H_pe-1 = I - V T V^T (forward, column-wise):
// T[c][c] = tau_c, T[0:c, c] = -tau_c T[0:c, 0:c] (V^T v_c).  S = V^T V comes from a GEMM.  One block per stamp.
__global__ __launch_bounds__(128) void trd_larft_kernel(const double *__restrict__ S, const double *__restrict__ tauvec, int ld,
                                                        int ps, double *__restrict__ T)
{
    // T in LDS as [TP][TP + 1]: thread r walks row r, stride 129 doubles = conflict free; S[q][c] has the same address in every
    // lane (scalar loads).  Row r of T depends on row r alone, so the 128 column steps need no barrier.  (The first version
    // kept T unpadded and synchronised twice per step: 685 us per panel.)
    extern __shared__ double Ts[];
    const int s = blockIdx.x, r = threadIdx.x;
    const double *Ss = S + (long)s * TP * TP;
    for (int q = 0; q < TP; q++) Ts[r * (TP + 1) + q] = 0.0;
    for (int c = 0; c < TP; c++) {
        const int j = ps + c;
        const double tau = j < ld ? tauvec[(long)s * ld + j] : 0.0;
        double t = 0.0;
        if (r < c) {
            for (int q = 0; q < c; q++) t += Ts[r * (TP + 1) + q] * Ss[(long)q * TP + c];  // (T[r][q] = 0 for q < r: uniform trip count and addresses of S)
            t *= -tau;
        } else if (r == c) t = tau;
        if (r <= c) Ts[r * (TP + 1) + c] = t;  // row r of T depends on row r alone: no barrier in this loop
    }
    __syncthreads();
    for (int q = 0; q < TP; q++) T[(long)s * TP * TP + (long)q * TP + r] = Ts[q * (TP + 1) + r];
}

// T of one panel: the MFMA triangular inverse (chol_diag.hip); IMCOM_LARFT=serial keeps the column-by-column kernel above
static int launch_larft(imcom_ctx *ctx, const double *S, const double *tauvec, int ld, int ps, double *T, int batch)
{
    if (!larft_serial()) return launch_larft_inv(ctx, S, tauvec, ld, ps, T, batch);
    hipLaunchKernelGGL(trd_larft_kernel, dim3(batch), dim3(TP), LARFT_LDS, ctx->stream, S, tauvec, ld, ps, T);
    return check_launch("trd_larft_kernel");
}

// One Givens step of the implicit QR bulge chase at position k of the block [lo, hi].  Carried state: (x, z) the
// pair to be rotated, a1 = current d[k], b1 = current e[k].  Returns the rotation [c s; -s c].
struct QrCarry {
    double x, z, a1, b1;
};

__device__ inline double2 qr_step(double *d, double *e, int lo, int hi, int k, QrCarry &q)
{
    const double a2 = d[k + 1];
    const double en = (k < hi - 1) ? e[k + 1] : 0.0;
    const double h = q.x * q.x + q.z * q.z;
    double c = 1.0, sn = 0.0, r = 0.0;
    if (h > 0.0) {
        const double ir = rsqrt(h);
        c = q.x * ir;
        sn = q.z * ir;
        r = h * ir;
    }
    if (k > lo) e[k - 1] = r;
    const double cc = c * c, ss = sn * sn, csn = c * sn;
    d[k] = cc * q.a1 + 2.0 * csn * q.b1 + ss * a2;
    const double ek = csn * (a2 - q.a1) + (cc - ss) * q.b1;
    q.a1 = ss * q.a1 - 2.0 * csn * q.b1 + cc * a2;  // the new d[k+1]
    e[k] = ek;
    if (k == hi - 1) d[hi] = q.a1;
    q.x = ek;
    q.z = sn * en;
    q.b1 = c * en;
    return make_double2(c, sn);
}

// number of eigenvalues of the m x m tridiagonal (d, e) below x (Sturm sequence)
__device__ inline int sturm_count(const double *d, const double *e, int m, double x)
{
    double q = d[0] - x;
    int cnt = q < 0.0;
    for (int i = 1; i < m; i++) {
        if (q == 0.0) q = 1e-300;
        q = d[i] - x - e[i - 1] * e[i - 1] / q;
        cnt += q < 0.0;
    }
    return cnt;
}

// One chunk of QRS implicit-shift QR sweeps on the tridiagonal (d, e) of every stamp; one wave per stamp working
// out of LDS.  Two modes, chosen by the size of the bottom unreduced block [lo, hi]:
//   small block (<= QR_SMALL rows): Wilkinson-shift sweeps one after the other, continuing across deflations;
//   large block: the QRS sweeps of the chunk use the eigenvalues of the trailing QRS x QRS block as shifts (found by
//     bisection, one per lane) and run PIPELINED, lane sw chasing its bulge QR_LAG positions behind lane sw-1.
// Rotation k of sweep sw -- [c s; -s c] on rows/columns (k, k+1) -- is logged at cs[(s*QRS + sw)*ldr + ROTPAD + k];
// everything else in the log is the identity.
// state[s]: 0 = bottom of the active part (hi), 1 = done, 4 = sweeps so far, 5 = tolerance initialised.
// The log `cs` is one slot of a ring; info[s] = {LO, HI, dirty}: the rows this chunk touches, and the highest
// position of the slot that may hold a rotation from an earlier chunk (reset to the identity on entry).
constexpr int QRG = 2;             // rotation logs (groups of QRS sweeps) filled by one launch: QRG * QRS bulges in flight
constexpr int QRW = QRG * QRS;     // sweeps per launch
constexpr int QR_SMALL = 3 * QRW;
constexpr int QR_LAG = 3;

__global__ __launch_bounds__(64) void tql_chunk_kernel(double *__restrict__ dvec, double *__restrict__ evec,
                                                       double2 *__restrict__ cs, long slot_stride, int *__restrict__ state,
                                                       int *__restrict__ info, long info_stride, double *__restrict__ tolv,
                                                       const int *__restrict__ n, int ld, int ldr, int want_rot, int max_sweeps)
{
    extern __shared__ double sm[];
    const int s = blockIdx.x, ns = n[s], lane = threadIdx.x;
    int *st = state + s * 8;
    double *d = sm, *e = sm + ld;
    auto inf = [&](int g) { return info + g * info_stride + s * 3; };
    auto logrow = [&](int sw) { return cs + (sw / QRS) * slot_stride + ((long)s * QRS + sw % QRS) * ldr + ROTPAD; };
    if (st[1]) {  // finished earlier: nothing to apply from these slots
        if (lane < QRG) { inf(lane)[0] = 0; inf(lane)[1] = -1; }
        return;
    }
    double tmax = 0.0;
    for (int i = lane; i < ns; i += 64) {
        d[i] = dvec[(long)s * ld + i];
        e[i] = (i < ns - 1) ? evec[(long)s * ld + i] : 0.0;
        tmax = fmax(tmax, fmax(fabs(d[i]), fabs(e[i])));
    }
    const bool first = st[5] == 0;
    double tol;
    if (first) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tmax = fmax(tmax, __shfl_xor(tmax, off, 64));
        tol = 2.220446049250313e-16 * tmax;
    } else tol = tolv[s];
    if (want_rot) {
        for (int sw = 0; sw < QRW; sw++) {
            const int fill_hi = inf(sw / QRS)[2];
            double2 *row = logrow(sw);
            for (int i = lane; i <= fill_hi; i += 64) row[i] = make_double2(1.0, 0.0);
        }
    }
    __syncthreads();
    // every lane follows the same control flow (uniform values, LDS broadcast reads); lane 0 stores
    int hi = first ? ns - 1 : st[0];
    int LO[QRG], HI[QRG];
#pragma unroll
    for (int g = 0; g < QRG; g++) { LO[g] = ns; HI[g] = -1; }
    int sw = 0, total = first ? 0 : st[4];
    while (hi > 0 && sw < QRW) {
        if (fabs(e[hi - 1]) <= tol) {
            __syncthreads();
            if (lane == 0) e[hi - 1] = 0.0;
            hi--;
            continue;
        }
        int lo = hi - 1;
        while (lo > 0 && fabs(e[lo - 1]) > tol) lo--;
        __syncthreads();
        if (hi - lo + 1 <= QR_SMALL) {
            if (lane == 0) {
                const double dd = 0.5 * (d[hi - 1] - d[hi]), b = e[hi - 1];
                const double mu = d[hi] - b * b / (dd + copysign(sqrt(dd * dd + b * b), dd));
                QrCarry q{d[lo] - mu, e[lo], d[lo], e[lo]};
                double2 *row = logrow(sw);
                for (int k = lo; k < hi; k++) {
                    const double2 g = qr_step(d, e, lo, hi, k, q);
                    if (want_rot) row[k] = g;
                }
            }
#pragma unroll
            for (int g = 0; g < QRG; g++)
                if (sw / QRS == g) { LO[g] = min(LO[g], lo); HI[g] = max(HI[g], hi); }
            sw++;
            total++;
        } else {
            if (sw > 0) break;  // a large block starts its own launch (all QRW sweeps)
            // shifts: eigenvalue `lane` of the trailing QRW x QRW block, by bisection
            double mu = 0.0;
            if (lane < QRW) {
                const double *db = d + hi - QRW + 1, *eb = e + hi - QRW + 1;
                double gl = 1e300, gu = -1e300;
                for (int i = 0; i < QRW; i++) {
                    const double rad = (i > 0 ? fabs(eb[i - 1]) : 0.0) + (i < QRW - 1 ? fabs(eb[i]) : 0.0);
                    gl = fmin(gl, db[i] - rad);
                    gu = fmax(gu, db[i] + rad);
                }
                for (int it = 0; it < 56; it++) {
                    const double mid = 0.5 * (gl + gu);
                    if (sturm_count(db, eb, QRW, mid) > lane) gu = mid; else gl = mid;
                }
                mu = 0.5 * (gl + gu);
            }
            __syncthreads();
            QrCarry q{0.0, 0.0, 0.0, 0.0};
            double2 *row = logrow(lane < QRW ? lane : 0);
            const int nst = (hi - lo) + QR_LAG * (QRW - 1);
            for (int t = 0; t < nst; t++) {
                const int k = lo + t - QR_LAG * lane;
                if (lane < QRW && k >= lo && k < hi) {
                    if (k == lo) q = QrCarry{d[lo] - mu, e[lo], d[lo], e[lo]};
                    const double2 g = qr_step(d, e, lo, hi, k, q);
                    if (want_rot) row[k] = g;
                }
            }
#pragma unroll
            for (int g = 0; g < QRG; g++) { LO[g] = lo; HI[g] = hi; }
            sw = QRW;
            total += QRW;
        }
        __syncthreads();
    }
    // trailing deflations so that `done` is seen as early as possible
    while (hi > 0 && fabs(e[hi - 1]) <= tol) hi--;
    __syncthreads();
    if (lane == 0) {
        st[0] = hi;
        st[4] = total;
        st[5] = 1;
#pragma unroll
        for (int g = 0; g < QRG; g++) { inf(g)[0] = LO[g]; inf(g)[1] = HI[g]; inf(g)[2] = HI[g]; }
        tolv[s] = tol;
        if (hi <= 0) st[1] = 1;
        else if (total >= max_sweeps) st[1] = 2;
    }
    for (int i = lane; i < ns; i += 64) {
        dvec[(long)s * ld + i] = d[i];
        if (i < ns - 1) evec[(long)s * ld + i] = e[i];
    }
}

// Apply one chunk of logged rotations to the rows [LO, HI] of X.  One thread per column of X (coalesced over the
// component index), one wave per block; the thread keeps a sliding window of 2*S rows in registers and runs the S
// sweeps as a wavefront: at step t sweep sw applies its rotation at position t - 2 sw, so every row is loaded and
// stored once per chunk while 4 S multiply-adds are done on it.  Window slots are addressed statically (t mod 2S
// is a compile-time constant in the unrolled body).  Rows are prefetched ROT_PF steps ahead into a register ring;
// the rotation coefficients of a group of 2S steps are staged through LDS (double buffered) and read as broadcasts.
constexpr int ROT_PF = 8;

template <int S>
__global__ __launch_bounds__(64) void rot_apply_kernel(double *__restrict__ X, const double2 *__restrict__ cs,
                                                       const int *__restrict__ info, const int *__restrict__ n, int ld, int ldr)
{
    constexpr int W = 2 * S;
    static_assert(QR_RING % QRG == 0, "ring must hold whole launches");
    static_assert(W % ROT_PF == 0, "window must be a multiple of the prefetch ring");
    __shared__ double2 gl[2][W][S];
    const int s = blockIdx.y, lane = threadIdx.x;
    const int LO = info[s * 3], HI = info[s * 3 + 1];
    if (HI <= LO) return;
    const int ns = n[s];
    if ((int)blockIdx.x * 64 >= ns) return;
    // surplus lanes shadow the last column: same loads, same arithmetic, identical (benign) stores
    const int i = min((int)blockIdx.x * 64 + lane, ns - 1);
    double *x0 = X + (long)s * ld * ld + (long)LO * ld + i;
    // The unrolled body stays branch-free: the surplus steps of the last group store into a scratch row (X carries
    // one per stamp after the last matrix), the steps before row 0 is final store early values over row 0.
    const int sinkrow = ((int)gridDim.y - s) * ld + s - LO;
    const double2 *c0 = cs + (long)s * S * ldr + ROTPAD + LO;
    const int R = HI - LO, nsteps = R + 2 * S - 1;
    auto stage = [&](int buf, int t0) {  // lane-consecutive positions within one sweep's row: coalesced
#pragma unroll
        for (int q = 0; q < W * S / 64; q++) {
            const int idx = q * 64 + lane, sw = idx / W, u = idx % W;
            gl[buf][u][sw] = c0[(long)sw * ldr + (t0 + u - 2 * sw)];
        }
    };
    // rows beyond R are only ever touched by identity rotations and never stored: load row R in their place
    double w[W], ring[ROT_PF];
#pragma unroll
    for (int q = 0; q < W; q++) w[q] = 0.0;
    w[0] = x0[0];
#pragma unroll
    for (int q = 0; q < ROT_PF; q++) ring[q] = x0[(long)min(q + 1, R) * ld];
    stage(0, 0);
    int buf = 0;
    // The S coefficients of a step are consumed in two halves; the half after the current one is read from LDS while
    // the current one is being applied (software pipeline through the fully unrolled body, across the loop edge
    // too: the first half of the next group of steps lives in the other LDS buffer, staged a group ahead).
    constexpr int H = S / 2;
    double2 gc[H], gn[H];
#pragma unroll
    for (int q = 0; q < H; q++) gc[q] = gl[0][0][q];
    for (int t0 = 0; t0 < nsteps; t0 += W, buf ^= 1) {
        stage(buf ^ 1, min(t0 + W, nsteps));  // past the end: identity margin of the log
#pragma unroll
        for (int u = 0; u < W; u++) {
            const int t = t0 + u;
            w[(u + 1) % W] = ring[u % ROT_PF];  // row t+1
            ring[u % ROT_PF] = x0[(long)min(t + 1 + ROT_PF, R) * ld];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // next half: second half of this step, or the first half of the next step (next group: other buffer)
#pragma unroll
                for (int q = 0; q < H; q++)
                    gn[q] = h == 0 ? gl[buf][u][H + q] : (u + 1 < W ? gl[buf][u + 1][q] : gl[buf ^ 1][0][q]);
#pragma unroll
                for (int q = 0; q < H; q++) {
                    const int sw = h * H + q;
                    const double2 g = gc[q];
                    const int a = ((u - 2 * sw) % W + W) % W, b = (a + 1) % W;
                    const double xa = w[a], xb = w[b];
                    w[a] = g.x * xa + g.y * xb;
                    w[b] = g.x * xb - g.y * xa;
                }
#pragma unroll
                for (int q = 0; q < H; q++) gc[q] = gn[q];
                __builtin_amdgcn_sched_barrier(0);  // keep the halves apart: hoisting more LDS reads spills registers
            }
            const int rs = t - 2 * S + 2;
            const int row = rs > R ? sinkrow : max(rs, 0);
            x0[(long)row * ld] = w[(u + 2) % W];
        }
    }
}

__global__ void tql_state_init_kernel(int *state, int batch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch * 8) state[i] = 0;
}

// rotation log = identity everywhere (the margins are never written again)
__global__ void rot_identity_kernel(double2 *cs, long count)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < count) cs[i] = make_double2(1.0, 0.0);
}

// -------------------------------------------------------------------------------------------------
size_t tridiag_ws_bytes(int batch, int ld, bool vectors);

// The tridiagonalisation proper: At (working copy of A, already initialised) -> d, e, reflectors Vall / tauvec.
struct TrdScratch {
    double *At, *Wp, *part, *ubuf, *pvec, *wprime, *wv, *hd, *pnorm, *pdot;
};

static size_t trd_scratch_bytes(int batch, int ld)
{
    const int npart = (ld + 255) / 256;
    size_t t = 0;
    auto add = [&](size_t b) { t = align_up(t, 256) + b; };
    add((size_t)batch * ld * ld * 8);                      // At
    add((size_t)batch * TPL * ld * 8);                     // Wp
    add((size_t)batch * (ld / 32) * ld * 8);               // per-strip partials of the symmetric product
    for (int q = 0; q < 3; q++) add((size_t)batch * ld * 8);  // u, p, w'
    add((size_t)batch * 2 * TP * 8);                       // V^T v, W^T v
    add((size_t)batch * 4 * 8);                            // hd
    add((size_t)batch * npart * 8 * 2);                    // pnorm, pdot
    return t + 4096;
}

static bool trd_take_scratch(imcom_ctx *ctx, int batch, int ld, TrdScratch *t)
{
    const int npart = (ld + 255) / 256;
    const size_t vecb = (size_t)batch * ld * 8;
    t->At = (double *)ws_take(ctx, (size_t)batch * ld * ld * 8);
    t->Wp = (double *)ws_take(ctx, (size_t)batch * TPL * ld * 8);
    t->part = (double *)ws_take(ctx, (size_t)batch * (ld / 32) * ld * 8);
    t->ubuf = (double *)ws_take(ctx, vecb);
    t->pvec = (double *)ws_take(ctx, vecb);
    t->wprime = (double *)ws_take(ctx, vecb);
    t->wv = (double *)ws_take(ctx, (size_t)batch * 2 * TP * 8);
    t->hd = (double *)ws_take(ctx, (size_t)batch * 4 * 8);
    t->pnorm = (double *)ws_take(ctx, (size_t)batch * npart * 8);
    t->pdot = (double *)ws_take(ctx, (size_t)batch * npart * 8);
    return t->At && t->Wp && t->part && t->ubuf && t->pvec && t->wprime && t->wv && t->hd && t->pnorm && t->pdot;
}

// A (device, lda / strideA) -> At, then the column steps.  Vall, dvec, evec, tauvec [batch][ld(x ld)] are outputs; X
// (optional) is set to the identity by the same initialisation kernel.
static int trd_reduce(imcom_ctx *ctx, int batch, int nmax, int ld, const double *A, long lda, long strideA, const int *n_dev,
                      const TrdScratch &t, double *Vall, double *dvec, double *evec, double *tauvec, double *X)
{
    const int npart = (ld + 255) / 256;
    const size_t mat = (size_t)batch * ld * ld * 8, vecb = (size_t)batch * ld * 8;
    hipStream_t st = ctx->stream;
    double *At = t.At, *Wp = t.Wp, *part = t.part, *ubuf = t.ubuf, *pvec = t.pvec, *wprime = t.wprime, *wv = t.wv, *hd = t.hd,
           *pnorm = t.pnorm, *pdot = t.pdot;
    IMCOM_HIP_CHECK(hipMemsetAsync(Vall, 0, mat, st));
    for (double *v : {ubuf, pvec, wprime, dvec, evec, tauvec}) IMCOM_HIP_CHECK(hipMemsetAsync(v, 0, vecb, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(hd, 0, (size_t)batch * 32, st));
    hipLaunchKernelGGL(trd_init_kernel, dim3((ld + 255) / 256, ld, batch), dim3(256), 0, st, A, lda, strideA, n_dev, At, X, ld);
    IMCOM_TRY(check_launch("trd_init_kernel"));
    ProfScope ps_(ctx, "eigen_trd", nmax);
    for (int ps = 0; ps < nmax; ps += TPL) {
        const int pe = std::min(ps + TPL, nmax);
        IMCOM_HIP_CHECK(hipMemsetAsync(Wp, 0, (size_t)batch * TPL * ld * 8, st));
        for (int j = ps; j < pe; j++) {
            hipLaunchKernelGGL(trd_column_kernel, dim3(npart, batch), dim3(256), 0, st, At, Vall, Wp, wprime, ubuf, dvec, hd, pdot, pnorm,
                               n_dev, ld, j, ps, npart, 1);
            if (j + 1 >= nmax) break;
            const int nrowtiles = ((nmax - 1) >> 5) - ((j + 1) >> 5) + 1, ndot = (2 * (j - ps) + 3) / 4;
            hipLaunchKernelGGL(trd_symv_kernel, dim3(nrowtiles + ndot, batch), dim3(256), 0, st, At, Vall, Wp, ubuf, pnorm, pvec, part, hd,
                               evec, tauvec, wv, n_dev, ld, j, ps, npart, nrowtiles);
            hipLaunchKernelGGL(trd_w_kernel, dim3(npart, batch), dim3(256), 0, st, Vall, Wp, ubuf, pvec, hd, wv, part, wprime, pdot, n_dev, ld,
                               j, ps, npart);
        }
        IMCOM_TRY(check_launch("trd column step"));
        if (ps + TPL < nmax) {  // trailing two-sided update A[pe:, pe:] -= V W^T + W V^T
            const int pe2 = ps + TPL;
            hipLaunchKernelGGL(trd_column_kernel, dim3(npart, batch), dim3(256), 0, st, At, Vall, Wp, wprime, ubuf, dvec, hd, pdot, pnorm,
                               n_dev, ld, pe2, ps, npart, 0);
            // The GEMM tiles are 128-aligned: start at the tile boundary at or below pe.  The extra rows/columns
            // it touches are already reduced and only ever read again under a zero multiplier.
            const int pa = pe2 / NB * NB, rem = ld - pa;
            const double *Vp = Vall + (long)ps * ld + pa, *Wq = Wp + pa;
            double *C = At + (long)pa * ld + pa;
            IMCOM_TRY(launch_gemm(ctx, true, true, rem, rem, TPL, batch, Vp, ld, (long)ld * ld, Wq, ld, (long)TPL * ld, C, ld, (long)ld * ld, -1.0, 1.0));
            IMCOM_TRY(launch_gemm(ctx, true, true, rem, rem, TPL, batch, Wq, ld, (long)TPL * ld, Vp, ld, (long)ld * ld, C, ld, (long)ld * ld, -1.0, 1.0));
        }
    }
    return IMCOM_OK;
}

// ---- the tridiagonal basis for callers that never need eigenvectors (eigen.hip): A = Qh T Qh^T with T = (d, e) and Qh
// kept as its reflectors; trd_apply_q multiplies a tall matrix by Qh or Qh^T panel by panel (compact WY, three GEMMs each).
size_t trd_basis_ws_bytes(int batch, int ld, int mp)
{
    const int npanels = (ld + TP - 1) / TP;
    size_t t = 0;
    auto add = [&](size_t b) { t = align_up(t, 256) + b; };
    add((size_t)batch * ld * ld * 8);                      // Vall
    for (int q = 0; q < 3; q++) add((size_t)batch * ld * 8);  // d, e, tau
    add((size_t)batch * 4);                                // n
    add((size_t)batch * npanels * TP * TP * 8);            // T factors of all panels
    add((size_t)batch * TP * TP * 8);                      // S = V V^T of one panel
    add((size_t)batch * TP * mp * 8 * 2);                  // W1, W2
    return t + 4096 + trd_scratch_bytes(batch, ld);
}

int trd_panel_factors(imcom_ctx *ctx, TrdBasis *out, int batch);

int trd_basis_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, int mp, const double *A, long lda, long strideA, TrdBasis *out)
{
    IMCOM_REQUIRE(ld % NB == 0 && ld >= NB && mp % NB == 0, "tridiag: ld=%d, mp=%d must be multiples of %d", ld, mp, NB);
    const size_t mat = (size_t)batch * ld * ld * 8, vecb = (size_t)batch * ld * 8;
    const int npanels = (ld + TP - 1) / TP;
    out->Vall = (double *)ws_take(ctx, mat);
    out->dvec = (double *)ws_take(ctx, vecb);
    out->evec = (double *)ws_take(ctx, vecb);
    out->tauvec = (double *)ws_take(ctx, vecb);
    out->n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    out->Tm = (double *)ws_take(ctx, (size_t)batch * npanels * TP * TP * 8);
    out->Sm = (double *)ws_take(ctx, (size_t)batch * TP * TP * 8);
    out->W1 = (double *)ws_take(ctx, (size_t)batch * TP * mp * 8);
    out->W2 = (double *)ws_take(ctx, (size_t)batch * TP * mp * 8);
    out->ld = ld;
    out->nmax = 0;
    for (int s = 0; s < batch; s++) out->nmax = std::max(out->nmax, n_host[s]);
    out->npanels = (std::max(out->nmax - 2, 0) + TP - 1) / TP;
    if (!out->Vall || !out->dvec || !out->evec || !out->tauvec || !out->n_dev || !out->Tm || !out->Sm || !out->W1 || !out->W2) {
        set_error("internal: tridiag workspace");
        return IMCOM_ERR_NOMEM;
    }
    IMCOM_TRY(upload(ctx, out->n_dev, n_host, (size_t)batch));
    const size_t mark = ctx->ws_used;
    TrdScratch t;
    if (!trd_take_scratch(ctx, batch, ld, &t)) { set_error("internal: tridiag workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(trd_reduce(ctx, batch, out->nmax, ld, A, lda, strideA, out->n_dev, t, out->Vall, out->dvec, out->evec, out->tauvec, nullptr));
    ctx->ws_used = mark;  // the scratch is free again (everything queued so far runs before whatever reuses it, same stream)
    return trd_panel_factors(ctx, out, batch);
}

// the triangular factors of the 128-reflector panels of a basis (Vall, tauvec), once for both directions of trd_apply_q
int trd_panel_factors(imcom_ctx *ctx, TrdBasis *out, int batch)
{
    const int ld = out->ld;
    ProfScope ps_(ctx, "eigen_applyq");
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)trd_larft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LARFT_LDS));
    for (int p = 0; p < out->npanels; p++) {
        const int ps = p * TP, rem = ld - ps;
        const double *Vp = out->Vall + (long)ps * ld + ps;
        IMCOM_TRY(launch_gemm(ctx, false, false, TP, TP, rem, batch, Vp, ld, (long)ld * ld, Vp, ld, (long)ld * ld, out->Sm, TP, (long)TP * TP, 1.0, 0.0));
        IMCOM_TRY(launch_larft(ctx, out->Sm, out->tauvec, ld, ps, out->Tm + (size_t)p * batch * TP * TP, batch));
    }
    return trd_pair_factors(ctx, out, batch);
}

// One step of Qh^T C for a caller that overlaps it with the reduction: the triangular factor of panel p, then C <- (I - V T^T V^T) C
int trd_panel_step(imcom_ctx *ctx, const TrdBasis &b, int batch, int p, double *C, int mp)
{
    ProfScope ps_(ctx, "eigen_applyq");
    const int ld = b.ld, ps = p * TP, rem = ld - ps;
    const double *Vp = b.Vall + (long)ps * ld + ps;
    double *Tm = b.Tm + (size_t)p * batch * TP * TP;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)trd_larft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LARFT_LDS));
    IMCOM_TRY(launch_gemm(ctx, false, false, TP, TP, rem, batch, Vp, ld, (long)ld * ld, Vp, ld, (long)ld * ld, b.Sm, TP, (long)TP * TP, 1.0, 0.0));
    IMCOM_TRY(launch_larft(ctx, b.Sm, b.tauvec, ld, ps, Tm, batch));
    double *Cp = C + (long)ps * mp;
    IMCOM_TRY(launch_gemm(ctx, false, true, TP, mp, rem, batch, Vp, ld, (long)ld * ld, Cp, mp, (long)ld * mp, b.W1, mp, (long)TP * mp, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, true, true, TP, mp, TP, batch, Tm, TP, (long)TP * TP, b.W1, mp, (long)TP * mp, b.W2, mp, (long)TP * mp, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, true, true, rem, mp, TP, batch, Vp, ld, (long)ld * ld, b.W2, mp, (long)TP * mp, Cp, mp, (long)ld * mp, -1.0, 1.0));
    return IMCOM_OK;
}

// Two consecutive panels a, b as ONE block reflector of 256: H_a H_b = I - V T V^T with V = [V_a; V_b] (consecutive rows of Vall) and
// T = [[T_a, -T_a (V_a V_b^T) T_b], [0, T_b]].  The update C -= V^T W then has K = 256: half the trips of C through HBM per flop.
__global__ __launch_bounds__(2 * TP) void t2_assemble_kernel(const double *__restrict__ Ta, const double *__restrict__ Tb, double *__restrict__ T2)
{
    const int s = blockIdx.y, r = blockIdx.x, c = threadIdx.x;
    double v = 0.0;
    if (r < TP && c >= TP) return;  // the upper right block comes from the products
    if (r < TP) v = Ta[(long)s * TP * TP + r * TP + c];
    else if (c >= TP) v = Tb[(long)s * TP * TP + (r - TP) * TP + c - TP];
    T2[(long)s * 4 * TP * TP + r * 2 * TP + c] = v;
}

int trd_pair_factors(imcom_ctx *ctx, TrdBasis *out, int batch)
{
    if (!out->T2) return IMCOM_OK;
    ProfScope ps_(ctx, "eigen_applyq");
    const int ld = out->ld;
    const long tt = (long)TP * TP;
    for (int q = 0; 2 * q + 1 < out->npanels; q++) {
        const int psa = 2 * q * TP, psb = psa + TP, remb = ld - psb;
        const double *Va = out->Vall + (long)psa * ld + psb, *Vb = out->Vall + (long)psb * ld + psb;  // components psb.. of both
        const double *Ta = out->Tm + (size_t)(2 * q) * batch * tt, *Tb = Ta + (size_t)batch * tt;
        double *T2 = out->T2 + (size_t)q * batch * 4 * tt;
        IMCOM_TRY(launch_gemm(ctx, false, false, TP, TP, remb, batch, Va, ld, (long)ld * ld, Vb, ld, (long)ld * ld, out->Sm, TP, tt, 1.0, 0.0));  // V_a V_b^T
        IMCOM_TRY(launch_gemm(ctx, false, true, TP, TP, TP, batch, out->Sm, TP, tt, Tb, TP, tt, out->W1, TP, tt, 1.0, 0.0));                   // ... T_b
        IMCOM_TRY(launch_gemm(ctx, false, true, TP, TP, TP, batch, Ta, TP, tt, out->W1, TP, tt, T2 + TP, 2 * TP, 4 * tt, -1.0, 0.0));          // -T_a ...
        hipLaunchKernelGGL(t2_assemble_kernel, dim3(2 * TP, batch), dim3(2 * TP), 0, ctx->stream, Ta, Tb, T2);
        IMCOM_TRY(check_launch("t2_assemble_kernel"));
    }
    return IMCOM_OK;
}

// C [batch][ld][mp] <- Qh^T C (transpose) or Qh C: Qh = H_0 H_1 ... and a panel's H_ps ... H_pe-1 = I - V T V^T, so
//   Qh^T C: panels in ascending order, C[ps:] -= V^T (T^T (V C[ps:]));   Qh C: descending order, C[ps:] -= V^T (T (V C[ps:]))
// With pair factors (b.T2) two panels go as one of 256 reflectors; an odd last panel alone.
int trd_apply_q(imcom_ctx *ctx, const TrdBasis &b, int batch, double *C, int mp, bool transpose)
{
    ProfScope ps_(ctx, "eigen_applyq");
    const int ld = b.ld;
    auto apply = [&](int ps, int rows, const double *T) -> int {
        const int rem = ld - ps;
        const double *Vp = b.Vall + (long)ps * ld + ps;  // [rows][rem] with row stride ld: reflectors ps.., components ps..
        double *Cp = C + (long)ps * mp;
        IMCOM_TRY(launch_gemm(ctx, false, true, rows, mp, rem, batch, Vp, ld, (long)ld * ld, Cp, mp, (long)ld * mp, b.W1, mp, (long)rows * mp, 1.0, 0.0));
        IMCOM_TRY(launch_gemm(ctx, transpose, true, rows, mp, rows, batch, T, rows, (long)rows * rows, b.W1, mp, (long)rows * mp, b.W2, mp, (long)rows * mp, 1.0, 0.0));
        return launch_gemm(ctx, true, true, rem, mp, rows, batch, Vp, ld, (long)ld * ld, b.W2, mp, (long)rows * mp, Cp, mp, (long)ld * mp, -1.0, 1.0);
    };
    const int nunits = b.T2 ? (b.npanels + 1) / 2 : b.npanels;  // pairs (and an odd last panel), or single panels
    for (int q = 0; q < nunits; q++) {
        const int u = transpose ? q : nunits - 1 - q;
        if (b.T2 && 2 * u + 1 < b.npanels) IMCOM_TRY(apply(2 * u * TP, 2 * TP, b.T2 + (size_t)u * batch * 4 * TP * TP));
        else {
            const int p = b.T2 ? 2 * u : u;
            IMCOM_TRY(apply(p * TP, TP, b.Tm + (size_t)p * batch * TP * TP));
        }
    }
    return IMCOM_OK;
}

size_t tridiag_ws_bytes(int batch, int ld, bool vectors)
{
    const int ldr = ld + 2 * ROTPAD;
    size_t t = trd_scratch_bytes(batch, ld);
    auto add = [&](size_t b) { t = align_up(t, 256) + b; };
    add((size_t)batch * ld * ld * 8);                      // Vall
    if (vectors) add((size_t)batch * ld * ld * 8 + (size_t)batch * ld * 8);  // X + one scratch row per stamp
    if (vectors) { add((size_t)batch * ld * TP * 8); add((size_t)batch * ld * TP * 8); }  // W1, W2
    if (vectors) { add((size_t)batch * TP * TP * 8); add((size_t)batch * TP * TP * 8); }  // S, T
    for (int q = 0; q < 3; q++) add((size_t)batch * ld * 8);  // d, e, tau
    if (vectors) add((size_t)batch * QRS * ldr * 16 * QR_RING);  // rotation logs
    add((size_t)QR_RING * batch * 3 * 4);                  // per-slot {LO, HI, dirty}
    add((size_t)batch * 8 * 4 + (size_t)batch * 8 + (size_t)batch * 4);  // state, tol, n
    add((size_t)batch * ld * 4);                           // rank
    return t + 8192;
}

// Same contract as jacobi_eigh_device: A [batch] matrices (lda, strideA) on the device, n_host ragged, ld the
// padded size (multiple of 128).  lam[s*ldlam + k] ascending; Q[s*strideQ + i*ldq + k] eigenvectors in columns
// (Q == nullptr: eigenvalues only).
int tridiag_eigh_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, const double *A, long lda, long strideA,
                        double *lam, long ldlam, double *Q, long ldq, long strideQ, int *sweeps_out)
{
    IMCOM_REQUIRE(ld % NB == 0 && ld >= NB, "tridiag: ld=%d must be a multiple of %d", ld, NB);
    const bool vectors = Q != nullptr;
    const int ldr = ld + 2 * ROTPAD;
    const size_t mat = (size_t)batch * ld * ld * 8, vecb = (size_t)batch * ld * 8;
    TrdScratch t;
    const bool have_scratch = trd_take_scratch(ctx, batch, ld, &t);
    double *Vall = (double *)ws_take(ctx, mat);
    double *X = vectors ? (double *)ws_take(ctx, mat + (size_t)batch * ld * 8) : nullptr;
    double *W1 = nullptr, *W2 = nullptr, *Sm = nullptr, *Tm = nullptr;
    if (vectors) {
        W1 = (double *)ws_take(ctx, (size_t)batch * ld * TP * 8);
        W2 = (double *)ws_take(ctx, (size_t)batch * ld * TP * 8);
        Sm = (double *)ws_take(ctx, (size_t)batch * TP * TP * 8);
        Tm = (double *)ws_take(ctx, (size_t)batch * TP * TP * 8);
    }
    double *dvec = (double *)ws_take(ctx, vecb), *evec = (double *)ws_take(ctx, vecb), *tauvec = (double *)ws_take(ctx, vecb);
    double2 *cs = vectors ? (double2 *)ws_take(ctx, (size_t)batch * QRS * ldr * 16 * QR_RING) : nullptr;
    int *info = (int *)ws_take(ctx, (size_t)QR_RING * batch * 3 * 4);
    int *state = (int *)ws_take(ctx, (size_t)batch * 8 * 4);
    double *tolv = (double *)ws_take(ctx, (size_t)batch * 8);
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    int *rank = (int *)ws_take(ctx, (size_t)batch * ld * 4);
    if (!have_scratch || !Vall || (vectors && (!X || !W1 || !W2 || !Sm || !Tm || !cs)) || !dvec || !evec || !tauvec || !info || !state || !tolv ||
        !n_dev || !rank) {
        set_error("internal: tridiag workspace");
        return IMCOM_ERR_NOMEM;
    }
    int nmax = 0;
    for (int s = 0; s < batch; s++) nmax = std::max(nmax, n_host[s]);
    hipStream_t st = ctx->stream;
    IMCOM_HIP_CHECK(hipMemcpyAsync(n_dev, n_host, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    IMCOM_HIP_CHECK(hipStreamSynchronize(st));  // n_host may be a caller local
    IMCOM_TRY(trd_reduce(ctx, batch, nmax, ld, A, lda, strideA, n_dev, t, Vall, dvec, evec, tauvec, X));

    // ---- X = Qh^T: X <- X (I - V T^T V^T) panel by panel, last panel first
    if (vectors) {
        ProfScope ps_(ctx, "eigen_orgtr");
        IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)trd_larft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LARFT_LDS));
        const int npanels = (std::max(nmax - 2, 0) + TP - 1) / TP;
        for (int p = npanels - 1; p >= 0; p--) {
            const int ps = p * TP, rem = ld - ps;
            const double *Vp = Vall + (long)ps * ld + ps;  // rows (reflectors) ps.., components ps..
            double *Xb = X + (long)ps * ld + ps;
            IMCOM_TRY(launch_gemm(ctx, false, false, TP, TP, rem, batch, Vp, ld, (long)ld * ld, Vp, ld, (long)ld * ld, Sm, TP, (long)TP * TP, 1.0, 0.0));
            IMCOM_TRY(launch_larft(ctx, Sm, tauvec, ld, ps, Tm, batch));
            IMCOM_TRY(launch_gemm(ctx, false, false, rem, TP, rem, batch, Xb, ld, (long)ld * ld, Vp, ld, (long)ld * ld, W1, TP, (long)ld * TP, 1.0, 0.0));
            IMCOM_TRY(launch_gemm(ctx, false, false, rem, TP, TP, batch, W1, TP, (long)ld * TP, Tm, TP, (long)TP * TP, W2, TP, (long)ld * TP, 1.0, 0.0));
            IMCOM_TRY(launch_gemm(ctx, false, true, rem, rem, TP, batch, W2, TP, (long)ld * TP, Vp, ld, (long)ld * ld, Xb, ld, (long)ld * ld, -1.0, 1.0));
        }
    }

    // ---- implicit QR on the tridiagonal (a serial chain of one-wave kernels on the main stream), rotations applied
    // chunk by chunk on the second stream: the two chains overlap through a ring of QR_RING rotation logs.
    hipLaunchKernelGGL(tql_state_init_kernel, dim3((batch * 8 + 255) / 256), dim3(256), 0, st, state, batch);
    if (vectors) {
        const long count = (long)batch * QRS * ldr * QR_RING;
        hipLaunchKernelGGL(rot_identity_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, cs, count);
        IMCOM_HIP_CHECK(hipMemsetAsync(info, 0xff, (size_t)QR_RING * batch * 3 * 4, st));  // {LO, HI, dirty} = -1
    }
    IMCOM_TRY(check_launch("tql init"));
    const size_t qr_lds = (size_t)2 * ld * 8;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)tql_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qr_lds));
    while ((int)ctx->sync_events.size() < 2 * QR_RING + 1) {
        hipEvent_t e;
        IMCOM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_events.push_back(e);
    }
    hipEvent_t *ev_qr = ctx->sync_events.data(), *ev_ap = ev_qr + QR_RING, ev_x = ctx->sync_events[2 * QR_RING];
    IMCOM_TRY(ensure_aux(ctx));
    hipStream_t sb = ctx->aux_stream;
    // whatever path leaves this function (an IMCOM_TRY / IMCOM_HIP_CHECK early return included), nothing may still be
    // running on the second stream: the next API call hands the same workspace out again
    struct AuxDrain {
        hipStream_t s;
        bool armed = true;
        ~AuxDrain() { if (armed) hipStreamSynchronize(s); }
    } aux_drain{sb};
    if (vectors) {  // X (Qh^T) is ready when everything queued so far has run
        IMCOM_HIP_CHECK(hipEventRecord(ev_x, st));
        IMCOM_HIP_CHECK(hipStreamWaitEvent(sb, ev_x, 0));
    }
    const int max_sweeps = 30 * std::max(nmax, 1);
    const int max_chunks = max_sweeps / QRW + 2;
    std::vector<int> sth((size_t)batch * 8);
    int chunks = 0;
    bool done = nmax <= 1;
    {
        ProfScope ps_(ctx, "eigen_qr");
        while (!done && chunks < max_chunks) {
            const int group = 16;
            for (int g = 0; g < group; g++, chunks++) {
                const int slot0 = (chunks * QRG) % QR_RING;  // QRG consecutive slots per launch
                const long slot_stride = (long)batch * QRS * ldr, info_stride = (long)batch * 3;
                double2 *log = vectors ? cs + (size_t)slot0 * slot_stride : nullptr;
                int *inf = info + (size_t)slot0 * info_stride;
                if (vectors && chunks * QRG >= QR_RING)
                    for (int q = 0; q < QRG; q++) IMCOM_HIP_CHECK(hipStreamWaitEvent(st, ev_ap[slot0 + q], 0));  // slots free again
                hipLaunchKernelGGL(tql_chunk_kernel, dim3(batch), dim3(64), qr_lds, st, dvec, evec, log, slot_stride, state, inf, info_stride,
                                   tolv, n_dev, ld, ldr, vectors ? 1 : 0, max_sweeps);
                if (vectors) {
                    IMCOM_HIP_CHECK(hipEventRecord(ev_qr[slot0], st));
                    IMCOM_HIP_CHECK(hipStreamWaitEvent(sb, ev_qr[slot0], 0));
                    for (int q = 0; q < QRG; q++) {
                        hipLaunchKernelGGL(rot_apply_kernel<QRS>, dim3((nmax + 63) / 64, batch), dim3(64), 0, sb, X, log + q * slot_stride,
                                           inf + q * info_stride, n_dev, ld, ldr);
                        IMCOM_HIP_CHECK(hipEventRecord(ev_ap[slot0 + q], sb));
                    }
                }
            }
            IMCOM_TRY(check_launch("tql chunk"));
            IMCOM_HIP_CHECK(hipMemcpyAsync(sth.data(), state, sth.size() * 4, hipMemcpyDeviceToHost, st));
            IMCOM_HIP_CHECK(hipStreamSynchronize(st));
            done = true;
            for (int s = 0; s < batch; s++) {
                if (n_host[s] > 1 && sth[(size_t)s * 8 + 1] == 0) done = false;
                if (sth[(size_t)s * 8 + 1] == 2) {
                    hipStreamSynchronize(sb);  // nothing of this call may still be running when the workspace is reused
                    set_error("tridiagonal QR did not converge for stamp %d", s);
                    return IMCOM_ERR_NUMERIC;
                }
            }
        }
        if (vectors && chunks > 0) {  // join: the main stream continues once the last rotations have been applied
            IMCOM_HIP_CHECK(hipEventRecord(ev_x, sb));
            IMCOM_HIP_CHECK(hipStreamWaitEvent(st, ev_x, 0));
        }
        if (done) aux_drain.armed = false;  // joined through the event: the main stream orders everything that follows
    }
    if (!done) { hipStreamSynchronize(sb); set_error("tridiagonal QR did not converge"); return IMCOM_ERR_NUMERIC; }
    if (sweeps_out) {
        int mx = 0;
        for (int s = 0; s < batch; s++) mx = std::max(mx, sth[(size_t)s * 8 + 4]);
        *sweeps_out = mx;
    }
    return launch_eig_sort_scatter(ctx, X, ld, dvec, n_dev, rank, lam, ldlam, Q, ldq, strideQ, batch);
}

// -------------------------------------------------------------------------------------------------
// Eigensolver used by the library.  The tridiagonal QR path is the product; a developer build (make DEV=1) also holds the
// one-sided block Jacobi solver (jacobi.hip), an independent cross-check selected by IMCOM_EIGH=jacobi.
#ifdef IMCOM_DEV
size_t jacobi_ws_bytes(int batch, int ld);
int jacobi_eigh_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, const double *A, long lda, long strideA,
                       double *lam, long ldlam, double *Q, long ldq, long strideQ, int *sweeps_out);

bool eigh_uses_jacobi()
{
    const char *e = getenv("IMCOM_EIGH");
    return e && strcmp(e, "jacobi") == 0;
}
#else
static size_t jacobi_ws_bytes(int, int) { return 0; }
static int jacobi_eigh_device(imcom_ctx *, int, const int *, int, const double *, long, long, double *, long, double *, long, long, int *)
{
    return IMCOM_ERR_ARG;
}
bool eigh_uses_jacobi() { return false; }
#endif

size_t eigh_ws_bytes(int batch, int ld, bool vectors)
{
    return eigh_uses_jacobi() ? jacobi_ws_bytes(batch, ld) : tridiag_ws_bytes(batch, ld, vectors);
}

int eigh_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, const double *A, long lda, long strideA, double *lam,
                long ldlam, double *Q, long ldq, long strideQ, int *sweeps_out)
{
    if (eigh_uses_jacobi()) {
        IMCOM_REQUIRE(Q != nullptr, "jacobi eigensolver needs the eigenvector output");
        return jacobi_eigh_device(ctx, batch, n_host, ld, A, lda, strideA, lam, ldlam, Q, ldq, strideQ, sweeps_out);
    }
    return tridiag_eigh_device(ctx, batch, n_host, ld, A, lda, strideA, lam, ldlam, Q, ldq, strideQ, sweeps_out);
}

}  // namespace imcom
