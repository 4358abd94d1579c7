// jacobi.hip -- batched symmetric eigendecomposition by one-sided block Jacobi (Hestenes), fp64.
//
// Replaces numpy.linalg.eigh as used by the reference's LA kernels (src/pyimcom/lakernel.py:162, 201,
// 266).  LAPACK's tridiagonalisation is a poor fit for a GPU (half its flops are memory-bound symv);
// Jacobi is all independent 64-column problems and small GEMMs:
//
//   G = A (symmetric), V = I.  Repeat sweeps over a round-robin schedule of column-block pairs (p,q),
//   32 columns each:  W = [Gp Gq]^T [Gp Gq] (64x64 Gram, fp64 MFMA),  W = J Lambda J^T (two-sided cyclic
//   Jacobi on W inside LDS),  [Gp Gq] <- [Gp Gq] J,  [Vp Vq] <- [Vp Vq] J.  When all columns of G are
//   mutually orthogonal, G = A V = V Lambda:  lambda_i = v_i . g_i (signed), eigenvectors = columns of V.
//
// Storage: Gt = G^T and Vt = V^T row-major, so a column block is a contiguous band of rows.  One
// workgroup owns one pair for a whole round (Gram pass, inner eigenproblem, update pass); all pairs of a
// round and all stamps of the batch run in one launch.  Rows >= n of Gt are zero and never rotate.
#include "common.h"
#include "launchers.h"

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int JB = 32;        // columns per block
constexpr int JP = 2 * JB;    // columns per pair problem (64)
constexpr int JCH = 32;       // Gram pass: columns staged per step
constexpr int JLD_R = JCH + 1;  // row stride of the staged row band (odd: conflict-free fragment reads)
constexpr int JLD_W = JP + 1;   // stride of W and J in LDS
constexpr int JUC = 64;       // update pass: columns per step
constexpr int JLD_X = JUC + 16;  // k-major stride of the staged band in the update pass

// inner problem: two-sided cyclic Jacobi on the 64x64 symmetric W (LDS), J accumulates the rotations.
// Parallel ordering: 63 steps of 32 disjoint index pairs (round-robin tournament); every 2x2 block
// W[{i,j},{k,l}] is updated by the rotations of its row pair and of its column pair.
__device__ __forceinline__ void tournament_pair(int step, int k, int &i, int &j)
{
    // players 0..62 rotate, player 63 is fixed
    if (k == 0) { i = 63; j = step; }
    else { i = (step + k) % 63; j = (step - k + 63) % 63; }
    if (i > j) { const int t = i; i = j; j = t; }
}

// returns (in LDS flag) whether any rotation was applied; tol2 = squared relative threshold
__device__ void inner_jacobi(double *W, double *J, double *cs, int *flags, double tol, double tiny)
{
    const int tid = threadIdx.x;
    for (int t = tid; t < JP * JP; t += 256) J[(t >> 6) * JLD_W + (t & 63)] = ((t >> 6) == (t & 63)) ? 1.0 : 0.0;
    if (tid == 0) flags[1] = 0;
    __syncthreads();
    for (int sweep = 0; sweep < 30; sweep++) {
        if (tid == 0) flags[0] = 0;
        __syncthreads();
        for (int step = 0; step < 63; step++) {
            if (tid < 32) {
                int i, j;
                tournament_pair(step, tid, i, j);
                const double wii = W[i * JLD_W + i], wjj = W[j * JLD_W + j], wij = W[i * JLD_W + j];
                double c = 1.0, s = 0.0;
                if (fabs(wij) > tol * sqrt(fabs(wii * wjj)) && fabs(wij) > tiny) {
                    // symmetric Schur: tan(2 theta) = 2 wij / (wjj - wii)
                    const double tau = (wjj - wii) / (2.0 * wij);
                    const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + tt * tt);
                    s = tt * c;
                    flags[0] = 1;
                    flags[1] = 1;
                }
                cs[2 * tid] = c;
                cs[2 * tid + 1] = s;
            }
            __syncthreads();
            // W <- R^T W R over 32 x 32 blocks of 2x2; J <- J R over 64 rows x 32 column pairs
            for (int b = tid; b < 32 * 32; b += 256) {
                const int kr = b >> 5, kc = b & 31;
                int i, j, k, l;
                tournament_pair(step, kr, i, j);
                tournament_pair(step, kc, k, l);
                const double cr = cs[2 * kr], sr = cs[2 * kr + 1], cc = cs[2 * kc], sc = cs[2 * kc + 1];
                const double a = W[i * JLD_W + k], bb = W[i * JLD_W + l], c2 = W[j * JLD_W + k], d = W[j * JLD_W + l];
                // rows: [a b; c d] <- [cr -sr; sr cr]^T-style rotation  (x_i' = c x_i - s x_j ; x_j' = s x_i + c x_j)
                const double a1 = cr * a - sr * c2, b1 = cr * bb - sr * d, c1 = sr * a + cr * c2, d1 = sr * bb + cr * d;
                // columns
                W[i * JLD_W + k] = cc * a1 - sc * b1;
                W[i * JLD_W + l] = sc * a1 + cc * b1;
                W[j * JLD_W + k] = cc * c1 - sc * d1;
                W[j * JLD_W + l] = sc * c1 + cc * d1;
            }
            for (int b = tid; b < 64 * 32; b += 256) {
                const int r = b >> 5, kc = b & 31;
                int k, l;
                tournament_pair(step, kc, k, l);
                const double cc = cs[2 * kc], sc = cs[2 * kc + 1];
                const double x = J[r * JLD_W + k], y = J[r * JLD_W + l];
                J[r * JLD_W + k] = cc * x - sc * y;
                J[r * JLD_W + l] = sc * x + cc * y;
            }
            __syncthreads();
        }
        if (flags[0] == 0) break;
        __syncthreads();
    }
}

// One round: grid (pairs, batch).  pairs_dev[round][pair] = (p, q) block indices.
__global__ __launch_bounds__(256, 2) void jacobi_round_kernel(double *__restrict__ Gt, double *__restrict__ Vt, int ld,
                                                              const int *__restrict__ nblk2,  // blocks (of 32) per stamp
                                                              const int *__restrict__ pairs, double tol,
                                                              const double *__restrict__ tiny, int *__restrict__ changed)
{
    extern __shared__ double sm[];
    double *Xs = sm;                       // staging: Gram [64][33] (2112) / update [64][80] (5120)
    double *Wm = sm;                       // [64][65], aliases the staging area (live only between the passes)
    double *Jm = sm + JP * JLD_X;          // [64][65]
    double *cs = Jm + JP * JLD_W;          // [64]
    int *flags = (int *)(cs + 64);
    const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = pairs[2 * blockIdx.x], q = pairs[2 * blockIdx.x + 1];
    const int nb = nblk2[s];
    if (p >= nb || q >= nb) return;
    const int ncol = nb * JB;  // columns of G in use (multiple of 32)
    double *G = Gt + (long)s * ld * ld, *V = Vt + (long)s * ld * ld;
    // row r (0..63) of the pair band lives at global row grow(r)
    auto grow = [&](int r) { return (r < JB ? p * JB + r : q * JB + (r - JB)); };
    const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lk = lane >> 4;

    // ---- Gram pass: W = band band^T, K = ncol in chunks of 32
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    {
        // 64 rows x 32 cols per chunk = 2048 doubles: 8 per thread (row = tid>>2, 8 consecutive cols)
        const int lr = tid >> 2, lc = (tid & 3) * 8;
        const double *src = G + (long)grow(lr) * ld + lc;
        double rg[8];
#pragma unroll
        for (int e = 0; e < 8; e++) rg[e] = src[e];
        for (int c0 = 0; c0 < ncol; c0 += JCH) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; e++) Xs[lr * JLD_R + lc + e] = rg[e];
            __syncthreads();
            if (c0 + JCH < ncol) {
#pragma unroll
                for (int e = 0; e < 8; e++) rg[e] = src[c0 + JCH + e];
            }
#pragma unroll
            for (int kk = 0; kk < JCH / 4; kk++) {
                double a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    a[i] = Xs[(wm * 32 + i * 16 + li) * JLD_R + kk * 4 + lk];
                    b[i] = Xs[(wn * 32 + i * 16 + li) * JLD_R + kk * 4 + lk];
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // Wm aliases the staging area
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) Wm[(wm * 32 + i * 16 + lk + 4 * r) * JLD_W + wn * 32 + j * 16 + li] = acc[i][j][r];
    __syncthreads();
    // exact symmetry (the two triangles come from different MFMA orders only in principle)
    for (int t = tid; t < JP * JP; t += 256) {
        const int r = t >> 6, c = t & 63;
        if (c > r) Wm[c * JLD_W + r] = Wm[r * JLD_W + c];
    }
    __syncthreads();

    // ---- inner eigenproblem
    inner_jacobi(Wm, Jm, cs, flags, tol, tiny[s]);
    __syncthreads();
    if (flags[1] == 0) return;  // band already orthogonal: nothing to rotate
    if (tid == 0) changed[s] = 1;

    // ---- update pass: band <- J^T band for Gt and Vt, 64 columns per step
    //      out[a][n] = sum_b J[b][a] X[b][n]:  A-op (m=a, k=b) = Jm[b][a] (k-major), B-op (k=b, n) = Xs[b][n]
    for (int arr = 0; arr < 2; arr++) {
        double *M = arr == 0 ? G : V;
        const int width = ncol;  // rows of Vt in use only ever mix with each other: nonzeros stay below ncol
        for (int c0 = 0; c0 < width; c0 += JUC) {
            __syncthreads();
            for (int t = tid; t < JP * JUC; t += 256) {
                const int r = t >> 6, c = t & 63;
                Xs[r * JLD_X + c] = (c0 + c < width) ? M[(long)grow(r) * ld + c0 + c] : 0.0;
            }
            __syncthreads();
            f64x4 o[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) o[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int kk = 0; kk < JP / 4; kk++) {
                double a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    a[i] = Jm[(kk * 4 + lk) * JLD_W + wm * 32 + i * 16 + li];
                    b[i] = Xs[(kk * 4 + lk) * JLD_X + wn * 32 + i * 16 + li];
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) o[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], o[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = wm * 32 + i * 16 + lk + 4 * r, col = c0 + wn * 32 + j * 16 + li;
                        if (col < width) M[(long)grow(row) * ld + col] = o[i][j][r];
                    }
        }
    }
}

// Gt = (A + sigma I)^T on the leading n x n, zero elsewhere; Vt = I.
// One-sided Jacobi orthogonalises the columns of G = A V, i.e. diagonalises A^2: for an INDEFINITE A a pair
// of eigenvalues +x / -x would be degenerate there and their eigenvectors could mix.  The PSF-overlap
// matrices are positive semi-definite only up to rounding, so the iteration runs on A + sigma I with
// sigma = ||A||_inf >= |lambda|_max (Gershgorin): same eigenvectors, all eigenvalues in [0, 2 sigma].  The
// eigenvalues themselves are taken afterwards as Rayleigh quotients of the unshifted A.
__global__ void jacobi_init_kernel(const double *__restrict__ A, long lda, long strideA, const int *__restrict__ n,
                                   double *__restrict__ Gt, double *__restrict__ Vt, int ld,
                                   const double *__restrict__ sigma)
{
    const int s = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const int ns = n[s];
    const long o = (long)s * ld * ld + (long)i * ld + j;
    double v = (i < ns && j < ns) ? A[s * strideA + (long)i * lda + j] : 0.0;
    if (sigma && i == j && i < ns) v += sigma[s];
    Gt[o] = v;
    if (Vt) Vt[o] = (i == j) ? 1.0 : 0.0;
}

// tiny[s] = (eps * sigma * n)^2 guards against rotating pure-noise columns; tiny[batch + s] = sigma = ||A||_inf
__global__ void jacobi_tiny_kernel(const double *__restrict__ A, long lda, long strideA, const int *__restrict__ n,
                                   double *__restrict__ tiny, int batch)
{
    __shared__ double red[256];
    const int s = blockIdx.x, ns = n[s];
    double m = 0.0;
    for (int j = threadIdx.x; j < ns; j += 256) {  // column sums (= row sums, A symmetric), coalesced over j
        double c = 0.0;
        for (int i = 0; i < ns; i++) c += fabs(A[s * strideA + (long)i * lda + j]);
        m = fmax(m, c);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sg = red[0] > 0.0 ? red[0] : 1.0;
        const double t = 2.2e-16 * sg * (double)(ns > 0 ? ns : 1);
        tiny[s] = t * t;
        tiny[batch + s] = sg;
    }
}

// lambda_a = (v_a . g_a) / (v_a . v_a) with g = A v; the row of Vt is renormalised in place (thousands of
// rotations let |v| drift from 1 by ~1e-13, which would otherwise show up directly in lambda).  One wave per row.
__global__ __launch_bounds__(256) void jacobi_lambda_kernel(const double *__restrict__ Gt, double *__restrict__ Vt, int ld,
                                                            const int *__restrict__ n, double *__restrict__ lam_raw)
{
    const int s = blockIdx.y, a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ns = n[s];
    if (a >= ns) return;
    const double *g = Gt + (long)s * ld * ld + (long)a * ld;
    double *v = Vt + (long)s * ld * ld + (long)a * ld;
    double acc = 0.0, vv = 0.0;
    for (int i = lane; i < ns; i += 64) { acc += g[i] * v[i]; vv += v[i] * v[i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { acc += __shfl_xor(acc, off, 64); vv += __shfl_xor(vv, off, 64); }
    const double inv = 1.0 / sqrt(vv);
    for (int i = lane; i < ns; i += 64) v[i] *= inv;
    if (lane == 0) lam_raw[(long)s * ld + a] = acc / vv;
}

int launch_eig_sort_scatter(imcom_ctx *ctx, const double *Vt, int ld, const double *lam_raw, const int *n_dev, int *rank,
                            double *lam, long ldlam, double *Q, long ldq, long strideQ, int batch);  // tridiag.hip

// -------------------------------------------------------------------------------------------------
size_t jacobi_ws_bytes(int batch, int ld)
{
    const int nbk = ld / JB;
    size_t t = 0;
    auto add = [&](size_t b) { t = align_up(t, 256) + b; };
    add((size_t)batch * ld * ld * 8);  // Gt
    add((size_t)batch * ld * ld * 8);  // Vt
    add((size_t)batch * ld * ld * 8);  // G2 = Vt A for the final Rayleigh quotients
    add((size_t)(nbk - 1) * (nbk / 2) * 2 * 4);  // schedule
    add((size_t)batch * 4 * 3);        // n, nblk2, changed
    add((size_t)batch * 16);           // tiny, sigma
    add((size_t)batch * ld * 8);       // lam_raw
    add((size_t)batch * ld * 4);       // rank
    return t + 4096;
}

// A: [batch] matrices (lda, strideA), device.  n_host ragged.  ld = padded size (multiple of 64).
// Outputs (device): lam[s*ldlam + k] ascending for k < n[s]; Q[s*strideQ + i*ldq + k].  Q must be
// zero-initialised by the caller where it wants zero padding.
int jacobi_eigh_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, const double *A, long lda, long strideA,
                       double *lam, long ldlam, double *Q, long ldq, long strideQ, int *sweeps_out)
{
    IMCOM_REQUIRE(ld % NB == 0 && ld >= NB, "jacobi: ld=%d must be a multiple of %d", ld, NB);
    const int nbk = ld / JB;  // even
    double *Gt = (double *)ws_take(ctx, (size_t)batch * ld * ld * 8);
    double *Vt = (double *)ws_take(ctx, (size_t)batch * ld * ld * 8);
    double *G2 = (double *)ws_take(ctx, (size_t)batch * ld * ld * 8);
    const int rounds = nbk - 1, ppr = nbk / 2;
    int *sched = (int *)ws_take(ctx, (size_t)rounds * ppr * 2 * 4);
    int *ints = (int *)ws_take(ctx, (size_t)batch * 4 * 3);
    double *tiny = (double *)ws_take(ctx, (size_t)batch * 16);
    double *lam_raw = (double *)ws_take(ctx, (size_t)batch * ld * 8);
    int *rank = (int *)ws_take(ctx, (size_t)batch * ld * 4);
    if (!Gt || !Vt || !G2 || !sched || !ints || !tiny || !lam_raw || !rank) { set_error("internal: jacobi workspace"); return IMCOM_ERR_NOMEM; }
    int *n_dev = ints, *nblk2 = ints + batch, *changed = ints + 2 * batch;
    // round-robin schedule over nbk blocks (block nbk-1 fixed)
    std::vector<int> sc((size_t)rounds * ppr * 2);
    for (int r = 0; r < rounds; r++)
        for (int k = 0; k < ppr; k++) {
            int a, b;
            if (k == 0) { a = nbk - 1; b = r; }
            else { a = (r + k) % (nbk - 1); b = (r - k + (nbk - 1)) % (nbk - 1); }
            if (a > b) std::swap(a, b);
            sc[((size_t)r * ppr + k) * 2] = a;
            sc[((size_t)r * ppr + k) * 2 + 1] = b;
        }
    std::vector<int> nb2(batch);
    int nbmax = 0;
    for (int s = 0; s < batch; s++) {
        nb2[s] = (n_host[s] + JP - 1) / JP * 2;  // whole pairs of blocks
        nbmax = std::max(nbmax, nb2[s]);
    }
    IMCOM_HIP_CHECK(hipMemcpyAsync(sched, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(n_dev, n_host, (size_t)batch * 4, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipMemcpyAsync(nblk2, nb2.data(), (size_t)batch * 4, hipMemcpyHostToDevice, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // host vectors are locals
    hipLaunchKernelGGL(jacobi_tiny_kernel, dim3(batch), dim3(256), 0, ctx->stream, A, lda, strideA, n_dev, tiny, batch);
    hipLaunchKernelGGL(jacobi_init_kernel, dim3((ld + 255) / 256, ld, batch), dim3(256), 0, ctx->stream, A, lda, strideA, n_dev, Gt, Vt, ld,
                       (const double *)(tiny + batch));
    IMCOM_TRY(check_launch("jacobi_init"));
    const size_t lds = (size_t)(JP * JLD_W + JP * JLD_X + 64) * 8 + 16;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)jacobi_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const double tol = 1e-15 * sqrt((double)std::max(1, *std::max_element(n_host, n_host + batch)));
    std::vector<int> ch(batch);
    int sweep = 0;
    const int max_sweeps = 40;
    for (; sweep < max_sweeps; sweep++) {
        IMCOM_HIP_CHECK(hipMemsetAsync(changed, 0, (size_t)batch * 4, ctx->stream));
        {
            ProfScope ps(ctx, "eigen_jacobi", rounds);
            for (int r = 0; r < rounds; r++)
                hipLaunchKernelGGL(jacobi_round_kernel, dim3(ppr, batch), dim3(256), lds, ctx->stream, Gt, Vt, ld, nblk2,
                                   sched + (size_t)r * ppr * 2, tol, tiny, changed);
            IMCOM_TRY(check_launch("jacobi_round_kernel"));
        }
        IMCOM_HIP_CHECK(hipMemcpyAsync(ch.data(), changed, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        bool any = false;
        for (int s = 0; s < batch; s++) any |= ch[s] != 0;
        if (!any) { sweep++; break; }
    }
    if (sweeps_out) *sweeps_out = sweep;
    if (sweep >= max_sweeps) { set_error("jacobi eigensolver did not converge in %d sweeps", max_sweeps); return IMCOM_ERR_NUMERIC; }
    // The rotated G has accumulated the rounding of every sweep: take the eigenvalues as Rayleigh quotients
    // of the ORIGINAL matrix, lambda_a = v_a^T A v_a, with one GEMM G2 = Vt A (A re-packed into Gt).
    hipLaunchKernelGGL(jacobi_init_kernel, dim3((ld + 255) / 256, ld, batch), dim3(256), 0, ctx->stream, A, lda, strideA, n_dev, Gt,
                       (double *)nullptr, ld, (const double *)nullptr);
    IMCOM_TRY(launch_gemm(ctx, false, true, ld, ld, ld, batch, Vt, ld, (long)ld * ld, Gt, ld, (long)ld * ld, G2, ld, (long)ld * ld, 1.0, 0.0));
    hipLaunchKernelGGL(jacobi_lambda_kernel, dim3((ld + 3) / 4, batch), dim3(256), 0, ctx->stream, G2, Vt, ld, n_dev, lam_raw);
    return launch_eig_sort_scatter(ctx, Vt, ld, lam_raw, n_dev, rank, lam, ldlam, Q, ldq, strideQ, batch);
}

}  // namespace imcom
