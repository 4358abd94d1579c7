// iter_block.hip -- IterKernel (reference src/pyimcom/lakernel.py:533-744, conjugate_gradient 397-442) for 16 output pixels at a
// time.
//
// The reference solves, per output pixel a, the sub-system of the input pixels within rho_acc of a by conjugate gradients.
// Neighbouring output pixels select almost the same input pixels, so one workgroup takes a 4 x 4 patch of output pixels:
//   * the union U of their selections (ascending, <= BCG_UMAX), a 16-bit mask per member (which of the 16 pixels select it);
//   * the dense sub-matrix AU = AA[U][U] once, in workspace (iter_block_setup_kernel);
//   * 16 conjugate-gradient recurrences in step (iter_block_cg_kernel): vectors outside a pixel's own selection are kept at zero
//     (masked p, r, q), which makes each recurrence exactly the CG on its own sub-matrix; the products q = AU p for all 16
//     pixels are ONE 16-column matrix product per step on the fp64 MFMA (v_mfma_f64_16x16x4_f64), so the sub-matrix is read once
//     per step for 16 pixels instead of once per pixel -- the per-pixel kernel (iter_empir.hip) streams 90 k gathered elements per
//     pixel and step and runs at the rate of those gathers;
//   * every recurrence stops on its own test (|r| < rtol |b|, checked at the top of a step as the reference does) and is frozen
//     from then on.
// Sums run in another order than numpy's; what that means for a recurrence that does not converge is described in DESIGN.md
// ("Iterative kernel and rounding") and is the same statement as for the per-pixel kernel.
#include "common.h"
#include "launchers.h"

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int BCG_R = 16;                  // output pixels (right-hand sides) per block
constexpr int BCG_UMAX = 512;              // members of a block's union selection (multiple of 16)
constexpr int BCG_TILES = BCG_UMAX / 16;   // 16-row tiles of the union
constexpr int BCG_TPW = BCG_TILES / 4;     // tiles per wave (4 waves)
constexpr int BCG_CH = 4;                  // k-tiles of a row strip fetched ahead

// block b of stamp s covers output pixels (4 by + dy) W + 4 bx + dx; pix[r] = its index or -1
__device__ __forceinline__ int bcg_pixel(int b, int r, int W, int H, int m)
{
    const int nbx = (W + 3) / 4, by = b / nbx, bx = b - by * nbx;
    const int y = 4 * by + (r >> 2), x = 4 * bx + (r & 3);
    const int a = y * W + x;
    return (y < H && x < W && a < m) ? a : -1;
}

// ------------------------------------------------------------------------------------------------
// Union selection, masks, dense AU (diagonal from `diag`: the reference's in-place kappa adds), masked right-hand sides.
//   usel [blocks][UMAX] int, umask [blocks][UMAX] ushort, nu [blocks] int, AU [blocks][UMAX][UMAX], BU [blocks][UMAX][16]
__global__ __launch_bounds__(256) void iter_block_setup_kernel(const double *__restrict__ A, long lda, long strideA,
                                                               const double *__restrict__ diag, long ldd,
                                                               const double *__restrict__ B, long ldb,
                                                               const double *__restrict__ oyx, const double *__restrict__ iy,
                                                               const double *__restrict__ ix, long ldxy, const int *__restrict__ n,
                                                               int m, int W, int H, double rho, int *__restrict__ usel,
                                                               unsigned short *__restrict__ umask, int *__restrict__ nu,
                                                               double *__restrict__ AU, double *__restrict__ BU)
{
    __shared__ int sel[BCG_UMAX];
    __shared__ unsigned short msk[BCG_UMAX];
    __shared__ double oys[BCG_R], oxs[BCG_R];
    __shared__ int pix[BCG_R];
    __shared__ int wcnt[4], total;
    const int s = blockIdx.y, b = blockIdx.x, ns = n[s];
    const long blk = (long)s * gridDim.x + b;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < BCG_R) {
        const int a = bcg_pixel(b, threadIdx.x, W, H, m);
        pix[threadIdx.x] = a;
        oys[threadIdx.x] = a >= 0 ? oyx[((long)s * 2 + 0) * m + a] : 0.0;
        oxs[threadIdx.x] = a >= 0 ? oyx[((long)s * 2 + 1) * m + a] : 0.0;
    }
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const double *py = iy + s * ldxy, *px = ix + s * ldxy;
    // ordered compaction of the input pixels that at least one of the 16 output pixels accepts
    for (int c0 = 0; c0 < ns; c0 += 256) {
        const int i = c0 + threadIdx.x;
        unsigned mk = 0;
        if (i < ns) {
            const double yi = py[i], xi = px[i];
#pragma unroll
            for (int r = 0; r < BCG_R; r++)
                if (pix[r] >= 0 && hypot(oys[r] - yi, oxs[r] - xi) < rho) mk |= 1u << r;
        }
        const bool in = mk != 0;
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wave; w++) off += wcnt[w];
        off += __popcll(bal & ((1ull << lane) - 1ull));
        if (in && off < BCG_UMAX) { sel[off] = i; msk[off] = (unsigned short)mk; }
        __syncthreads();
        if (threadIdx.x == 0) total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    const int nsel = total;
    if (threadIdx.x == 0) nu[blk] = nsel;
    if (nsel > BCG_UMAX || nsel == 0) return;  // too large: the host falls back to the per-pixel kernel; empty: nothing to solve
    const int up = (nsel + 15) / 16 * 16;
    for (int j = threadIdx.x; j < up; j += 256) {
        usel[blk * BCG_UMAX + j] = j < nsel ? sel[j] : 0;
        umask[blk * BCG_UMAX + j] = j < nsel ? msk[j] : 0;
    }
    const double *As = A + s * strideA, *dg = diag + s * ldd;
    double *AUb = AU + blk * (long)BCG_UMAX * BCG_UMAX, *BUb = BU + blk * (long)BCG_UMAX * BCG_R;
    // dense AU, zero padded to `up`: a wave per row, lanes over the columns
    for (int j = wave; j < up; j += 4) {
        const int gj = j < nsel ? sel[j] : -1;
        const double *row = As + (long)(gj < 0 ? 0 : gj) * lda;
        for (int i = lane; i < up; i += 64) {
            double v = 0.0;
            if (gj >= 0 && i < nsel) {
                const int gi = sel[i];
                v = gi == gj ? dg[gj] : row[gi];
            }
            AUb[(long)j * BCG_UMAX + i] = v;
        }
    }
    // right-hand sides: BU[j][r] = -B/2 [pixel r][U[j]] where pixel r selects U[j], else zero
    for (int t = threadIdx.x; t < up * BCG_R; t += 256) {
        const int j = t >> 4, r = t & 15;
        double v = 0.0;
        if (j < nsel && (msk[j] >> r & 1)) v = B[((long)s * m + pix[r]) * ldb + sel[j]];
        BUb[t] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// 16 conjugate-gradient recurrences in step.  Four waves; wave w owns the 16-row tiles w, w + 4, ...; in a tile, lane
// (li = lane & 15, lk = lane >> 4) owns rows lk + 4 r' (r' < 4) of right-hand side li -- the C/D layout of the MFMA, so that the
// products land where the vectors live.  p is shared through LDS ([row][16]) as the B operand of the next product.
__global__ __launch_bounds__(256, 1) void iter_block_cg_kernel(const double *__restrict__ AU, const double *__restrict__ BU,
                                                               const int *__restrict__ usel, const unsigned short *__restrict__ umask,
                                                               const int *__restrict__ nu, int m, int W, int H, double rtol,
                                                               int maxiter, float *__restrict__ T, long ldt)
{
    extern __shared__ double lds[];  // P [up][16], then red [4][16] doubles
    const int s = blockIdx.y, b = blockIdx.x;
    const long blk = (long)s * gridDim.x + b;
    const int nsel = nu[blk];
    if (nsel > BCG_UMAX || nsel == 0) return;
    const int up = (nsel + 15) / 16 * 16, ntile = up / 16;
    double *P = lds, *red = lds + (long)up * BCG_R;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double *AUb = AU + blk * (long)BCG_UMAX * BCG_UMAX, *BUb = BU + blk * (long)BCG_UMAX * BCG_R;
    const unsigned short *mk = umask + blk * BCG_UMAX;
    const bool valid = bcg_pixel(b, li, W, H, m) >= 0;

    // block sum per right-hand side over all rows: lanes with the same li, then the four waves (fixed order)
    auto rhs_sum = [&](double v) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        __syncthreads();
        if (lk == 0) red[wave * BCG_R + li] = v;
        __syncthreads();
        return (red[li] + red[BCG_R + li]) + (red[2 * BCG_R + li] + red[3 * BCG_R + li]);
    };

    double x[BCG_TPW][4], r[BCG_TPW][4], p[BCG_TPW][4];
    unsigned own = 0;  // bit 4 q + e: this right-hand side selects the row
    double bb = 0.0;
#pragma unroll
    for (int q = 0; q < BCG_TPW; q++) {
        const int tt = wave + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 16 * tt + lk + 4 * e;
            const bool in = tt < ntile;
            if (in && (mk[j] >> li & 1)) own |= 1u << (4 * q + e);
            x[q][e] = 0.0;
            r[q][e] = in ? BUb[(long)j * BCG_R + li] : 0.0;  // already masked
            p[q][e] = r[q][e];
            bb += r[q][e] * r[q][e];
        }
    }
    const double atol = sqrt(rhs_sum(bb)) * rtol;
    double rho_prev = 0.0;
    bool done = !valid;
    for (int it = 0; it < maxiter; it++) {
        double rr = 0.0;
#pragma unroll
        for (int q = 0; q < BCG_TPW; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) rr += r[q][e] * r[q][e];
        const double rho_cur = rhs_sum(rr);
        if (!done && sqrt(rho_cur) < atol) done = true;  // "Are we done?" at the top of the step, as the reference
        if (__syncthreads_and(done)) break;
        const bool act = !done;
        if (it > 0 && act) {
            const double beta = rho_cur / rho_prev;
#pragma unroll
            for (int q = 0; q < BCG_TPW; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) p[q][e] = p[q][e] * beta + r[q][e];
        }
#pragma unroll
        for (int q = 0; q < BCG_TPW; q++) {
            const int tt = wave + 4 * q;
            if (tt < ntile)
#pragma unroll
                for (int e = 0; e < 4; e++) P[(long)(16 * tt + lk + 4 * e) * BCG_R + li] = act ? p[q][e] : 0.0;
        }
        __syncthreads();
        // q = AU P, tile by tile: A operand lane (row li of the tile, k = 4 lk + kk of a 16-wide k block), B operand P[k][li]
        double pq = 0.0;
        double qv[BCG_TPW][4];
#pragma unroll
        for (int q = 0; q < BCG_TPW; q++) {
            const int tt = wave + 4 * q;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            if (tt < ntile) {
                // the row strip of AU streams through registers in chunks of BCG_CH k-tiles, the next chunk's loads in flight
                // while this one is multiplied (one workgroup per CU: the wave has 512 registers, and two loads in flight per lane
                // left every k-tile waiting on a memory round trip)
                const double *arow = AUb + (long)(16 * tt + li) * BCG_UMAX + 4 * lk;
                f64x2 nx[BCG_CH][2];
                auto fetch = [&](int kt0) {
#pragma unroll
                    for (int u = 0; u < BCG_CH; u++) {
                        const int kt = min(kt0 + u, ntile - 1);  // the tail re-reads the last tile (unused)
                        nx[u][0] = *(const f64x2 *)(arow + 16 * kt);
                        nx[u][1] = *(const f64x2 *)(arow + 16 * kt + 2);
                    }
                };
                fetch(0);
                for (int kt0 = 0; kt0 < ntile; kt0 += BCG_CH) {
                    f64x2 cu[BCG_CH][2];
#pragma unroll
                    for (int u = 0; u < BCG_CH; u++) { cu[u][0] = nx[u][0]; cu[u][1] = nx[u][1]; }
                    if (kt0 + BCG_CH < ntile) fetch(kt0 + BCG_CH);
#pragma unroll
                    for (int u = 0; u < BCG_CH; u++) {
                        if (kt0 + u < ntile) {
                            const double *pk = P + (long)(16 * (kt0 + u) + 4 * lk) * BCG_R + li;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[u][0].x, pk[0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[u][0].y, pk[BCG_R], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[u][1].x, pk[2 * BCG_R], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[u][1].y, pk[3 * BCG_R], acc, 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                qv[q][e] = (own >> (4 * q + e) & 1) ? acc[e] : 0.0;
                pq += p[q][e] * qv[q][e];
            }
        }
        const double pqs = rhs_sum(pq);
        if (act) {
            const double alpha = rho_cur / pqs;
#pragma unroll
            for (int q = 0; q < BCG_TPW; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    x[q][e] += alpha * p[q][e];
                    r[q][e] -= alpha * qv[q][e];
                }
            rho_prev = rho_cur;
        }
    }
    const int a = bcg_pixel(b, li, W, H, m);
    if (a >= 0) {
        float *Trow = T + ((long)s * m + a) * ldt;
        const int *us = usel + blk * BCG_UMAX;
#pragma unroll
        for (int q = 0; q < BCG_TPW; q++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (own >> (4 * q + e) & 1) Trow[us[16 * (wave + 4 * q) + lk + 4 * e]] = (float)x[q][e];
    }
}

size_t iter_block_ws_bytes(int batch, int nblocks)
{
    const size_t nb = (size_t)batch * nblocks;
    return nb * ((size_t)BCG_UMAX * BCG_UMAX * 8 + (size_t)BCG_UMAX * BCG_R * 8 + (size_t)BCG_UMAX * 6 + 4) + 4096;
}

int iter_block_count(int m, int W) { const int H = (m + W - 1) / W; return ((W + 3) / 4) * ((H + 3) / 4); }

// One kappa node.  Returns in *max_union the largest union selection (callers fall back to the per-pixel kernel when it
// exceeds iter_block_umax()); T must have been zeroed.
int launch_iter_block(imcom_ctx *ctx, const double *A, long lda, long strideA, const double *diag, long ldd, const double *B, long ldb,
                      const double *oyx, const double *iy, const double *ix, long ldxy, const int *n, int m, int W, int batch,
                      double rho, double rtol, int maxiter, float *T, long ldt, void *ws, int *max_union)
{
    const int H = (m + W - 1) / W, nblocks = iter_block_count(m, W);
    const size_t nb = (size_t)batch * nblocks;
    char *w = (char *)ws;
    double *AU = (double *)w; w += nb * (size_t)BCG_UMAX * BCG_UMAX * 8;
    double *BU = (double *)w; w += nb * (size_t)BCG_UMAX * BCG_R * 8;
    int *usel = (int *)w; w += nb * (size_t)BCG_UMAX * 4;
    unsigned short *umask = (unsigned short *)w; w += nb * (size_t)BCG_UMAX * 2;
    int *nu = (int *)w;
    hipLaunchKernelGGL(iter_block_setup_kernel, dim3(nblocks, batch), dim3(256), 0, ctx->stream, A, lda, strideA, diag, ldd, B, ldb, oyx, iy, ix,
                       ldxy, n, m, W, H, rho, usel, umask, nu, AU, BU);
    IMCOM_TRY(check_launch("iter_block_setup_kernel"));
    std::vector<int> nu_h(nb);
    IMCOM_HIP_CHECK(hipMemcpyAsync(nu_h.data(), nu, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int mx = 0;
    for (size_t i = 0; i < nb; i++) mx = std::max(mx, nu_h[i]);
    *max_union = mx;
    if (mx > BCG_UMAX) return IMCOM_OK;  // nothing solved: the caller uses the per-pixel kernel
    const size_t lds = ((size_t)((mx + 15) / 16 * 16) * BCG_R + 4 * BCG_R) * 8;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)iter_block_cg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(iter_block_cg_kernel, dim3(nblocks, batch), dim3(256), lds, ctx->stream, (const double *)AU, (const double *)BU,
                       (const int *)usel, (const unsigned short *)umask, (const int *)nu, m, W, H, rtol, maxiter, T, ldt);
    return check_launch("iter_block_cg_kernel");
}

int iter_block_umax() { return BCG_UMAX; }

}  // namespace imcom
