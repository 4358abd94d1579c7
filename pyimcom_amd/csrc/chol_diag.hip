// chol_diag.hip -- the 128x128 diagonal block of the blocked Cholesky: P[k] -> L[k,k] (in place) and
// Linv[k] = L[k,k]^-1, one 256-thread workgroup per stamp, everything in LDS.
//
// This kernel is the serial spine of the factorisation (18 dependent launches per cfg-2 stamp), so it is
// organised around 16x16 sub-blocks to keep the dependent chain short:
//   factor:  for b = 0..7: (a) 16x16 diagonal sub-block by 16 lanes of one wave (a row per lane in registers, v_readlane),
//            (b) panel rows below solved against it (one thread per row), (c) trailing update of the
//            remaining sub-blocks with v_mfma_f64_16x16x4_f64, tiles dealt over the 4 waves.
//   invert:  16x16 diagonal inverses (one thread per column), then recursive doubling
//            [[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]] for h = 16, 32, 64 with MFMA products,
//            in place over the (already written out) L; the scratch for B A^-1 lives in the unused upper
//            triangle of the LDS image.
// A non-positive pivot (LAPACK dpotrf's failure; scipy raises LinAlgError, reference lakernel.py:262-264)
// records fail[s] = global column + 1 and leaves the block unfinished.
#include "common.h"
#include "launchers.h"

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int SLD = NB + 2;   // LDS row stride: 130 -> fragment reads hit distinct banks
constexpr int XLD = 17;       // stride of the 16x16 diagonal inverses

// C(16x16 at Cp, row stride ldc) (+)= sign * A(16 x 16k) * B(16k x 16), operands read through functors
//   a(r, c): element (r, c) of the A operand tile row-block, c in [0, 16 kt)
//   b(r, c): element (r, c) of the B operand, r in [0, 16 kt)
template <typename FA, typename FB>
__device__ __forceinline__ f64x4 mfma_16(FA a, FB b, int kt, f64x4 acc)
{
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    for (int k4 = 0; k4 < 4 * kt; k4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(li, 4 * k4 + lk), b(4 * k4 + lk, li), acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ double readlane_f64(double v, int l)  // l: wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__global__ __launch_bounds__(256) void chol_diag_kernel(double *__restrict__ L, double *__restrict__ Dinv, int ldn, int k,
                                                        const int *__restrict__ nblk, int *__restrict__ fail)
{
    extern __shared__ double S[];            // [128][SLD]
    double *Xd = S + NB * SLD;               // [8][16][XLD] inverses of the diagonal sub-blocks
    double *dg = Xd + 8 * 16 * XLD;          // [128] diagonal of L
    double *rdg = dg + NB;                   // [128] its reciprocal: the dependent chains below multiply instead of dividing
    __shared__ int bad;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (k >= nblk[s] || fail[s] != 0) return;
    double *Lkk = L + (long)s * ldn * ldn + (long)k * NB * ldn + k * NB;
    // 128 x 128 doubles as 16-byte chunks, eight loads in flight per thread (one 8-byte load per trip was 15 % of the kernel)
    {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll 1
        for (int t0 = tid; t0 < NB * NB / 2; t0 += 8 * 256) {
            f64x2 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int t = t0 + 256 * q;
                v[q] = *(const f64x2 *)&Lkk[(long)(t >> 6) * ldn + 2 * (t & 63)];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int t = t0 + 256 * q;
                *(f64x2 *)&S[(t >> 6) * SLD + 2 * (t & 63)] = v[q];
            }
        }
    }
    if (tid == 0) bad = 0;
    __syncthreads();

    // ------------------------------------------------------------------ factor
    for (int b = 0; b < 8; b++) {
        const int o = 16 * b;
        // (a) diagonal 16x16 sub-block: lanes 0..15 of wave 0 hold one row each in registers; right-looking, the column just
        //     finished is handed round by v_readlane (everything unrolled: no LDS round trip inside the 16 dependent steps; the
        //     left-looking form through LDS, a dot product of j dependent loads per column, was 40 % of this kernel).  Each
        //     element still sees the same subtractions in the same order: S[r][c] - L[r][0] L[c][0] - L[r][1] L[c][1] - ...
        if (wave == 0) {
            double row[16];
#pragma unroll
            for (int c = 0; c < 16; c++) row[c] = lane < 16 ? S[(o + lane) * SLD + o + c] : 1.0;
            int failj = -1;  // first non-positive pivot (wave-uniform); the steps after it run on, their results are discarded
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const double d = readlane_f64(row[j], j);
                if (failj < 0 && !(d > 0.0)) failj = j;  // also NaN
                const double ljj = sqrt(d), rl = 1.0 / ljj;
                const double xr = lane == j ? ljj : row[j] * rl;  // L[r][j] for r >= j (rows above j: unused)
                row[j] = xr;
                if (lane == j) { dg[o + j] = ljj; rdg[o + j] = rl; }
#pragma unroll
                for (int c = j + 1; c < 16; c++) row[c] -= xr * readlane_f64(xr, c);
            }
            if (failj >= 0) {
                if (lane == 0) { bad = 1; fail[s] = k * NB + o + failj + 1; }
            } else if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; c++)
                    if (c <= lane) S[(o + lane) * SLD + o + c] = row[c];
            }
        }
        __syncthreads();
        if (bad) return;
        // (b) panel: rows r > o+15 solve x L_bb^T = p   (one thread per row)
        {
            const int r = o + 16 + tid;
            if (r < NB) {
                double x[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    double v = S[r * SLD + o + j];
#pragma unroll
                    for (int c = 0; c < 16; c++)
                        if (c < j) v -= x[c] * S[(o + j) * SLD + o + c];
                    x[j] = v * rdg[o + j];
                }
#pragma unroll
                for (int j = 0; j < 16; j++) S[r * SLD + o + j] = x[j];
            }
        }
        __syncthreads();
        // (c) trailing update: tile (ti, tc), b < tc <= ti < 8:  S_tile -= P[ti] P[tc]^T over the 16 new columns
        {
            const int nt = 7 - b;                    // sub-blocks below
            const int ntiles = nt * (nt + 1) / 2;
            const int li = lane & 15, lk = lane >> 4;
            for (int t = wave; t < ntiles; t += 4) {
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
                const int tc = t - ti * (ti + 1) / 2;
                const int r0 = o + 16 + 16 * ti, c0 = o + 16 + 16 * tc;
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(r0 + li) * SLD + o + 4 * k4 + lk], S[(c0 + li) * SLD + o + 4 * k4 + lk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) S[(r0 + lk + 4 * r) * SLD + c0 + li] -= acc[r];
            }
        }
        __syncthreads();
    }
    // L[k,k] out (strict upper part zeroed)
    for (int t = tid; t < NB * NB / 2; t += 256) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const int r = t >> 6, c = 2 * (t & 63);
        f64x2 v = *(const f64x2 *)&S[r * SLD + c];
        if (c > r) v.x = 0.0;
        if (c + 1 > r) v.y = 0.0;
        *(f64x2 *)&Lkk[(long)r * ldn + c] = v;
    }
    // ------------------------------------------------------------------ invert
    // 16x16 diagonal inverses: thread (blk = tid>>4, col = tid&15) for tid < 128 computes column `col`
    if (tid < 128) {
        const int blk = tid >> 4, c = tid & 15, o = 16 * blk;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double v = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++)
                if (l < i && l >= c) v -= S[(o + i) * SLD + o + l] * x[l];
            x[i] = (i >= c) ? v * rdg[o + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) Xd[(blk * 16 + i) * XLD + c] = x[i];
    }
    __syncthreads();
    // inverse-so-far accessor: element (r, c) of the lower-triangular inverse assembled from Xd (diagonal
    // 16-blocks) and the already overwritten off-diagonal blocks of S
    auto Xinv = [&](int r, int c) -> double {
        if ((r >> 4) == (c >> 4)) return Xd[r * XLD + (c & 15)];
        return (r > c) ? S[r * SLD + c] : 0.0;
    };
    const int li = lane & 15, lk = lane >> 4;
    for (int h = 16; h < NB; h *= 2) {
        const int hb = h / 16;               // sub-blocks per half
        const int npair = NB / (2 * h);
        // tmp = B A^-1 for every pair, into the upper-right scratch: pair p uses rows [2hp, 2hp+h) x cols [2hp+h, 2hp+2h)
        // (strictly above the diagonal of S, never part of L or of the inverse)
        for (int t = wave; t < npair * hb * hb; t += 4) {
            const int p = t / (hb * hb), q = t % (hb * hb), ti = q / hb, tj = q % hb;
            const int base = 2 * h * p;
            // B = L[base+h .. base+2h) x [base .. base+h); A^-1 = inverse of the upper-left h-block
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb = tj; kb < hb; kb++)  // A^-1 is lower triangular: column block tj only has rows >= tj
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int kk = base + 16 * kb + 4 * k4 + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(base + h + 16 * ti + li) * SLD + kk], Xinv(kk, base + 16 * tj + li), acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; r++) S[(base + 16 * ti + lk + 4 * r) * SLD + base + h + 16 * tj + li] = acc[r];
        }
        __syncthreads();
        // X_BA = -C^-1 tmp, written over B
        f64x4 res[4];  // a wave handles at most (npair*hb*hb)/4 <= 4 tiles per level
        int cnt = 0;
        for (int t = wave; t < npair * hb * hb; t += 4, cnt++) {
            const int p = t / (hb * hb), q = t % (hb * hb), ti = q / hb, tj = q % hb;
            const int base = 2 * h * p;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb = 0; kb <= ti; kb++)  // C^-1 lower triangular: row block ti only has columns <= ti
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int kk = 16 * kb + 4 * k4 + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xinv(base + h + 16 * ti + li, base + h + kk), S[(base + kk) * SLD + base + h + 16 * tj + li], acc, 0, 0, 0);
                }
            res[cnt & 3] = acc;
        }
        __syncthreads();  // every wave has read the B blocks / tmp it needs before anything is overwritten
        cnt = 0;
        for (int t = wave; t < npair * hb * hb; t += 4, cnt++) {
            const int p = t / (hb * hb), q = t % (hb * hb), ti = q / hb, tj = q % hb;
            const int base = 2 * h * p;
            const f64x4 acc = res[cnt & 3];
#pragma unroll
            for (int r = 0; r < 4; r++) S[(base + h + 16 * ti + lk + 4 * r) * SLD + base + 16 * tj + li] = -acc[r];
        }
        __syncthreads();
    }
    double *Di = Dinv + ((long)s * (ldn / NB) + k) * NB * NB;
    for (int t = tid; t < NB * NB / 2; t += 256) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const int r = t >> 6, c = 2 * (t & 63);
        f64x2 v;
        v.x = (c <= r) ? Xinv(r, c) : 0.0;
        v.y = (c + 1 <= r) ? Xinv(r, c + 1) : 0.0;
        *(f64x2 *)&Di[2 * t] = v;
    }
}

// Triangular factor of a block reflector of 128 Householder reflectors, H_ps ... H_ps+127 = I - V T V^T (forward, column-wise):
// T^-1 = diag(1 / tau) + striu(S) with S = V V^T (from a GEMM, symmetric), i.e. T = (Lm^-1)^T for the LOWER triangular
// Lm = diag(1 / tau) + stril(S) -- inverted here exactly as chol_diag_kernel inverts L[k,k]: 16 x 16 diagonal inverses (a column per
// thread), then recursive doubling with MFMA products, everything in LDS.  Only tau itself is ever used (as the reciprocal of Lm's
// diagonal), never 1 / tau: a reflector with tau = 0 (H = I: padding columns) gives a zero row and column of T by itself.
// (tridiag.hip's trd_larft_kernel builds the same T column by column, a row per thread: 128 dependent steps of scalar loads,
// 0.44 - 1.25 ms per panel where this takes tens of microseconds; it stays as IMCOM_LARFT=serial for cross-checks.)
__global__ __launch_bounds__(256) void larft_inv_kernel(const double *__restrict__ Sg, const double *__restrict__ tauvec, int ld, int ps,
                                                        double *__restrict__ T)
{
    extern __shared__ double S[];            // [128][SLD]: strict lower part of S = V V^T, overwritten by the inverse's off-diagonal blocks
    double *Xd = S + NB * SLD;               // [8][16][XLD] inverses of the diagonal sub-blocks
    double *rdg = Xd + 8 * 16 * XLD;         // [128] reciprocal of Lm's diagonal = tau
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Ss = Sg + (long)s * NB * NB;
    {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll 1
        for (int t0 = tid; t0 < NB * NB / 2; t0 += 8 * 256) {
            f64x2 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = *(const f64x2 *)&Ss[2 * (t0 + 256 * q)];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int t = t0 + 256 * q;
                *(f64x2 *)&S[(t >> 6) * SLD + 2 * (t & 63)] = v[q];
            }
        }
    }
    if (tid < NB) rdg[tid] = ps + tid < ld ? tauvec[(long)s * ld + ps + tid] : 0.0;
    __syncthreads();
    if (tid < 128) {
        const int blk = tid >> 4, c = tid & 15, o = 16 * blk;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double v = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++)
                if (l < i && l >= c) v -= S[(o + i) * SLD + o + l] * x[l];
            x[i] = (i >= c) ? v * rdg[o + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) Xd[(blk * 16 + i) * XLD + c] = x[i];
    }
    __syncthreads();
    auto Xinv = [&](int r, int c) -> double {
        if ((r >> 4) == (c >> 4)) return Xd[r * XLD + (c & 15)];
        return (r > c) ? S[r * SLD + c] : 0.0;
    };
    const int li = lane & 15, lk = lane >> 4;
    for (int h = 16; h < NB; h *= 2) {
        const int hb = h / 16, npair = NB / (2 * h);
        for (int t = wave; t < npair * hb * hb; t += 4) {  // tmp = B A^-1 into the (unused) upper right of the pair
            const int p = t / (hb * hb), q = t % (hb * hb), ti = q / hb, tj = q % hb;
            const int base = 2 * h * p;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb = tj; kb < hb; kb++)
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int kk = base + 16 * kb + 4 * k4 + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(base + h + 16 * ti + li) * SLD + kk], Xinv(kk, base + 16 * tj + li), acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; r++) S[(base + 16 * ti + lk + 4 * r) * SLD + base + h + 16 * tj + li] = acc[r];
        }
        __syncthreads();
        f64x4 res[4];
        int cnt = 0;
        for (int t = wave; t < npair * hb * hb; t += 4, cnt++) {  // X_BA = -C^-1 tmp
            const int p = t / (hb * hb), q = t % (hb * hb), ti = q / hb, tj = q % hb;
            const int base = 2 * h * p;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb = 0; kb <= ti; kb++)
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int kk = 16 * kb + 4 * k4 + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xinv(base + h + 16 * ti + li, base + h + kk), S[(base + kk) * SLD + base + h + 16 * tj + li], acc, 0, 0, 0);
                }
            res[cnt & 3] = acc;
        }
        __syncthreads();
        cnt = 0;
        for (int t = wave; t < npair * hb * hb; t += 4, cnt++) {
            const int p = t / (hb * hb), q = t % (hb * hb), ti = q / hb, tj = q % hb;
            const int base = 2 * h * p;
            const f64x4 acc = res[cnt & 3];
#pragma unroll
            for (int r = 0; r < 4; r++) S[(base + h + 16 * ti + lk + 4 * r) * SLD + base + 16 * tj + li] = -acc[r];
        }
        __syncthreads();
    }
    // T = (Lm^-1)^T, upper triangular, row-major [128][128]
    double *To = T + (long)s * NB * NB;
    for (int t = tid; t < NB * NB; t += 256) {
        const int r = t >> 7, c = t & 127;
        To[t] = c >= r ? Xinv(c, r) : 0.0;
    }
}

int launch_larft_inv(imcom_ctx *ctx, const double *S, const double *tauvec, int ld, int ps, double *T, int batch)
{
    static bool attr_set = false;
    const size_t bytes = (size_t)(NB * SLD + 8 * 16 * XLD + NB) * sizeof(double);
    if (!attr_set) {
        IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)larft_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(larft_inv_kernel, dim3(batch), dim3(256), bytes, ctx->stream, S, tauvec, ld, ps, T);
    return check_launch("larft_inv_kernel");
}

int launch_chol_diag(imcom_ctx *ctx, double *L, double *Dinv, int ldn, int k, int batch, const int *nblk, int *fail)
{
    static bool attr_set = false;
    const size_t bytes = (size_t)(NB * SLD + 8 * 16 * XLD + 2 * NB) * sizeof(double);
    if (!attr_set) {
        IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)chol_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(chol_diag_kernel, dim3(batch), dim3(256), bytes, ctx->stream, L, Dinv, ldn, k, nblk, fail);
    return check_launch("chol_diag_kernel");
}

}  // namespace imcom
