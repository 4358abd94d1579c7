// probe_gemm.hip -- diagnostic only (imcom_ctx_gemm_probe): the k loop of the fp64 tile engine as a plain batched product
// C[M][N] = A[M][K] (row-major) x B[K][N] (k-major rows), in two tilings:
//   variant 0: the production engine of mma_dma.h, 128 x 128 tiles, 8-wave workgroups, two per CU;
//   variant 1: 256 x 128 tiles, ONE 16-wave workgroup per CU: the same 4 waves per SIMD and the same LDS, but every B (here:
//              the Y of the solves) slice is fetched once for 256 rows instead of once per 128 -- half the B traffic per flop.
// What the comparison says about taller tiles for the solves is recorded in profiles/ and DESIGN.md.
#include "common.h"
#include "launchers.h"
#include "mma_dma.h"

namespace imcom {

constexpr int PW = 16, PTHREADS = 64 * PW, PTM = 256, PTN = 128;
constexpr int PA_IMG = PTM * DBK, PB_IMG = DBK * DKM_LD, PSTAGE = PA_IMG + PB_IMG;  // doubles

__global__ __launch_bounds__(PTHREADS, 4) void gemm_probe16_kernel(const double *__restrict__ A, long lda, long strideA,
                                                                   const double *__restrict__ B, long ldb, long strideB,
                                                                   double *__restrict__ C, long ldc, long strideC, int K)
{
    __shared__ __attribute__((aligned(16))) double lds[2 * PSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, li = lane & 15, lk = lane >> 4;
    const double *Ag = A + blockIdx.z * strideA + (long)blockIdx.y * PTM * lda;
    const double *Bg = B + blockIdx.z * strideB + (long)blockIdx.x * PTN;
    // LDS-DMA sources: A row-major [256][16], instruction u = 2 wave + q covers rows 8u..8u+7 (chunk swizzle as in mma_dma.h);
    // B k-major [16][144], this wave's k-row = wave
    const double *ga[2], *gb;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int u = 2 * wave + q, row = 8 * u + (lane >> 3), ch = (lane & 7) ^ dma_key(row);
        ga[q] = Ag + (long)row * lda + 2 * ch;
    }
    gb = Bg + (long)wave * ldb + 2 * lane;
    auto issue = [&](int slot) {
        double *st = lds + slot * PSTAGE;
#pragma unroll
        for (int q = 0; q < 2; q++) { IMCOM_GLDS16(ga[q], st + (2 * wave + q) * 8 * DBK); ga[q] += DBK; }
        IMCOM_GLDS16(gb, st + PA_IMG + wave * DKM_LD);
        gb += (long)DBK * ldb;
    };
    int ra[4][4], rb[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = wm * 64 + i * 16 + li;
            ra[kk][i] = row * DBK + ((((lk >> 1) + 2 * kk) ^ dma_key(row)) << 1) + (lk & 1);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) rb[kk][j] = PA_IMG + (lk + 4 * kk) * DKM_LD + wn * 32 + j * 16 + li;
    }
    f64x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nt = K / DBK;
    issue(0);
    if (nt > 1) { issue(1); asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < nt; t++) {
        const double *st = lds + (t & 1) * PSTAGE;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = st[ra[kk][i]];
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = st[rb[kk][j]];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nt) issue(t & 1);
        }
    }
    double *Co = C + blockIdx.z * strideC + (long)blockIdx.y * PTM * ldc + (long)blockIdx.x * PTN;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) Co[(long)(wm * 64 + i * 16 + lk + 4 * r) * ldc + wn * 32 + j * 16 + li] = acc[i][j][r];
}

int launch_gemm_probe16(imcom_ctx *ctx, int M, int N, int K, int batch, const double *A, const double *B, double *C)
{
    IMCOM_REQUIRE(M % PTM == 0 && N % PTN == 0 && K % DBK == 0 && DBK == 16, "gemm probe: M % 256, N % 128, K % 16");
    hipLaunchKernelGGL(gemm_probe16_kernel, dim3(N / PTN, M / PTM, batch), dim3(PTHREADS), 0, ctx->stream, A, (long)K, (long)M * K, B, (long)N,
                       (long)K * N, C, (long)N, (long)M * N, K);
    return check_launch("gemm_probe16_kernel");
}

}  // namespace imcom
