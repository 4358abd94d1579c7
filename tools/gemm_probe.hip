// gemm_probe.hip -- ablation probe for the fp64 MFMA tile engine (tuning aid, not part of the library).
// Times C[128x128 tiles] += A(row-major) * B(k-major) with pieces of the pipeline switched off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "mma_dma_4wave.h"
using imcom::f64x4;
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int NB = 128, BK = 16, LDS_RM = 17, LDS_KM = 144, TILE_WORDS = BK * LDS_KM;

// MODE bits: 1 = skip global loads after the first tile, 2 = skip barriers, 4 = skip LDS stores after first,
// 8 = skip MFMAs (keep ds_reads alive)
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const double *__restrict__ A, long lda, const double *__restrict__ B,
                                                long ldb, double *__restrict__ C, long ldc, int K, int ntile)
{
    __shared__ double smem[2 * TILE_WORDS];
    double *sA = smem, *sB = smem + TILE_WORDS;
    const int s = blockIdx.y, c = blockIdx.x;
    const double *Ag = A + (long)s * NB * lda;
    const double *Bg = B + (long)s * K * ldb + c * NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 15, lk = lane >> 4;
    f64x4 acc[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = f64x4{0, 0, 0, 0};
    f64x2 ra[4], rb[4];
    const int nt = K / BK;
    auto gload = [&](int t) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cc = tid + 256 * q;
            ra[q] = *(const f64x2 *)(Ag + (long)(cc >> 3) * lda + t * BK + (cc & 7) * 2);
            rb[q] = *(const f64x2 *)(Bg + (long)(t * BK + (cc >> 6)) * ldb + (cc & 63) * 2);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cc = tid + 256 * q;
            double *d = sA + (cc >> 3) * LDS_RM + (cc & 7) * 2; d[0] = ra[q][0]; d[1] = ra[q][1];
            *(f64x2 *)(sB + (cc >> 6) * LDS_KM + (cc & 63) * 2) = rb[q];
        }
    };
    gload(0);
    for (int t = 0; t < nt; t++) {
        if (!(MODE & 2)) __syncthreads();
        if (!(MODE & 4) || t == 0) sstore();
        if (!(MODE & 2)) __syncthreads();
        if (t + 1 < nt && !(MODE & 1)) gload(t + 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                a[i] = sA[(wm * 64 + i * 16 + li) * LDS_RM + kk * 4 + lk];
                b[i] = sB[(kk * 4 + lk) * LDS_KM + wn * 64 + i * 16 + li];
            }
            if (MODE & 8) {
#pragma unroll
                for (int i = 0; i < 4; i++) { acc[i][0][0] += a[i]; acc[0][i][1] += b[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    double *Co = C + ((long)s * ntile + c) * NB * NB;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                Co[(wm * 64 + i * 16 + (lane >> 4) + 4 * r) * NB + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
}

template <bool AKM>
__global__ __launch_bounds__(256, 2) void probe_dma(const double *__restrict__ A, long lda, const double *__restrict__ B,
                                                    long ldb, double *__restrict__ C, long ldc, int K, int ntile, long bstride)
{
    __shared__ __attribute__((aligned(16))) double lds[imcom::DMA_LDS_DOUBLES];
    const int s = blockIdx.y, c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    // AKM: A is stored [K][128] per stamp (k-major) at the same base; else row-major [128][lda]
    const double *Ag = A + (long)s * NB * lda;
    const double *Bg = B + (long)s * bstride + c * NB;
    f64x4 acc[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = f64x4{0, 0, 0, 0};
    imcom::mma_tile_dma<AKM, true>(acc, Ag, AKM ? 128 : lda, Bg, ldb, K, lds);
    double *Co = C + ((long)s * ntile + c) * NB * NB;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                Co[(wm * 64 + i * 16 + (lane >> 4) + 4 * r) * NB + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
}

template <bool AKM>
float run_dma(const double *A, const double *B, double *C, int K, int ntile, int batch, int ldn, int ldm, long bstride)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe_dma<AKM>, dim3(ntile, batch), dim3(256), 0, 0, A, (long)ldn, B, (long)ldm, C, (long)NB, K, ntile, bstride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

template <int MODE>
float run(const double *A, const double *B, double *C, int K, int ntile, int batch, int ldn, int ldm)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(ntile, batch), dim3(256), 0, 0, A, (long)ldn, B, (long)ldm, C, (long)NB, K, ntile);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int K = 2048, ntile = 18, batch = argc > 1 ? atoi(argv[1]) : 256, ldn = 2304, ldm = 2304;
    double *A, *B, *C;
    const size_t nA = (size_t)batch * NB * ldn, nB = (size_t)batch * K * ldm, nC = (size_t)batch * ntile * NB * NB;
    hipMalloc(&A, nA * 8); hipMalloc(&B, nB * 8); hipMalloc(&C, nC * 8);
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (size_t o = 0; o < nA; o += h.size()) hipMemcpy(A + o, h.data(), std::min(h.size(), nA - o) * 8, hipMemcpyHostToDevice);
    for (size_t o = 0; o < nB; o += h.size()) hipMemcpy(B + o, h.data(), std::min(h.size(), nB - o) * 8, hipMemcpyHostToDevice);
    const double flops = 2.0 * batch * ntile * NB * NB * (double)K;
    double *C2; hipMalloc(&C2, nC * 8);
    const char *names[] = {"full", "no global loads", "no barriers", "no loads+no barriers", "no LDS stores", "", "", "no loads/barriers/stores"};
    float t;
    t = run<0>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  %.1f TF\n", names[0], t, flops / t * 1e-9);
    {
        float td = run_dma<false>(A, B, C2, K, ntile, batch, ldn, ldm, (long)K * ldm);
        std::vector<double> h1(1 << 18), h2(1 << 18);
        hipMemcpy(h1.data(), C, h1.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), C2, h2.size() * 8, hipMemcpyDeviceToHost);
        double e = 0, m = 0;
        for (size_t i = 0; i < h1.size(); i++) { e = std::max(e, std::fabs(h1[i] - h2[i])); m = std::max(m, std::fabs(h1[i])); }
        printf("%-28s %.3f ms  %.1f TF   max|diff| vs full = %.3g (max|C| %.3g)\n", "LDS-DMA ring (A row-major)", td, flops / td * 1e-9, e, m);
        td = run_dma<true>(A, B, C2, K, ntile, batch, ldn, ldm, (long)K * ldm);
        printf("%-28s %.3f ms  %.1f TF\n", "LDS-DMA ring (A k-major)", td, flops / td * 1e-9);
        td = run_dma<false>(A, B, C2, K, ntile, batch, ldn, ldm, 0L);
        printf("%-28s %.3f ms  %.1f TF\n", "LDS-DMA, B shared (L2 hits)", td, flops / td * 1e-9);
    }
    t = run<1>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  %.1f TF\n", names[1], t, flops / t * 1e-9);
    t = run<2>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  %.1f TF\n", names[2], t, flops / t * 1e-9);
    t = run<3>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  %.1f TF\n", names[3], t, flops / t * 1e-9);
    t = run<5>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  %.1f TF\n", "no loads, no LDS stores", t, flops / t * 1e-9);
    t = run<7>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  %.1f TF\n", names[7], t, flops / t * 1e-9);
    t = run<8>(A, B, C, K, ntile, batch, ldn, ldm); printf("%-28s %.3f ms  (no MFMA: memory+LDS pipeline alone)\n", "no mfma", t);
    return 0;
}
