// mfma_f64_bench.hip -- measures the issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on gfx950
// (one wave per SIMD and two), to fix the fp64 roofline used by bench.py.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_loop(double *out, int iters, double a0, double b0)
{
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void fma_loop(double *out, int iters, double a0, double b0)
{
    double acc[16];
    for (int i = 0; i < 16; i++) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    double *out;
    hipMalloc(&out, sizeof(double) * 1024 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; wps++) {
        const int threads = 256 * wps;  // wps waves per SIMD
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 0.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)cus * (threads / 64) * iters * 8.0 * 2048.0;
        double cyc_per = (ms * 1e-3 * 2.4e9) / ((double)iters * 8.0 * wps);
        printf("mfma_f64_16x16x4 %d wave/SIMD: %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wps, flops / ms * 1e-9, cyc_per);
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma_loop, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)cus * threads * iters * 16.0 * 2.0;
        printf("v_fma_f64 %d wave/SIMD: %.2f TFLOP/s\n", wps, flops / ms * 1e-9);
    }
    return 0;
}
