// l1_access_probe.hip -- how the L1 (TCP) counts and serves the lanes of one global_load_lds_dwordx4 (64 lanes x 16 B), for the
// gather patterns the A builder could use.  Standalone:  hipcc --offload-arch=gfx950 -O3 -o tools/bin/l1_access_probe tools/l1_access_probe.hip
//   tools/bin/l1_access_probe            prints ns per wave-instruction and GB/s of useful bytes per pattern
// under rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum the accesses per instruction follow from the counter / launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(ldsptr), 16, 0, 0)

constexpr int ITER = 64;     // DMA instructions per wave
constexpr long TAB = 34L << 20;  // bytes of the gather target (MALL / L2 resident like a PSF group's tables)

__device__ __forceinline__ unsigned long mix(unsigned long h) { h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ul; h ^= h >> 32; return h; }

// byte offset of this lane's 16-byte chunk for instruction `it` of wave `w`
template <int P>
__device__ __forceinline__ long lane_offset(int lane, unsigned long seed)
{
    auto rnd = [&](int group, int salt) { return mix(seed * 1315423911ul + group * 2654435761ul + salt); };
    if (P == 0) { return (long)(rnd(lane, 0) % (TAB / 128)) * 128 + 16 * (lane & 7); }                       // every lane its own line
    if (P == 1) { const int g = lane >> 2; return (long)(rnd(g, 1) % (TAB / 64)) * 64 + 16 * (lane & 3); }   // quads: aligned 64-byte pieces
    if (P == 2) { const int g = lane >> 3; return (long)(rnd(g, 2) % (TAB / 128)) * 128 + 16 * (lane & 7); } // octets: whole aligned lines
    if (P == 3) { const int g = lane / 5, c = lane % 5; return (long)(rnd(g, 3) % (TAB / 8 - 32)) * 8 + 16 * c; }  // 80-byte segments at any double (the A builder)
    if (P == 4) { const int g = lane >> 4; return (long)(rnd(g, 4) % (TAB / 256)) * 256 + 16 * (lane & 15); } // 16 lanes: two neighbouring lines
    if (P == 5) { const int g = lane >> 2; return (long)(rnd(g, 5) % (TAB / 128)) * 128 + ((lane & 2) ? 64 : 0) + 16 * (lane & 1); }  // quad = 32 B + 32 B of ONE line
    if (P == 6) { const int g = lane >> 2; return (long)(rnd(g >> 1, 6) % (TAB / 128)) * 128 + 64 * (g & 1) + 16 * (lane & 3); }    // as 2, via the quad index
    if (P == 7) { const int g = lane >> 2, c = lane & 3; return (long)(rnd(g, 7) % (TAB / 8 - 32)) * 8 + 16 * c; }  // quads: 64 contiguous bytes at any double
    if (P == 8) { const int g = lane >> 3; return (long)(rnd(g, 8) % (TAB / 128)) * 128 + 16 * (((lane & 7) * 5) & 7); }            // whole lines, lanes permuted inside
    return 0;
}

template <int P>
__global__ __launch_bounds__(256, 3) void probe_kernel(const char *__restrict__ tab, double *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) double buf[2][256 * 2];
    const int lane = threadIdx.x & 63;
    const unsigned long wid = (unsigned long)blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = 0.0;
    for (int it = 0; it < ITER; it++) {
        const long off = lane_offset<P>(lane, wid * ITER + it);
        GLDS16(tab + off, &buf[it & 1][(threadIdx.x & ~63) * 2]);
        if (it) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            acc += buf[(it - 1) & 1][threadIdx.x * 2];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += buf[(ITER - 1) & 1][threadIdx.x * 2];
    if (acc == 0.123456) sink[0] = acc;
}

template <int P> static void run(const char *tab, double *sink, const char *name, double useful_per_instr)
{
    const int nwg = 256 * 3 * 40;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe_kernel<P>, dim3(nwg), dim3(256), 0, 0, tab, sink);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe_kernel<P>, dim3(nwg), dim3(256), 0, 0, tab, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ninstr = (double)nwg * 4 * ITER;
    printf("pattern %d %-58s %7.3f ms  %6.2f clk/instr/CU at 2.4 GHz  %6.1f useful B/clk/CU\n", P, name, ms,
           ms * 1e-3 * 2.4e9 / (ninstr / 256), useful_per_instr * ninstr / 256 / (ms * 1e-3 * 2.4e9));
}

int main()
{
    char *tab; double *sink;
    CHECK(hipMalloc(&tab, TAB + 4096)); CHECK(hipMemset(tab, 0, TAB + 4096)); CHECK(hipMalloc(&sink, 64));
    run<0>(tab, sink, "64 lanes in 64 different lines", 1024);
    run<1>(tab, sink, "quads: 16 aligned 64-byte pieces", 1024);
    run<2>(tab, sink, "octets: 8 whole aligned 128-byte lines", 1024);
    run<8>(tab, sink, "octets: 8 whole lines, lanes permuted inside the octet", 1024);
    run<4>(tab, sink, "16-lane groups: 4 aligned 256-byte pieces", 1024);
    run<5>(tab, sink, "quads: 32 B + 32 B of one line (16 lines)", 1024);
    run<7>(tab, sink, "quads: 64 contiguous bytes at any double", 1024);
    run<3>(tab, sink, "80-byte segments at any double, 5 lanes each (A builder)", 64.0 / 5 * 80);
    return 0;
}
