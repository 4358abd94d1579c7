// mma_dma_4wave.h -- the 4-wave (2x2, 64x64 per wave) edition of ../mma_dma.h that gemm_probe.hip measured; kept for the probe only.
//
// acc(128x128) += sum_k Aop[m][k] * Bop[k][n] for one 256-thread workgroup (4 waves, 2x2, 64x64 per wave as
// 4x4 v_mfma_f64_16x16x4_f64 tiles).  Operand slices of BK = 8 are streamed global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of NS = 4 stages, so three
// slices (~6000 MFMA cycles) are in flight behind the one being consumed -- the probe
// (tools/gemm_probe.hip) showed HBM latency, not barriers or LDS, was what held the register-staged
// engine at 60 of 78.6 TFLOP/s.  One raw s_barrier per slice with a counted vmcnt (never 0 in steady state).
//
// LDS images (all in one __shared__ array; an LDS-DMA wave-instruction writes 64 x 16 B contiguously):
//   k-major operand  (element (r,k) at p[k*ld + r]):  [8][128 + 16]; one instruction = one 1 KB k-row; the
//                    16-word row pad makes lane (i = l&15, k = l>>4) reads hit 32 distinct banks.
//   row-major operand (element (r,k) at p[r*ld + k]): [128][8], no pad; one instruction = 16 rows x 64 B.
//                    Bank conflicts are removed by XOR-swizzling the 16-byte chunk index with bits 2..3 of
//                    the row -- applied to the per-lane GLOBAL address (the LDS side of a DMA is linear)
//                    and again on the read.
#pragma once
#include <hip/hip_runtime.h>

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int DBK = 8;                       // k-slice per stage
constexpr int DNS = 4;                       // ring stages
constexpr int DKM_LD = 128 + 16;             // k-major image row stride (doubles)
constexpr int DIMG = DBK * DKM_LD;           // doubles per operand image slot (1152; row-major needs 1024)
constexpr int DSTAGE = 2 * DIMG;             // doubles per stage
constexpr int DMA_LDS_DOUBLES = DNS * DSTAGE;  // 9216 doubles = 73,728 B per workgroup

#define IMCOM_GLDS16(gptr, ldsptr)                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),     \
                                     (__attribute__((address_space(3))) void *)(ldsptr), 16, 0, 0)

template <bool AKM, bool BKM>
__device__ __forceinline__ void mma_tile_dma(f64x4 (&acc)[4][4], const double *__restrict__ Ag, long lda,
                                             const double *__restrict__ Bg, long ldb, int K, double *lds)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int nt = K / DBK;
    if (nt <= 0) return;

    // per-lane global source of this wave's two DMA instructions per operand, for slice 0
    //   k-major: instruction q loads k-row (2*wave + q): lane -> doubles [2*lane, 2*lane+1] of that row
    //   row-major: instruction q loads rows 16*(2*wave+q) .. +15: lane -> row 16u + (lane>>2), chunk (lane&3)^swz
    const double *ga[2], *gb[2];
    long ainc, binc;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int u = 2 * wave + q;
        if (AKM) ga[q] = Ag + (long)u * lda + 2 * lane;
        else { const int row = 16 * u + (lane >> 2); ga[q] = Ag + (long)row * lda + 2 * ((lane & 3) ^ ((row >> 2) & 3)); }
        if (BKM) gb[q] = Bg + (long)u * ldb + 2 * lane;
        else { const int row = 16 * u + (lane >> 2); gb[q] = Bg + (long)row * ldb + 2 * ((lane & 3) ^ ((row >> 2) & 3)); }
    }
    ainc = AKM ? (long)DBK * lda : DBK;
    binc = BKM ? (long)DBK * ldb : DBK;
    // wave-uniform LDS destinations inside a stage
    const int da0 = AKM ? (2 * wave) * DKM_LD : (2 * wave) * 128;          // + q * (DKM_LD or 128)
    const int dastep = AKM ? DKM_LD : 128;
    const int db0 = DIMG + (BKM ? (2 * wave) * DKM_LD : (2 * wave) * 128);
    const int dbstep = BKM ? DKM_LD : 128;

    auto issue = [&](int slot) {
        double *st = lds + slot * DSTAGE;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            IMCOM_GLDS16(ga[q], st + da0 + q * dastep);
            IMCOM_GLDS16(gb[q], st + db0 + q * dbstep);
            ga[q] += ainc;
            gb[q] += binc;
        }
    };

    // fragment read offsets (doubles) inside a stage for kk = 0; kk = 1 adds 4 k
    int ra[4], rb[4], ra1[4], rb1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (AKM) { ra[i] = lk * DKM_LD + wm * 64 + i * 16 + li; ra1[i] = ra[i] + 4 * DKM_LD; }
        else {
            const int row = wm * 64 + i * 16 + li, sw = (row >> 2) & 3;
            ra[i] = row * 8 + (((lk >> 1) ^ sw) << 1) + (lk & 1);
            ra1[i] = row * 8 + (((2 + (lk >> 1)) ^ sw) << 1) + (lk & 1);
        }
        if (BKM) { rb[i] = DIMG + lk * DKM_LD + wn * 64 + i * 16 + li; rb1[i] = rb[i] + 4 * DKM_LD; }
        else {
            const int row = wn * 64 + i * 16 + li, sw = (row >> 2) & 3;
            rb[i] = DIMG + row * 8 + (((lk >> 1) ^ sw) << 1) + (lk & 1);
            rb1[i] = DIMG + row * 8 + (((2 + (lk >> 1)) ^ sw) << 1) + (lk & 1);
        }
    }

    // prologue: up to three slices in flight, slice 0 landed
    issue(0);
    if (nt > 1) issue(1);
    if (nt > 2) issue(2);
    if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nt; t++) {
        // slot (t+3)&3 == (t-1)&3 was last read in iteration t-1; every wave passed that iteration's barrier
        if (t + 3 < nt) issue((t + 3) & 3);
        const double *st = lds + (t & 3) * DSTAGE;
        double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { a0[i] = st[ra[i]]; b0[i] = st[rb[i]]; }
#pragma unroll
        for (int i = 0; i < 4; i++) { a1[i] = st[ra1[i]]; b1[i] = st[rb1[i]]; }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
        if (t + 1 < nt) {
            // slice t+1 must have landed (this wave's part) before the barrier publishes it; slices t+2, t+3
            // stay in flight
            if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the caller's epilogue may reuse LDS: make sure every wave is done reading the last stage
    __builtin_amdgcn_s_barrier();
}

}  // namespace imcom
