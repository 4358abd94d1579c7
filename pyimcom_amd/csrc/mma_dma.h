// mma_dma.h -- fp64 MFMA tile engine, LDS-DMA edition.
//
// acc(128x128) += sum_k Aop[m][k] * Bop[k][n] for one 512-thread workgroup (8 waves, 2x4, 64x32 per wave as
// 4x2 v_mfma_f64_16x16x4_f64 tiles: 64 accumulator registers, so two workgroups = four waves per SIMD are
// resident -- with one workgroup per CU the engine ran at 38 TFLOP/s, with two 4-wave ones at 62).  Operand slices of BK = 16 are streamed global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into two stages of BK = 16: the next slice
// streams in behind the 32 MFMAs per wave (x 4 waves per SIMD, ~8000 cycles) of the current one.  One raw
// s_barrier per slice.
//
// LDS images (all in one __shared__ array; an LDS-DMA wave-instruction writes 64 x 16 B contiguously):
//   k-major operand  (element (r,k) at p[k*ld + r]):  [16][128 + 16]; one instruction = one 1 KB k-row; the
//                    row pad makes lane (i = l&15, k = l>>4) reads hit 32 distinct banks.
//   row-major operand (element (r,k) at p[r*ld + k]): [128][16], no pad; one instruction = 8 rows x 128 B.
//                    Bank conflicts are removed by XOR-swizzling the 16-byte chunk index (0..7) with bits 1..3
//                    of the row -- applied to the per-lane GLOBAL address (the LDS side of a DMA is linear) and
//                    again on the read.  A ds_read_b64 is banked over lanes 0-31 / 32-63 and 64 banks: the 16 rows
//                    x 2 k of a half-wave then hit (row & 1) * 32 + 4 * (chunk ^ (row >> 1 & 7)) + 2 * (k & 1), all
//                    distinct (with the low three row bits as the key, rows r and r + 8 collided: two-way
//                    conflicts on every row-major fragment read).
#pragma once
#include <hip/hip_runtime.h>

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));

#ifndef IMCOM_MMA_BK
#define IMCOM_MMA_BK 16
#endif
constexpr int DBK = IMCOM_MMA_BK;            // k-slice per stage: 16, or 8 (half the LDS per workgroup: three 4-wave workgroups per CU)
constexpr int DCH = DBK / 2;                 // 16-byte chunks per row of a row-major image
constexpr int DRPI = 64 / DCH;               // rows of a row-major image per LDS-DMA instruction
constexpr int DNS = 2;                       // stages (double buffer)
constexpr int DKM_LD = 128 + 16;             // k-major image row stride (doubles)
constexpr int DIMG = DBK * DKM_LD;           // doubles per operand image slot (2304; row-major needs 2048)
constexpr int DSTAGE = 2 * DIMG;             // doubles per stage
constexpr int DMA_LDS_DOUBLES = DNS * DSTAGE;  // 9216 doubles = 73,728 B per workgroup
#ifndef IMCOM_MMA_WAVES
#define IMCOM_MMA_WAVES 8
#endif
#ifndef IMCOM_MMA_PIPE
#define IMCOM_MMA_PIPE 0                     // software-pipelined fragment reads in the full-tile k loop (0: the compiler's own schedule)
#endif
constexpr int MMA_WAVES = IMCOM_MMA_WAVES;   // 8: wave (wm = w >> 2, wn = w & 3) owns rows 64 wm.., columns 32 wn..; 4: 2 x 2 waves of 64 x 64
constexpr int MMA_THREADS = 64 * MMA_WAVES;
constexpr int MMA_WN = MMA_WAVES / 2;        // waves along the columns
constexpr int MMA_NJ = 8 / MMA_WN;           // 16-column MFMA tiles per wave
constexpr int MMA_IQ = DBK / MMA_WAVES;      // LDS-DMA instructions per operand, wave and stage
static_assert(MMA_IQ >= 1 && MMA_IQ * MMA_WAVES == DBK, "k-slice must be a multiple of the wave count");
// swizzle key of a row-major image row (see above): distinct over any 16 consecutive rows together with the row's bank phase
__device__ __forceinline__ int dma_key(int row) { return DCH == 8 ? (row >> 1) & 7 : (row >> 2) & 3; }

#define IMCOM_GLDS16(gptr, ldsptr)                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),     \
                                     (__attribute__((address_space(3))) void *)(ldsptr), 16, 0, 0)

// PARTIAL: only the first `mrows` rows of the tile carry data (the last 128-block of a stamp is padded): 16-row
// groups beyond it are neither read from LDS nor multiplied, their accumulators stay zero.  A separate
// instantiation, so that the full-tile loop stays free of the test.
// TRI (K = 128 only): the A operand is a triangular 128 x 128 block -- 1: lower (element (r,k) is zero for k > r), 2: upper.
// 16-row group g only meets k-slices t <= g (lower) or t >= g (upper); the other products are skipped.  So that both
// halves of the workgroup carry the same share, the wave's four row groups are then interleaved, g = 2 i + wm
// (IMCOM_FOR_ACC_TRI is the matching accumulator map): 20 instead of 32 MFMA rounds on the critical path.
// ABL (diagnostic, imcom_ctx_gemm_probe variants 5-7; results are then meaningless): bit 0 = no LDS-DMA after the prologue,
// bit 1 = no per-slice wait / barrier.
template <bool AKM, bool BKM, bool PARTIAL = false, int TRI = 0, int ABL = 0>
__device__ __forceinline__ void mma_tile_dma(f64x4 (&acc)[4][MMA_NJ], const double *__restrict__ Ag, long lda,
                                             const double *__restrict__ Bg, long ldb, int K, double *lds, int mrows = 128)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / MMA_WN, wn = wave % MMA_WN;
    const int li = lane & 15, lk = lane >> 4;
    const int nt = K / DBK;
    if (nt <= 0) return;
    // PARTIAL: number of this wave's 16-row groups that hold data (wave-uniform)
    const int ni = !PARTIAL ? 4 : (mrows - wm * 64 <= 0 ? 0 : (mrows - wm * 64 >= 64 ? 4 : (mrows - wm * 64 + 15) / 16));

    // per-lane global source of this wave's two DMA instructions per operand, for slice 0
    //   k-major: instruction q loads k-row 2*wave + q: lane -> doubles [2*lane, 2*lane+1] of that row
    //   row-major ([128][16], 8 chunks of 16 B per row): instruction q loads rows 16 wave + 8 q .. +7:
    //             lane -> row + (lane>>3), chunk (lane&7) ^ (row&7)
    const double *ga[MMA_IQ], *gb[MMA_IQ];
#pragma unroll
    for (int q = 0; q < MMA_IQ; q++) {
        const int u = MMA_IQ * wave + q, row = DRPI * u + lane / DCH, ch = (lane % DCH) ^ dma_key(row);
        ga[q] = AKM ? Ag + (long)u * lda + 2 * lane : Ag + (long)row * lda + 2 * ch;
        gb[q] = BKM ? Bg + (long)u * ldb + 2 * lane : Bg + (long)row * ldb + 2 * ch;
    }
    const long ainc = AKM ? (long)DBK * lda : DBK, binc = BKM ? (long)DBK * ldb : DBK;
    // wave-uniform LDS destinations inside a stage
    const int da0 = AKM ? MMA_IQ * wave * DKM_LD : DRPI * MMA_IQ * wave * DBK, dastep = AKM ? DKM_LD : DRPI * DBK;
    const int db0 = DIMG + (BKM ? MMA_IQ * wave * DKM_LD : DRPI * MMA_IQ * wave * DBK), dbstep = BKM ? DKM_LD : DRPI * DBK;

    auto issue = [&](int slot) {
        double *st = lds + slot * DSTAGE;
#pragma unroll
        for (int q = 0; q < MMA_IQ; q++) {
            IMCOM_GLDS16(ga[q], st + da0 + q * dastep);
            IMCOM_GLDS16(gb[q], st + db0 + q * dbstep);
            ga[q] += ainc;
            gb[q] += binc;
        }
    };

    // fragment read offsets (doubles) inside a stage for the four k-quads kk (k = lk + 4 kk)
    constexpr int NKK = DBK / 4;  // MFMA k-quads per slice
    int ra[NKK][4], rb[NKK][MMA_NJ];
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = TRI ? (2 * i + wm) * 16 + li : wm * 64 + i * 16 + li;
            ra[kk][i] = AKM ? (lk + 4 * kk) * DKM_LD + row : row * DBK + ((((lk >> 1) + 2 * kk) ^ dma_key(row)) << 1) + (lk & 1);
        }
#pragma unroll
        for (int i = 0; i < MMA_NJ; i++) {
            const int row = wn * (16 * MMA_NJ) + i * 16 + li;
            rb[kk][i] = DIMG + (BKM ? (lk + 4 * kk) * DKM_LD + row : row * DBK + ((((lk >> 1) + 2 * kk) ^ dma_key(row)) << 1) + (lk & 1));
        }
    }

    // prologue: slice 0 landed, slice 1 in flight
    issue(0);
    if (nt > 1) issue(1);
    if (nt > 1) { if (MMA_IQ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }  // 2 MMA_IQ DMA instructions per wave and slice
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#if IMCOM_MMA_PIPE
    if constexpr (TRI == 0 && !PARTIAL && MMA_NJ == 2 && NKK % 2 == 0) {
        // Software-pipelined fragment reads, pinned by scheduling barriers.  Left alone, the machine scheduler puts the six
        // reads of a k-quad in front of its eight MFMAs behind one lgkmcnt(0), and every wave stalls an LDS round trip per
        // k-quad.  (Counted lgkmcnt waits are not to be had from the compiler: with an LDS-DMA in flight its waitcnt pass
        // treats the counters as "pending FLAT" and waits for zero.)  So the fragments are double buffered: the reads of k-quad
        // kk+1 are issued in front of the MFMAs of kk and have long returned when the wait for zero comes.  The per-slice
        // barrier sits in front of the LAST k-quad's MFMAs -- every LDS read of the slice has returned by then -- so that the
        // first fragments of the next slice are read behind those MFMAs as well.
        double fa[2][4], fb[2][2];
        auto rd = [&](const double *src, int kn, double (&a)[4], double (&b)[2]) {
            b[0] = src[rb[kn][0]]; b[1] = src[rb[kn][1]];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = src[ra[kn][i]];
        };
        auto mm = [&](const double (&a)[4], const double (&b)[2], int i0, int i1) {
#pragma unroll
            for (int i = i0; i < i1; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        rd(lds, 0, fa[0], fb[0]);
        for (int t = 0; t < nt; t++) {
            const double *st = lds + (t & 1) * DSTAGE, *sn = lds + ((t + 1) & 1) * DSTAGE;
#pragma unroll
            for (int kk = 0; kk < NKK; kk++) {
                const int cur = kk & 1, nxt = cur ^ 1;
                // the wait the compiler puts here (for zero, see above) covers reads issued six MFMAs ago
                __builtin_amdgcn_sched_barrier(0);
                mm(fa[cur], fb[cur], 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < NKK - 1) rd(st, kk + 1, fa[nxt], fb[nxt]);
                else if (t + 1 < nt) {
                    // slice t+1 (this wave's part) has landed and every LDS read of slice t has returned: the barrier publishes
                    // the one and frees the other
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 2 < nt) issue(t & 1);
                    rd(sn, 0, fa[nxt], fb[nxt]);
                }
                __builtin_amdgcn_sched_barrier(0);
                mm(fa[cur], fb[cur], 1, 4);
            }
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
#endif
    for (int t = 0; t < nt; t++) {
        const double *st = lds + (t & 1) * DSTAGE;
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) {
            if constexpr (TRI != 0) {
                double b[MMA_NJ];
#pragma unroll
                for (int i = 0; i < MMA_NJ; i++) b[i] = st[rb[kk][i]];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int g = 2 * i + wm;
                    if (TRI == 1 ? DBK * t <= 16 * g + 15 : DBK * t + DBK - 1 >= 16 * g) {  // wave-uniform: the slice meets the group's triangle
                        const double a = st[ra[kk][i]];
#pragma unroll
                        for (int j = 0; j < MMA_NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[i][j], 0, 0, 0);
                    }
                }
            } else if constexpr (!PARTIAL) {
                double a[4], b[MMA_NJ];
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = st[ra[kk][i]];
#pragma unroll
                for (int i = 0; i < MMA_NJ; i++) b[i] = st[rb[kk][i]];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < MMA_NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            } else if (ni > 0) {
                double b[MMA_NJ];
#pragma unroll
                for (int i = 0; i < MMA_NJ; i++) b[i] = st[rb[kk][i]];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (i < ni) {
                        const double a = st[ra[kk][i]];
#pragma unroll
                        for (int j = 0; j < MMA_NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (t + 1 < nt) {
            // slice t+1 (this wave's part) has landed; the barrier publishes it and tells everybody that slot t&1 is free
            if constexpr (!(ABL & 2)) {
                if constexpr (!(ABL & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bit 2: the barrier without the wait for the DMA
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 1))
                if (t + 2 < nt) issue(t & 1);  // slice t+2 streams in behind the 32 MFMAs per wave of slice t+1
        }
    }
    // the caller's epilogue may reuse LDS: make sure every wave is done reading the last stage
    __builtin_amdgcn_s_barrier();
}

}  // namespace imcom
