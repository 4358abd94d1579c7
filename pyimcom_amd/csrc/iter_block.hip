// iter_block.hip -- IterKernel (reference src/pyimcom/lakernel.py:533-744, conjugate_gradient 397-442) for 16 output pixels at a
// time.
//
// The reference solves, per output pixel a, the sub-system of the input pixels within rho_acc of a by conjugate gradients.
// Neighbouring output pixels select almost the same input pixels, so one workgroup takes a 4 x 4 patch of output pixels:
//   * the union U of their selections (ascending, <= BCG_UMAX), a 16-bit mask per member (which of the 16 pixels select it)
//     (iter_block_select_kernel);
//   * the dense sub-matrix AU = AA[U][U] once, in workspace, as 16 x 16 tiles of 2 KB each, tile-row major
//     (iter_block_gather_kernel): the solver streams a tile row as one contiguous run;
//   * 16 conjugate-gradient recurrences in step (iter_block_cg_kernel): vectors outside a pixel's own selection are kept at zero
//     (masked p, r, q), which makes each recurrence exactly the CG on its own sub-matrix; the products q = AU p for all 16
//     pixels are ONE 16-column matrix product per step on the fp64 MFMA (v_mfma_f64_16x16x4_f64), so the sub-matrix is read once
//     per step for 16 pixels instead of once per pixel -- the per-pixel kernel (iter_empir.hip) streams 90 k gathered elements per
//     pixel and step and runs at the rate of those gathers;
//   * every recurrence stops on its own test (|r| < rtol |b|, checked at the top of a step as the reference does) and is frozen
//     from then on.
// At the reference's default configuration (configs/default_config.json: OUTSIZE [., 32, 0.0390625], INPAD 0.6 -> rho = 15.36
// output pixels, six exposures, KAPPAC [0.0]) a pixel selects ~560 input pixels and a patch's union ~700: the sub-matrix is 4 MB,
// no recurrence converges in 30 steps, and a step is bound by reading those 4 MB out of HBM (2 x 700^2 x 16 flops against them is
// 0.4 of that time on the matrix pipe).  Round 6: unions up to 1024 (until then 512: the default configuration fell back to the
// per-pixel kernel), tile-major storage, loads that run ahead across the tile rows, per-pixel step counts.
// Sums run in another order than numpy's; what that means for a recurrence that does not converge is described in DESIGN.md
// ("Iterative kernel and rounding") and is the same statement as for the per-pixel kernel.
#include <algorithm>
#include <utility>

#include "common.h"
#include "launchers.h"

namespace imcom {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int BCG_R = 16;       // output pixels (right-hand sides) per block
constexpr int BCG_UMAX = 1024;  // members of a block's union selection (multiple of 16)
#ifndef IMCOM_BCG_CH
#define IMCOM_BCG_CH 8
#endif
constexpr int BCG_CH = IMCOM_BCG_CH;  // k-tiles fetched ahead: 16 KB per wave in flight (with 4 the stream ran at 3.5 TB/s, bound by the round trip)

// block b of stamp s covers output pixels (4 by + dy) W + 4 bx + dx; pix[r] = its index or -1
__device__ __forceinline__ int bcg_pixel(int b, int r, int W, int H, int m)
{
    const int nbx = (W + 3) / 4, by = b / nbx, bx = b - by * nbx;
    const int y = 4 * by + (r >> 2), x = 4 * bx + (r & 3);
    const int a = y * W + x;
    return (y < H && x < W && a < m) ? a : -1;
}

// ------------------------------------------------------------------------------------------------
// Union selection and masks of every patch: usel [blocks][UMAX] int, umask [blocks][UMAX] ushort, nu [blocks] int (the true
// count, also when it exceeds UMAX: the host then takes the per-pixel kernel).
__global__ __launch_bounds__(256) void iter_block_select_kernel(const double *__restrict__ oyx, const double *__restrict__ iy,
                                                                const double *__restrict__ ix, long ldxy, const int *__restrict__ n,
                                                                int m, int W, int H, double rho, int *__restrict__ usel,
                                                                unsigned short *__restrict__ umask, int *__restrict__ nu)
{
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
    int *sel = usel + blk * BCG_UMAX;
    unsigned short *msk = umask + blk * BCG_UMAX;
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
    if (nsel > BCG_UMAX) return;
    for (int j = nsel + threadIdx.x; j < (nsel + 15) / 16 * 16; j += 256) { sel[j] = 0; msk[j] = 0; }  // padding rows of the last tile
}

// Dense AU of the patches [blk0, blk0 + gridDim.x) (diagonal from `diag`: the reference's in-place kappa adds), zero padded to the
// patch's own `up`, as tiles: element (j, i) at ((j >> 4) * nts + (i >> 4)) * 256 + (j & 15) * 16 + (i & 15), nts = ups / 16; and
// the masked right-hand sides BU[j][r] = -B/2 [pixel r][U[j]] where pixel r selects U[j], else zero.
// SYM: only the tiles on and below the diagonal, packed column panel by column panel -- tile (I, K), I >= K, at
// (K ntile - K (K - 1) / 2 + I - K) * 256 with ntile the patch's own tile count (iter_block_cg_sym_kernel).
template <bool SYM>
__global__ __launch_bounds__(256) void iter_block_gather_kernel(const double *__restrict__ A, long lda, long strideA,
                                                                const double *__restrict__ diag, long ldd,
                                                                const double *__restrict__ B, long ldb, int m, int W, int H,
                                                                int nblocks, long blk0, int ups, long au_stride, const int *__restrict__ usel,
                                                                const unsigned short *__restrict__ umask, const int *__restrict__ nu,
                                                                double *__restrict__ AU, double *__restrict__ BU)
{
    __shared__ int sel[BCG_UMAX];
    __shared__ int pix[BCG_R];
    const long blk = blk0 + blockIdx.x;
    const int s = (int)(blk / nblocks), b = (int)(blk - (long)s * nblocks);
    const int nsel = nu[blk];
    if (nsel == 0) return;
    const int up = (nsel + 15) / 16 * 16, nts = ups / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < up; j += 256) sel[j] = j < nsel ? usel[blk * BCG_UMAX + j] : -1;
    if (threadIdx.x < BCG_R) pix[threadIdx.x] = bcg_pixel(b, threadIdx.x, W, H, m);
    __syncthreads();
    const double *As = A + s * strideA, *dg = diag + s * ldd;
    double *AUb = AU + (long)blockIdx.x * au_stride, *BUb = BU + (long)blockIdx.x * ups * BCG_R;
    const unsigned short *msk = umask + blk * BCG_UMAX;
    const int ntile = up / 16;
    // a wave per row, lanes over the columns: 16 consecutive lanes write the 128 bytes of a tile row
    for (int j = wave; j < up; j += 4) {
        const int gj = sel[j];
        const double *row = As + (long)(gj < 0 ? 0 : gj) * lda;
        const int J = j >> 4;
        double *out = AUb + (SYM ? 0 : (long)J * nts * 256) + (j & 15) * 16;
        const int iend = SYM ? 16 * (J + 1) : up;  // (SYM: the tiles up to the diagonal one)
        for (int i = lane; i < iend; i += 64) {
            double v = 0.0;
            const int gi = sel[i];
            if (gj >= 0 && gi >= 0) v = gi == gj ? dg[gj] : row[gi];
            const int K = i >> 4;
            const long tile = SYM ? (long)K * ntile - (long)K * (K - 1) / 2 + (J - K) : K;
            out[tile * 256 + (i & 15)] = v;
        }
    }
    for (int t = threadIdx.x; t < up * BCG_R; t += 256) {
        const int j = t >> 4, r = t & 15;
        double v = 0.0;
        if (j < nsel && (msk[j] >> r & 1)) v = B[((long)s * m + pix[r]) * ldb + sel[j]];
        BUb[t] = v;
    }
    if (SYM) AUb[((long)ntile * (ntile + 1) / 2) * 256 + threadIdx.x] = 0.0;  // the all-zero tile behind the packed ones
}

// ------------------------------------------------------------------------------------------------
// 16 conjugate-gradient recurrences in step.  Four waves; wave w owns the 16-row tiles w, w + 4, ... (TPW of them at most); in a
// tile, lane (li = lane & 15, lk = lane >> 4) owns rows lk + 4 r' (r' < 4) of right-hand side li -- the C/D layout of the MFMA, so
// that the products land where the vectors live.  Where the vectors live: the residual r and the step's product q in registers; the
// search direction p in LDS ([row][16], fp64: it is the B operand of the product, and its owner lane reads its own entries back
// from there); the iterate x in workspace, updated by atomic adds without return (one lane per entry: a fixed order, no waiting).
// (What made the first version spill at 700 rows was not the vectors: the optimiser hoisted the 64-bit addresses of every own entry
// out of the step loop, 6 registers per entry -- the opaque offset `ob` below stops that -- and overlapped the unrolled tile rows.)
// stats: [0] patches, [1] sum up^2 steps, [2] sum steps, [3] sum up^2 (unsigned long long, atomics); steps: per output pixel.
template <int TPW>
__global__ __launch_bounds__(256, 1) void iter_block_cg_kernel(const double *__restrict__ AU, const double *__restrict__ BU,
                                                               const int *__restrict__ usel, const unsigned short *__restrict__ umask,
                                                               const int *__restrict__ nu, int m, int W, int H, int nblocks, long blk0,
                                                               int ups, double rtol, int maxiter, float *__restrict__ T, long ldt,
                                                               double *__restrict__ XW, int *__restrict__ steps,
                                                               unsigned long long *__restrict__ stats)
{
    extern __shared__ double lds[];  // P [up][16], then red [4][16] doubles
    const long blk = blk0 + blockIdx.x;
    const int s = (int)(blk / nblocks), b = (int)(blk - (long)s * nblocks);
    const int nsel = nu[blk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int a = bcg_pixel(b, li, W, H, m);
    if (nsel == 0) {
        if (steps && threadIdx.x < BCG_R && a >= 0) steps[(long)s * m + a] = 0;
        return;
    }
    const int up = (nsel + 15) / 16 * 16, ntile = up / 16, nts = ups / 16;
    double *P = lds, *red = lds + (long)up * BCG_R;
    const double *AUb = AU + (long)blockIdx.x * ups * ups, *BUb = BU + (long)blockIdx.x * ups * BCG_R;
    double *Xb = XW + (long)blockIdx.x * ups * BCG_R;  // the iterate x
    const unsigned short *mk = umask + blk * BCG_UMAX;
    const bool valid = a >= 0;
    const int own0 = (16 * wave + lk) * BCG_R + li;  // this lane's entry (q, e) of a [row][16] vector: own0 + (64 q + 4 e) * 16

    // block sum per right-hand side over all rows: lanes with the same li, then the four waves (fixed order)
    auto rhs_sum = [&](double v) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        __syncthreads();
        if (lk == 0) red[wave * BCG_R + li] = v;
        __syncthreads();
        return (red[li] + red[BCG_R + li]) + (red[2 * BCG_R + li] + red[3 * BCG_R + li]);
    };

    double r[TPW][4];
    unsigned long long own = 0;  // bit 4 q + e: this right-hand side selects the row
    double bb = 0.0;
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        const int tt = wave + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 16 * tt + lk + 4 * e;
            const bool in = tt < ntile;
            if (in && (mk[j] >> li & 1)) own |= 1ull << (4 * q + e);
            r[q][e] = in ? BUb[(long)j * BCG_R + li] : 0.0;  // already masked
            if (in) Xb[own0 + (64 * q + 4 * e) * BCG_R] = 0.0;
            bb += r[q][e] * r[q][e];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const double atol = sqrt(rhs_sum(bb)) * rtol;
    double rho_prev = 0.0;
    bool done = !valid;
    int used = 0, block_steps = 0;
    // a wave's tile rows as runs of ntile tiles of 256 doubles; lane (li, lk) reads columns 4 lk .. 4 lk + 3 of row li of a tile
    const double *strip0 = AUb + (long)wave * nts * 256 + li * 16 + 4 * lk;
    const long strip_step = 4L * nts * 256;
    for (int it = 0; it < maxiter; it++) {
        double rr = 0.0;
#pragma unroll
        for (int q = 0; q < TPW; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) rr += r[q][e] * r[q][e];
        const double rho_cur = rhs_sum(rr);
        if (!done && sqrt(rho_cur) < atol) done = true;  // "Are we done?" at the top of the step, as the reference
        if (__syncthreads_and(done)) break;
        const bool act = !done;
        used += act ? 1 : 0;
        block_steps++;
        // (opaque to the optimiser: it otherwise hoists the 64-bit addresses of every own entry of x, q and p out of this loop and keeps
        // them in registers -- 6 registers per entry, which is what made unions above 700 rows spill)
        int ob = own0;
        asm volatile("" : "+v"(ob));
        // p = r (first step), p = p beta + r (later); a recurrence that has stopped multiplies zeros from now on
        const double beta = (it > 0 && act) ? rho_cur / rho_prev : 0.0;
#pragma unroll
        for (int q = 0; q < TPW; q++) {
            if (wave + 4 * q < ntile)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    double *pp = P + ob + (64 * q + 4 * e) * BCG_R;
                    const double pold = it > 0 ? *pp : 0.0;
                    *pp = act ? pold * beta + r[q][e] : 0.0;
                }
            __builtin_amdgcn_sched_barrier(0);  // (one tile at a time: with every tile's loads issued up front the unrolled loop took 16 TPW registers)
        }
        __syncthreads();
        // q = AU P, tile row by tile row: A operand lane (row li of the tile, k = 4 lk + kk of a 16-wide k block), B operand P[k][li].
        // The tiles of this wave's rows stream through registers BCG_CH at a time; the next chunk's loads -- the first chunk of the
        // NEXT tile row at the end of a row -- are in flight while this one is multiplied.
        double pq = 0.0;
        double qv[TPW][4];
#pragma unroll
        for (int q = 0; q < TPW; q++) qv[q][0] = qv[q][1] = qv[q][2] = qv[q][3] = 0.0;
        f64x2 nx[BCG_CH][2];
        auto fetch = [&](const double *strip, int kt0) {
#pragma unroll
            for (int u = 0; u < BCG_CH; u++) {
                const int kt = min(kt0 + u, ntile - 1);  // the tail re-reads the last tile (unused)
                nx[u][0] = *(const f64x2 *)(strip + (long)kt * 256);
                nx[u][1] = *(const f64x2 *)(strip + (long)kt * 256 + 2);
            }
        };
        if (wave < ntile) fetch(strip0, 0);
        // (the loop over this wave's tile rows is NOT unrolled: unrolled, the compiler overlaps the rows' address arithmetic and loads and
        // spills 300 registers; the row's four results reach the register array qv through a switch on the row number)
#pragma unroll 1
        for (int q = 0; wave + 4 * q < ntile; q++) {
            const double *strip = strip0 + q * strip_step;
            const bool more = wave + 4 * (q + 1) < ntile;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int kt0 = 0; kt0 < ntile; kt0 += BCG_CH) {
                f64x2 cu[BCG_CH][2];
#pragma unroll
                for (int u = 0; u < BCG_CH; u++) { cu[u][0] = nx[u][0]; cu[u][1] = nx[u][1]; }
                if (kt0 + BCG_CH < ntile) fetch(strip, kt0 + BCG_CH);
                else if (more) fetch(strip + strip_step, 0);
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
            double v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = (own >> (4 * q + e) & 1) ? acc[e] : 0.0;
                pq += P[ob + (64 * q + 4 * e) * BCG_R] * v[e];
            }
#define BCG_PUT(Q) case Q: if (Q < TPW) { qv[Q < TPW ? Q : 0][0] = v[0]; qv[Q < TPW ? Q : 0][1] = v[1]; qv[Q < TPW ? Q : 0][2] = v[2]; qv[Q < TPW ? Q : 0][3] = v[3]; } break;
            switch (q) {
                BCG_PUT(0) BCG_PUT(1) BCG_PUT(2) BCG_PUT(3) BCG_PUT(4) BCG_PUT(5) BCG_PUT(6) BCG_PUT(7)
                BCG_PUT(8) BCG_PUT(9) BCG_PUT(10) BCG_PUT(11) BCG_PUT(12) BCG_PUT(13) BCG_PUT(14) BCG_PUT(15)
                default: break;
            }
#undef BCG_PUT
        }
        const double pqs = rhs_sum(pq);
        if (act) {
            const double alpha = rho_cur / pqs;
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                if (wave + 4 * q < ntile)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int o = ob + (64 * q + 4 * e) * BCG_R;
                        // x lives in workspace: an atomic add without return -- only this lane ever touches the entry, so the order of the
                        // sums is fixed, and nothing waits for a load (a read-modify-write cost a memory round trip per tile and step)
                        unsafeAtomicAdd(Xb + o, alpha * P[o]);
                        r[q][e] -= alpha * qv[q][e];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            rho_prev = rho_cur;
        }
    }
    __threadfence();  // this lane's atomic adds to x have reached L2 (the loads below bypass L1: device-scope atomic loads)
    if (a >= 0) {
        float *Trow = T + ((long)s * m + a) * ldt;
        const int *us = usel + blk * BCG_UMAX;
#pragma unroll
        for (int q = 0; q < TPW; q++) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (own >> (4 * q + e) & 1) Trow[us[16 * (wave + 4 * q) + lk + 4 * e]] = (float)__hip_atomic_load(Xb + own0 + (64 * q + 4 * e) * BCG_R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (steps && wave == 0 && lk == 0) steps[(long)s * m + a] = used;
    }
    if (stats && threadIdx.x == 0) {
        atomicAdd(stats + 0, 1ull);
        atomicAdd(stats + 1, (unsigned long long)up * up * block_steps);
        atomicAdd(stats + 2, (unsigned long long)block_steps);
        atomicAdd(stats + 3, (unsigned long long)up * up);
    }
}


// ------------------------------------------------------------------------------------------------
// The same recurrences on HALF the sub-matrix (round 6).  AU is symmetric; a step of the kernel above streams both triangles -- 4 MB
// per patch at the default configuration against 2 x 700^2 x 16 flops: 2.5 x the matrix pipe's time at the HBM peak.  Here only the
// tiles on and below the diagonal are stored and read, and every tile T = (I, K), I > K, is used twice while it is in registers:
// q_I += T P_K and q_K += T^T P_I (the transposed operand is the same 2 KB read again, through L1, with the lanes' roles swapped).
// The first product lands where its vector lives (wave I mod 4 owns tile row I).  The second belongs to another wave's tile row K:
// the tiles are visited column panel by column panel (K outer), a wave keeps ONE accumulator for its share of panel K's transposed
// products, leaves it in an LDS ring when the panel is done, and after a barrier per `ring` panels the owner of tile row K adds the four
// waves' pieces in wave order -- a fixed order of sums: results are reproducible.  LDS: P [ups][16], the ring 2 x ring x 4 x 2 KB.
// MEASURED (128 default-configuration stamps): 134 ms against the full-storage kernel's 142, the gather 8.6 against 16.5 -- the default for
// unions up to 768 rows since round 6 (IMCOM_ITER_SYM=0 turns it off).
// RG: the residual r lives in workspace beside x (the twelve-row variant: r, the accumulators and the fetch slots are 3 x 96 registers, and
// with r among them the compiler spilled 200): read twice per step, updated by atomic adds without return.
template <class F, int... Q>
__device__ __forceinline__ void bcg_for_each(F &&f, std::integer_sequence<int, Q...>) { (f(std::integral_constant<int, Q>{}), ...); }

// NW: waves per workgroup, 4 or 8.  Eight (two per SIMD: 256 registers each, half the tile rows per wave) let the hardware issue one wave's
// MFMAs under the other's waits.
template <int TPW, bool RG, int NW>
__global__ __launch_bounds__(64 * NW, 1) void iter_block_cg_sym_kernel(const double *__restrict__ AU, const double *__restrict__ BU,
                                                                   const int *__restrict__ usel, const unsigned short *__restrict__ umask,
                                                                   const int *__restrict__ nu, int m, int W, int H, int nblocks, long blk0,
                                                                   int ups, long au_stride, int ring, int ntw, double rtol, int maxiter,
                                                                   float *__restrict__ T, long ldt, double *__restrict__ XW,
                                                                   int *__restrict__ steps, unsigned long long *__restrict__ stats)
{
    extern __shared__ double lds[];  // P [ups][16], red [NW][16], ring [2][ring][NW][256], Tw [NW][ntw][16][18]
    const long blk = blk0 + blockIdx.x;
    const int s = (int)(blk / nblocks), b = (int)(blk - (long)s * nblocks);
    const int nsel = nu[blk];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int a = bcg_pixel(b, li, W, H, m);
    if (nsel == 0) {
        if (steps && threadIdx.x < BCG_R && a >= 0) steps[(long)s * m + a] = 0;
        return;
    }
    const int up = (nsel + 15) / 16 * 16, ntile = up / 16;
    double *P = lds, *red = lds + (long)ups * BCG_R, *Rg = red + NW * BCG_R, *Tw = Rg + (long)2 * ring * NW * 256;
    const double *AUb = AU + (long)blockIdx.x * au_stride, *BUb = BU + (long)blockIdx.x * ups * BCG_R;
    double *Xb = XW + (long)blockIdx.x * 2 * ups * BCG_R, *Rb = Xb + (long)ups * BCG_R;  // iterate x; residual r (RG)
    const unsigned short *mk = umask + blk * BCG_UMAX;
    const bool valid = a >= 0;
    const int own0 = (16 * wave + lk) * BCG_R + li;

    auto rhs_sum = [&](double v) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        __syncthreads();
        if (lk == 0) red[wave * BCG_R + li] = v;
        __syncthreads();
        double t = (red[li] + red[BCG_R + li]) + (red[2 * BCG_R + li] + red[3 * BCG_R + li]);
        if (NW == 8) t += (red[4 * BCG_R + li] + red[5 * BCG_R + li]) + (red[6 * BCG_R + li] + red[7 * BCG_R + li]);
        return t;
    };

    double r[RG ? 1 : TPW][4];
    unsigned long long own = 0;
    double bb = 0.0;
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        const int tt = wave + NW * q;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 16 * tt + lk + 4 * e;
            const bool in = tt < ntile;
            if (in && (mk[j] >> li & 1)) own |= 1ull << (4 * q + e);
            const double rv = in ? BUb[(long)j * BCG_R + li] : 0.0;
            if (RG) { if (in) Rb[own0 + (16 * NW * q + 4 * e) * BCG_R] = rv; }
            else r[q][e] = rv;
            if (in) Xb[own0 + (16 * NW * q + 4 * e) * BCG_R] = 0.0;
            bb += rv * rv;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const double atol = sqrt(rhs_sum(bb)) * rtol;
    double rho_prev = 0.0;
    bool done = !valid;
    int used = 0, block_steps = 0;
    // the tiles of this wave, column panel by column panel: (K, I) with I = first(K), first(K) + 4, ... < ntile
    for (int it = 0; it < maxiter; it++) {
        if (RG) __threadfence();  // (this lane's atomic adds to r of the step before have reached L2: the loads below bypass L1)
        auto rget = [&](int q, int e, int base) -> double {
            if (!RG) return r[RG ? 0 : q][e];
            return (wave + NW * q < ntile) ? __hip_atomic_load(Rb + base + (16 * NW * q + 4 * e) * BCG_R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        };
        double rr = 0.0;
#pragma unroll
        for (int q = 0; q < TPW; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) { const double rv = rget(q, e, own0); rr += rv * rv; }
        const double rho_cur = rhs_sum(rr);
        if (!done && sqrt(rho_cur) < atol) done = true;
        if (__syncthreads_and(done)) break;
        const bool act = !done;
        used += act ? 1 : 0;
        block_steps++;
        int ob = own0;
        asm volatile("" : "+v"(ob));
        const double beta = (it > 0 && act) ? rho_cur / rho_prev : 0.0;
#pragma unroll
        for (int q = 0; q < TPW; q++) {
            if (wave + NW * q < ntile)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    double *pp = P + ob + (16 * NW * q + 4 * e) * BCG_R;
                    const double pold = it > 0 ? *pp : 0.0;
                    *pp = act ? pold * beta + rget(q, e, ob) : 0.0;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---- q = AU P from the lower tiles.  Column panels K = 0 .. ntile - 1 in order; this wave's tile rows I = wave + NW q in an unrolled
        // loop (q is a compile-time index: the accumulators are plain registers).  Every tile row keeps its own fetch slot across the
        // panels: right after tile (I, K) has been used its slot is refilled with tile (I, K + 1) -- one panel ahead, behind the MFMAs of
        // the wave's other rows.  The panels are grouped in PHASES by the first row that still has a tile in them (qmin: it moves up one row
        // every NW panels), so that inside a phase the rows above qmin run as straight-line code -- no test per tile, the scheduler
        // overlaps one row's LDS round trips with another's MFMAs; only row qmin (whose last tile of the phase is the diagonal one) tests.
        // Rows beyond the patch (the last row of three of the four waves) multiply an all-zero tile.
        f64x4 accv[TPW];
#pragma unroll
        for (int q = 0; q < TPW; q++) accv[q] = f64x4{0.0, 0.0, 0.0, 0.0};
        f64x2 sn[TPW][2];
        const double *lane_n = AUb + li * 16 + 4 * lk;  // this lane's four elements of a tile: row li, columns 4 lk ..
        const long ztile = ((long)ntile * (ntile + 1) / 2) * 256;  // the all-zero tile behind the packed ones (iter_block_gather_kernel<true>)
        double *tw = Tw + wave * (ntw * 16 * 18);                  // this wave's ntw (1 or 2) transposition tiles in LDS ([16][18] each)
#pragma unroll
        for (int q = 0; q < TPW; q++) {  // panel 0: tile (I, 0) at tile index I
            const int I = wave + NW * q;
            const long t = I < ntile ? (long)I * 256 : ztile;
            sn[q][0] = *(const f64x2 *)(lane_n + t);
            sn[q][1] = *(const f64x2 *)(lane_n + t + 2);
        }
        f64x4 part = {0.0, 0.0, 0.0, 0.0};
        long cs1 = ntile;  // first tile index of panel K + 1
        auto panel_end = [&](int K) __attribute__((always_inline)) {
            // panel K is complete for this wave: its transposed sums go to the ring; after every `ring` panels (and after the last) the
            // owners of those tile rows add the four waves' pieces, in wave order
            const int slot = ((K / ring) & 1) * ring + K % ring;
            double *dst = Rg + ((long)slot * NW + wave) * 256 + lane;
#pragma unroll
            for (int e = 0; e < 4; e++) dst[64 * e] = part[e];
            part = f64x4{0.0, 0.0, 0.0, 0.0};
            if ((K + 1) % ring == 0 || K == ntile - 1) {
                __syncthreads();
                for (int Kg = K / ring * ring; Kg <= K; Kg++) {
                    if (Kg % NW != wave) continue;
                    const double *src = Rg + ((long)(((Kg / ring) & 1) * ring + Kg % ring) * NW) * 256 + lane;
                    f64x4 add;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        add[e] = (src[64 * e] + src[256 + 64 * e]) + (src[512 + 64 * e] + src[768 + 64 * e]);
                        if (NW == 8) add[e] += (src[1024 + 64 * e] + src[1280 + 64 * e]) + (src[1536 + 64 * e] + src[1792 + 64 * e]);
                    }
                    // (every row updated, by the piece times one or zero: written as "if (q == Kg / 4) accv[q] += add" the compiler turns
                    // the unrolled tests into ONE dynamically indexed access and moves the whole accumulator array to scratch memory)
#pragma unroll
                    for (int q = 0; q < TPW; q++) {
                        const double w1 = q == Kg / NW ? 1.0 : 0.0;
#pragma unroll
                        for (int e = 0; e < 4; e++) accv[q][e] = fma(add[e], w1, accv[q][e]);
                    }
                }
            }
            cs1 += ntile - K - 1;
        };
        auto phase = [&](auto qm) __attribute__((always_inline)) {
            constexpr int qmin = decltype(qm)::value;
            const int Kbeg = qmin == 0 ? 0 : wave + NW * (qmin - 1) + 1, Kend = min(wave + NW * qmin, ntile - 1);  // the panels of this phase
            for (int K = Kbeg; K <= Kend; K++) {
                double bK[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) bK[kk] = P[(long)(16 * K + 4 * lk + kk) * BCG_R + li];
                const bool last = K + 1 >= ntile;
#pragma unroll
                for (int q = qmin; q < TPW; q++) {
                    const int I = wave + NW * q;
                    const bool rowok = I < ntile, below = I > K;  // (q > qmin: always below the diagonal when the row exists)
                    const f64x2 t0 = sn[q][0], t1 = sn[q][1];
                    f64x4 an = accv[q];
                    an = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bK[0], an, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bK[1], an, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.x, bK[2], an, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.y, bK[3], an, 0, 0, 0);
                    accv[q] = an;
                    if (q > qmin || below) {
                        // the transposed operand through one of the wave's two LDS tiles: element (li, 4 lk + c) in, (4 lk + kk, li) out
                        double *tq = tw + (q & (ntw - 1)) * (16 * 18);
                        *(f64x2 *)(tq + li * 18 + 4 * lk) = t0;
                        *(f64x2 *)(tq + li * 18 + 4 * lk + 2) = t1;
                        // the slot again: tile (I, K + 1), tile index cs1 + (I - K - 1); the zero tile for a row beyond the patch / after the last panel
                        const long t = (rowok && !last) ? (cs1 + (I - K - 1)) * 256 : ztile;
                        sn[q][0] = *(const f64x2 *)(lane_n + t);
                        sn[q][1] = *(const f64x2 *)(lane_n + t + 2);
                        const double *pi = P + (long)(16 * (rowok ? I : ntile - 1) + 4 * lk) * BCG_R + li;
                        const double *tr = tq + (4 * lk) * 18 + li;
                        part = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[0], pi[0], part, 0, 0, 0);
                        part = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[18], pi[BCG_R], part, 0, 0, 0);
                        part = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[36], pi[2 * BCG_R], part, 0, 0, 0);
                        part = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[54], pi[3 * BCG_R], part, 0, 0, 0);
                    }
                }
                panel_end(K);
            }
        };
        bcg_for_each(phase, std::make_integer_sequence<int, TPW>{});
        // panels beyond this wave's last row (ntile > wave + 4 TPW - 3): nothing to multiply, but every wave ends EVERY panel -- its (zero)
        // piece of the ring and the barriers (the first build lacked this: right at ntile = 45 = 4 x 12 - 3, wrong from 46 and for the
        // eight-row variant from 30; tests/test_gpu_iter_default.py::test_point_source_known_answer_on_a_block caught it)
        for (int K = wave + NW * (TPW - 1) + 1; K < ntile; K++) panel_end(K);
        double pq = 0.0;
#pragma unroll
        for (int q = 0; q < TPW; q++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                accv[q][e] = (own >> (4 * q + e) & 1) ? accv[q][e] : 0.0;
                if (wave + NW * q < ntile) pq += P[ob + (16 * NW * q + 4 * e) * BCG_R] * accv[q][e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double pqs = rhs_sum(pq);
        if (act) {
            const double alpha = rho_cur / pqs;
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                if (wave + NW * q < ntile)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int o = ob + (16 * NW * q + 4 * e) * BCG_R;
                        unsafeAtomicAdd(Xb + o, alpha * P[o]);
                        if (RG) unsafeAtomicAdd(Rb + o, -alpha * accv[q][e]);
                        else r[RG ? 0 : q][e] -= alpha * accv[q][e];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            rho_prev = rho_cur;
        }
    }
    __threadfence();
    if (a >= 0) {
        float *Trow = T + ((long)s * m + a) * ldt;
        const int *us = usel + blk * BCG_UMAX;
#pragma unroll
        for (int q = 0; q < TPW; q++) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (own >> (4 * q + e) & 1)
                    Trow[us[16 * (wave + NW * q) + lk + 4 * e]] = (float)__hip_atomic_load(Xb + own0 + (16 * NW * q + 4 * e) * BCG_R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (steps && wave == 0 && lk == 0) steps[(long)s * m + a] = used;
    }
    if (stats && threadIdx.x == 0) {
        atomicAdd(stats + 0, 1ull);
        atomicAdd(stats + 1, (unsigned long long)up * up * block_steps);
        atomicAdd(stats + 2, (unsigned long long)block_steps);
        atomicAdd(stats + 3, (unsigned long long)up * up);
        atomicAdd(stats + 4, (unsigned long long)ntile * (ntile + 1) / 2 * 2048ull * block_steps);  // bytes of sub-matrix streamed
    }
}

int iter_block_count(int m, int W) { const int H = (m + W - 1) / W; return ((W + 3) / 4) * ((H + 3) / 4); }
int iter_block_umax() { return BCG_UMAX; }

// workspace of a call: the selection of every patch of the batch + `budget` bytes for the dense sub-matrices of a sub-batch
size_t iter_block_select_bytes(int batch, int nblocks) { return (size_t)batch * nblocks * ((size_t)BCG_UMAX * 6 + 4) + 4096; }
size_t iter_block_patch_bytes(int ups) { return (size_t)ups * ups * 8 + 2 * (size_t)ups * BCG_R * 8; }  // AU + BU + x

// One kappa node for the whole batch.  *max_union = the largest union selection: above iter_block_umax() nothing was solved and
// the caller takes the per-pixel kernel.  T must have been zeroed.  `ws` holds iter_block_select_bytes() + `budget` bytes: the
// patches go through the solver in groups whose sub-matrices fit the budget.  stats (device, 4 x u64) and steps (device,
// [batch][m]) are optional.
int launch_iter_block(imcom_ctx *ctx, const double *A, long lda, long strideA, const double *diag, long ldd, const double *B, long ldb,
                      const double *oyx, const double *iy, const double *ix, long ldxy, const int *n, int m, int W, int batch,
                      double rho, double rtol, int maxiter, float *T, long ldt, void *ws, size_t budget, int *max_union, int *steps,
                      unsigned long long *stats, int *sym_used)
{
    const int H = (m + W - 1) / W, nblocks = iter_block_count(m, W);
    const size_t nb = (size_t)batch * nblocks;
    char *w = (char *)ws;
    int *usel = (int *)w; w += nb * (size_t)BCG_UMAX * 4;
    unsigned short *umask = (unsigned short *)w; w += nb * (size_t)BCG_UMAX * 2;
    int *nu = (int *)w; w += (nb * 4 + 4095) / 4096 * 4096;
    {
        ProfScope ps(ctx, "iter_gather");
        hipLaunchKernelGGL(iter_block_select_kernel, dim3(nblocks, batch), dim3(256), 0, ctx->stream, oyx, iy, ix, ldxy, n, m, W, H, rho, usel, umask, nu);
        IMCOM_TRY(check_launch("iter_block_select_kernel"));
    }
    std::vector<int> nu_h(nb);
    IMCOM_HIP_CHECK(hipMemcpyAsync(nu_h.data(), nu, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int mx = 0;
    for (size_t i = 0; i < nb; i++) mx = std::max(mx, nu_h[i]);
    *max_union = mx;
    if (sym_used) *sym_used = 0;
    if (mx > BCG_UMAX) return IMCOM_OK;  // nothing solved: the caller uses the per-pixel kernel
    if (mx == 0) {
        if (steps) IMCOM_HIP_CHECK(hipMemsetAsync(steps, 0, (size_t)batch * m * 4, ctx->stream));
        return IMCOM_OK;
    }
    const int ups = (mx + 15) / 16 * 16, ntile = ups / 16;
    // The half-storage kernel where its LDS ring fits beside P (unions up to 864 rows with eight waves), the full-storage one beyond; IMCOM_ITER_SYM=0: the
    // full-storage kernel always (tests/test_gpu_iter_default.py runs both).  The first half-storage kernel (a flat tile list, accumulators
    // through a switch) was slower than the full one; the second (phases, rows unrolled, transposition through LDS) is faster:
    // profiles/r06_negative_results.txt item 1.
    static const bool sym_off = getenv("IMCOM_ITER_SYM") && strcmp(getenv("IMCOM_ITER_SYM"), "0") == 0;
    const size_t lds_p = ((size_t)ups * BCG_R + 4 * BCG_R) * 8, lds_max = 160 * 1024;
    // the half-storage kernel: NW waves (IMCOM_ITER_WAVES=4|8), their transposition tiles and ring pieces
    static const int NWs = getenv("IMCOM_ITER_WAVES") && atoi(getenv("IMCOM_ITER_WAVES")) == 4 ? 4 : 8;
    const size_t lds_ps = ((size_t)ups * BCG_R + NWs * BCG_R) * 8;
    int ntw = 2;  // transposition tiles per wave: two (consecutive rows of a wave overlap) where they fit beside P and one ring unit, else one
    if (lds_ps + (size_t)NWs * ntw * 16 * 18 * 8 + 2 * NWs * 2048 > lds_max) ntw = 1;
    const size_t lds_tw = (size_t)NWs * ntw * 16 * 18 * 8;
    const int ring = lds_ps + lds_tw + 2 * NWs * 2048 <= lds_max ? (int)std::min<size_t>(8, (lds_max - lds_ps - lds_tw) / (2 * NWs * 2048)) : 0;
    const bool sym = !sym_off && ring >= 1 && ntile <= (NWs == 8 ? 56 : 48);  // (rows per wave: 7 x 8 or 12 x 4; the LDS beside P decides first: 864 rows)
    if (sym_used) *sym_used = sym ? 1 : 0;
    const long au_stride = sym ? ((long)ntile * (ntile + 1) / 2 + 1) * 256 : (long)ups * ups;  // (half storage: + the all-zero tile)
    const size_t per = (size_t)au_stride * 8 + (sym ? 3 : 2) * (size_t)ups * BCG_R * 8;  // sub-matrix, right-hand sides, x (+ r: half-storage kernel)
    const size_t group = std::max<size_t>(1, std::min<size_t>(nb, budget / per));
    IMCOM_REQUIRE(per <= budget, "iterative kernel: a patch's sub-matrix (%zu bytes) exceeds the workspace share of %zu", per, budget);
    double *AU = (double *)w, *BU = AU + group * (size_t)au_stride, *XW = BU + group * (size_t)ups * BCG_R;  // (XW: [group][2][ups][16] for the half-storage kernel)
    const size_t lds = sym ? lds_ps + (size_t)2 * ring * NWs * 2048 + lds_tw : lds_p;
    const int tpw = (ntile + 3) / 4;
    auto cg = tpw <= 8 ? iter_block_cg_kernel<8> : tpw <= 12 ? iter_block_cg_kernel<12> : iter_block_cg_kernel<16>;
    auto cgs = NWs == 4 ? (tpw <= 8 ? iter_block_cg_sym_kernel<8, false, 4> : iter_block_cg_sym_kernel<12, true, 4>)
                        : (ntile <= 32 ? iter_block_cg_sym_kernel<4, false, 8> : ntile <= 48 ? iter_block_cg_sym_kernel<6, false, 8> : iter_block_cg_sym_kernel<7, true, 8>);
    if (sym) IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)cgs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)cg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (size_t b0 = 0; b0 < nb; b0 += group) {
        const unsigned g = (unsigned)std::min(group, nb - b0);
        {
            ProfScope ps(ctx, "iter_gather");
            if (sym)
                hipLaunchKernelGGL(iter_block_gather_kernel<true>, dim3(g), dim3(256), 0, ctx->stream, A, lda, strideA, diag, ldd, B, ldb, m, W, H, nblocks, (long)b0,
                                   ups, au_stride, (const int *)usel, (const unsigned short *)umask, (const int *)nu, AU, BU);
            else
                hipLaunchKernelGGL(iter_block_gather_kernel<false>, dim3(g), dim3(256), 0, ctx->stream, A, lda, strideA, diag, ldd, B, ldb, m, W, H, nblocks, (long)b0,
                                   ups, au_stride, (const int *)usel, (const unsigned short *)umask, (const int *)nu, AU, BU);
            IMCOM_TRY(check_launch("iter_block_gather_kernel"));
        }
        ProfScope ps(ctx, "iter_cg");
        if (sym)
            hipLaunchKernelGGL(cgs, dim3(g), dim3(64 * NWs), lds, ctx->stream, (const double *)AU, (const double *)BU, (const int *)usel, (const unsigned short *)umask,
                               (const int *)nu, m, W, H, nblocks, (long)b0, ups, au_stride, ring, ntw, rtol, maxiter, T, ldt, XW, steps, stats);
        else
            hipLaunchKernelGGL(cg, dim3(g), dim3(256), lds, ctx->stream, (const double *)AU, (const double *)BU, (const int *)usel, (const unsigned short *)umask,
                               (const int *)nu, m, W, H, nblocks, (long)b0, ups, rtol, maxiter, T, ldt, XW, steps, stats);
        IMCOM_TRY(check_launch("iter_block_cg_kernel"));
    }
    return IMCOM_OK;
}

}  // namespace imcom
