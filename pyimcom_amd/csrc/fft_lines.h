// fft_lines.h -- line FFTs, one wavefront per line (device side of the butterfly path of psf_overlap.hip).
//
// A line of n <= 1024 complex doubles is transformed by ONE wave: Stockham stages of radix 16 / 8 / 4 / 2 / 3 / 5 whose
// butterflies sit in registers (a lane holds ceil(n / R / 64) butterflies of R values), with the line exchanged between
// stages through the wave's own slice of LDS.  Because a wave executes its LDS instructions in order there is no
// workgroup barrier anywhere in the transform: waves of a workgroup only share the twiddle tables, and the LDS pipe --
// the binding resource -- is kept busy by whichever waves have their operands (the earlier engine, 8 lines per 16-wave
// workgroup with two __syncthreads per radix-4 stage, ran at a sixth of the LDS rate).
//   * the first stage reads its inputs straight from the caller's loader (global memory: spectra products, packed real
//     rows, zero padding), the last one hands its outputs to the caller's store (global memory again where the result
//     needs no partner element): nst - 1 LDS round trips per line, 2 for n = 768 = 16 x 16 x 3;
//   * LDS index i lives at i ^ ((i >> 4) & 15): the Stockham scatter of a radix-16 stage (lane stride 16 elements =
//     256 B, i.e. one bank group) becomes conflict free, contiguous reads stay so;
//   * twiddles per stage as tables tw[off_s + (t - 1) Ns + k] = exp(-2 pi i t k / (Ns R)), contiguous in k.
#pragma once
#include "fft_radix.h"

namespace imcom {

constexpr int WF_MAXN = 1024;
constexpr int WF_MAXST = 8;
constexpr int WF_MAXWAVES = 12;  // waves per workgroup: three per SIMD leave a wave 168 registers (a radix-16 butterfly with its twiddles needs ~140)

struct FftPlan {
    int n, npad, nst, radix[WF_MAXST], twoff[WF_MAXST], twn, waves;  // npad: n rounded up to 16; waves per workgroup
};

__device__ __forceinline__ int wf_swz(int i) { return i ^ ((i >> 4) & 15); }

template <int R> struct WfQ { static constexpr int value = (WF_MAXN / R + 63) / 64; };  // butterflies per lane

// one stage: butterfly j (< n / R) takes load(j + t n/R), t < R, times the stage twiddles, and delivers
// store((j - k) R + k + u Ns), u < R, with k = j mod Ns (Ns = product of the earlier radices)
template <int R, bool INV, class Load, class Store>
__device__ __forceinline__ void wf_stage(int n, int Ns, const cplx *tws, Load load, Store store)
{
    constexpr int Q = WfQ<R>::value;
    const int lane = threadIdx.x & 63, nb = n / R;
    const bool pow2 = (Ns & (Ns - 1)) == 0;
    cplx v[Q][R];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = lane + 64 * q;
        if (j < nb) {
#pragma unroll
            for (int t = 0; t < R; t++) v[q][t] = load(j + t * nb);
            if (Ns > 1) {
                const int k = pow2 ? (j & (Ns - 1)) : j % Ns;
#pragma unroll
                for (int t = 1; t < R; t++) {
                    cplx w = tws[(t - 1) * Ns + k];
                    if (INV) w.y = -w.y;
                    v[q][t] = cmulf(v[q][t], w);
                }
            }
            SmallDft<R, INV>::run(v[q]);
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = lane + 64 * q;
        if (j < nb) {
            const int k = pow2 ? (j & (Ns - 1)) : j % Ns, base = (j - k) * R + k;
#pragma unroll
            for (int u = 0; u < R; u++) store(base + u * Ns, v[q][u]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <bool INV, class Load, class Store>
__device__ __forceinline__ void wf_stage_r(int r, int n, int Ns, const cplx *tws, Load load, Store store)
{
    switch (r) {
    case 16: wf_stage<16, INV>(n, Ns, tws, load, store); break;
    case 8: wf_stage<8, INV>(n, Ns, tws, load, store); break;
    case 4: wf_stage<4, INV>(n, Ns, tws, load, store); break;
    case 2: wf_stage<2, INV>(n, Ns, tws, load, store); break;
    case 3: wf_stage<3, INV>(n, Ns, tws, load, store); break;
    default: wf_stage<5, INV>(n, Ns, tws, load, store); break;
    }
}

// the whole line: load0(i) delivers input element i, storeN(i, value) takes output element i (natural order both);
// `line` is this wave's npad-element LDS slice, `twl` the stage tables (LDS).  pl.nst >= 2.
template <bool INV, class Load0, class StoreN>
__device__ __forceinline__ void wf_line(cplx *line, const cplx *twl, const FftPlan &pl, Load0 load0, StoreN storeN)
{
    auto ld = [line](int i) { return line[wf_swz(i)]; };
    auto st = [line](int i, cplx v) { line[wf_swz(i)] = v; };
    const int last = pl.nst - 1;
    wf_stage_r<INV>(pl.radix[0], pl.n, 1, twl, load0, st);
    int Ns = pl.radix[0];
    for (int s = 1; s < last; s++) {
        wf_stage_r<INV>(pl.radix[s], pl.n, Ns, twl + pl.twoff[s], ld, st);
        Ns *= pl.radix[s];
    }
    wf_stage_r<INV>(pl.radix[last], pl.n, Ns, twl + pl.twoff[last], ld, storeN);
}

// ---------------------------------------------------------------------------------------------------------
// Static shape n = 256 r = 16 x 16 x r (r = 2, 3, 4: nfft 512, 768, 1024 -- what PSFGrp.setup produces for the usual
// npixpsf x oversamp): every LDS address is a per-lane base plus a compile-time offset.  Here the line is stored padded,
// element i at i + (i >> 4) (17 n / 16 elements per line), which makes the radix-16 scatter conflict free without any
// per-element index arithmetic.  Same stage tables as the general plan {16, 16, r}.
template <int R2> struct Wf16 {
    static constexpr int N = 256 * R2, NB = 16 * R2, LINE = 272 * R2, TW2 = 240, TWN = 240 + (R2 - 1) * 256;
};
__device__ __forceinline__ int wf_pad16(int i) { return i + (i >> 4); }

#ifndef IMCOM_FFT_ABL
#define IMCOM_FFT_ABL 0  // timing experiments (tools/ab_fft_abl.sh; results are then garbage): bit 0 no global loads, bit 1 no global stores,
                         // bit 2 twiddles as constants (no LDS reads for them), bit 3 no LDS stores between the stages
#endif
// tw1(t), t = 1 .. 15: the lane's stage-1 twiddle exp(-2 pi i t k / 256), k = lane & 15; tw2(q, t), t = 1 .. R2 - 1: the stage-2 twiddle
// exp(-2 pi i t j / n) of butterfly j = lane + 64 q -- from the tables in LDS (wf16_line) or from registers the caller filled once
// (wf16_line_regs: a lane's twiddles are the same for every line)
template <int R2, bool INV, class Load0, class StoreN, class Tw1, class Tw2>
__device__ __forceinline__ void wf16_line_impl(cplx *line, Load0 load0, StoreN storeN, Tw1 tw1, Tw2 tw2)
{
    constexpr int NB = Wf16<R2>::NB;
    constexpr int ABL = IMCOM_FFT_ABL;
    const int lane = threadIdx.x & 63;
    double sink = 0.0;
    if (lane < NB) {  // stage 0: butterfly `lane`, inputs lane + NB t, outputs 16 lane + u
        cplx v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            v[t] = (ABL & 1) ? make_double2(1e-3 * lane + t, 0.5 * t - 1e-3 * lane) : load0(lane + NB * t);
            if (t == 7) __builtin_amdgcn_sched_barrier(0);  // two batches of loads: a loader may need two fetches per element
        }
        SmallDft<16, INV>::run(v);
        cplx *dst = line + 17 * lane;
#pragma unroll
        for (int u = 0; u < 16; u++) { if (ABL & 8) sink += v[u].x + v[u].y; else dst[u] = v[u]; }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < NB) {  // stage 1 (Ns = 16): k = lane & 15; inputs lane + NB t, outputs 256 (lane >> 4) + k + 16 u
        const int k = lane & 15, a = lane >> 4;
        const cplx *src = line + lane + a;
        cplx v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = src[17 * R2 * t];
#pragma unroll
        for (int t = 1; t < 16; t++) {
            cplx w = (ABL & 4) ? make_double2(0.6 + 0.01 * t, 0.8 - 0.01 * t) : tw1(t);
            if (INV) w.y = -w.y;
            v[t] = cmulf(v[t], w);
        }
        SmallDft<16, INV>::run(v);
        cplx *dst = line + 272 * a + k;
#pragma unroll
        for (int u = 0; u < 16; u++) { if (ABL & 8) sink += v[u].x + v[u].y; else dst[17 * u] = v[u]; }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {  // stage 2 (Ns = 256): butterfly j = lane + 64 q, inputs j + 256 t, outputs j + 256 u
        const int j = lane + 64 * q;
        const cplx *src = line + j + (j >> 4);
        cplx v[R2];
#pragma unroll
        for (int t = 0; t < R2; t++) v[t] = src[272 * t];
#pragma unroll
        for (int t = 1; t < R2; t++) {
            cplx w = (ABL & 4) ? make_double2(0.6 + 0.01 * t, 0.8 - 0.01 * t) : tw2(q, t);
            if (INV) w.y = -w.y;
            v[t] = cmulf(v[t], w);
        }
        SmallDft<R2, INV>::run(v);
#pragma unroll
        for (int u = 0; u < R2; u++) {
            if (ABL & 8) v[u].x += sink;
            if (!(ABL & 2) || v[u].x == 1.2345e300) storeN(j + 256 * u, v[u]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int R2, bool INV, class Load0, class StoreN>
__device__ __forceinline__ void wf16_line(cplx *line, const cplx *twl, Load0 load0, StoreN storeN)
{
    const int lane = threadIdx.x & 63;
    const cplx *t1 = twl + (lane & 15), *t2 = twl + Wf16<R2>::TW2 + lane;
    wf16_line_impl<R2, INV>(line, load0, storeN, [t1](int t) { return t1[16 * (t - 1)]; }, [t2](int q, int t) { return t2[64 * q + 256 * (t - 1)]; });
}

// a lane's twiddles in registers: 15 + 4 (R2 - 1) complex values (92 registers at n = 768), for kernels that run two waves per SIMD anyway
template <int R2> struct WfTw {
    cplx t1[15], t2[4][R2 - 1];
};
template <int R2>
__device__ __forceinline__ void wf16_load_tw(WfTw<R2> &w, const cplx *tw)  // tw: the stage tables (global memory or LDS)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 1; t < 16; t++) w.t1[t - 1] = tw[(lane & 15) + 16 * (t - 1)];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int t = 1; t < R2; t++) w.t2[q][t - 1] = tw[Wf16<R2>::TW2 + lane + 64 * q + 256 * (t - 1)];
}
template <int R2, bool INV, class Load0, class StoreN>
__device__ __forceinline__ void wf16_line_regs(cplx *line, const WfTw<R2> &w, Load0 load0, StoreN storeN)
{
    wf16_line_impl<R2, INV>(line, load0, storeN, [&w](int t) { return w.t1[t - 1]; }, [&w](int q, int t) { return w.t2[q][t - 1]; });
}

}  // namespace imcom
