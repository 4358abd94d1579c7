// partition.hip -- input pixels of one image binned into postage stamps on the device.
//
// Replaces the binning loop of InImage.partition_pixels (reference src/pyimcom/coadd.py:329-358): pixels are
// visited in a fixed order (sparse-grid cell by cell, row-major inside a cell), dropped when outside
// (pix_lower, pix_upper), masked, or in a stamp the block does not use, and appended to their stamp
// (j_st, i_st) = floor((pos - pix_lower) / n2).  The order inside every stamp is the visiting order, so this is a
// STABLE partition by stamp: a stable least-significant-digit radix sort of (stamp key, visit index) pairs (own kernels
// below: 6-bit digits, two passes for up to 4095 stamps), segment starts from the sorted keys, then a gather.  The WCS
// evaluation that produces the positions stays on the host.
#include "common.h"

namespace imcom {

// ---- stable LSD radix sort of 32-bit keys with their 32-bit payloads, 6 bits per pass --------------------------------
// A workgroup of 256 threads owns RS_TILE consecutive elements; wave w the RS_TILE / 4 consecutive ones from w * RS_TILE / 4,
// lane-consecutive in every step of 64, so "earlier in the array" = (earlier wave, earlier step, lower lane).
constexpr int RS_BITS = 6, RS_BINS = 1 << RS_BITS, RS_STEPS = 8, RS_TILE = 4 * 64 * RS_STEPS;

// hist[d * nblk + b] = number of elements of workgroup b with digit d (digit-major: one exclusive scan over the whole array
// then gives every (digit, workgroup) its first output position)
__global__ __launch_bounds__(256) void radix_hist_kernel(const unsigned int *__restrict__ keys, long n, int shift, int nblk,
                                                         unsigned int *__restrict__ hist)
{
    __shared__ unsigned int h[RS_BINS];
    if (threadIdx.x < RS_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * RS_TILE;
    for (int q = threadIdx.x; q < RS_TILE; q += 256) {
        const long p = base + q;
        if (p < n) atomicAdd(&h[(keys[p] >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RS_BINS) hist[(long)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `count` values in place, one workgroup (the array has 64 x nblk entries: a few hundred thousand)
__global__ __launch_bounds__(1024) void radix_scan_kernel(unsigned int *__restrict__ a, long count)
{
    __shared__ unsigned int part[1024];
    const int t = threadIdx.x;
    const long per = (count + 1023) / 1024, lo = (long)t * per, hi = lo + per < count ? lo + per : count;
    unsigned int sum = 0;
    for (long i = lo; i < hi; i++) sum += a[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned int v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned int run = part[t] - sum;  // exclusive prefix of this thread's range
    for (long i = lo; i < hi; i++) { const unsigned int v = a[i]; a[i] = run; run += v; }
}

// vals_in == nullptr: the payload is the element's index (first pass)
__global__ __launch_bounds__(256) void radix_scatter_kernel(const unsigned int *__restrict__ keys_in, const unsigned int *__restrict__ vals_in,
                                                            long n, int shift, int nblk, const unsigned int *__restrict__ base,
                                                            unsigned int *__restrict__ keys_out, unsigned int *__restrict__ vals_out)
{
    __shared__ unsigned int cnt[4][RS_BINS];  // running count per wave and digit, then the wave's exclusive offset
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cnt[tid >> 6][tid & 63] = 0;
    __syncthreads();
    const long p0 = (long)blockIdx.x * RS_TILE + (long)wave * (RS_TILE / 4);
    const unsigned long long lt = (1ULL << lane) - 1;
    unsigned int key[RS_STEPS], rank[RS_STEPS];
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const long p = p0 + s * 64 + lane;
        const bool on = p < n;
        key[s] = on ? keys_in[p] : 0xffffffffu;
        const unsigned int d = (key[s] >> shift) & (RS_BINS - 1);
        unsigned long long peers = __ballot(on);  // lanes that carry an element with my digit
#pragma unroll
        for (int b = 0; b < RS_BITS; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        rank[s] = cnt[wave][d] + __popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();  // every lane has read the running count before the leader advances it
        if (on && (peers & lt) == 0) cnt[wave][d] += __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < RS_BINS) {  // totals of the four waves -> exclusive offsets among them
        unsigned int run = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const unsigned int c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const long p = p0 + s * 64 + lane;
        if (p < n) {
            const unsigned int d = (key[s] >> shift) & (RS_BINS - 1);
            const unsigned int o = base[(long)d * nblk + blockIdx.x] + cnt[wave][d] + rank[s];
            keys_out[o] = key[s];
            vals_out[o] = vals_in ? vals_in[p] : (unsigned int)p;
        }
    }
}

// sorts (k0, index) by the low `bits` bits of the keys; the result is in (*ko, *vo), which point into the buffers given
static int radix_sort_pairs(imcom_ctx *ctx, unsigned int *k0, unsigned int *k1, unsigned int *v0, unsigned int *v1, long n, int bits,
                            unsigned int *hist, unsigned int **ko, unsigned int **vo)
{
    const int nblk = (int)((n + RS_TILE - 1) / RS_TILE);
    unsigned int *kin = k0, *kout = k1, *vin = nullptr, *vout = v0;
    for (int shift = 0; shift < bits; shift += RS_BITS) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(256), 0, ctx->stream, kin, n, shift, nblk, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, hist, (long)RS_BINS * nblk);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(256), 0, ctx->stream, kin, vin, n, shift, nblk, hist, kout, vout);
        IMCOM_TRY(check_launch("radix sort pass"));
        unsigned int *t = kin; kin = kout; kout = t;
        vin = vout;
        vout = vout == v0 ? v1 : v0;
    }
    *ko = kin;
    *vo = vin;
    return IMCOM_OK;
}

// Python's float floor division a // b for b > 0 (numpy npy_divmod): exact remainder first
__device__ inline double py_floordiv(double a, double b)
{
    const double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0 && mod < 0.0) div -= 1.0;
    double fl = floor(div);
    if (div - fl > 0.5) fl += 1.0;
    return fl;
}

__global__ void partition_key_kernel(const double *__restrict__ ox, const double *__restrict__ oy,
                                     const unsigned char *__restrict__ mask, const unsigned char *__restrict__ use, int nst,
                                     double n2, double lo, double hi, long npix, unsigned int *__restrict__ keys)
{
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= npix) return;
    unsigned int key = (unsigned int)(nst * nst);  // dropped pixels sort behind every stamp
    const double x = ox[p], y = oy[p];
    if (lo < x && x < hi && lo < y && y < hi && (!mask || mask[p])) {
        const int ist = (int)py_floordiv(x - lo, n2), jst = (int)py_floordiv(y - lo, n2);
        if (ist >= 0 && ist < nst && jst >= 0 && jst < nst && use[jst * nst + ist]) key = (unsigned int)(jst * nst + ist);
    }
    keys[p] = key;
}

// start[k] = first sorted position of key k (npix where absent); one thread per sorted position
__global__ void partition_bounds_kernel(const unsigned int *__restrict__ keys, long npix, int nkeys, unsigned int *__restrict__ start)
{
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const unsigned int k = keys[p];
    if (p == 0 || keys[p - 1] != k) {
        if (k <= (unsigned int)nkeys) start[k] = (unsigned int)p;
    }
}

__global__ void partition_count_kernel(const unsigned int *__restrict__ start, int nkeys, long npix, unsigned int *__restrict__ count,
                                       int npixmax, int *__restrict__ status)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkeys) return;
    unsigned int s = start[k], e = (unsigned int)npix;
    if (s == 0xffffffffu) { count[k] = 0; return; }
    for (int q = k + 1; q <= nkeys; q++)
        if (start[q] != 0xffffffffu) { e = start[q]; break; }
    count[k] = e - s;
    if ((long)(e - s) > npixmax) atomicMax(status, (int)(e - s));
}

__global__ void partition_gather_kernel(const unsigned int *__restrict__ keys, const unsigned int *__restrict__ vals,
                                        const unsigned int *__restrict__ start, long npix, int nkeys, int npixmax,
                                        const double *__restrict__ ox, const double *__restrict__ oy,
                                        const unsigned short *__restrict__ ix, const unsigned short *__restrict__ iy,
                                        unsigned short *__restrict__ y_idx, unsigned short *__restrict__ x_idx,
                                        double *__restrict__ y_val, double *__restrict__ x_val)
{
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const unsigned int k = keys[p];
    if (k >= (unsigned int)nkeys) return;
    const long slot = p - start[k];
    if (slot >= npixmax) return;
    const unsigned int v = vals[p];
    const long o = (long)k * npixmax + slot;
    y_idx[o] = iy[v];
    x_idx[o] = ix[v];
    y_val[o] = oy[v];
    x_val[o] = ox[v];
}

}  // namespace imcom

using namespace imcom;

extern "C" int imcom_partition_pixels(imcom_ctx *ctx, long npix, const double *out_x, const double *out_y, const unsigned short *in_x,
                                      const unsigned short *in_y, const unsigned char *mask, const unsigned char *use_instamps, int nst,
                                      int n2, double pix_lower, double pix_upper, int npixmax, unsigned short *y_idx,
                                      unsigned short *x_idx, double *y_val, double *x_val, unsigned int *pix_count)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(npix >= 0 && npix < (1L << 32) - 2 && nst >= 1 && n2 >= 1 && npixmax >= 1, "bad sizes");
    IMCOM_REQUIRE(use_instamps && y_idx && x_idx && y_val && x_val && pix_count, "null pointer");
    IMCOM_REQUIRE(npix == 0 || (out_x && out_y && in_x && in_y), "null input pointer");
    const int nkeys = nst * nst;
    hipStream_t st = ctx->stream;
    int bits = 1;
    while ((1L << bits) <= nkeys) bits++;  // keys 0 .. nkeys (nkeys = dropped)
    const long nblk = (npix + RS_TILE - 1) / RS_TILE;
    IMCOM_TRY(ws_reserve(ctx, (size_t)npix * 16 + (size_t)nblk * RS_BINS * 4 + (size_t)(nkeys + 2) * 4 + 8192));
    unsigned int *k0 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4), *k1 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4);
    unsigned int *v0 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4), *v1 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4);
    unsigned int *hist = (unsigned int *)ws_take(ctx, (size_t)nblk * RS_BINS * 4 + 16);
    unsigned int *start = (unsigned int *)ws_take(ctx, (size_t)(nkeys + 1) * 4);
    int *status = (int *)ws_take(ctx, 4);
    if (!k0 || !k1 || !v0 || !v1 || !hist || !start || !status) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    unsigned int *ks = k0, *vs = v0;  // the sorted pairs
    IMCOM_HIP_CHECK(hipMemsetAsync(start, 0xff, (size_t)(nkeys + 1) * 4, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(status, 0, 4, st));
    if (npix > 0) {
        const unsigned nb = (unsigned)((npix + 255) / 256);
        hipLaunchKernelGGL(partition_key_kernel, dim3(nb), dim3(256), 0, st, out_x, out_y, mask, use_instamps, nst, (double)n2, pix_lower,
                           pix_upper, npix, k0);
        IMCOM_TRY(check_launch("partition_key_kernel"));
        IMCOM_TRY(radix_sort_pairs(ctx, k0, k1, v0, v1, npix, bits, hist, &ks, &vs));
        hipLaunchKernelGGL(partition_bounds_kernel, dim3(nb), dim3(256), 0, st, ks, npix, nkeys, start);
    }
    hipLaunchKernelGGL(partition_count_kernel, dim3((nkeys + 255) / 256), dim3(256), 0, st, start, nkeys, npix, pix_count, npixmax, status);
    if (npix > 0) {
        const unsigned nb = (unsigned)((npix + 255) / 256);
        hipLaunchKernelGGL(partition_gather_kernel, dim3(nb), dim3(256), 0, st, ks, vs, start, npix, nkeys, npixmax, out_x, out_y, in_x, in_y,
                           y_idx, x_idx, y_val, x_val);
    }
    IMCOM_TRY(check_launch("partition kernels"));
    int sth = 0;
    IMCOM_HIP_CHECK(hipMemcpyAsync(&sth, status, 4, hipMemcpyDeviceToHost, st));
    IMCOM_HIP_CHECK(hipStreamSynchronize(st));
    IMCOM_REQUIRE(sth == 0, "a stamp receives %d pixels of this image, more than npixmax=%d (coadd.py:270-279)", sth, npixmax);
    return IMCOM_OK;
}
