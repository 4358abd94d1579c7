// partition.hip -- input pixels of one image binned into postage stamps on the device.
//
// Replaces the binning loop of InImage.partition_pixels (reference src/pyimcom/coadd.py:329-358): pixels are
// visited in a fixed order (sparse-grid cell by cell, row-major inside a cell), dropped when outside
// (pix_lower, pix_upper), masked, or in a stamp the block does not use, and appended to their stamp
// (j_st, i_st) = floor((pos - pix_lower) / n2).  The order inside every stamp is the visiting order, so this is a
// STABLE partition by stamp: a stable radix sort of (stamp key, visit index) pairs (hipCUB), segment starts from
// the sorted keys, then a gather.  The WCS evaluation that produces the positions stays on the host.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace imcom {

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
                                     double n2, double lo, double hi, long npix, unsigned int *__restrict__ keys,
                                     unsigned int *__restrict__ vals)
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
    vals[p] = (unsigned int)p;
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
    size_t temp_bytes = 0;
    unsigned int *dummy = nullptr;
    int end_bit = 1;
    while ((1L << end_bit) <= nkeys) end_bit++;
    if (npix > 0)
        IMCOM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, dummy, dummy, dummy, dummy, (int)npix, 0, end_bit, st));
    IMCOM_TRY(ws_reserve(ctx, (size_t)npix * 16 + temp_bytes + (size_t)(nkeys + 2) * 4 + 8192));
    unsigned int *k0 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4), *k1 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4);
    unsigned int *v0 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4), *v1 = (unsigned int *)ws_take(ctx, (size_t)npix * 4 + 4);
    void *temp = ws_take(ctx, temp_bytes + 16);
    unsigned int *start = (unsigned int *)ws_take(ctx, (size_t)(nkeys + 1) * 4);
    int *status = (int *)ws_take(ctx, 4);
    if (!k0 || !k1 || !v0 || !v1 || !temp || !start || !status) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_HIP_CHECK(hipMemsetAsync(start, 0xff, (size_t)(nkeys + 1) * 4, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(status, 0, 4, st));
    if (npix > 0) {
        const unsigned nb = (unsigned)((npix + 255) / 256);
        hipLaunchKernelGGL(partition_key_kernel, dim3(nb), dim3(256), 0, st, out_x, out_y, mask, use_instamps, nst, (double)n2, pix_lower,
                           pix_upper, npix, k0, v0);
        IMCOM_TRY(check_launch("partition_key_kernel"));
        IMCOM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k0, k1, v0, v1, (int)npix, 0, end_bit, st));
        hipLaunchKernelGGL(partition_bounds_kernel, dim3(nb), dim3(256), 0, st, k1, npix, nkeys, start);
    }
    hipLaunchKernelGGL(partition_count_kernel, dim3((nkeys + 255) / 256), dim3(256), 0, st, start, nkeys, npix, pix_count, npixmax, status);
    if (npix > 0) {
        const unsigned nb = (unsigned)((npix + 255) / 256);
        hipLaunchKernelGGL(partition_gather_kernel, dim3(nb), dim3(256), 0, st, k1, v1, start, npix, nkeys, npixmax, out_x, out_y, in_x, in_y,
                           y_idx, x_idx, y_val, x_val);
    }
    IMCOM_TRY(check_launch("partition kernels"));
    int sth = 0;
    IMCOM_HIP_CHECK(hipMemcpyAsync(&sth, status, 4, hipMemcpyDeviceToHost, st));
    IMCOM_HIP_CHECK(hipStreamSynchronize(st));
    IMCOM_REQUIRE(sth == 0, "a stamp receives %d pixels of this image, more than npixmax=%d (coadd.py:270-279)", sth, npixmax);
    return IMCOM_OK;
}
