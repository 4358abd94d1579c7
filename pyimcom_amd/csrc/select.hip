// select.hip -- input-pixel selection of an output stamp on the device.
//
// Replaces OutStamp._process_input_stamps (reference src/pyimcom/coadd.py:886-977) with
// InStamp.make_selection (716-749): the nine InStamps around an output stamp contribute, in order, the pixels
// with (x - x_pivot)^2 + (y - y_pivot)^2 < radius^2 (a missing pivot coordinate drops its term; no pivot at all
// takes every pixel), concatenated into x / y / indata / exposure index and the ten running counts
// (inpix_cumsum).  Pure index work apart from the distance test, which is evaluated exactly as numpy does
// (square, add, compare; no fused multiply-add), so the selection is bit-identical.
#include "common.h"

namespace imcom {

__global__ __launch_bounds__(256) void select_pixels_kernel(const double *__restrict__ pool_x, const double *__restrict__ pool_y,
                                                            const float *__restrict__ pool_data, long npool, int n_inframe,
                                                            const int *__restrict__ pool_expo, const long *__restrict__ inst_off,
                                                            const int *__restrict__ inst_id, const double *__restrict__ pivot_x,
                                                            const double *__restrict__ pivot_y, double radius, int ldn,
                                                            double *__restrict__ x, double *__restrict__ y,
                                                            float *__restrict__ indata, int *__restrict__ expo,
                                                            int *__restrict__ cumsum, int *__restrict__ status)
{
    __shared__ int wcnt[4], total;
    const int s = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double r2 = __dmul_rn(radius, radius);
    if (threadIdx.x == 0) { total = 0; cumsum[s * 10] = 0; }
    __syncthreads();
    for (int idx = 0; idx < 9; idx++) {
        const int inst = inst_id[s * 9 + idx];
        if (inst >= 0) {
            const long o0 = inst_off[inst], o1 = inst_off[inst + 1];
            const double px = pivot_x[s * 9 + idx], py = pivot_y[s * 9 + idx];
            const bool all = (isnan(px) && isnan(py)) || isnan(radius);
            for (long c0 = o0; c0 < o1; c0 += 256) {
                const long i = c0 + threadIdx.x;
                bool in = false;
                double xi = 0.0, yi = 0.0;
                if (i < o1) {
                    xi = pool_x[i];
                    yi = pool_y[i];
                    double d = 0.0;
                    if (!isnan(px)) { const double dx = __dsub_rn(xi, px); d = __dadd_rn(d, __dmul_rn(dx, dx)); }
                    if (!isnan(py)) { const double dy = __dsub_rn(yi, py); d = __dadd_rn(d, __dmul_rn(dy, dy)); }
                    in = all || d < r2;
                }
                const unsigned long long mask = __ballot(in);
                if (lane == 0) wcnt[wave] = __popcll(mask);
                __syncthreads();
                int off = total;
                for (int w = 0; w < wave; w++) off += wcnt[w];
                off += __popcll(mask & ((1ull << lane) - 1ull));
                if (in && off < ldn) {
                    const long o = (long)s * ldn + off;
                    x[o] = xi;
                    y[o] = yi;
                    expo[o] = pool_expo[i];
                    for (int f = 0; f < n_inframe; f++) indata[((long)s * n_inframe + f) * ldn + off] = pool_data[f * npool + i];
                }
                __syncthreads();
                if (threadIdx.x == 0) total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) cumsum[s * 10 + idx + 1] = total;
    }
    __syncthreads();
    const int n = total;
    if (n > ldn) {
        if (threadIdx.x == 0) atomicMax(status, n);
        return;
    }
    for (int i = n + threadIdx.x; i < ldn; i += 256) {  // padding: zeros
        const long o = (long)s * ldn + i;
        x[o] = 0.0;
        y[o] = 0.0;
        expo[o] = 0;
        for (int f = 0; f < n_inframe; f++) indata[((long)s * n_inframe + f) * ldn + i] = 0.0f;
    }
}

}  // namespace imcom

using namespace imcom;

extern "C" int imcom_select_pixels(imcom_ctx *ctx, int batch, const double *pool_x, const double *pool_y, const float *pool_data,
                                   long npool, int n_inframe, const int *pool_expo, const long *inst_off, int n_inst,
                                   const int *inst_id, const double *pivot_x, const double *pivot_y, double radius, int ldn,
                                   double *x, double *y, float *indata, int *expo, int *cumsum, int memspace)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(batch >= 1 && npool >= 0 && n_inframe >= 1 && n_inst >= 1 && ldn >= 1, "bad sizes");
    IMCOM_REQUIRE(inst_off && inst_id && pivot_x && pivot_y && x && y && indata && expo && cumsum, "null pointer");
    IMCOM_REQUIRE(npool == 0 || (pool_x && pool_y && pool_data && pool_expo), "null pool pointer");
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t nb = (size_t)batch;
    size_t total = 65536;
    if (host)
        total += (size_t)npool * (16 + 4 * (size_t)n_inframe + 4) + (size_t)(n_inst + 1) * 8 + nb * 9 * (4 + 16) +
                 nb * ldn * (16 + 4 * (size_t)n_inframe + 4) + nb * 40 + 8192;
    IMCOM_TRY(ws_reserve(ctx, total));
    auto take = [&](size_t bytes) { return ws_take(ctx, bytes); };
    auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
        *dst = src;
        if (!host || bytes == 0) return IMCOM_OK;
        void *d = take(bytes);
        if (!d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        *dst = d;
        return IMCOM_OK;
    };
    const void *px_, *py_, *pd_, *pe_, *io_, *ii_, *vx_, *vy_;
    IMCOM_TRY(up(pool_x, (size_t)npool * 8, &px_));
    IMCOM_TRY(up(pool_y, (size_t)npool * 8, &py_));
    IMCOM_TRY(up(pool_data, (size_t)npool * 4 * n_inframe, &pd_));
    IMCOM_TRY(up(pool_expo, (size_t)npool * 4, &pe_));
    IMCOM_TRY(up(inst_off, (size_t)(n_inst + 1) * 8, &io_));
    IMCOM_TRY(up(inst_id, nb * 9 * 4, &ii_));
    IMCOM_TRY(up(pivot_x, nb * 9 * 8, &vx_));
    IMCOM_TRY(up(pivot_y, nb * 9 * 8, &vy_));
    double *x_d = x, *y_d = y;
    float *d_d = indata;
    int *e_d = expo, *c_d = cumsum;
    if (host) {
        x_d = (double *)take(nb * ldn * 8);
        y_d = (double *)take(nb * ldn * 8);
        d_d = (float *)take(nb * ldn * 4 * n_inframe);
        e_d = (int *)take(nb * ldn * 4);
        c_d = (int *)take(nb * 40);
        if (!x_d || !y_d || !d_d || !e_d || !c_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    }
    int *status = (int *)take(4);
    if (!status) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_HIP_CHECK(hipMemsetAsync(status, 0, 4, ctx->stream));
    { ProfScope ps(ctx, "select");
    hipLaunchKernelGGL(select_pixels_kernel, dim3(batch), dim3(256), 0, ctx->stream, (const double *)px_, (const double *)py_,
                       (const float *)pd_, npool, n_inframe, (const int *)pe_, (const long *)io_, (const int *)ii_, (const double *)vx_,
                       (const double *)vy_, radius, ldn, x_d, y_d, d_d, e_d, c_d, status); }
    IMCOM_TRY(check_launch("select_pixels_kernel"));
    int st = 0;
    IMCOM_HIP_CHECK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(x, x_d, nb * ldn * 8, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(y, y_d, nb * ldn * 8, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(indata, d_d, nb * ldn * 4 * n_inframe, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(expo, e_d, nb * ldn * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(cumsum, c_d, nb * 40, hipMemcpyDeviceToHost, ctx->stream));
    }
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    IMCOM_REQUIRE(st == 0, "a stamp selects %d input pixels, more than ldn=%d", st, ldn);
    return IMCOM_OK;
}
