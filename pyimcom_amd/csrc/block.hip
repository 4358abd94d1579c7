// block.hip -- block-level accumulation of finished stamps on the device.
//
// Replaces the map updates of Block._output_stamp_wrapper (reference src/pyimcom/coadd.py:1975-1993:
// out_map / UC / Sigma / kappa / Tsum / Neff maps += the stamp's n2f x n2f tile at
// rows (j_st-1)*n2 .., cols (i_st-1)*n2 .., neighbouring stamps overlapping by 2*fade pixels) and the
// boundary recovery of Block.build_output_file (coadd.py:2163-2181: OutStamp.trapezoid(..., recover_mode=True,
// pad_widths)).  All maps are float32 as in the reference (coadd.py:2031-2047); float64 per-stamp inputs
// (Tsum_inpix, Neff) are added in double and rounded once, as numpy's `f32 += f64` does.
#include "common.h"
#include "launchers.h"

namespace imcom {

template <typename SRC>
__global__ void block_accumulate_kernel(int npass, const int *__restrict__ list, const int *__restrict__ jst,
                                        const int *__restrict__ ist, int n2, int n2f, int nlayer,
                                        const SRC *__restrict__ src, float *__restrict__ dst, int nside)
{
    // grid: (pixels of one tile, stamps of this parity pass, layers)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n2f * n2f || (int)blockIdx.y >= npass) return;
    const int s = list[blockIdx.y], layer = blockIdx.z;
    const int r = t / n2f, c = t - r * n2f;
    const int row = (jst[s] - 1) * n2 + r, col = (ist[s] - 1) * n2 + c;
    float *d = dst + ((long)layer * nside + row) * nside + col;
    const SRC v = src[((long)s * nlayer + layer) * n2f * n2f + t];
    *d = (float)((double)*d + (double)v);
}

// Overlapping stamps, any visiting order (fade > 0).  A pixel of the block belongs to at most four stamps, one of each index
// parity (2 * fade <= n2).  Every stamp's tile is STORED -- in the dtype it arrives in -- into the layer of its parity
// (block_place_kernel; tiles of one parity never overlap, so nothing is summed there), and block_combine_kernel then adds a
// pixel's layers in the order in which the reference's loop meets their stamps (coadd.py:2056-2059: cells of 2 x 2 stamps from
// (j_st_min, i_st_min) on, row by row of cells; inside a cell dj outer, di inner), each addition rounded as numpy's
// `f32_map[...] += tile` rounds it.  The block's maps therefore do not depend on how the
// stamps were dealt to batches, passes or processes, and carry the reference's own rounding.
template <typename SRC>
__global__ void block_place_kernel(int batch, const int *__restrict__ jst, const int *__restrict__ ist, int n2, int n2f,
                                   int nlayer, const SRC *__restrict__ src, SRC *__restrict__ layers, int nside)
{
    // grid: (pixels of one tile, stamps, layers)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n2f * n2f || (int)blockIdx.y >= batch) return;
    const int s = blockIdx.y, layer = blockIdx.z;
    const int r = t / n2f, c = t - r * n2f;
    const int row = (jst[s] - 1) * n2 + r, col = (ist[s] - 1) * n2 + c;
    const int par = ((jst[s] & 1) << 1) | (ist[s] & 1);
    layers[(((long)par * nlayer + layer) * nside + row) * nside + col] = src[((long)s * nlayer + layer) * n2f * n2f + t];
}

template <typename SRC>
__global__ void block_combine_kernel(int n1P, int n2, int n2f, long nlayer, const SRC *__restrict__ layers,
                                     float *__restrict__ dst, int nside, int cells, int pj, int pi)
{
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= nlayer * nside * nside) return;
    const int col = (int)(t % nside), row = (int)((t / nside) % nside);
    const long layer = t / ((long)nside * nside);
    // 1-based stamps covering a row: (j-1) n2 <= row <= (j-1) n2 + n2f - 1
    auto first = [&](int r) { const int lo = r - n2f + 1; return (lo <= 0 ? 0 : (lo + n2 - 1) / n2) + 1; };
    auto last = [&](int r) { const int hi = r / n2 + 1; return hi < n1P ? hi : n1P; };
    // the visiting order of the reference's loop as a sort key: cells of 2 x 2 stamps starting at (pj, pi), row by row of cells,
    // inside a cell dj outer, di inner (coadd.py:2056-2059); cells = 0: plain rows (j outer, i inner)
    int key[4], nk = 0;
    SRC val[4];
    for (int j = first(row); j <= last(row); j++)
        for (int i = first(col); i <= last(col); i++) {
            const int par = ((j & 1) << 1) | (i & 1);
            const int k = cells ? (((((j - pj) >> 1) + 1) * 32768 + (((i - pi) >> 1) + 1)) * 2 + ((j - pj) & 1)) * 2 + ((i - pi) & 1) : j * 32768 + i;
            int q = nk++;
            const SRC v = layers[(((long)par * nlayer + layer) * nside + row) * nside + col];
            while (q > 0 && key[q - 1] > k) { key[q] = key[q - 1]; val[q] = val[q - 1]; q--; }
            key[q] = k;
            val[q] = v;
        }
    float d = 0.0f;
    for (int q = 0; q < nk; q++) d = (float)((double)d + (double)val[q]);
    dst[t] = d;
}

// coadd.py:1284-1292 with pad widths (1262-1267): divide the 2f boundary rows/cols by the taper, B, T, L, R
__global__ void trapezoid_recover_kernel(float *__restrict__ maps, long nmaps, int ny, int nx, int fade, int pb, int pt,
                                         int pl, int pr)
{
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= nmaps * ny * nx) return;
    const int x = (int)(t % nx), y = (int)((t / nx) % ny);
    const int fk2 = 2 * fade;
    const double two_pi = 6.283185307179586;
    auto taper = [&](int k) { double s = (double)(k + 1) / (fk2 + 1); return s - sin(two_pi * s) / two_pi; };
    float v = maps[t];
    const int it = ny - pt - 1, ir = nx - pr - 1;
    if (y >= pb && y < pb + fk2) v = (float)((double)v / taper(y - pb));
    if (y <= it && y > it - fk2) v = (float)((double)v / taper(it - y));
    if (x >= pl && x < pl + fk2) v = (float)((double)v / taper(x - pl));
    if (x <= ir && x > ir - fk2) v = (float)((double)v / taper(ir - x));
    maps[t] = v;
}

}  // namespace imcom

using namespace imcom;

extern "C" int imcom_block_accumulate(imcom_ctx *ctx, int batch, const int *jst_host, const int *ist_host, int n2, int fade,
                                      int nlayer, const void *src, int src_is_f64, float *dst, int nside_pf)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(batch >= 1 && jst_host && ist_host && src && dst && n2 >= 1 && fade >= 0 && nlayer >= 1, "bad arguments");
    const int n2f = n2 + 2 * fade;
    IMCOM_REQUIRE(2 * fade <= n2, "fade=%d: neighbouring stamps must overlap by less than a stamp", fade);
    for (int s = 0; s < batch; s++)
        IMCOM_REQUIRE(jst_host[s] >= 1 && ist_host[s] >= 1 && jst_host[s] * n2 + 2 * fade <= nside_pf && ist_host[s] * n2 + 2 * fade <= nside_pf,
                      "stamp %d (%d,%d) outside the block", s, jst_host[s], ist_host[s]);
    IMCOM_TRY(ws_reserve(ctx, (size_t)batch * 12 + 1024));
    int *jd = (int *)ws_take(ctx, (size_t)batch * 4), *id = (int *)ws_take(ctx, (size_t)batch * 4), *ld = (int *)ws_take(ctx, (size_t)batch * 4);
    // stamps of equal index parity never overlap: four ordered passes make the overlap sums deterministic
    std::vector<int> order;
    int cnt[4] = {0, 0, 0, 0};
    for (int p = 0; p < 4; p++)
        for (int s = 0; s < batch; s++)
            if ((((jst_host[s] & 1) << 1) | (ist_host[s] & 1)) == p) { order.push_back(s); cnt[p]++; }
    // through the pinned ring: the copies read it, not the caller's arrays, so the call returns without draining the stream
    IMCOM_TRY(upload(ctx, jd, jst_host, (size_t)batch));
    IMCOM_TRY(upload(ctx, id, ist_host, (size_t)batch));
    IMCOM_TRY(upload(ctx, ld, order.data(), (size_t)batch));
    ProfScope ps(ctx, "block_acc");
    int off = 0;
    for (int p = 0; p < 4; p++) {
        if (cnt[p] == 0) continue;
        dim3 grid((n2f * n2f + 255) / 256, cnt[p], nlayer);
        if (src_is_f64)
            hipLaunchKernelGGL(block_accumulate_kernel<double>, grid, dim3(256), 0, ctx->stream, cnt[p], ld + off, jd, id, n2, n2f, nlayer,
                               (const double *)src, dst, nside_pf);
        else
            hipLaunchKernelGGL(block_accumulate_kernel<float>, grid, dim3(256), 0, ctx->stream, cnt[p], ld + off, jd, id, n2, n2f, nlayer,
                               (const float *)src, dst, nside_pf);
        off += cnt[p];
    }
    return check_launch("block_accumulate_kernel");
}

extern "C" int imcom_block_place(imcom_ctx *ctx, int batch, const int *jst_host, const int *ist_host, int n2, int fade, int nlayer,
                                 const void *src, int src_is_f64, void *layers, int nside_pf)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(batch >= 1 && jst_host && ist_host && src && layers && n2 >= 1 && fade >= 0 && nlayer >= 1, "bad arguments");
    const int n2f = n2 + 2 * fade;
    IMCOM_REQUIRE(2 * fade <= n2, "fade=%d: neighbouring stamps must overlap by less than a stamp", fade);
    for (int s = 0; s < batch; s++)
        IMCOM_REQUIRE(jst_host[s] >= 1 && ist_host[s] >= 1 && jst_host[s] * n2 + 2 * fade <= nside_pf && ist_host[s] * n2 + 2 * fade <= nside_pf,
                      "stamp %d (%d,%d) outside the block", s, jst_host[s], ist_host[s]);
    IMCOM_TRY(ws_reserve(ctx, (size_t)batch * 8 + 1024));
    int *jd = (int *)ws_take(ctx, (size_t)batch * 4), *id = (int *)ws_take(ctx, (size_t)batch * 4);
    IMCOM_TRY(upload(ctx, jd, jst_host, (size_t)batch));
    IMCOM_TRY(upload(ctx, id, ist_host, (size_t)batch));
    ProfScope ps(ctx, "block_acc");
    dim3 grid((n2f * n2f + 255) / 256, batch, nlayer);
    if (src_is_f64)
        hipLaunchKernelGGL(block_place_kernel<double>, grid, dim3(256), 0, ctx->stream, batch, jd, id, n2, n2f, nlayer, (const double *)src,
                           (double *)layers, nside_pf);
    else
        hipLaunchKernelGGL(block_place_kernel<float>, grid, dim3(256), 0, ctx->stream, batch, jd, id, n2, n2f, nlayer, (const float *)src,
                           (float *)layers, nside_pf);
    return check_launch("block_place_kernel");
}

extern "C" int imcom_block_combine(imcom_ctx *ctx, int n1P, int n2, int fade, long nlayer, const void *layers, int src_is_f64, float *dst,
                                   int nside_pf, int order, int j_st_min, int i_st_min)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(n1P >= 1 && n1P < 32000 && n2 >= 1 && fade >= 0 && nlayer >= 1 && layers && dst && (order == 0 || order == 1) && j_st_min >= 1 && i_st_min >= 1,
                  "bad arguments");
    const int cells = order, pj = j_st_min & 1, pi = i_st_min & 1;  // (only the parity of the window's origin decides which stamps share a cell)
    IMCOM_REQUIRE(2 * fade <= n2 && nside_pf == n1P * n2 + 2 * fade, "nside_pf = %d is not n1P * n2 + 2 fade (or 2 fade > n2)", nside_pf);
    const long tot = nlayer * nside_pf * nside_pf;
    if (src_is_f64)
        hipLaunchKernelGGL(block_combine_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n1P, n2, n2 + 2 * fade,
                           nlayer, (const double *)layers, dst, nside_pf, cells, pj, pi);
    else
        hipLaunchKernelGGL(block_combine_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n1P, n2, n2 + 2 * fade,
                           nlayer, (const float *)layers, dst, nside_pf, cells, pj, pi);
    return check_launch("block_combine_kernel");
}

extern "C" int imcom_trapezoid_recover_f32(imcom_ctx *ctx, float *maps, long nmaps, int ny, int nx, int fade, int pad_b, int pad_t,
                                           int pad_l, int pad_r)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(maps && nmaps >= 0 && ny >= 1 && nx >= 1 && fade >= 0 && pad_b >= 0 && pad_t >= 0 && pad_l >= 0 && pad_r >= 0, "bad arguments");
    if (fade == 0 || nmaps == 0) return IMCOM_OK;
    const long tot = nmaps * ny * nx;
    hipLaunchKernelGGL(trapezoid_recover_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, maps, nmaps, ny, nx, fade,
                       pad_b, pad_t, pad_l, pad_r);
    return check_launch("trapezoid_recover_kernel");
}

// Block.compress_map (reference src/pyimcom/coadd.py:2087-2138): float32 map -> (u)int16 of coef * log10(map),
//   clip(floor(coef * log10(clip(map, 1e-32, None)) + 0.5), a_min, a_max)
// evaluated in float32 as numpy evaluates it for a float32 map (the python scalars are weakly typed); log10 is
// rounded correctly to float32 (computed in double).  NaN inputs map to a_min.
namespace imcom {
__global__ void compress_map_kernel(const float *__restrict__ map, long count, float coef, float a_min, float a_max,
                                    short *__restrict__ out)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float x = fmaxf(map[i], 1e-32f);
    const float l = (float)log10((double)x);
    float v = floorf(__fadd_rn(__fmul_rn(coef, l), 0.5f));
    v = fminf(fmaxf(v, a_min), a_max);
    if (!(v == v)) v = a_min;
    const int iv = (int)v;
    out[i] = (short)(unsigned short)(iv & 0xffff);  // two's complement bits serve both int16 and uint16
}
}  // namespace imcom

extern "C" int imcom_compress_map_f32(imcom_ctx *ctx, const float *map, long count, int coef, int is_unsigned, void *out)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(map && out && count >= 0, "bad arguments");
    if (count == 0) return IMCOM_OK;
    hipLaunchKernelGGL(compress_map_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, map, count, (float)coef,
                       is_unsigned ? 0.0f : -32768.0f, is_unsigned ? 65535.0f : 32767.0f, (short *)out);
    return check_launch("compress_map_kernel");
}
