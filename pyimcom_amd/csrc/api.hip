// api.hip -- extern "C" entry points of libimcom_hip.so (see include/imcom_hip.h) and the host-side
// orchestration of the batched blocked Cholesky solve.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>

#include "common.h"
#include "launchers.h"

namespace imcom {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ws_reserve(imcom_ctx *ctx, size_t bytes)
{
    ctx->ws_used = 0;
    ctx->ws_limit = 0;
    ctx->ws_need = bytes;
    if (bytes <= ctx->ws_bytes) return IMCOM_OK;
    if (ctx->ws_external) {  // the caller owns device memory: say what is needed (imcom_ctx_workspace_needed) and let it decide
        set_error("caller-provided workspace of %zu bytes: this call needs %zu (imcom_ctx_set_workspace)", ctx->ws_bytes, bytes);
        return IMCOM_ERR_NOMEM;
    }
    if (ctx->ws) {
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        IMCOM_HIP_CHECK(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    const size_t want = align_up(bytes + (bytes >> 3), (size_t)1 << 21);
    hipError_t e = hipMalloc((void **)&ctx->ws, want);
    if (e != hipSuccess) {
        set_error("device workspace of %zu bytes: %s", want, hipGetErrorString(e));
        return IMCOM_ERR_NOMEM;
    }
    ctx->ws_bytes = want;
    return IMCOM_OK;
}

void *ws_take(imcom_ctx *ctx, size_t bytes)
{
    const size_t off = align_up(ctx->ws_used, 256);
    const size_t end = ctx->ws_limit ? std::min(ctx->ws_limit, ctx->ws_bytes) : ctx->ws_bytes;
    if (off + bytes > end) return nullptr;  // callers reserve the exact total first
    ctx->ws_used = off + bytes;
    return ctx->ws + off;
}

struct WsPlan {
    size_t total = 0;
    size_t add(size_t bytes) { total = align_up(total, 256) + bytes; return total; }
};

ProfScope::ProfScope(imcom_ctx *c, const char *fam, long n, bool fine) : ctx(c), family(fam), launches(n)
{
    if (!ctx->profile || (fine && !ctx->profile_fine)) return;
    auto get = [&]() {
        hipEvent_t e;
        if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else hipEventCreate(&e);
        return e;
    };
    start = get();
    stop = get();
    hipEventRecord(start, ctx->stream);
}

ProfScope::~ProfScope()
{
    if (!ctx->profile || !start) return;
    hipEventRecord(stop, ctx->stream);
    ctx->pending.push_back({family, start, stop, launches});
}

int profile_collect(imcom_ctx *ctx)
{
    if (ctx->pending.empty()) return IMCOM_OK;
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        IMCOM_HIP_CHECK(hipEventElapsedTime(&ms, p.start, p.stop));
        auto &slot = ctx->prof[p.family];
        slot.ms += ms;
        slot.launches += p.launches;
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->pending.clear();
    return IMCOM_OK;
}

int pin_reserve(imcom_ctx *ctx, size_t bytes)
{
    if (ctx->pin && bytes <= ctx->pin_bytes) return IMCOM_OK;
    if (ctx->pin) { IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream)); IMCOM_HIP_CHECK(hipHostFree(ctx->pin)); ctx->pin = nullptr; }
    IMCOM_HIP_CHECK(hipHostMalloc((void **)&ctx->pin, bytes, hipHostMallocDefault));
    ctx->pin_bytes = bytes;
    ctx->pin_used = 0;
    return IMCOM_OK;
}

// -------------------------------------------------------------------------------------------------
// Batched blocked Cholesky solve on padded, device-resident operands.
//   A  [batch][Np][Np]   identity-padded, never modified
//   Bt [batch][Np][mp]   input-pixel-major -B/2, zero padded
// Produces Tt (float32 [batch][Np][mp]) and the per-pixel maps.
constexpr int REPAIR_GROUP = 32;  // failed stamps whose smallest eigenvalues are computed in one batch of the eigensolver (repair path, small matrices / fallback)
constexpr int CHOL_MAXNV = 8;  // kappa nodes of the multi-kappa Cholesky kernel (launch_multi's MAXNV; 3 nv diagonal increments <= MAX_INC)
static_assert(3 * CHOL_MAXNV <= MAX_INC_HOST, "diagonal increments of the repair sequence");
constexpr int LMIN_P = NB;        // vectors of the subspace iteration for the smallest eigenvalue (one tile column of the solves)
constexpr int LMIN_MIN_N = 1024;  // smaller matrices go through the eigensolver itself (cheap there; the block needs n >> LMIN_P)

// Split-K for small batches: with fewer than ~256 tiles per launch (the kernel-class seam hands over one stamp: 18) the K
// loop of every tile is dealt to up to 8 workgroups, so that a launch has ~512 of them (gemm_f64.hip).
static int splitk_parts(int batch, int tiles)
{
    const long wgs = (long)batch * tiles;
    if (wgs >= 256) return 1;
    return (int)std::min<long>(8, std::max<long>(1, 512 / std::max<long>(wgs, 1)));
}

static size_t splitk_bytes(int batch, int Np, int mp)
{
    const int tiles = std::max(Np, mp) / NB;
    return splitk_parts(batch, std::min(Np, mp) / NB) > 1 ? (size_t)batch * tiles * 8 * NB * NB * 8 + 256 : 0;
}

// Multi-kappa on a small batch: the nv factorisations and solves of a stamp are independent, so they are run as ONE pass over
// nv x batch "stamps" (node-major: exactly the layout of Y) instead of nv passes that each fill a fraction of the chip.
static int nodes_per_pass(int batch, int mp, int nv)
{
    return (nv > 1 && (long)batch * nv * (mp / NB) <= 1024) ? nv : 1;
}

static bool lmin_subspace_enabled()
{
    static const bool off = getenv("IMCOM_LMIN") && strcmp(getenv("IMCOM_LMIN"), "eigh") == 0;  // (cross-check: the eigensolver for every repair)
    return !off;
}

// workspace of lambda_min_subspace on top of the factorisation's own L / Dinv / dshift
static size_t lmin_ws_bytes(int batch, int Np)
{
    if (!lmin_subspace_enabled() || Np < LMIN_MIN_N) return 0;
    WsPlan p;
    for (int q = 0; q < 3; q++) p.add((size_t)batch * Np * LMIN_P * 8);      // X, Y, Z = A X
    for (int q = 0; q < 4; q++) p.add((size_t)batch * LMIN_P * LMIN_P * 8);  // G, its inverse Cholesky factor, H = X^T A X, H's eigenvectors
    p.add((size_t)batch * LMIN_P * 8);                                       // eigenvalues of H
    p.add((size_t)batch * LMIN_RESID_GROUPS * 2 * 8);                        // residuals of the two lowest Ritz pairs, by row group
    p.add((size_t)batch * 8 * NB * NB * 8);                                  // split-K partial products of the 128-column solves
    p.add((size_t)batch * 8);                                                // largest diagonal entry
    for (int q = 0; q < 4; q++) p.add((size_t)batch * 4);                    // want, ones, flags of the Gram factorisations, masked nblk
    p.add(eigh_ws_bytes(batch, LMIN_P, true));
    return p.total + 8192;
}

static size_t chol_core_bytes(int batch, int Np, int m, int mp, int nv)
{
    WsPlan p;
    const size_t nb = Np / NB, eb = (size_t)batch * nodes_per_pass(batch, mp, nv);
    p.add(eb * Np * Np * 8);                     // L
    p.add(eb * nb * NB * NB * 8);                // Dinv
    p.add((size_t)nv * batch * Np * mp * 8);     // Y / X per node
    p.add(eb * Np * 8);                          // dshift
    p.add(eb * 4 * 3 + (size_t)batch * 4 * nv);  // n, nblk, ninc, fail[nv]
    p.add((size_t)batch * 4 * 4);                // attempt masks: stamps to factor, their block counts, block counts of the solves, solved flags
    p.add(eb * MAX_INC_HOST * 8);                // inc
    p.add((size_t)batch * 8 * 2);                // kap, C
    p.add((size_t)nv * 8);                       // kappaC
    if (nv > 1) {
        p.add((size_t)batch * m * nv * 8);       // Dp
        p.add((size_t)batch * m * nv * nv * 8);  // Npq
        p.add((size_t)batch * m * nv * 8);       // W
    }
    // repair path (lakernel.py:262-279): the smallest eigenvalue of every failed stamp's A -- the subspace iteration on the stamps in
    // place (lambda_min_subspace), the eigensolver on up to REPAIR_GROUP gathered copies for small matrices and as its fallback
    const int rg = std::min(batch, REPAIR_GROUP);
    p.add(std::max(eigh_ws_bytes(rg, Np, false) + (size_t)rg * Np * Np * 8 * (eigh_uses_jacobi() ? 2 : 1) + (size_t)rg * Np * 8 + 4096, lmin_ws_bytes(batch, Np)));
    p.add(splitk_bytes((int)eb, Np, mp));
    if (nv == 1) p.add((size_t)batch * 2 * nb * mp * 8 * 2);  // per-block-row column sums of Y^2 and X^2
    return p.total + 4096;
}

// smallest eigenvalues of the matrices A[idx[q]] (leading n x n of the padded ones), computed on the device: the stamps are
// gathered into one batch of the eigensolver (eigenvalues only).  (One call per failed stamp, as the first version did, made a
// batch in which every factorisation fails -- PSFs cut too tight -- take 0.3 s per stamp.)
static int lambda_min_group(imcom_ctx *ctx, const double *A, const int *n_host, int Np, const int *idx, int count, double *w0)
{
    const size_t mark = ctx->ws_used;
    const size_t mat = (size_t)Np * Np;
    double *G = (double *)ws_take(ctx, (size_t)count * mat * 8);
    double *lam = (double *)ws_take(ctx, (size_t)count * Np * 8);
    double *Q = eigh_uses_jacobi() ? (double *)ws_take(ctx, (size_t)count * mat * 8) : nullptr;  // eigenvalues only otherwise
    if (!G || !lam || (eigh_uses_jacobi() && !Q)) { set_error("internal: workspace (repair)"); return IMCOM_ERR_NOMEM; }
    std::vector<int> ng(count);
    for (int q = 0; q < count; q++) {
        ng[q] = n_host[idx[q]];
        IMCOM_HIP_CHECK(hipMemcpyAsync(G + (size_t)q * mat, A + (size_t)idx[q] * mat, mat * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (Q) IMCOM_HIP_CHECK(hipMemsetAsync(Q, 0, (size_t)count * mat * 8, ctx->stream));
    int rc = eigh_device(ctx, count, ng.data(), Np, G, Np, (long)mat, lam, Np, Q, Np, (long)mat, nullptr);
    if (rc == IMCOM_OK) {
        hipError_t e = hipMemcpy2DAsync(w0, 8, lam, (size_t)Np * 8, 8, count, hipMemcpyDeviceToHost, ctx->stream);  // lam[q][0]
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { set_error("repair: %s", hipGetErrorString(e)); rc = IMCOM_ERR_HIP; }
    }
    ctx->ws_used = mark;
    return rc;
}

// The smallest eigenvalue of the stamps `idx` WITHOUT an eigendecomposition (the reference's repair needs w[0] of eigh(A) and nothing
// else, lakernel.py:266-268).  A production stamp (configs/paper4_configs/H158_Chol_benchmark.json: N ~ 6.2k pixels) has thousands of
// eigenvalues within 1e-5 |A| of zero and a smallest one of either sign; the tridiagonalisation of a 6.2k matrix costs 4 N^3 / 3 flops at
// memory speed (95 ms per stamp), the kernels this library is fast at are the blocked Cholesky and the triangular solves.  So:
//   1. a shift sigma with A + sigma I positive definite (trial factorisations, sigma x 8 per failure; the first trial is 4 x the
//      increment the failed factorisation had);
//   2. subspace iteration with the inverse, X <- orth((A + sigma I)^-1 X), on a block of 128 vectors = one tile column of the solve
//      kernels (CholQR twice per step: Gram matrix on the tile engine, its Cholesky factor and the factor's inverse from the diagonal-block
//      kernel), then Rayleigh-Ritz with A ITSELF: theta = lambda_min(X^T A X) >= lambda_min(A), accurate to eps |A| once the block has
//      converged -- the accuracy class of LAPACK's own w[0] -- with the residuals of the two lowest Ritz pairs as the error bound;
//   3. re-factor at sigma' = |theta| (1 + eta) just above the estimate (eta from that bound; a sigma' that is not above
//      |lambda_min| makes the factorisation fail: eta x 8), where the block converges by 1e-3 and more per step; stop when the bound is
//      below 1e-11 |theta| (or two successive Rayleigh-Ritz values agree to that).
// On a paper4 stamp: 2 factorisations, 6 + 3 steps, 2 Rayleigh-Ritz steps = 5.4 ms, theta within 2e-13 of LAPACK's w[0].
// Everything runs on the stamps in place (L, Dinv, dshift of the caller's factorisation; stamps that are not wanted have no blocks in
// these launches).  factor(shift, mask, fail): L L^T = A + shift[s] I for the stamps of mask, fail[s] != 0 where that is not positive
// definite; solve(mask, X, Y): Y = (L L^T)^-1 X on LMIN_P columns.  ok[s] = 0: no answer (the caller takes the eigensolver).
static int lambda_min_subspace(imcom_ctx *ctx, int batch, const int *n_host, const int *n_dev, int Np, const double *A, const std::vector<int> &idx,
                               const std::vector<double> &inc_failed, double hint, const std::vector<char> &may_decide, std::vector<char> &decided,
                               const std::function<int(const std::vector<double> &, const std::vector<char> &, std::vector<int> &)> &factor,
                               const std::function<int(const std::vector<char> &, const double *, double *, double *, double *, int, int)> &solve,
                               std::vector<double> &w0, std::vector<char> &ok)
{
    const size_t mark = ctx->ws_used;
    // Blocks of 16 vectors instead of 128 (lmin_skinny.hip): with a shift a few per cent above |lambda_min| the block's width hardly
    // matters for the convergence (7-9 steps against 5-7 on a production stamp), and a 16-column sweep is one pass over the factor at
    // HBM speed instead of a 128-column product on the matrix pipe.  Two forms of the sweeps: a pass of more than 128 stamps runs a
    // workgroup per stamp; fewer stamps (the kernel-class seam hands over one or four; a block's short last pass) two short launches
    // per block row with the sums dealt to many workgroups (on a pass of 168 stamps the two are within 2 % of each other: 1960
    // launches per pass against 10).
    // IMCOM_LMIN_SKINNY=0: the 128-vector form of rounds 5 and 6a (kept as the cross-check: tests/test_gpu_stamps.py runs both).
    const char *sk_env = getenv("IMCOM_LMIN_SKINNY");  // (read per call)
    const bool skinny = sk_env ? atoi(sk_env) != 0 : true;
    const int P = skinny ? LMIN_SKINNY_P : LMIN_P;
    const size_t blk = (size_t)batch * Np * P * 8, sq = (size_t)batch * P * P * 8;
    double *X = (double *)ws_take(ctx, blk), *Y = (double *)ws_take(ctx, blk), *Z = (double *)ws_take(ctx, blk);
    double *G = (double *)ws_take(ctx, sq), *Gi = (double *)ws_take(ctx, sq), *H = (double *)ws_take(ctx, sq), *Qh = (double *)ws_take(ctx, sq);
    double *lam = (double *)ws_take(ctx, (size_t)batch * P * 8);
    double *rpart = (double *)ws_take(ctx, (size_t)batch * LMIN_RESID_GROUPS * 2 * 8);
    double *part = (double *)ws_take(ctx, (size_t)batch * 8 * NB * NB * 8);
    double *dmax_d = (double *)ws_take(ctx, (size_t)batch * 8);
    int *want_d = (int *)ws_take(ctx, (size_t)batch * 4), *ones_d = (int *)ws_take(ctx, (size_t)batch * 4), *gfail_d = (int *)ws_take(ctx, (size_t)batch * 4);
    int *nbrun_d = (int *)ws_take(ctx, (size_t)batch * 4);  // 128-blocks of the stamps a round runs (0: not in it), for the 16-vector kernels
    if (!X || !Y || !Z || !G || !Gi || !H || !Qh || !lam || !rpart || !part || !dmax_d || !want_d || !ones_d || !gfail_d || !nbrun_d) { set_error("internal: workspace (smallest eigenvalue)"); return IMCOM_ERR_NOMEM; }
    const size_t mark_eig = ctx->ws_used;
    hipStream_t st = ctx->stream;
    std::vector<char> want(batch, 0);
    std::vector<int> want_i(batch, 0), ones(batch, 1), nP(batch, P);
    for (int s : idx) { want[s] = 1; want_i[s] = 1; }
    IMCOM_TRY(upload(ctx, want_d, want_i.data(), (size_t)batch));
    IMCOM_TRY(upload(ctx, ones_d, ones.data(), (size_t)batch));
    IMCOM_HIP_CHECK(hipMemsetAsync(gfail_d, 0, (size_t)batch * 4, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(Y, 0, blk, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(Z, 0, blk, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(G, 0, sq, st));  // (stamps that are not wanted keep zero blocks: gram_guard_kernel puts ones on their diagonals)
    IMCOM_HIP_CHECK(hipMemsetAsync(H, 0, sq, st));
    IMCOM_HIP_CHECK(hipMemsetAsync(Qh, 0, sq, st));
    IMCOM_TRY(launch_diag_max(ctx, A, Np, n_dev, dmax_d, batch));
    std::vector<double> dmax(batch, 0.0);
    IMCOM_HIP_CHECK(hipMemcpyAsync(dmax.data(), dmax_d, (size_t)batch * 8, hipMemcpyDeviceToHost, st));
    IMCOM_HIP_CHECK(hipStreamSynchronize(st));
    static const bool dbg = getenv("IMCOM_LMIN_DEBUG") != nullptr;
    int nfac = 0, nfac_failed = 0, rounds_run = 0;
    // 1. a positive definite shift
    std::vector<double> sigma(batch, 0.0);
    std::vector<char> todo = want, act = want;
    std::vector<int> fail;
    // The caller's estimate of max |lambda_min| (imcom_ctx_set_repair_hint: what the pass before this one found -- over a production
    // block the largest |lambda_min| of a pass of 168 stamps stays within 1 % from pass to pass, the stamps of a pass within 6 % of
    // each other): the first shift is that estimate plus a margin instead of four times the failed increment, i.e. already as close as the
    // SECOND shift of a stamp that starts without one, and the iteration needs one factorisation instead of two.  An estimate that is too
    // small for a stamp costs that stamp one failed factorisation (then the shift it would have started with).
    static const bool hint_off = getenv("IMCOM_LMIN_HINT") && strcmp(getenv("IMCOM_LMIN_HINT"), "0") == 0;
    static const double hint_margin = getenv("IMCOM_LMIN_MARGIN") ? std::max(0.0, atof(getenv("IMCOM_LMIN_MARGIN"))) : 0.05;
    std::vector<char> hinted(batch, 0);
    std::vector<double> base(batch, 0.0);
    bool any_hinted = false;
    for (int s : idx) {
        base[s] = sigma[s] = std::max(std::max(4.0 * inc_failed[s], 1e-13 * dmax[s]), 1e-300);
        if (!hint_off && hint > 0.0 && std::isfinite(hint)) { sigma[s] = hint * (1.0 + hint_margin); hinted[s] = 1; any_hinted = true; }
    }
    for (int t = 0;; t++) {
        IMCOM_TRY(factor(sigma, todo, fail));
        nfac++;
        bool any = false;
        for (int s : idx) {
            if (!todo[s]) continue;
            if (fail[s] != 0 && hinted[s]) { sigma[s] = std::max(base[s], 2.0 * sigma[s]); hinted[s] = 0; any = true; }
            else if (fail[s] != 0 && std::isfinite(sigma[s] * 8.0)) { sigma[s] *= 8.0; any = true; }
            else if (fail[s] != 0) { todo[s] = 0; act[s] = 0; }  // (not finite: not a matrix this iteration can help)
            else todo[s] = 0;
        }
        if (any) nfac_failed++;
        if (!any) break;
        if (t >= 24) {
            for (int s : idx) if (todo[s]) { todo[s] = 0; act[s] = 0; }
            break;
        }
    }
    // 2. / 3. subspace iteration, Rayleigh-Ritz with A, closer shifts
    IMCOM_TRY(launch_lmin_init(ctx, X, Np, P, n_dev, want_d, batch));
    const long sX = (long)Np * P, sG = (long)P * P;
    // a product per stamp of the batch, or -- when few stamps are wanted -- per wanted stamp (the others' operands are zero)
    const bool few = (long)idx.size() * 4 <= (long)batch;
    auto gemm = [&](bool akm, bool bkm, int M, int N, int K, const double *Ao, long lda, long sA, const double *Bo, long ldb, long sB, double *Co, long ldc, long sC) -> int {
        if (!few) return launch_gemm(ctx, akm, bkm, M, N, K, batch, Ao, lda, sA, Bo, ldb, sB, Co, ldc, sC, 1.0, 0.0);
        for (int s : idx) IMCOM_TRY(launch_gemm(ctx, akm, bkm, M, N, K, 1, Ao + s * sA, lda, sA, Bo + s * sB, ldb, sB, Co + s * sC, ldc, sC, 1.0, 0.0));
        return IMCOM_OK;
    };
    auto orth = [&]() -> int {  // X <- orth(Y): CholQR twice (X = Y R^-1 with R^T R = Y^T Y; the second pass takes the loss of the first back)
        for (int pass = 0; pass < 2; pass++) {
            double *src = pass == 0 ? Y : X, *dst = pass == 0 ? X : Y;
            if (skinny) { IMCOM_TRY(launch_skinny_orth(ctx, src, dst, Np, nbrun_d, gfail_d, batch)); continue; }
            IMCOM_TRY(gemm(true, true, P, P, Np, src, P, sX, src, P, sX, G, P, sG));
            IMCOM_TRY(launch_gram_guard(ctx, G, P, want_d, batch));
            IMCOM_TRY(launch_chol_diag(ctx, G, Gi, P, 0, batch, ones_d, gfail_d));
            IMCOM_TRY(gemm(false, false, Np, P, P, src, P, sX, Gi, P, sG, dst, P, sX));  // dst[i][c] = sum_j src[i][j] Linv[c][j]
        }
        std::swap(X, Y);  // the orthonormal block is in the buffer pass 1 wrote
        return IMCOM_OK;
    };
    std::vector<double> theta(batch, 0.0), prev(batch, 0.0), eta(batch, 1e-3), lam01((size_t)batch * 2, 0.0), rp((size_t)batch * LMIN_RESID_GROUPS * 2, 0.0);
    std::vector<double> est(batch, 0.0), est_simple(batch, 0.0), dbg_th(batch, 0.0), dbg_r1(batch, 0.0), dbg_r2(batch, 0.0), dbg_g2(batch, 0.0), dbg_gP(batch, 0.0), lamP(batch, 0.0);
    std::vector<char> conv(batch, 0);
    std::vector<int> gfail(batch, 0);
    // Per stamp two phases.  Coarse, at the first shift: six steps, then a Rayleigh-Ritz step that also yields the residuals r1, r2 of the
    // two lowest Ritz pairs (theta1, y1), (theta2, y2) -- there is an eigenvalue within |r1| of theta1, and once theta1 is separated from
    // the rest, (theta2 - |r2|) - theta1 = gap > 4 |r1|, theta1 - lambda_min <= |r1|^2 / gap (Kato-Temple).  With that bound below 15 % the
    // stamp gets ONE factorisation at |theta1| (1 + eta), eta = 1.5 x the bound (a shift that is not above |lambda_min| makes the
    // factorisation fail: eta x 8 -- every such trial costs eight steps' time), where the block converges by 1e-3 and more per step:
    // fine rounds of three steps until the bound is below 1e-11 |theta1|, which the first one reaches on a production stamp
    // (configs/paper4: six coarse steps leave 2e-3, three fine ones 1e-13).  The change between two successive values of theta1 -- the
    // criterion of the first version, which cost a round of three steps and a Rayleigh-Ritz step in each phase only to confirm -- still
    // ends either phase when the residuals cannot (theta1 inside a cluster closer than its residual).
    static const int coarse_steps_env = getenv("IMCOM_LMIN_COARSE") ? std::max(1, atoi(getenv("IMCOM_LMIN_COARSE"))) : 0;
    const int coarse_steps = coarse_steps_env ? coarse_steps_env : (LMIN_SKINNY_P == P ? 9 : 6);  // (16 vectors at a blind shift: six steps leave 25-40 %, nine a few per cent)
    static const int lmin_parts = getenv("IMCOM_LMIN_PARTS") ? std::min(8, std::max(1, atoi(getenv("IMCOM_LMIN_PARTS")))) : 0;  // (A/B: split-K parts of the 128-column solves)
    static const int hinted_steps_env = getenv("IMCOM_LMIN_HINTED") ? std::max(1, atoi(getenv("IMCOM_LMIN_HINTED"))) : 0;  // (34 x (7.6e-3)^7: see the hint above)
    const int hinted_steps = hinted_steps_env ? hinted_steps_env : (skinny ? 10 : 7);
    // how far the spectrum the block has NOT captured lies above lambda_min, in units of |lambda_min| (production stamps: lambda_129 is 0.85,
    // lambda_17 0.4 of the way to zero): the step counts below are planned with it
    const double bulk = skinny ? 0.4 : 1.0;
    static const int round_steps = getenv("IMCOM_LMIN_FINE") ? std::max(1, atoi(getenv("IMCOM_LMIN_FINE"))) : 3;
    static const bool by_change = getenv("IMCOM_LMIN_BOUND") && strcmp(getenv("IMCOM_LMIN_BOUND"), "change") == 0;  // (A/B: the first version's criteria alone)
    const int max_rounds = 14;
    std::vector<char> fine(batch, 0);
    std::vector<int> steps_wanted(batch, 0);
    std::vector<double> lastrel(batch, 1.0);
    for (int round = 0; round < max_rounds; round++) {
        std::vector<char> run(batch, 0);
        bool any = false;
        for (int s : idx) if (act[s] && !conv[s]) { run[s] = 1; any = true; }
        if (!any) break;
        int iters = round == 0 ? (any_hinted ? hinted_steps : coarse_steps) : round_steps;
        for (int s : idx) if (run[s]) iters = std::max(iters, steps_wanted[s]);  // (a stamp's first round at its closer shift: see below)
        std::fill(steps_wanted.begin(), steps_wanted.end(), 0);
        int nbrun_max = 0;
        if (skinny) {
            std::vector<int> nbr(batch, 0);
            for (int s : idx) if (run[s]) { nbr[s] = (n_host[s] + NB - 1) / NB; nbrun_max = std::max(nbrun_max, nbr[s]); }
            IMCOM_TRY(upload(ctx, nbrun_d, nbr.data(), (size_t)batch));
        }
        for (int it = 0; it < iters; it++) {
            IMCOM_TRY(solve(run, X, Y, Z, part, lmin_parts > 0 ? lmin_parts : splitk_parts(batch, 1), P));  // (Z: scratch here, the Rayleigh-Ritz step below fills it anew)
            IMCOM_TRY(orth());
        }
        // Z = A X, H = X^T Z, its eigenvalues and eigenvectors, the residuals of the two lowest pairs
        if (skinny) {
            IMCOM_TRY(launch_skinny_ax(ctx, A, X, Z, Np, nbrun_d, nbrun_max, batch));
            IMCOM_TRY(launch_skinny_rr(ctx, X, Z, Np, nbrun_d, lam, rpart, LMIN_RESID_GROUPS, batch));
        } else {
            IMCOM_TRY(gemm(false, true, Np, P, Np, A, Np, (long)Np * Np, X, P, sX, Z, P, sX));
            IMCOM_TRY(gemm(true, true, P, P, Np, X, P, sX, Z, P, sX, H, P, sG));
            ctx->ws_used = mark_eig;
            IMCOM_TRY(eigh_device(ctx, batch, nP.data(), P, H, P, sG, lam, P, Qh, P, sG, nullptr));
            ctx->ws_used = mark_eig;
            IMCOM_TRY(launch_ritz_residual(ctx, X, Z, Qh, lam, Np, P, n_dev, want_d, rpart, batch));
        }
        IMCOM_HIP_CHECK(hipMemcpy2DAsync(lam01.data(), 16, lam, (size_t)P * 8, 16, batch, hipMemcpyDeviceToHost, st));
        IMCOM_HIP_CHECK(hipMemcpyAsync(rp.data(), rpart, rp.size() * 8, hipMemcpyDeviceToHost, st));
        IMCOM_HIP_CHECK(hipMemcpy2DAsync(lamP.data(), 8, lam + P - 1, (size_t)P * 8, 8, batch, hipMemcpyDeviceToHost, st));
        IMCOM_HIP_CHECK(hipMemcpyAsync(gfail.data(), gfail_d, (size_t)batch * 4, hipMemcpyDeviceToHost, st));
        IMCOM_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<char> refac(batch, 0);
        bool any_refac = false;
        for (int s : idx) {
            if (!run[s]) continue;
            const double th1 = lam01[2 * (size_t)s], th2 = lam01[2 * (size_t)s + 1];
            double q1 = 0.0, q2 = 0.0;
            for (int g = 0; g < LMIN_RESID_GROUPS; g++) { q1 += rp[((size_t)s * LMIN_RESID_GROUPS + g) * 2]; q2 += rp[((size_t)s * LMIN_RESID_GROUPS + g) * 2 + 1]; }
            if (gfail[s] != 0 || !std::isfinite(th1) || !std::isfinite(q1) || !std::isfinite(q2)) { act[s] = 0; continue; }  // the block lost rank / not a number: the eigensolver's case
            const double r1 = sqrt(q1), r2 = sqrt(q2), gap = (th2 - r2) - th1, mag = std::max(fabs(th1), 1e-300);
            est_simple[s] = r1;
            if (dbg && round == 0) { dbg_th[s] = th1; dbg_r1[s] = r1; dbg_r2[s] = r2; dbg_g2[s] = th2 - th1; dbg_gP[s] = lamP[s] - th1; }
            // theta1 - lambda_min >= 0: at most |r1|, at most |r1|^2 / gap once theta1 is separated (both rigorous) -- and in fact close to
            // |r1|^2 / (theta_P - theta1), the distance to the part of the spectrum the block has NOT captured (the lowest eigenvalues of a
            // stamp's A come in near-degenerate pairs, so the rigorous gap is tiny while theta1's error is governed by the bulk near zero:
            // on 128 paper4 stamps the error was 1.03 ... 1.28 x that quotient after six coarse steps and after three fine ones)
            const double span = lamP[s] - th1;
            est[s] = gap > 4.0 * r1 ? r1 * r1 / gap : r1;
            if (span > 4.0 * r1) est[s] = std::min(est[s], 3.0 * r1 * r1 / span);
            prev[s] = theta[s];
            theta[s] = th1;
            const double rel = round == 0 ? 1.0 : fabs(theta[s] - prev[s]) / mag;
            const double bound = by_change ? 1.0 : est[s] / mag;
            // A stamp whose failure is the CALLER's expectation, not an observed one (may_decide): all that is wanted first is whether
            // A + inc I is positive definite after all, and theta1 - |r1| decides that long before theta1 has converged (NOT a rigorous bound:
            // the residual only says that SOME eigenvalue lies within |r1| of theta1; lambda_min can lie below if the block has not captured
            // it.  A stamp decided this way is factored plainly next, and should that fail the iteration runs again: a misjudgement costs
            // time, never a wrong shift -- and only stamps whose failure was EXPECTED, not observed, come here) --
            // lambda_min of a healthy PSF-overlap matrix is rounding noise around zero inside a dense cluster, which this iteration would
            // chase for all its rounds and then hand to the eigensolver.  (From the second Rayleigh-Ritz step on: ten steps bring a
            // lambda_min of the size of -inc out of a random block beyond doubt.)  theta1 is then no eigenvalue: `decided` says so.
            if (may_decide[s] && round >= 1 && inc_failed[s] > 0.0 && (th1 - r1) + inc_failed[s] > 1e-3 * inc_failed[s]) {
                conv[s] = 1; decided[s] = 1; lastrel[s] = rel;
                continue;
            }
            // (the quotient with theta_P is an estimate, not a bound: it ends the iteration only together with the rigorous |r1| <= 1e-6 |theta1|,
            // which keeps the worst case -- theta1 inside a cluster the block has not separated -- at the rounding level of the float32 T)
            if ((bound <= 1e-11 && r1 <= 1e-6 * mag) || (round >= 1 && rel <= 1e-11)) { conv[s] = 1; lastrel[s] = rel; continue; }
            // a shift just above |theta| (theta >= lambda_min: it must exceed |theta| by more than theta's error)
            // a first shift that is already close (a good hint): no second factorisation, the rounds go on where they are; a step multiplies
            // the error by ((lambda_min + sigma) / (|lambda_min| + lambda_min + sigma))^2 (the bulk of the spectrum is |lambda_min| away)
            const double close = theta[s] < 0.0 ? (sigma[s] + theta[s]) / mag : 1e300;
            if (!fine[s] && close > 0.0 && close <= 0.25) {
                fine[s] = 1;
                eta[s] = close;
                const double c = (close / (bulk + close)) * (close / (bulk + close));
                steps_wanted[s] = (int)std::min(6.0, std::max(1.0, ceil(log(std::max(1e-11 / std::max(bound, 1e-300), 1e-300)) / log(c))));
            }
            if (!fine[s] && theta[s] < 0.0 && sigma[s] > 4.0 * (mag + r1) && !(bound <= 0.15)) {
                // a first shift far above what is needed (the x 8 ladder overshot, or a hint from another regime): the block converges by
                // (1 - |lambda_min| / sigma)^2 per step there.  lambda_min >= theta1 - |r1| (rigorous): one factorisation at twice that, still coarse
                eta[s] = 1.0 + 2.0 * r1 / mag;
                refac[s] = 1; any_refac = true;
            } else if (!fine[s]) {
                if (theta[s] < 0.0 && (bound <= 0.15 || (round >= 1 && rel <= 2e-2))) {
                    fine[s] = 1;
                    eta[s] = std::min(std::max(bound <= 0.15 ? 1.5 * bound : 8.0 * rel, 1e-3), 0.25);
                    refac[s] = 1; any_refac = true;
                    // steps of the first fine round: with e = bound / 3 the error of theta1 and the bulk of the spectrum |theta1| away, the shift
                    // leaves lambda_min + sigma' = 3.5 e |theta1| and a step multiplies the error by (3.5 e)^2: e (12 e^2)^j <= 3e-12
                    steps_wanted[s] = (bound <= 1.5e-2 ? 3 : bound <= 4e-2 ? 4 : bound <= 7e-2 ? 5 : 6) + (skinny ? 1 : 0);
                }
            } else if (rel > 0.05 * lastrel[s] && 16.0 * rel < 0.25 * eta[s] && theta[s] < 0.0) {
                eta[s] = std::max(16.0 * rel, 1e-9);
                refac[s] = 1; any_refac = true;
            }
            lastrel[s] = rel;
        }
        for (int t = 0; any_refac; t++) {
            std::vector<double> sg = sigma;
            for (int s : idx) if (refac[s]) sg[s] = -theta[s] * (1.0 + eta[s]);
            IMCOM_TRY(factor(sg, refac, fail));
            nfac++;
            any_refac = false;
            for (int s : idx) {
                if (!refac[s]) continue;
                if (fail[s] == 0) { sigma[s] = sg[s]; refac[s] = 0; }
                else if (t >= 10) { refac[s] = 0; act[s] = 0; }
                else { eta[s] *= 8.0; any_refac = true; }
            }
            if (any_refac) nfac_failed++;
        }
        rounds_run = round + 1;
        if (dbg) {
            const int s = idx[0];
            fprintf(stderr, "[lmin] round %d (%d steps): stamp %d theta %.15e (prev %.15e) sigma %.6e eta %.3e fine %d conv %d   |r1| %.3e  bound %.3e (of |theta|)\n", round,
                    iters, s, theta[s], prev[s], sigma[s], eta[s], (int)fine[s], (int)conv[s], est_simple[s], est[s] / std::max(fabs(theta[s]), 1e-300));
            double rmin = 1e300, rmax = 0.0, bmin = 1e300, bmax = 0.0, smin = 1e300, smax_ = 0.0;
            int nfine = 0, nconv = 0, nrun = 0;
            for (int q : idx) {
                if (!run[q]) continue;
                const double mag = std::max(fabs(theta[q]), 1e-300);
                nrun++; nfine += fine[q]; nconv += conv[q];
                rmin = std::min(rmin, lastrel[q]); rmax = std::max(rmax, lastrel[q]);
                bmin = std::min(bmin, est[q] / mag); bmax = std::max(bmax, est[q] / mag);
                smin = std::min(smin, est_simple[q] / mag); smax_ = std::max(smax_, est_simple[q] / mag);
            }
            fprintf(stderr, "[lmin]   %d stamps ran: change of theta %.2e .. %.2e, bound %.2e .. %.2e, |r1| / |theta| %.2e .. %.2e, %d fine, %d converged\n", nrun, rmin, rmax, bmin,
                    bmax, smin, smax_, nfine, nconv);
        }
    }
    if (dbg) for (int s : idx) if (conv[s] && dbg_r1[s] > 0.0) {
        const double d = dbg_th[s] - theta[s];
        fprintf(stderr, "[lmin0] stamp %d theta0 %.9e final %.15e err %.3e r1 %.3e r1^2/err %.3e th2-th1 %.3e thP-th1 %.3e r2 %.3e\n", s, dbg_th[s], theta[s], d / fabs(theta[s]), dbg_r1[s],
                dbg_r1[s] * dbg_r1[s] / std::max(d, 1e-300), dbg_g2[s], dbg_gP[s], dbg_r2[s]);
    }
    for (int s : idx) {
        ok[s] = act[s] && conv[s];
        if (ok[s]) w0[s] = theta[s];
    }
    if (dbg) {
        double tmin = 1e300, tmax = -1e300;
        for (int s : idx) if (ok[s]) { tmin = std::min(tmin, w0[s]); tmax = std::max(tmax, w0[s]); }
        int nok = 0;
        for (int s : idx) nok += ok[s] ? 1 : 0;
        if (getenv("IMCOM_LMIN_DUMP")) {  // every stamp's value, in batch order
            fprintf(stderr, "[lmin-w0]");
            for (int s : idx) fprintf(stderr, " %.9e", ok[s] ? w0[s] : 0.0);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[lmin] %zu stamps: %d factorisations (%d of them failed for some stamp), %d rounds; lambda_min %.6e .. %.6e; %d without an answer (the eigensolver's)\n", idx.size(), nfac,
                nfac_failed, rounds_run, tmin, tmax, (int)idx.size() - nok);
    }
    ctx->ws_used = mark;
    return IMCOM_OK;
}

// before_solve (optional): called once, after the first factorisation's launches have been queued and before the first launch
// that reads Bt -- the host-buffer entry uploads -B/2 there, behind the factorisation instead of in front of it.
// what the coaddition needs besides T (coadd.py:1294-1363), for the entries that solve and coadd in one call
struct CoaddArgs {
    int n2f, fade, n2, n_inframe, n_expo;
    const float *indata;
    const int *expo;
    float *outimage;
    double *Tsum_stamp, *Tsum_inpix, *Neff;
};

static bool coadd_fusable(int nv, int fade)
{
    static const bool off = getenv("IMCOM_SOLVE_UNFUSED") != nullptr || getenv("IMCOM_EPILOGUE_UNFUSED") != nullptr;
    return !off && nv == 1 && fade == 0;  // one kappa node: T is final in the backward launches; no taper to apply to it first
}

static size_t coadd_fuse_bytes(int batch, int Np, int m, int mp, int nv, const CoaddArgs *co)
{
    if (!co) return 0;
    size_t t = (size_t)batch * m * co->n_expo * 8 + 512;  // Tsum_image
    if (coadd_fusable(nv, co->fade)) t += (size_t)batch * 2 * (Np / NB) * (co->n_expo + co->n_inframe) * mp * 8 + 512;
    return t;
}

// redo_host (optional, one kappa node only): per stamp 0 = leave the stamp and its outputs alone, 1 = solve it, 2 = solve it knowing that
// the factorisation of A + kappa I fails (imcom_solve_chol_resident_end OBSERVED it: straight to the repair), 3 = the same as the caller's
// EXPECTATION (the stamps before it failed): straight to the repair, but the smallest-eigenvalue iteration may find that A + kappa I is
// positive definite after all and end early (may_decide), the stamp is then factored plainly.  An observed failure never takes that
// shortcut (the bound it rests on is not rigorous, and for an observed failure it could only cost a second iteration).
// With ONE kappa node an attempt after the first works on the stamps whose factorisation failed and on nothing else: the launches
// see the other stamps with no blocks at all, and their outputs stay what the first attempt wrote (a batch of 256 in which one
// factorisation fails used to be factored and solved three times over).
static int chol_core(imcom_ctx *ctx, int batch, const int *n_host, int Np, int m, int mp, const double *A,
                     const double *Bt, const double *C_host, const double *kappaC_host, int nv, double ucmin,
                     double smax, float *Tt, float *UC, float *Sigma, float *kappa, int *info_host,
                     const std::function<int()> &before_solve = nullptr, const CoaddArgs *co = nullptr, bool defer = false,
                     const int *redo_host = nullptr)
{
    bool bt_ready = !before_solve;
    ctx->last_repair_count = 0;
    const int nbmax_all = Np / NB;
    const int pb = nodes_per_pass(batch, mp, nv), eb = batch * pb;  // nodes per pass, stamps x nodes of a pass
    const bool masked = nv == 1;  // (pb == 1 then)
    IMCOM_REQUIRE(!redo_host || masked, "a redo mask needs a single kappa node");
    double *L = (double *)ws_take(ctx, (size_t)eb * Np * Np * 8);
    double *Dinv = (double *)ws_take(ctx, (size_t)eb * nbmax_all * NB * NB * 8);
    const long node_stride = (long)batch * Np * mp;
    double *Y = (double *)ws_take(ctx, (size_t)nv * node_stride * 8);
    double *dshift = (double *)ws_take(ctx, (size_t)eb * Np * 8);
    int *ints = (int *)ws_take(ctx, (size_t)eb * 4 * 3 + (size_t)batch * 4 * nv);
    int *mints = (int *)ws_take(ctx, (size_t)batch * 4 * 4);
    double *inc = (double *)ws_take(ctx, (size_t)eb * MAX_INC_HOST * 8);
    double *dbl = (double *)ws_take(ctx, (size_t)batch * 8 * 2);
    double *kappaC_dev = (double *)ws_take(ctx, (size_t)nv * 8);
    double *Dp = nullptr, *Npq = nullptr, *W = nullptr;
    if (nv > 1) {
        Dp = (double *)ws_take(ctx, (size_t)batch * m * nv * 8);
        Npq = (double *)ws_take(ctx, (size_t)batch * m * nv * nv * 8);
        W = (double *)ws_take(ctx, (size_t)batch * m * nv * 8);
    }
    const size_t pbytes = splitk_bytes(eb, Np, mp);
    double *partial = pbytes ? (double *)ws_take(ctx, pbytes) : nullptr;
    const int parts_solve = partial ? splitk_parts(eb, mp / NB) : 1;
    // single kappa with the diagonal blocks fused into the solve launches: the launches leave T (float32) and the column sums
    // the maps need behind, and the pass of finalize_single_kernel over X and -B/2 (27 GB per 256 cfg-2 stamps) is not needed
    static const bool unfused_solve = getenv("IMCOM_SOLVE_UNFUSED") != nullptr;
    double *colsums = (nv == 1 && !unfused_solve) ? (double *)ws_take(ctx, (size_t)batch * 2 * nbmax_all * mp * 8 * 2) : nullptr;
    double *Dpart = colsums, *Npart = colsums ? colsums + (size_t)batch * 2 * nbmax_all * mp : nullptr;
    // the coaddition of the same call: its sums ride in the backward launches when T is final there (else the stand-alone epilogue)
    CoaddFuse cfuse;
    double *Tsum_image = nullptr;
    if (co) {
        Tsum_image = (double *)ws_take(ctx, (size_t)batch * m * co->n_expo * 8);
        if (colsums && coadd_fusable(nv, co->fade)) {
            cfuse.indata = co->indata; cfuse.expo = co->expo; cfuse.n_inframe = co->n_inframe; cfuse.n_expo = co->n_expo;
            cfuse.Epart = (double *)ws_take(ctx, (size_t)batch * 2 * nbmax_all * (co->n_expo + co->n_inframe) * mp * 8);
        }
        if (!Tsum_image || (colsums && coadd_fusable(nv, co->fade) && !cfuse.Epart)) { set_error("internal: workspace plan too small (coaddition)"); return IMCOM_ERR_NOMEM; }
    }
    if (!L || !Dinv || !Y || !dshift || !ints || !mints || !inc || !dbl || !kappaC_dev || (nv > 1 && (!Dp || !Npq || !W)) || (pbytes && !partial) || (nv == 1 && !unfused_solve && !colsums)) {
        set_error("internal: workspace plan too small");
        return IMCOM_ERR_NOMEM;
    }
    const size_t ws_after_plan = ctx->ws_used;  // (the repair's own scratch is taken from here on and handed back)
    int *n_dev = ints, *nblk_dev = ints + eb, *ninc_dev = ints + 2 * eb, *fail_dev = ints + 3 * eb;  // n, nblk, ninc [eb]; fail[nv][batch]
    int *fac_dev = mints, *nblk_fac = mints + batch, *nblk_sol = mints + 2 * batch, *act_dev = mints + 3 * batch;  // one kappa node: this attempt's stamps
    double *kap_dev = dbl, *C_dev = dbl + batch;

    std::vector<int> nblk(batch), ninc(batch, 0), fail((size_t)nv * batch);
    std::vector<double> inc_h((size_t)batch * MAX_INC_HOST, 0.0), kap_h(batch), rep(batch, 0.0);
    std::vector<char> repaired((size_t)nv * batch, 0), have_w0(batch, 0);
    std::vector<char> active(batch, 1), known(batch, 0), expected(batch, 0);  // the stamps of the coming attempt; stamps whose first factorisation is known
                                                                            // to fail; ... of which by the caller's expectation only (redo code 3)
    std::vector<char> expected_only(batch, 0);            // ... of which the eigenvalue says that they do not fail after all
    int nbmax = 0;
    for (int s = 0; s < batch; s++) {
        IMCOM_REQUIRE(n_host[s] >= 0 && n_host[s] <= Np, "n[%d]=%d outside [0,%d]", s, n_host[s], Np);
        nblk[s] = (n_host[s] + NB - 1) / NB;
        if (nblk[s] > nbmax) nbmax = nblk[s];
        if (redo_host) { active[s] = redo_host[s] != 0; known[s] = redo_host[s] >= 2; expected[s] = redo_host[s] == 3; }
        if (active[s]) info_host[s] = 0;
    }
    for (int q = 0; q < pb; q++) {  // the stamps of a pass: node-major copies
        IMCOM_TRY(upload(ctx, n_dev + q * batch, n_host, batch));
        IMCOM_TRY(upload(ctx, nblk_dev + q * batch, nblk.data(), batch));
    }
    IMCOM_TRY(upload(ctx, C_dev, C_host, batch));
    IMCOM_TRY(upload(ctx, kappaC_dev, kappaC_host, nv));

    // one factorisation over the stamps of `mask` with the diagonal A_ii + shift[s] (the smallest-eigenvalue iteration's trial
    // factorisations: lambda_min_subspace), flags read back
    auto factor_masked = [&](const std::vector<double> &shift, const std::vector<char> &mask, std::vector<int> &fail_h) -> int {
        std::vector<int> nb(batch), one(batch, 1);
        std::vector<double> ih((size_t)batch * MAX_INC_HOST, 0.0);
        int nbm = 0;
        for (int s = 0; s < batch; s++) {
            nb[s] = mask[s] ? nblk[s] : 0;
            nbm = std::max(nbm, nb[s]);
            ih[(size_t)s * MAX_INC_HOST] = shift[s];
        }
        IMCOM_TRY(upload(ctx, nblk_fac, nb.data(), (size_t)batch));
        IMCOM_TRY(upload(ctx, inc, ih.data(), ih.size()));
        IMCOM_TRY(upload(ctx, ninc_dev, one.data(), (size_t)batch));
        IMCOM_TRY(launch_diag_shift(ctx, A, Np, inc, ninc_dev, dshift, batch));
        IMCOM_HIP_CHECK(hipMemsetAsync(fail_dev, 0, (size_t)batch * 4, ctx->stream));
        for (int k = 0; k < nbm; k++) {
            IMCOM_TRY(launch_chol_update(ctx, A, L, Np, k, nbm, batch, batch, nblk_fac, dshift, partial, partial ? splitk_parts(batch, nbm - k) : 1));
            IMCOM_TRY(launch_chol_diag(ctx, L, Dinv, Np, k, batch, nblk_fac, fail_dev));
            IMCOM_TRY(launch_chol_trsm(ctx, L, Dinv, Np, k, nbm, batch, nblk_fac));
        }
        fail_h.assign(batch, 0);
        IMCOM_HIP_CHECK(hipMemcpyAsync(fail_h.data(), fail_dev, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return IMCOM_OK;
    };
    // Yv = (L L^T)^-1 Xv on LMIN_P columns for the stamps of `mask`; Wv: scratch of the same size (may be null).
    // Two forms.  Many stamps: the left-looking block rows of the solve kernels (a launch per block row, one 128-column tile per stamp,
    // the K loop dealt to up to 8 workgroups).  FEW stamps (the kernel-class seam hands over one, a 2 x 2 group four): that form keeps 9
    // workgroups per stamp busy and a 128-column solve of a production stamp (N = 6.2k) takes 10.8 ms -- 97 of the 135 ms of a seam call
    // (profiles/r06_seam_paper4_kernel_stats.csv).  There the solve runs RIGHT-looking on the generic product kernel: block k is finished by
    // its inverted diagonal block, then ALL block rows below it are updated at once, Y_i -= L_ik Y_k: nb - k - 1 workgroups per stamp
    // and launch instead of 9, the same flops.
    static const bool right_off = getenv("IMCOM_LMIN_RIGHT") && strcmp(getenv("IMCOM_LMIN_RIGHT"), "0") == 0;
    auto solve_block = [&](const std::vector<char> &mask, const double *Xv, double *Yv, double *Wv, double *part, int parts, int Pv) -> int {
        std::vector<int> nb(batch);
        int nbm = 0, cnt = 0;
        for (int s = 0; s < batch; s++) { nb[s] = mask[s] ? nblk[s] : 0; nbm = std::max(nbm, nb[s]); cnt += mask[s] ? 1 : 0; }
        if (Pv == LMIN_SKINNY_P) {  // blocks of 16 vectors (lmin_skinny.hip): one workgroup per stamp, both sweeps in one launch -- or, for few stamps, two launches per block row
            IMCOM_TRY(upload(ctx, nblk_sol, nb.data(), (size_t)batch));
            // (per-block-row launches: 1.8 ms of launches per solve + 44 us per production stamp at 7 TB/s; a workgroup per stamp: 12 ms whatever the
            // count -- they meet near 200 stamps, and at 168 the single launch is 2 % ahead on the wall clock.  Read per call: the tests run both forms.)
            const int few_max = getenv("IMCOM_LMIN_FEW_MAX") ? atoi(getenv("IMCOM_LMIN_FEW_MAX")) : 128;
            if (cnt <= few_max && part && skinny_few_partial_doubles(batch) <= (size_t)batch * 8 * NB * NB)
                return launch_skinny_solve_few(ctx, L, Dinv, Xv, Yv, Np, nblk_sol, nbm, batch, part);
            return launch_skinny_solve(ctx, L, Dinv, Xv, Yv, Np, nblk_sol, batch);
        }
        if (Wv && !right_off && cnt > 0 && cnt <= 8) {
            const int P = LMIN_P;
            const long sL = (long)Np * Np, sY = (long)Np * P, sD = (long)(Np / NB) * NB * NB;
            for (int s0 = 0; s0 < batch;) {  // runs of consecutive wanted stamps with the same number of blocks: one batched launch each
                if (!mask[s0] || nblk[s0] == 0) { s0++; continue; }
                int s1 = s0 + 1;
                while (s1 < batch && mask[s1] && nblk[s1] == nblk[s0]) s1++;
                const int bc = s1 - s0, nbs = nblk[s0];
                const double *L0 = L + s0 * sL, *D0 = Dinv + s0 * sD, *X0 = Xv + s0 * sY;
                double *Y0 = Yv + s0 * sY, *W0 = Wv + s0 * sY;
                IMCOM_HIP_CHECK(hipMemcpyAsync(W0, X0, (size_t)bc * sY * 8, hipMemcpyDeviceToDevice, ctx->stream));
                for (int k = 0; k < nbs; k++) {  // L Y = X
                    IMCOM_TRY(launch_gemm(ctx, false, true, NB, P, NB, bc, D0 + (long)k * NB * NB, NB, sD, W0 + (long)k * NB * P, P, sY, Y0 + (long)k * NB * P, P, sY, 1.0, 0.0));
                    if (k + 1 < nbs)
                        IMCOM_TRY(launch_gemm(ctx, false, true, (nbs - k - 1) * NB, P, NB, bc, L0 + (long)(k + 1) * NB * Np + (long)k * NB, Np, sL, Y0 + (long)k * NB * P, P, sY,
                                              W0 + (long)(k + 1) * NB * P, P, sY, -1.0, 1.0));
                }
                IMCOM_HIP_CHECK(hipMemcpyAsync(W0, Y0, (size_t)bc * sY * 8, hipMemcpyDeviceToDevice, ctx->stream));
                for (int k = nbs - 1; k >= 0; k--) {  // L^T Z = Y
                    IMCOM_TRY(launch_gemm(ctx, true, true, NB, P, NB, bc, D0 + (long)k * NB * NB, NB, sD, W0 + (long)k * NB * P, P, sY, Y0 + (long)k * NB * P, P, sY, 1.0, 0.0));
                    if (k > 0)
                        IMCOM_TRY(launch_gemm(ctx, true, true, k * NB, P, NB, bc, L0 + (long)k * NB * Np, Np, sL, Y0 + (long)k * NB * P, P, sY, W0, P, sY, -1.0, 1.0));
                }
                s0 = s1;
            }
            return IMCOM_OK;
        }
        IMCOM_TRY(upload(ctx, nblk_sol, nb.data(), (size_t)batch));
        for (int k = 0; k < nbm; k++) IMCOM_TRY(launch_solve_fwd(ctx, L, Xv, Yv, Np, LMIN_P, k, batch, batch, nblk_sol, n_dev, Dinv, part, parts, nullptr));
        for (int k = nbm - 1; k >= 0; k--) IMCOM_TRY(launch_solve_bwd(ctx, L, Yv, Np, LMIN_P, k, nbm, batch, nblk_sol, n_dev, Dinv, part, parts, nullptr, nullptr));
        return IMCOM_OK;
    };

    for (int attempt = 0;; attempt++) {
        IMCOM_HIP_CHECK(hipMemsetAsync(fail_dev, 0, (size_t)nv * batch * 4, ctx->stream));
        const int *nb_fac = nblk_dev, *nb_sol = nblk_dev, *act_fin = nullptr;
        std::vector<char> fac(batch, 1);  // the stamps this attempt factors
        if (masked) {
            std::vector<int> f_i(batch), nb_i(batch);
            for (int s = 0; s < batch; s++) {
                fac[s] = active[s] && !(attempt == 0 && known[s]);
                f_i[s] = fac[s];
                nb_i[s] = fac[s] ? nblk[s] : 0;
            }
            IMCOM_TRY(upload(ctx, fac_dev, f_i.data(), (size_t)batch));
            IMCOM_TRY(upload(ctx, nblk_fac, nb_i.data(), (size_t)batch));
            nb_fac = nblk_fac; nb_sol = nblk_sol; act_fin = act_dev;
        }
        bool any_fac = false;
        for (int s = 0; s < batch; s++) any_fac |= fac[s] != 0;
        for (int p0 = 0; any_fac && p0 < nv; p0 += pb) {
            for (int q = 0; q < pb; q++) {
                const int p = p0 + q;
                // the diagonal of AA at node p as the reference builds it: a sequence of in-place adds
                // (lakernel.py:298 single kappa; 356 node differences; 268/277 repair add and restore)
                for (int s = 0; s < batch; s++) {
                    double *ih = &inc_h[(size_t)s * MAX_INC_HOST];
                    int c = 0;
                    if (nv == 1) { if (kappaC_host[0] * C_host[s] != 0.0) ih[c++] = kappaC_host[0] * C_host[s]; }
                    else
                        for (int qq = 0; qq <= p; qq++) {
                            ih[c++] = kappaC_host[qq] * C_host[s] - (qq > 0 ? kappaC_host[qq - 1] * C_host[s] : 0.0);
                            if (qq < p && repaired[(size_t)qq * batch + s]) { ih[c++] = rep[s]; ih[c++] = -rep[s]; }
                        }
                    if (repaired[(size_t)p * batch + s]) ih[c++] = rep[s];  // c <= 3 nv <= MAX_INC: nv <= CHOL_MAXNV is checked at entry
                    ninc[s] = c;
                    kap_h[s] = kappaC_host[p] * C_host[s];
                }
                IMCOM_TRY(upload(ctx, inc + (size_t)q * batch * MAX_INC_HOST, inc_h.data(), inc_h.size()));
                IMCOM_TRY(upload(ctx, ninc_dev + q * batch, ninc.data(), batch));
                if (q == 0) IMCOM_TRY(upload(ctx, kap_dev, kap_h.data(), batch));  // used by the single-kappa finalize only
                IMCOM_TRY(launch_diag_shift(ctx, A, Np, inc + (size_t)q * batch * MAX_INC_HOST, ninc_dev + q * batch, dshift + (size_t)q * batch * Np, batch));
            }
            int *failp = fail_dev + (size_t)p0 * batch;
            for (int k = 0; k < nbmax; k++) {
                { ProfScope ps(ctx, "chol_gemm"); IMCOM_TRY(launch_chol_update(ctx, A, L, Np, k, nbmax, eb, batch, nb_fac, dshift, partial, partial ? splitk_parts(eb, nbmax - k) : 1)); }
                { ProfScope ps(ctx, "chol_diag"); IMCOM_TRY(launch_chol_diag(ctx, L, Dinv, Np, k, eb, nb_fac, failp)); }
                { ProfScope ps(ctx, "chol_gemm"); IMCOM_TRY(launch_chol_trsm(ctx, L, Dinv, Np, k, nbmax, eb, nb_fac)); }
            }
            // one kappa node: a stamp whose factorisation has just failed is left out of the solves (its tiles return at once)
            if (masked) IMCOM_TRY(launch_solve_mask(ctx, nblk_dev, fac_dev, failp, nblk_sol, act_dev, batch));
            if (!bt_ready) { IMCOM_TRY(before_solve()); bt_ready = true; }
            double *Yp = Y + p0 * node_stride;
            // the diagonal blocks are applied inside the update launches; IMCOM_SOLVE_UNFUSED=1 keeps them apart (A/B runs)
            static const bool unfused = getenv("IMCOM_SOLVE_UNFUSED") != nullptr;
            const double *Dfused = unfused ? nullptr : Dinv;
            for (int k = 0; k < nbmax; k++) {
                { ProfScope ps(ctx, "solve_gemm"); IMCOM_TRY(launch_solve_fwd(ctx, L, Bt, Yp, Np, mp, k, eb, batch, nb_sol, n_dev, Dfused, partial, parts_solve, Dpart)); }
                if (unfused) { ProfScope ps(ctx, "solve_dinv"); IMCOM_TRY(launch_solve_dinv(ctx, Dinv, Yp, Np, mp, k, eb, nb_sol, false)); }
            }
            for (int k = nbmax - 1; k >= 0; k--) {
                if (k < nbmax - 1 || !unfused) { ProfScope ps(ctx, "solve_gemm"); IMCOM_TRY(launch_solve_bwd(ctx, L, Yp, Np, mp, k, nbmax, eb, nb_sol, n_dev, Dfused, partial, parts_solve, Npart, Tt, cfuse.Epart ? &cfuse : nullptr)); }
                if (unfused) { ProfScope ps(ctx, "solve_dinv"); IMCOM_TRY(launch_solve_dinv(ctx, Dinv, Yp, Np, mp, k, eb, nb_sol, true)); }
            }
        }
        if (any_fac) {
            ProfScope ps(ctx, "finalize");
            if (nv == 1 && colsums)
                IMCOM_TRY(launch_finalize_fused(ctx, Dpart, Npart, Np, mp, m, n_dev, nblk_dev, kap_dev, C_dev, Tt, UC, Sigma, kappa, batch, act_fin));
            else if (nv == 1)
                IMCOM_TRY(launch_finalize_single(ctx, Y, Bt, Np, mp, m, n_dev, kap_dev, C_dev, Tt, UC, Sigma, kappa, batch, act_fin));
            else
                IMCOM_TRY(launch_multi(ctx, Y, node_stride, Bt, Np, mp, m, n_dev, nv, kappaC_dev, C_dev, ucmin, smax, Dp, Npq, W, Tt, UC, Sigma, kappa, batch));
        }
        if (defer) {
            // imcom_solve_chol_resident_begin: everything of the first attempt is queued; the failure flags travel to page-locked memory
            // behind it (in stream order: before whatever the caller queues next may reuse the workspace) and ..._end reads them
            const size_t cnt = (size_t)nv * batch;
            if (ctx->flag_pin_count < cnt) {
                if (ctx->flag_pin) { IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream)); IMCOM_HIP_CHECK(hipHostFree(ctx->flag_pin)); ctx->flag_pin = nullptr; ctx->flag_pin_count = 0; }
                IMCOM_HIP_CHECK(hipHostMalloc((void **)&ctx->flag_pin, std::max<size_t>(cnt, 256) * 4, hipHostMallocDefault));
                ctx->flag_pin_count = std::max<size_t>(cnt, 256);
            }
            IMCOM_HIP_CHECK(hipMemcpyAsync(ctx->flag_pin, fail_dev, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
            while (ctx->sync_events.size() < 3) {
                hipEvent_t e;
                IMCOM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ctx->sync_events.push_back(e);
            }
            IMCOM_HIP_CHECK(hipEventRecord(ctx->sync_events[2], ctx->stream));
            ctx->deferred_flags = (long)cnt;
            return IMCOM_OK;
        }
        IMCOM_HIP_CHECK(hipMemcpyAsync(fail.data(), fail_dev, fail.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (masked && attempt == 0)
            for (int s = 0; s < batch; s++)
                if (active[s] && known[s]) fail[s] = 1;  // (the caller's knowledge stands for the factorisation that was not repeated)
        bool any = false;
        std::vector<int> need;  // stamps whose smallest eigenvalue is wanted: w, v = eigh(A); shift by |w[0]| + 1e-16
        for (int s = 0; s < batch; s++) {
            bool f = false;
            for (int p = 0; p < nv; p++) {
                if (fail[(size_t)p * batch + s] == 0) continue;
                if (repaired[(size_t)p * batch + s]) {
                    set_error("stamp %d node %d: Cholesky failed again after the lakernel.py:262-279 repair (pivot %d)", s, p,
                              fail[(size_t)p * batch + s]);
                    return IMCOM_ERR_NUMERIC;
                }
                f = true;
            }
            if (f && !have_w0[s]) need.push_back(s);
        }
        if (!need.empty()) {
            ProfScope ps(ctx, "eigen_repair");
            std::vector<double> w0v(batch, 0.0);
            std::vector<char> got(batch, 0);
            std::vector<int> big;
            std::vector<char> may_decide(batch, 0), decided(batch, 0);
            if (lmin_subspace_enabled())
                for (int s : need)
                    if (n_host[s] >= LMIN_MIN_N) big.push_back(s);
            if (!big.empty()) {
                // the largest diagonal increment whose factorisation failed: the first trial shift is four times that
                std::vector<double> incf(batch, 0.0);
                for (int s : big)
                    for (int p = 0; p < nv; p++)
                        if (fail[(size_t)p * batch + s] != 0) incf[s] = std::max(incf[s], kappaC_host[p] * C_host[s]);
                ctx->ws_used = ws_after_plan;
                for (int s : big) may_decide[s] = masked && attempt == 0 && expected[s];
                IMCOM_TRY(lambda_min_subspace(ctx, batch, n_host, n_dev, Np, A, big, incf, ctx->repair_hint, may_decide, decided, factor_masked, solve_block, w0v, got));
                ctx->ws_used = ws_after_plan;
            }
            std::vector<int> rest;
            for (int s : need)
                if (!got[s]) rest.push_back(s);
            for (size_t g0 = 0; g0 < rest.size(); g0 += REPAIR_GROUP) {
                const int cnt = (int)std::min<size_t>(REPAIR_GROUP, rest.size() - g0);
                std::vector<double> w0(cnt, 0.0);
                IMCOM_TRY(lambda_min_group(ctx, A, n_host, Np, rest.data() + g0, cnt, w0.data()));
                for (int q = 0; q < cnt; q++) { w0v[rest[g0 + q]] = w0[q]; got[rest[g0 + q]] = 1; }
            }
            for (int s : need) {
                rep[s] = fabs(w0v[s]) + 1e-16;
                have_w0[s] = decided[s] ? 0 : 1;  // (decided: w0v is a lower bound's witness, not the eigenvalue -- should the plain factorisation fail after all, the iteration runs again)
                if (!decided[s]) {
                    ctx->last_w0_min = ctx->last_repair_count ? std::min(ctx->last_w0_min, w0v[s]) : w0v[s];
                    ctx->last_w0_max = ctx->last_repair_count ? std::max(ctx->last_w0_max, w0v[s]) : w0v[s];
                    ctx->last_repair_count++;
                }
                // A stamp whose failure was the CALLER's expectation (redo = 2 without a factorisation having failed here): the smallest
                // eigenvalue says whether A + kappa I is positive definite after all.  If it clearly is (w0 + kappa above 1e-6 kappa: the
                // eigenvalue is good to 1e-11) the reference's cholesky() succeeds and nothing is repaired: the stamp is factored plainly in
                // the next attempt -- and repaired then, should that factorisation fail in spite of the eigenvalue.
                const double kap_s = kappaC_host[0] * C_host[s];
                if (masked && attempt == 0 && expected[s] && w0v[s] + kap_s > 1e-6 * fabs(kap_s)) expected_only[s] = 1;
            }
        }
        std::vector<char> next(batch, 0);
        for (int p = 0; p < nv; p++)
            for (int s = 0; s < batch; s++) {
                if (fail[(size_t)p * batch + s] == 0) continue;
                next[s] = 1;
                any = true;
                if (masked && attempt == 0 && expected_only[s]) continue;  // (solved in the next attempt, without the repair)
                repaired[(size_t)p * batch + s] = 1;
                if (info_host[s] == 0) info_host[s] = p + 1;
            }
        if (!any) break;
        if (masked) active = next;  // the next attempt: the stamps that failed, nothing else
        IMCOM_REQUIRE(attempt <= nv + 1, "repair loop did not terminate");
    }
    if (co) {
        ProfScope ps(ctx, "epilogue");
        if (cfuse.Epart)
            IMCOM_TRY(launch_coadd_from_partials(ctx, batch, n_dev, nblk_dev, Np, m, mp, co->n2, cfuse, co->outimage, Tsum_image, co->Tsum_stamp, co->Tsum_inpix,
                                                 co->Neff));
        else
            IMCOM_TRY(launch_epilogue(ctx, batch, n_dev, Np, m, mp, co->n2f, co->fade, co->n2, Tt, co->indata, co->n_inframe, co->expo, co->n_expo, co->outimage,
                                      Tsum_image, co->Tsum_stamp, co->Tsum_inpix, co->Neff));
    }
    return IMCOM_OK;
}

// The context's second queue, created when a call first needs one (most never do: a stream that exists takes a share of the HIP
// runtime's hardware queues whether or not it carries work).  It carries work that FILLS gaps of the main stream (copies, the Eigen
// path's reflector products): lowest priority, so that a short kernel of the main stream's dependent chain does not queue behind its
// long tiles.
int ensure_aux(imcom_ctx *ctx)
{
    if (ctx->aux_stream) return IMCOM_OK;
    int least = 0, greatest = 0;
    IMCOM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // IMCOM_AUX_CUS = k: the second queue confined to k of the CUs (bit i of the mask = CU i / 8 of XCD i % 8), so that the
    // main stream's one-workgroup-per-stamp kernels always find CUs without a product tile on them (A/B runs)
    const char *cus = getenv("IMCOM_AUX_CUS");
    const int k = cus ? atoi(cus) : 0;
    if (k > 0 && k < ctx->cu_count) {
        std::vector<uint32_t> mask((ctx->cu_count + 31) / 32, 0u);
        for (int i = 0; i < k; i++) mask[i / 32] |= 1u << (i % 32);
        IMCOM_HIP_CHECK(hipExtStreamCreateWithCUMask(&ctx->aux_stream, (uint32_t)mask.size(), mask.data()));
    } else
        IMCOM_HIP_CHECK(hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, least));
    return IMCOM_OK;
}

static int check_ctx(imcom_ctx *ctx)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { set_error("hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e)); return IMCOM_ERR_HIP; }
    return IMCOM_OK;
}

// staging helper: device view of a caller buffer (copy in when host)
struct Staged {
    void *dev = nullptr;
};

}  // namespace imcom

using namespace imcom;

extern "C" {

int imcom_version(void) { return IMCOM_HIP_VERSION; }
int imcom_dev_build(void)
{
#ifdef IMCOM_DEV
    return 1;
#else
    return 0;
#endif
}

const char *imcom_last_error(void) { return g_err; }

int imcom_device_count(int *count)
{
    IMCOM_REQUIRE(count, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return IMCOM_ERR_HIP; }
    *count = c;
    return IMCOM_OK;
}

int imcom_ctx_create(int device, imcom_ctx **out)
{
    IMCOM_REQUIRE(out, "null ctx pointer");
    *out = nullptr;
    int c = 0;
    IMCOM_HIP_CHECK(hipGetDeviceCount(&c));
    IMCOM_REQUIRE(device >= 0 && device < c, "device %d out of range (have %d)", device, c);
    IMCOM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    IMCOM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libimcom_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return IMCOM_ERR_UNSUPPORTED;
    }
    imcom_ctx *ctx = new imcom_ctx();
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    IMCOM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return IMCOM_OK;
}

int imcom_ctx_destroy(imcom_ctx *ctx)
{
    if (!ctx) return IMCOM_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending) { hipEventDestroy(p.start); hipEventDestroy(p.stop); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->ws && !ctx->ws_external) hipFree(ctx->ws);
    if (ctx->pin) hipHostFree(ctx->pin);
    for (auto e : ctx->sync_events) hipEventDestroy(e);
    if (ctx->stream_event) hipEventDestroy(ctx->stream_event);
    if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
    for (auto s_ : ctx->sub_streams) hipStreamDestroy(s_);
    for (auto s_ : ctx->part_streams) hipStreamDestroy(s_);
    if (ctx->flag_pin) hipHostFree(ctx->flag_pin);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return IMCOM_OK;
}

int imcom_ctx_set_stream(imcom_ctx *ctx, void *hip_stream)
{
    IMCOM_TRY(check_ctx(ctx));
    hipStream_t next = (hipStream_t)hip_stream;  // NULL = the legacy default stream, which is what torch's default is
    if (next != ctx->stream) {
        // work queued on the old stream may still be using the context's bump workspace, which the next call on the new
        // stream hands out again from offset 0: order the new stream behind it
        if (!ctx->stream_event) IMCOM_HIP_CHECK(hipEventCreateWithFlags(&ctx->stream_event, hipEventDisableTiming));
        IMCOM_HIP_CHECK(hipEventRecord(ctx->stream_event, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamWaitEvent(next, ctx->stream_event, 0));
        ctx->stream = next;
    }
    return IMCOM_OK;
}

int imcom_ctx_sync(imcom_ctx *ctx)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

int imcom_ctx_workspace_bytes(imcom_ctx *ctx, size_t *bytes)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(bytes, "null bytes");
    *bytes = ctx->ws_bytes;
    return IMCOM_OK;
}

int imcom_ctx_workspace_release(imcom_ctx *ctx)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
    if (ctx->ws && !ctx->ws_external) IMCOM_HIP_CHECK(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = ctx->ws_used = 0;
    return IMCOM_OK;
}

// One owner for device memory: from this call on the context works in the caller's buffer and never allocates device memory itself
// (a call that needs more returns IMCOM_ERR_NOMEM, imcom_ctx_workspace_needed says how much).  The analogue in the reference is its
// TEMPFILE "virtual memory" knob (psfutil.py:2056-2085): the user decides where the sub-blocks live, not luck.
int imcom_ctx_set_workspace(imcom_ctx *ctx, void *ptr, size_t bytes)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE((ptr != nullptr) == (bytes > 0), "workspace pointer and size must come together (NULL, 0: none yet)");
    IMCOM_REQUIRE(((uintptr_t)ptr & 255) == 0, "the workspace must be aligned to 256 bytes");
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
    for (auto s_ : ctx->sub_streams) IMCOM_HIP_CHECK(hipStreamSynchronize(s_));
    if (ctx->ws && !ctx->ws_external) IMCOM_HIP_CHECK(hipFree(ctx->ws));
    ctx->ws = (char *)ptr;
    ctx->ws_bytes = bytes;
    ctx->ws_used = 0;
    ctx->ws_external = true;
    return IMCOM_OK;
}

int imcom_ctx_workspace_needed(imcom_ctx *ctx, size_t *bytes)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(bytes, "null bytes");
    *bytes = ctx->ws_need;
    return IMCOM_OK;
}

int imcom_ctx_set_repair_hint(imcom_ctx *ctx, double lmin_abs)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(lmin_abs >= 0.0 && std::isfinite(lmin_abs), "repair hint: a finite |lambda_min| >= 0 (0 clears it)");
    ctx->repair_hint = lmin_abs;
    return IMCOM_OK;
}

int imcom_ctx_set_repair_expect(imcom_ctx *ctx, int expect)
{
    IMCOM_TRY(check_ctx(ctx));
    ctx->repair_expect = expect != 0;
    return IMCOM_OK;
}

int imcom_ctx_last_repair(imcom_ctx *ctx, int *count, double *w0_min, double *w0_max)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(count && w0_min && w0_max, "null output");
    *count = ctx->last_repair_count;
    *w0_min = ctx->last_repair_count ? ctx->last_w0_min : 0.0;
    *w0_max = ctx->last_repair_count ? ctx->last_w0_max : 0.0;
    return IMCOM_OK;
}

// Bytes of workspace imcom_solve_chol_resident (and _begin / _redo) takes: pure arithmetic, for a planner
int imcom_solve_chol_workspace(int batch, int ldn, int m, int ldm, int nv, size_t *bytes)
{
    IMCOM_REQUIRE(batch >= 1 && ldn >= NB && ldn % NB == 0 && ldm >= NB && ldm % NB == 0 && m >= 1 && m <= ldm && nv >= 1 && nv <= CHOL_MAXNV && bytes,
                  "bad sizes (ldn, ldm multiples of 128)");
    *bytes = chol_core_bytes(batch, ldn, m, ldm, nv);
    return IMCOM_OK;
}

int imcom_ctx_profile_enable(imcom_ctx *ctx, int on)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_TRY(profile_collect(ctx));
    ctx->profile = on != 0;
    ctx->profile_fine = on >= 2;
    return IMCOM_OK;
}

int imcom_ctx_profile_reset(imcom_ctx *ctx)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_TRY(profile_collect(ctx));
    ctx->prof.clear();
    return IMCOM_OK;
}

int imcom_ctx_profile_get(imcom_ctx *ctx, const char *family, double *ms, long *launches)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(family && ms && launches, "null argument");
    IMCOM_TRY(profile_collect(ctx));
    auto it = ctx->prof.find(family);
    *ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return IMCOM_OK;
}

int imcom_ctx_mfma_probe(imcom_ctx *ctx, double millis, double *tflops)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(tflops && millis > 0.0 && millis <= 5000.0, "bad arguments (millis in (0, 5000])");
    const int nwg = 2 * ctx->cu_count;  // two 8-wave workgroups per CU: four waves per SIMD, all resident at once
    IMCOM_TRY(ws_reserve(ctx, 4096));
    double *sink = (double *)ws_take(ctx, 8);
    if (!sink) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    hipEvent_t e0, e1;
    IMCOM_HIP_CHECK(hipEventCreate(&e0));
    IMCOM_HIP_CHECK(hipEventCreate(&e1));
    int iters = 200, waves = 8;  // calibrated on a short first run
    double ms = 0.0;
    for (int pass = 0; pass < 2; pass++) {
        IMCOM_HIP_CHECK(hipEventRecord(e0, ctx->stream));
        IMCOM_TRY(launch_mfma_probe(ctx, nwg, iters, sink, &waves));
        IMCOM_HIP_CHECK(hipEventRecord(e1, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        float f = 0.f;
        IMCOM_HIP_CHECK(hipEventElapsedTime(&f, e0, e1));
        ms = f;
        if (pass == 0) iters = (int)std::max(200.0, std::min(2.0e7, iters * millis / std::max(ms, 1e-3)));
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *tflops = (double)nwg * waves * 8.0 * iters * 2048.0 / (ms * 1e-3) / 1e12;  // 16 x 16 x 4 x 2 flop per MFMA and wave
    return IMCOM_OK;
}

int imcom_ctx_gemm_probe(imcom_ctx *ctx, int variant, int M, int N, int K, int batch, int reps, double *tflops)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(tflops && variant >= 0 && variant <= 8 && M >= 256 && N >= 128 && K >= 16 && batch >= 1 && reps >= 1, "bad arguments");
    IMCOM_REQUIRE(M % 256 == 0 && N % 128 == 0 && K % 16 == 0, "gemm probe: M % 256, N % 128, K % 16");
    const size_t a = (size_t)batch * M * K * 8, b = (size_t)batch * K * N * 8, c = (size_t)batch * M * N * 8;
    IMCOM_TRY(ws_reserve(ctx, a + b + c + 4096));
    double *A = (double *)ws_take(ctx, a), *B = (double *)ws_take(ctx, b), *C = (double *)ws_take(ctx, c);
    if (!A || !B || !C) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(launch_probe_fill(ctx, A, (long)(a / 8), 1u));  // pseudo-random operands: the power (and clock) of real data
    IMCOM_TRY(launch_probe_fill(ctx, B, (long)(b / 8), 2u));
    hipEvent_t e0, e1;
    IMCOM_HIP_CHECK(hipEventCreate(&e0));
    IMCOM_HIP_CHECK(hipEventCreate(&e1));
    auto run = [&]() -> int {
#ifdef IMCOM_DEV
        if (variant == 1) return launch_gemm_probe16(ctx, M, N, K, batch, A, B, C);  // probe_gemm.hip
#else
        if (variant == 1) { set_error("gemm probe variant 1 (256 x 128 tiles) is part of the developer build (make DEV=1)"); return IMCOM_ERR_ARG; }
#endif
        // variants 2-4: the engine's other operand layouts (2: both row-major, the Cholesky updates' form C = A B^T; 3: both k-major,
        // the backward solves'; 4: A k-major, B row-major) on the same buffers
        if (variant >= 5) return launch_gemm_abl(ctx, variant - 4, M, N, K, batch, A, B, C);  // 5: no DMA, 6: no barrier, 7: neither, 8: DMA and barrier but no wait for the DMA
        if (variant == 2) return launch_gemm(ctx, false, false, M, N, K, batch, A, K, (long)M * K, B, K, (long)K * N, C, N, (long)M * N, 1.0, 0.0);
        if (variant == 3) return launch_gemm(ctx, true, true, M, N, K, batch, A, M, (long)M * K, B, N, (long)K * N, C, N, (long)M * N, 1.0, 0.0);
        if (variant == 4) return launch_gemm(ctx, true, false, M, N, K, batch, A, M, (long)M * K, B, K, (long)K * N, C, N, (long)M * N, 1.0, 0.0);
        return launch_gemm(ctx, false, true, M, N, K, batch, A, K, (long)M * K, B, N, (long)K * N, C, N, (long)M * N, 1.0, 0.0);
    };
    IMCOM_TRY(run());  // warm-up
    IMCOM_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; r++) IMCOM_TRY(run());
    IMCOM_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    IMCOM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *tflops = 2.0 * M * N * K * batch * reps / (ms * 1e-3) / 1e12;
    return IMCOM_OK;
}

// ---------------------------------------------------------------------------------------------
// native-routine seam: host pointers are staged through the workspace; device pointers used as is.
#define IMCOM_STAGE_IN(T, name, src, count)                                                   \
    T *name = (T *)(src);                                                                     \
    if (host) {                                                                               \
        name = (T *)ws_take(ctx, (size_t)(count) * sizeof(T));                                \
        if (!name) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; } \
        IMCOM_HIP_CHECK(hipMemcpyAsync(name, src, (size_t)(count) * sizeof(T), hipMemcpyHostToDevice, ctx->stream)); \
    }
#define IMCOM_STAGE_OUT(T, name, dst, count)                                                  \
    if (host) {                                                                               \
        IMCOM_HIP_CHECK(hipMemcpyAsync(dst, name, (size_t)(count) * sizeof(T), hipMemcpyDeviceToHost, ctx->stream)); \
    }

int imcom_d5512_getw(imcom_ctx *ctx, const double *fh, long n, double *w, int memspace)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(n >= 0 && (n == 0 || (fh && w)), "bad arguments");
    if (n == 0) return IMCOM_OK;
    const bool host = memspace == IMCOM_MEM_HOST;
    if (host) IMCOM_TRY(ws_reserve(ctx, (size_t)n * 11 * 8 + 1024));
    IMCOM_STAGE_IN(double, fh_d, fh, n);
    IMCOM_STAGE_IN(double, w_d, w, n * 10);
    IMCOM_TRY(launch_getw(ctx, fh_d, n, w_d));
    IMCOM_STAGE_OUT(double, w_d, w, n * 10);
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

int imcom_interp_d5512(imcom_ctx *ctx, const double *infunc, int nlayer, int ngy, int ngx, const double *xpos,
                       const double *ypos, long nout, double *fhatout, int sym, int memspace)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(infunc && xpos && ypos && fhatout, "null pointer");
    IMCOM_REQUIRE(nlayer >= 1 && ngy >= 1 && ngx >= 1 && nout >= 0, "bad sizes");
    if (sym) {
        long sq = (long)sqrt((double)(nout + 1));
        IMCOM_REQUIRE(sq * sq <= nout, "iD5512C_sym: nout=%ld smaller than its square side^2", nout);
    }
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t ntab = (size_t)nlayer * ngy * ngx, no = (size_t)nlayer * nout;
    if (host) IMCOM_TRY(ws_reserve(ctx, (ntab + 2 * (size_t)nout + no) * 8 + 4096));
    IMCOM_STAGE_IN(double, f_d, infunc, ntab);
    IMCOM_STAGE_IN(double, x_d, xpos, nout);
    IMCOM_STAGE_IN(double, y_d, ypos, nout);
    IMCOM_STAGE_IN(double, o_d, fhatout, no);  // in-place semantics: untouched elements keep their values
    { ProfScope ps(ctx, "interp"); IMCOM_TRY(launch_interp(ctx, f_d, nlayer, ngy, ngx, x_d, y_d, nout, o_d, sym)); }
    IMCOM_STAGE_OUT(double, o_d, fhatout, no);
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

int imcom_grid_d5512(imcom_ctx *ctx, const double *infunc, int ngy, int ngx, const double *xpos, const double *ypos,
                     long npi, int nxo, int nyo, double *fhatout, int memspace)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(infunc && xpos && ypos && fhatout, "null pointer");
    IMCOM_REQUIRE(ngy >= 10 && ngx >= 10 && npi >= 0 && nxo >= 1 && nyo >= 1, "bad sizes");
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t ntab = (size_t)ngy * ngx, no = (size_t)npi * nxo * nyo;
    if (host) IMCOM_TRY(ws_reserve(ctx, (ntab + (size_t)npi * (nxo + nyo) + no) * 8 + 4096));
    IMCOM_STAGE_IN(double, f_d, infunc, ntab);
    IMCOM_STAGE_IN(double, x_d, xpos, (size_t)npi * nxo);
    IMCOM_STAGE_IN(double, y_d, ypos, (size_t)npi * nyo);
    double *o_d = fhatout;
    if (host) { o_d = (double *)ws_take(ctx, no * 8); if (!o_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; } }
    { ProfScope ps(ctx, "interp"); IMCOM_TRY(launch_grid(ctx, f_d, ngy, ngx, x_d, y_d, npi, nxo, nyo, o_d)); }
    IMCOM_STAGE_OUT(double, o_d, fhatout, no);
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

int imcom_lakernel1(imcom_ctx *ctx, const double *lam, const double *mPhalf, long m, long n, double C, double targetleak,
                    double kCmin, double kCmax, int nbis, double *kappa, double *Sigma, double *UC, double *T, double smax,
                    int memspace)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(lam && mPhalf && kappa && Sigma && UC && T, "null pointer");
    IMCOM_REQUIRE(m >= 0 && n >= 0 && nbis >= 0, "bad sizes");
    const bool host = memspace == IMCOM_MEM_HOST;
    if (host) IMCOM_TRY(ws_reserve(ctx, ((size_t)n + 2 * (size_t)m * n + 3 * (size_t)m) * 8 + 8192));
    IMCOM_STAGE_IN(double, lam_d, lam, n);
    IMCOM_STAGE_IN(double, p_d, mPhalf, (size_t)m * n);
    double *k_d = kappa, *S_d = Sigma, *U_d = UC, *T_d = T;
    if (host) {
        k_d = (double *)ws_take(ctx, (size_t)m * 8); S_d = (double *)ws_take(ctx, (size_t)m * 8);
        U_d = (double *)ws_take(ctx, (size_t)m * 8); T_d = (double *)ws_take(ctx, (size_t)m * n * 8);
        if (!k_d || !S_d || !U_d || !T_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    }
    { ProfScope ps(ctx, "lakernel1"); IMCOM_TRY(launch_lakernel1(ctx, lam_d, p_d, m, n, n, C, targetleak, kCmin, kCmax, nbis, k_d, S_d, U_d, T_d, n, smax)); }
    IMCOM_STAGE_OUT(double, k_d, kappa, m);
    IMCOM_STAGE_OUT(double, S_d, Sigma, m);
    IMCOM_STAGE_OUT(double, U_d, UC, m);
    IMCOM_STAGE_OUT(double, T_d, T, (size_t)m * n);
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

int imcom_build_reduced_T(imcom_ctx *ctx, const double *Nflat, const double *Dflat, const double *Eflat, const double *kappa,
                          int nv, long m, double ucmin, double smax, double *out_kappa, double *out_Sigma, double *out_UC,
                          double *out_w, int memspace)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(Nflat && Dflat && Eflat && kappa && out_kappa && out_Sigma && out_UC && out_w, "null pointer");
    IMCOM_REQUIRE(m >= 0 && nv >= 2, "bad sizes (nv must be >= 2, routine.py:537)");
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t nv2 = (size_t)nv * nv;
    if (host) IMCOM_TRY(ws_reserve(ctx, ((size_t)m * (2 * nv2 + 2 * nv + 3) + nv) * 8 + 8192));
    IMCOM_STAGE_IN(double, N_d, Nflat, (size_t)m * nv2);
    IMCOM_STAGE_IN(double, D_d, Dflat, (size_t)m * nv);
    IMCOM_STAGE_IN(double, E_d, Eflat, (size_t)m * nv2);
    IMCOM_STAGE_IN(double, k_d, kappa, nv);
    double *ok = out_kappa, *oS = out_Sigma, *oU = out_UC, *ow = out_w;
    if (host) {
        ok = (double *)ws_take(ctx, (size_t)m * 8); oS = (double *)ws_take(ctx, (size_t)m * 8);
        oU = (double *)ws_take(ctx, (size_t)m * 8); ow = (double *)ws_take(ctx, (size_t)m * nv * 8);
        if (!ok || !oS || !oU || !ow) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    }
    { ProfScope ps(ctx, "reduced_T"); IMCOM_TRY(launch_build_reduced_T(ctx, N_d, D_d, E_d, k_d, nv, m, ucmin, smax, ok, oS, oU, ow)); }
    IMCOM_STAGE_OUT(double, ok, out_kappa, m);
    IMCOM_STAGE_OUT(double, oS, out_Sigma, m);
    IMCOM_STAGE_OUT(double, oU, out_UC, m);
    IMCOM_STAGE_OUT(double, ow, out_w, (size_t)m * nv);
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

// ---------------------------------------------------------------------------------------------
int imcom_solve_chol(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A, const double *mBhalf,
                     const double *C, const double *kappaC, int nv, double ucmin, double smax, float *T, float *UC,
                     float *Sigma, float *kappa, int *info, int memspace)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n && C && kappaC && UC && Sigma && kappa && info, "null pointer / empty batch");
    IMCOM_REQUIRE(m >= 1 && nv >= 1 && ldn >= 0, "bad sizes");
    IMCOM_REQUIRE(nv <= CHOL_MAXNV, "nv=%d kappa nodes: at most %d", nv, CHOL_MAXNV);
    int nmax = 0;
    for (int s = 0; s < batch; s++) {
        IMCOM_REQUIRE(n[s] >= 0 && n[s] <= ldn, "n[%d]=%d exceeds ldn=%d", s, n[s], ldn);
        if (n[s] > nmax) nmax = n[s];
    }
    IMCOM_REQUIRE(nmax == 0 || (A && mBhalf && T), "null matrix pointer");
    const bool host = memspace == IMCOM_MEM_HOST;
    const int Np = (int)align_up((size_t)(nmax > 0 ? nmax : 1), NB), mp = (int)align_up((size_t)m, NB);
    const size_t szA = (size_t)batch * ldn * ldn, szB = (size_t)batch * m * ldn, szM = (size_t)batch * m;
    WsPlan plan;
    if (host) { plan.add(szA * 8); plan.add(szB * 8); plan.add(szB * 4); plan.add(szM * 4 * 3); }
    plan.add((size_t)batch * Np * Np * 8);  // Ap
    plan.add((size_t)batch * Np * mp * 8);  // Bt
    plan.add((size_t)batch * Np * mp * 4);  // Tt
    plan.add((size_t)batch * 4);            // n
    IMCOM_TRY(ws_reserve(ctx, plan.total + chol_core_bytes(batch, Np, m, mp, nv) + 8192));
    IMCOM_STAGE_IN(double, A_d, A, szA);
    // -B/2 is not needed before the triangular solves: from host memory it is uploaded on the context's second queue while the
    // factorisation runs (a third of a stamp's 100 MB over PCIe moves behind 3 ms of Cholesky)
    double *B_d = (double *)mBhalf;
    if (host) {
        B_d = (double *)ws_take(ctx, szB * 8);
        if (!B_d) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; }
    }
    float *T_d = T, *UC_d = UC, *Sig_d = Sigma, *kap_d = kappa;
    if (host) {
        T_d = (float *)ws_take(ctx, szB * 4);
        UC_d = (float *)ws_take(ctx, szM * 4 * 3);
        Sig_d = UC_d + szM;
        kap_d = Sig_d + szM;
    }
    double *Ap = (double *)ws_take(ctx, (size_t)batch * Np * Np * 8);
    double *Bt = (double *)ws_take(ctx, (size_t)batch * Np * mp * 8);
    float *Tt = (float *)ws_take(ctx, (size_t)batch * Np * mp * 4);
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    if (!Ap || !Bt || !Tt || !n_dev || (host && (!T_d || !UC_d))) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(upload(ctx, n_dev, n, batch));
    {
        ProfScope ps(ctx, "pack");
        IMCOM_TRY(launch_pack_A(ctx, A_d, ldn, n_dev, Ap, Np, batch));
    }
    hipEvent_t ev_entry = nullptr;
    if (host && szB > 0) {
        while (ctx->sync_events.size() < 2) {
            hipEvent_t e;
            IMCOM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->sync_events.push_back(e);
        }
        ev_entry = ctx->sync_events[1];
        IMCOM_HIP_CHECK(hipEventRecord(ev_entry, ctx->stream));  // everything earlier calls left on the main stream
    }
    auto stage_B = [&]() -> int {
        if (host && szB > 0) {
            hipEvent_t ev = ctx->sync_events[0];
            // the copy lands in workspace that kernels of an EARLIER, un-synchronised device-mode call on this context may still be
            // using: it waits for the point the main stream had reached when this call began (ev_entry) -- not for this call's own
            // factorisation, behind which it is meant to hide
            IMCOM_TRY(ensure_aux(ctx));
            IMCOM_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ev_entry, 0));
            IMCOM_HIP_CHECK(hipMemcpyAsync(B_d, mBhalf, szB * 8, hipMemcpyHostToDevice, ctx->aux_stream));
            IMCOM_HIP_CHECK(hipEventRecord(ev, ctx->aux_stream));
            IMCOM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev, 0));
        }
        ProfScope ps(ctx, "pack");
        return launch_pack_Bt(ctx, B_d, ldn, m, n_dev, Bt, Np, mp, batch);
    };
    // imcom_ctx_set_repair_expect: the caller has seen the factorisation of A + kappa I fail on the stamps before these (the reference's
    // production shape: every stamp) -- straight to _cholesky_wrapper's repair, as the resident path's redo code 2
    const std::vector<int> expect_all(batch, 3);
    const int rc_core = chol_core(ctx, batch, n, Np, m, mp, Ap, Bt, C, kappaC, nv, ucmin, smax, Tt, UC_d, Sig_d, kap_d, info, stage_B, nullptr, false,
                                  (ctx->repair_expect && nv == 1) ? expect_all.data() : nullptr);
    if (rc_core != IMCOM_OK) {
        if (host && ctx->aux_stream) hipStreamSynchronize(ctx->aux_stream);  // nothing of this call may still be copying into the workspace
        return rc_core;
    }
    if (ldn > 0) { ProfScope ps(ctx, "pack"); IMCOM_TRY(launch_unpack_T(ctx, Tt, Np, mp, n_dev, m, T_d, ldn, batch)); }
    IMCOM_STAGE_OUT(float, T_d, T, szB);
    IMCOM_STAGE_OUT(float, UC_d, UC, szM);
    IMCOM_STAGE_OUT(float, Sig_d, Sigma, szM);
    IMCOM_STAGE_OUT(float, kap_d, kappa, szM);
    if (host) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMCOM_OK;
}

// The same kernel for SEVERAL OutStamps per call, each with its own host arrays (the four OutStamps of a 2 x 2 group, say): one
// batched factorisation / solve for all of them, -B/2 uploaded behind the factorisation.  See imcom_hip.h.
int imcom_solve_chol_stamps(imcom_ctx *ctx, int nst, const int *n, int m, const double *const *A, const double *const *mBhalf,
                            const double *C, const double *kappaC, int nv, double ucmin, double smax, float *const *T,
                            float *const *UC, float *const *Sigma, float *const *kappa, int *info)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(nst >= 1 && n && A && mBhalf && C && kappaC && T && UC && Sigma && kappa && info, "null pointer / no stamps");
    IMCOM_REQUIRE(m >= 1 && nv >= 1, "bad sizes");
    IMCOM_REQUIRE(nv <= CHOL_MAXNV, "nv=%d kappa nodes: at most %d", nv, CHOL_MAXNV);
    int nmax = 0;
    size_t totA = 0, totB = 0;
    for (int s = 0; s < nst; s++) {
        IMCOM_REQUIRE(n[s] >= 0, "n[%d]=%d", s, n[s]);
        IMCOM_REQUIRE(UC[s] && Sigma[s] && kappa[s] && (n[s] == 0 || (A[s] && mBhalf[s] && T[s])), "null pointer for stamp %d", s);
        nmax = std::max(nmax, n[s]);
        totA += align_up((size_t)n[s] * n[s], 32);
        totB += align_up((size_t)m * n[s], 32);
    }
    const int Np = (int)align_up((size_t)(nmax > 0 ? nmax : 1), NB), mp = (int)align_up((size_t)m, NB);
    const size_t szM = (size_t)nst * m;
    WsPlan plan;
    plan.add(totA * 8); plan.add(totB * 8); plan.add(totB * 4); plan.add(szM * 4 * 3);
    plan.add((size_t)nst * Np * Np * 8);  // Ap
    plan.add((size_t)nst * Np * mp * 8);  // Bt
    plan.add((size_t)nst * Np * mp * 4);  // Tt
    plan.add((size_t)nst * 4);            // n
    IMCOM_TRY(ws_reserve(ctx, plan.total + chol_core_bytes(nst, Np, m, mp, nv) + 8192));
    double *rawA = (double *)ws_take(ctx, totA * 8), *rawB = (double *)ws_take(ctx, totB * 8);
    float *rawT = (float *)ws_take(ctx, totB * 4), *maps = (float *)ws_take(ctx, szM * 4 * 3);
    double *Ap = (double *)ws_take(ctx, (size_t)nst * Np * Np * 8), *Bt = (double *)ws_take(ctx, (size_t)nst * Np * mp * 8);
    float *Tt = (float *)ws_take(ctx, (size_t)nst * Np * mp * 4);
    int *n_dev = (int *)ws_take(ctx, (size_t)nst * 4);
    if (!rawA || !rawB || !rawT || !maps || !Ap || !Bt || !Tt || !n_dev) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    float *UC_d = maps, *Sig_d = maps + szM, *kap_d = maps + 2 * szM;
    IMCOM_TRY(upload(ctx, n_dev, n, nst));
    std::vector<size_t> offA(nst), offB(nst);
    {
        size_t a = 0, b = 0;
        for (int s = 0; s < nst; s++) { offA[s] = a; offB[s] = b; a += align_up((size_t)n[s] * n[s], 32); b += align_up((size_t)m * n[s], 32); }
    }
    while (ctx->sync_events.size() < 2) {
        hipEvent_t e;
        IMCOM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_events.push_back(e);
    }
    const bool timing = getenv("IMCOM_SEAM_TIMING") != nullptr;  // host-side timeline of the call on stderr (adds a synchronisation)
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double tA = 0, tB0 = 0, tB1 = 0, tcore = 0, tsync = 0;
    // the system matrices one after the other on the main stream, each packed as soon as it is there
    for (int s = 0; s < nst; s++) {
        if (n[s] == 0) continue;
        IMCOM_HIP_CHECK(hipMemcpyAsync(rawA + offA[s], A[s], (size_t)n[s] * n[s] * 8, hipMemcpyHostToDevice, ctx->stream));
        ProfScope ps(ctx, "pack");
        IMCOM_TRY(launch_pack_A(ctx, rawA + offA[s], n[s], n_dev + s, Ap + (size_t)s * Np * Np, Np, 1));
    }
    hipEvent_t ev_entry = ctx->sync_events[1];
    IMCOM_HIP_CHECK(hipEventRecord(ev_entry, ctx->stream));
    tA = now();
    // -B/2 is called for when the factorisation's launches are queued: its upload (host-blocking from pageable memory) runs on the
    // second queue while the GPU factors
    auto stage_B = [&]() -> int {
        hipEvent_t ev = ctx->sync_events[0];
        tB0 = now();
        IMCOM_TRY(ensure_aux(ctx));
        IMCOM_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ev_entry, 0));
        for (int s = 0; s < nst; s++)
            if (n[s] > 0) IMCOM_HIP_CHECK(hipMemcpyAsync(rawB + offB[s], mBhalf[s], (size_t)m * n[s] * 8, hipMemcpyHostToDevice, ctx->aux_stream));
        IMCOM_HIP_CHECK(hipEventRecord(ev, ctx->aux_stream));
        IMCOM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev, 0));
        tB1 = now();
        ProfScope ps(ctx, "pack");
        for (int s = 0; s < nst; s++)
            IMCOM_TRY(launch_pack_Bt(ctx, rawB + offB[s], n[s], m, n_dev + s, Bt + (size_t)s * Np * mp, Np, mp, 1));
        return IMCOM_OK;
    };
    const std::vector<int> expect_all(nst, 3);  // (imcom_ctx_set_repair_expect: see imcom_solve_chol)
    const int rc_core = chol_core(ctx, nst, n, Np, m, mp, Ap, Bt, C, kappaC, nv, ucmin, smax, Tt, UC_d, Sig_d, kap_d, info, stage_B, nullptr, false,
                                  (ctx->repair_expect && nv == 1) ? expect_all.data() : nullptr);
    tcore = now();
    if (timing) { hipStreamSynchronize(ctx->stream); tsync = now(); }
    if (rc_core != IMCOM_OK) {
        if (ctx->aux_stream) hipStreamSynchronize(ctx->aux_stream);  // nothing of this call may still be copying into the workspace
        return rc_core;
    }
    for (int s = 0; s < nst; s++) {
        if (n[s] > 0) {
            { ProfScope ps(ctx, "pack"); IMCOM_TRY(launch_unpack_T(ctx, Tt + (size_t)s * Np * mp, Np, mp, n_dev + s, m, rawT + offB[s], n[s], 1)); }
            IMCOM_HIP_CHECK(hipMemcpyAsync(T[s], rawT + offB[s], (size_t)m * n[s] * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        IMCOM_HIP_CHECK(hipMemcpyAsync(UC[s], UC_d + (size_t)s * m, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(Sigma[s], Sig_d + (size_t)s * m, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpyAsync(kappa[s], kap_d + (size_t)s * m, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (timing)
        fprintf(stderr, "[imcom] solve_chol_stamps nst=%d: A up + pack queued %.2f ms, B copies start %.2f end %.2f, core queued %.2f, compute done %.2f, T down %.2f\n",
                nst, tA - t0, tB0 - t0, tB1 - t0, tcore - t0, tsync - t0, now() - t0);
    return IMCOM_OK;
}

int imcom_solve_chol_resident(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm, const double *A,
                              const double *Bt, const double *C_host, const double *kappaC_host, int nv, double ucmin,
                              double smax, float *Tt, float *UC, float *Sigma, float *kappa, int *info_host)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && A && Bt && C_host && kappaC_host && Tt && UC && Sigma && kappa && info_host, "null pointer");
    IMCOM_REQUIRE(ldn >= NB && ldn % NB == 0 && ldm % NB == 0 && m >= 1 && m <= ldm && nv >= 1, "ldn=%d / ldm=%d must be multiples of %d", ldn, ldm, NB);
    IMCOM_REQUIRE(nv <= CHOL_MAXNV, "nv=%d kappa nodes: at most %d", nv, CHOL_MAXNV);
    IMCOM_TRY(ws_reserve(ctx, chol_core_bytes(batch, ldn, m, ldm, nv)));
    return chol_core(ctx, batch, n_host, ldn, m, ldm, A, Bt, C_host, kappaC_host, nv, ucmin, smax, Tt, UC, Sigma, kappa, info_host);
}

// The same solve in two halves, for a caller that has host work to do while the device factors and solves (blockrun.coadd_block
// prepares the next pass in between): _begin queues everything of the FIRST attempt and returns; _end waits for it and reads the
// factorisations' failure flags.  All positive definite (the normal case): info = 0, IMCOM_OK, the outputs are final.  Otherwise _end
// returns 1 and has written nothing: the caller runs imcom_solve_chol_resident on the same arguments, which repairs as the reference does
// (lakernel.py:262-279).  Between the two calls the caller may queue other work on the context (it runs behind the solve and may reuse
// the workspace), but no other solve.
int imcom_solve_chol_resident_begin(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm, const double *A, const double *Bt,
                                    const double *C_host, const double *kappaC_host, int nv, double ucmin, double smax, float *Tt, float *UC,
                                    float *Sigma, float *kappa)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && A && Bt && C_host && kappaC_host && Tt && UC && Sigma && kappa, "null pointer");
    IMCOM_REQUIRE(ldn >= NB && ldn % NB == 0 && ldm % NB == 0 && m >= 1 && m <= ldm && nv >= 1, "ldn=%d / ldm=%d must be multiples of %d", ldn, ldm, NB);
    IMCOM_REQUIRE(nv <= CHOL_MAXNV, "nv=%d kappa nodes: at most %d", nv, CHOL_MAXNV);
    if (ctx->deferred_flags != 0) {  // a begin whose end never came (the caller failed in between): its work is waited for and forgotten
        IMCOM_HIP_CHECK(hipEventSynchronize(ctx->sync_events[2]));
        ctx->deferred_flags = 0;
    }
    IMCOM_TRY(ws_reserve(ctx, chol_core_bytes(batch, ldn, m, ldm, nv)));
    std::vector<int> info(batch, 0);
    return chol_core(ctx, batch, n_host, ldn, m, ldm, A, Bt, C_host, kappaC_host, nv, ucmin, smax, Tt, UC, Sigma, kappa, info.data(), nullptr, nullptr, true);
}

int imcom_solve_chol_resident_end(imcom_ctx *ctx, int batch, int *info_host)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(info_host && batch >= 1, "null pointer");
    IMCOM_REQUIRE(ctx->deferred_flags > 0 && ctx->deferred_flags % batch == 0, "imcom_solve_chol_resident_end without a matching begin");
    const long cnt = ctx->deferred_flags;
    ctx->deferred_flags = 0;
    IMCOM_HIP_CHECK(hipEventSynchronize(ctx->sync_events[2]));
    // a factorisation failed: info says whose (info[s] = 1 + the first kappa node that failed), and imcom_solve_chol_resident_redo (one kappa
    // node) or imcom_solve_chol_resident (all stamps again) repairs as the reference does
    bool any = false;
    for (int s = 0; s < batch; s++) info_host[s] = 0;
    for (long q = 0; q < cnt; q++)
        if (ctx->flag_pin[q] != 0) {
            any = true;
            const int s = (int)(q % batch), p = (int)(q / batch);
            if (info_host[s] == 0) info_host[s] = p + 1;
        }
    return any ? 1 : IMCOM_OK;
}

// The solve for SOME stamps of a resident batch (one kappa node): redo_host[s] = 0 leaves stamp s and all its outputs alone, 1 solves
// it, 2 solves it knowing that the plain factorisation fails -- what imcom_solve_chol_resident_end has just reported -- so that the repair
// of lakernel.py:262-279 starts at once.  A batch in which a few factorisations fail costs those stamps again, not the batch.
int imcom_solve_chol_resident_redo(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm, const double *A, const double *Bt,
                                   const double *C_host, const double *kappaC_host, int nv, double ucmin, double smax, float *Tt, float *UC,
                                   float *Sigma, float *kappa, const int *redo_host, int *info_host)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && A && Bt && C_host && kappaC_host && Tt && UC && Sigma && kappa && redo_host && info_host, "null pointer");
    IMCOM_REQUIRE(ldn >= NB && ldn % NB == 0 && ldm % NB == 0 && m >= 1 && m <= ldm, "ldn=%d / ldm=%d must be multiples of %d", ldn, ldm, NB);
    IMCOM_REQUIRE(nv == 1, "imcom_solve_chol_resident_redo: one kappa node (nv=%d: call imcom_solve_chol_resident)", nv);
    if (ctx->deferred_flags != 0) {
        IMCOM_HIP_CHECK(hipEventSynchronize(ctx->sync_events[2]));
        ctx->deferred_flags = 0;
    }
    IMCOM_TRY(ws_reserve(ctx, chol_core_bytes(batch, ldn, m, ldm, nv)));
    return chol_core(ctx, batch, n_host, ldn, m, ldm, A, Bt, C_host, kappaC_host, nv, ucmin, smax, Tt, UC, Sigma, kappa, info_host, nullptr, nullptr, false, redo_host);
}

// CholKernel on the device layouts followed by the coaddition of the same stamps, in one call: with one kappa node and fade 0
// the coaddition's sums are taken from the tiles of T while the backward launches still hold them (no pass over T afterwards);
// otherwise the call is imcom_solve_chol_resident + the map tapers' caller + imcom_coadd_epilogue back to back.
int imcom_solve_chol_resident_coadd(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm, const double *A, const double *Bt,
                                    const double *C_host, const double *kappaC_host, int nv, double ucmin, double smax, float *Tt, float *UC,
                                    float *Sigma, float *kappa, int *info_host, int n2f, int fade, int n2, const float *indata, int n_inframe,
                                    const int *expo, int n_expo, float *outimage, double *Tsum_stamp, double *Tsum_inpix, double *Neff)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && A && Bt && C_host && kappaC_host && Tt && UC && Sigma && kappa && info_host, "null pointer");
    IMCOM_REQUIRE(indata && expo && outimage && Tsum_stamp && Tsum_inpix && Neff, "null pointer (coaddition)");
    IMCOM_REQUIRE(ldn >= NB && ldn % NB == 0 && ldm % NB == 0 && m >= 1 && m <= ldm && nv >= 1, "ldn=%d / ldm=%d must be multiples of %d", ldn, ldm, NB);
    IMCOM_REQUIRE(nv <= CHOL_MAXNV, "nv=%d kappa nodes: at most %d", nv, CHOL_MAXNV);
    IMCOM_REQUIRE(m == n2f * n2f && n_inframe >= 1 && n_expo >= 1 && fade >= 0 && n2 >= 1, "bad sizes (coaddition)");
    IMCOM_REQUIRE(fade == 0, "imcom_solve_chol_resident_coadd: fade = %d -- the map tapers of coadd.py:1118-1122 come between the solve and the coaddition: "
                             "call imcom_solve_chol_resident, imcom_trapezoid_f32, imcom_coadd_epilogue", fade);
    const CoaddArgs co{n2f, fade, n2, n_inframe, n_expo, indata, expo, outimage, Tsum_stamp, Tsum_inpix, Neff};
    IMCOM_TRY(ws_reserve(ctx, chol_core_bytes(batch, ldn, m, ldm, nv) + coadd_fuse_bytes(batch, ldn, m, ldm, nv, &co)));
    return chol_core(ctx, batch, n_host, ldn, m, ldm, A, Bt, C_host, kappaC_host, nv, ucmin, smax, Tt, UC, Sigma, kappa, info_host, nullptr, &co);
}

// ---------------------------------------------------------------------------------------------
int imcom_build_A(imcom_ctx *ctx, int batch, const int *n_host, int ldn, const double *x, const double *y, const int *psf,
                  const double *tables, int ntab, const imcom_table_geom *geom, const int *pair_tab, const double *pair_pen,
                  int npsf_max, double *A)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && x && y && psf && tables && geom && pair_tab && pair_pen && A, "null pointer");
    IMCOM_REQUIRE(ldn >= 1 && ntab >= 1 && npsf_max >= 1 && geom->nsamp >= 1 && geom->dscale > 0, "bad sizes");
    const long nt_ = (ldn + 15) / 16, tiles_ = nt_ * (nt_ + 1) / 2 * batch;  // per-stamp tile order by PSF pair: key + descriptor per tile, bins per stamp
    IMCOM_TRY(ws_reserve(ctx, (size_t)batch * 4 + (size_t)tiles_ * 8 + (size_t)batch * ((size_t)npsf_max * npsf_max + 1) * 4 + 8192));
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    IMCOM_TRY(upload(ctx, n_dev, n_host, batch));
    ProfScope ps(ctx, "build_A");
    return launch_build_A(ctx, batch, n_dev, ldn, x, y, psf, tables, ntab, geom->nsamp + 12, geom->nc, geom->dscale, pair_tab,
                          pair_pen, npsf_max, A);
}

int imcom_build_B(imcom_ctx *ctx, int batch, const int *n_host, int ldn, const double *x, const double *y, const int *psf,
                  const double *tables, int ntab, const imcom_table_geom *geom, const int *io_tab, int npsf_max,
                  const double *out_x0, const double *out_y0, int n2f, int ldm, double *Bt)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && x && y && psf && tables && geom && io_tab && out_x0 && out_y0 && Bt, "null pointer");
    IMCOM_REQUIRE(ldn >= 1 && ntab >= 1 && npsf_max >= 1 && n2f >= 1 && ldm >= n2f * n2f, "bad sizes");
    IMCOM_TRY(ws_reserve(ctx, (size_t)batch * 4 + 1024));
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    IMCOM_TRY(upload(ctx, n_dev, n_host, batch));
    ProfScope ps(ctx, "build_B");
    return launch_build_B(ctx, batch, n_dev, ldn, x, y, psf, tables, geom->nsamp + 12, geom->nc, geom->dscale, io_tab,
                          npsf_max, out_x0, out_y0, n2f, ldm, Bt);
}

int imcom_coadd_epilogue(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm, int n2f, int fade, int n2,
                         float *Tt, const float *indata, int n_inframe, const int *expo, int n_expo, float *outimage,
                         double *Tsum_stamp, double *Tsum_inpix, double *Neff)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(batch >= 1 && n_host && Tt && indata && expo && outimage && Tsum_stamp && Tsum_inpix && Neff, "null pointer");
    IMCOM_REQUIRE(m == n2f * n2f && m <= ldm && n_inframe >= 1 && n_expo >= 1 && fade >= 0 && n2 >= 1, "bad sizes");
    IMCOM_TRY(ws_reserve(ctx, (size_t)batch * 4 + (size_t)batch * m * n_expo * 8 + 2048));
    int *n_dev = (int *)ws_take(ctx, (size_t)batch * 4);
    double *Tsum_image = (double *)ws_take(ctx, (size_t)batch * m * n_expo * 8);
    IMCOM_TRY(upload(ctx, n_dev, n_host, batch));
    ProfScope ps(ctx, "epilogue");
    return launch_epilogue(ctx, batch, n_dev, ldn, m, ldm, n2f, fade, n2, Tt, indata, n_inframe, expo, n_expo, outimage,
                           Tsum_image, Tsum_stamp, Tsum_inpix, Neff);
}

int imcom_trapezoid_f32(imcom_ctx *ctx, float *maps, long nmaps, int n2f, int fade)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(maps && nmaps >= 0 && n2f >= 1 && fade >= 0, "bad arguments");
    return launch_trapezoid_f32(ctx, maps, nmaps, n2f, fade);
}

int imcom_clamp_min_f32(imcom_ctx *ctx, float *maps, long count, float lo)
{
    IMCOM_TRY(check_ctx(ctx));
    IMCOM_REQUIRE(count >= 0 && (count == 0 || maps), "bad arguments");
    return launch_clamp_min_f32(ctx, maps, count, lo);
}

}  // extern "C"
