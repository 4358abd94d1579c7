// launchers.h -- host-side launch functions implemented next to their kernels.
#pragma once
#include <functional>

#include "common.h"

namespace imcom {

// gemm_f64.hip
int launch_chol_update(imcom_ctx *ctx, const double *A, double *L, int ldn, int k, int nbmax, int batch, int abatch,
                       const int *nblk, const double *dshift, double *partial, int nparts);  // nparts > 1: split-K through `partial`
int launch_gemm_abl(imcom_ctx *ctx, int abl, int M, int N, int K, int batch, const double *A, const double *B, double *C);  // diagnostic
int launch_chol_trsm(imcom_ctx *ctx, double *L, const double *Dinv, int ldn, int k, int nbmax, int batch,
                     const int *nblk);
int launch_solve_fwd(imcom_ctx *ctx, const double *L, const double *Bt, double *Y, int ldn, int ldm, int k,
                     int batch, int bbatch, const int *nblk, const int *n, const double *Dinv, double *partial, int nparts,
                     double *Dpart);  // Dinv != null: Linv[k] applied in the same launch; Dpart: column sums of Y^2 per block row
// The coaddition sums (coadd.py:1320-1354) riding in the backward launches: per block row and wave row of a tile, the column sums
// sum_j T[j][a] w_q[j] over the tile's rows with w_q = the indicator of exposure q (q < n_expo) or input frame q - n_expo --
// Epart [batch][2 ldn / 128][n_expo + n_inframe][ldm]; launch_coadd_from_partials adds them up in a fixed order.
struct CoaddFuse {
    const float *indata = nullptr;  // [batch][n_inframe][ldn]
    const int *expo = nullptr;      // [batch][ldn]
    int n_inframe = 0, n_expo = 0;
    double *Epart = nullptr;
};
int launch_solve_bwd(imcom_ctx *ctx, const double *L, double *Y, int ldn, int ldm, int k, int nbmax, int batch,
                     const int *nblk, const int *n, const double *Dinv, double *partial, int nparts, double *Npart, float *Tt,
                     const CoaddFuse *cf = nullptr);
int launch_coadd_from_partials(imcom_ctx *ctx, int batch, const int *n_dev, const int *nblk_dev, int ldn, int m, int ldm, int n2, const CoaddFuse &cf,
                               float *outimage, double *Tsum_image, double *Tsum_stamp, double *Tsum_inpix, double *Neff);
int launch_solve_dinv(imcom_ctx *ctx, const double *Dinv, double *Y, int ldn, int ldm, int k, int batch,
                      const int *nblk, bool trans);
int launch_probe_fill(imcom_ctx *ctx, double *p, long count, unsigned seed);
int launch_gemm_probe16(imcom_ctx *ctx, int M, int N, int K, int batch, const double *A, const double *B, double *C);
int launch_mfma_probe(imcom_ctx *ctx, int nwg, int iters, double *sink, int *waves_per_wg);
int launch_syr2k_lower(imcom_ctx *ctx, int N, int K, int batch, const double *V, long ldv, long strideV, const double *W, long ldw, long strideW,
                       double *C, long ldc, long strideC, double alpha);  // lower 128-tiles of C += alpha (V^T W + W^T V), V, W k-major
int launch_gemm(imcom_ctx *ctx, bool akm, bool bkm, int M, int N, int K, int batch, const double *A, long lda,
                long strideA, const double *B, long ldb, long strideB, double *C, long ldc, long strideC,
                double alpha, double beta);

// tridiag.hip (dispatch between the tridiagonal QR eigensolver and the Jacobi cross-check, jacobi.hip)
bool eigh_uses_jacobi();
size_t eigh_ws_bytes(int batch, int ld, bool vectors);
int eigh_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, const double *A, long lda, long strideA, double *lam,
                long ldlam, double *Q, long ldq, long strideQ, int *sweeps_out);

// tridiag.hip: the tridiagonal basis A = Qh T Qh^T without eigenvectors (eigen.hip works in it)
constexpr int BAND_BW = 4;  // bandwidth of band.hip's reduction
struct TrdBasis {
    double *Vall, *dvec, *evec, *tauvec;  // reflectors [batch][ld][ld] (row j = v_j), T's diagonal / off-diagonal, tau [batch][ld]
    double *Tm, *Sm, *W1, *W2;            // triangular factors of the 128-reflector panels [npanels][batch][128][128]; scratch of trd_apply_q
    double *band = nullptr;               // band.hip: [batch][bw + 1][ld], band[t][i] = B[i + t][i]
    double *T2 = nullptr;                 // factors of PAIRS of panels (256 reflectors) [npanels / 2][batch][256][256]; W1, W2 then hold 256 rows
    int *n_dev;
    int ld, nmax, npanels, bw = 1;        // bw: rows between a reflector's column and its pivot (1: tridiagonal basis)
};
size_t trd_basis_ws_bytes(int batch, int ld, int mp);
int trd_basis_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, int mp, const double *A, long lda, long strideA, TrdBasis *out);
int trd_apply_q(imcom_ctx *ctx, const TrdBasis &b, int batch, double *C, int mp, bool transpose);
int trd_pair_factors(imcom_ctx *ctx, TrdBasis *out, int batch);  // T2 from Tm (all panels' factors must be there)
// band.hip: the same with A = Q B Q^T, B of bandwidth BAND_BW (a quarter of the passes over the matrix); ld up to what the panel's LDS holds
bool band_basis_fits(int ld);
size_t band_basis_ws_bytes(int batch, int ld, int mp);
size_t band_basis_keep_bytes(int batch, int ld, int mp);  // the part of it that outlives band_basis_device (reflectors, factors, W1 / W2)
// on_panel(p) (optional) is called on the host as soon as the launches that complete the 128 reflectors of panel p have been queued:
// the caller may start applying them on another stream; the panels' T factors are then the caller's job (trd_panel_step)
int band_basis_device(imcom_ctx *ctx, int batch, const int *n_host, int ld, int mp, const double *A, long lda, long strideA, TrdBasis *out,
                      const std::function<int(int)> &on_panel = nullptr);
int trd_panel_step(imcom_ctx *ctx, const TrdBasis &b, int batch, int p, double *C, int mp);  // T factor of panel p, then C <- (I - V T^T V^T) C (one step of Qh^T C)

// la_kernels.hip
int launch_chol_diag(imcom_ctx *ctx, double *L, double *Dinv, int ldn, int k, int batch, const int *nblk, int *fail);
// chol_diag.hip: T [batch][128][128] of the block reflector of reflectors ps .. ps+127 from S = V V^T [batch][128][128] and tau [batch][ld]
int launch_larft_inv(imcom_ctx *ctx, const double *S, const double *tauvec, int ld, int ps, double *T, int batch);
int launch_diag_shift(imcom_ctx *ctx, const double *A, int ldn, const double *inc, const int *ninc, double *dshift,
                      int batch);
int launch_pack_A(imcom_ctx *ctx, const double *A, long lda, const int *n, double *Ap, int ldp, int batch);
int launch_pack_Bt(imcom_ctx *ctx, const double *B, long ldb, int m, const int *n, double *Bt, int ldp, int ldm,
                   int batch);
int launch_unpack_T(imcom_ctx *ctx, const float *Tt, int ldp, int ldm, const int *n, int m, float *T, long ldt,
                    int batch);
int launch_finalize_fused(imcom_ctx *ctx, const double *Dpart, const double *Npart, int ldn, int ldm, int m, const int *n,
                          const int *nblk, const double *kap, const double *Cs, float *Tt, float *UC, float *Sigma, float *kappa,
                          int batch, const int *act = nullptr);  // act != null: only the stamps with act[s] != 0
int launch_solve_mask(imcom_ctx *ctx, const int *nblk, const int *fac, const int *fail, int *nblk_sol, int *act, int batch);
// lmin_skinny.hip: the smallest-eigenvalue iteration on blocks of 16 vectors [batch][ldn][16] (one workgroup per stamp streams the factor;
// nblk[s] = 0: the stamp is left alone)
constexpr int LMIN_SKINNY_P = 16;
int launch_skinny_solve(imcom_ctx *ctx, const double *L, const double *Dinv, const double *X, double *Y, int ldn, const int *nblk, int batch);  // Y = (L L^T)^-1 X (X may be Y)
int launch_skinny_solve_few(imcom_ctx *ctx, const double *L, const double *Dinv, const double *X, double *Y, int ldn, const int *nblk, int nbmax, int batch,
                            double *partial);  // the same for FEW stamps: two short launches per block row and sweep; partial: skinny_few_partial_doubles(batch) doubles
size_t skinny_few_partial_doubles(int batch);
int launch_skinny_ax(imcom_ctx *ctx, const double *A, const double *X, double *Z, int ldn, const int *nblk, int nbmax, int batch);              // Z = A X
int launch_skinny_orth(imcom_ctx *ctx, const double *src, double *dst, int ldn, const int *nblk, int *fail, int batch);                        // one CholQR pass
int launch_skinny_rr(imcom_ctx *ctx, const double *X, const double *Z, int ldn, const int *nblk, double *lam, double *part, int ngroups, int batch);  // eigenvalues of X^T Z [batch][16], residuals of the two lowest pairs
constexpr int LMIN_RESID_GROUPS = 32;  // row groups of launch_ritz_residual: part is [batch][32][2]
int launch_ritz_residual(imcom_ctx *ctx, const double *X, const double *Z, const double *Qh, const double *lam, int ldn, int P, const int *n,
                         const int *want, double *part, int batch);
int launch_lmin_init(imcom_ctx *ctx, double *X, int ldn, int P, const int *n, const int *want, int batch);
int launch_diag_max(imcom_ctx *ctx, const double *A, int ldn, const int *n, double *dmax, int batch);
int launch_gram_guard(imcom_ctx *ctx, double *G, int P, const int *want, int batch);
int launch_finalize_single(imcom_ctx *ctx, const double *X, const double *Bt, int ldn, int ldm, int m, const int *n,
                           const double *kap, const double *Cs, float *Tt, float *UC, float *Sigma, float *kappa,
                           int batch, const int *act = nullptr);
int launch_multi(imcom_ctx *ctx, const double *Xs, long node_stride, const double *Bt, int ldn, int ldm, int m,
                 const int *n, int nv, const double *kappaC_dev, const double *Cs, double ucmin, double smax,
                 double *Dp, double *Npq, double *W, float *Tt, float *UC, float *Sigma, float *kappa, int batch);
int launch_build_reduced_T(imcom_ctx *ctx, const double *Nf, const double *Df, const double *Ef, const double *kappa,
                           int nv, long m, double ucmin, double smax, double *ok, double *oS, double *oU, double *ow);
int launch_lakernel1(imcom_ctx *ctx, const double *lam, const double *mPhalf, long m, long n, long ldp, double C,
                     double targetleak, double kCmin, double kCmax, int nbis, double *kappa, double *Sigma,
                     double *UC, double *T, long ldt, double smax);
int launch_trapezoid_f32(imcom_ctx *ctx, float *maps, long nmaps, int n2f, int fade);
int launch_clamp_min_f32(imcom_ctx *ctx, float *maps, long count, float lo);
int launch_epilogue(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, int m, int ldm, int n2f, int fade, int n2,
                    float *Tt, const float *indata, int n_inframe, const int *expo, int n_expo, float *outimage,
                    double *Tsum_image, double *Tsum_stamp, double *Tsum_inpix, double *Neff);

// interp.hip
int launch_getw(imcom_ctx *ctx, const double *fh, long n, double *w);
int launch_interp(imcom_ctx *ctx, const double *infunc, int nlayer, int ngy, int ngx, const double *xpos,
                  const double *ypos, long nout, double *fhatout, int sym);
int launch_grid(imcom_ctx *ctx, const double *infunc, int ngy, int ngx, const double *xpos, const double *ypos,
                long npi, int nxo, int nyo, double *fhatout);
int launch_build_A(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, const double *x, const double *y,
                   const int *psf, const double *tables, int ntab, int ng, double nc, double dscale,
                   const int *pair_tab, const double *pair_pen, int npsf_max, double *A);
int launch_build_B(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, const double *x, const double *y,
                   const int *psf, const double *tables, int ng, double nc, double dscale, const int *io_tab,
                   int npsf_max, const double *out_x0, const double *out_y0, int n2f, int ldm, double *Bt);

}  // namespace imcom
