/*
 * imcom_hip.h -- C-ABI of libimcom_hip.so: the MI355X (gfx950) IMCOM postage-stamp path.
 *
 * This is the drop-in boundary for the ONE hot path of pyimcom (reference checkout paths below are
 * relative to src/pyimcom/): per-postage-stamp construction of the PSF-overlap system matrix A and
 * the target cross-correlation -B/2, the Cholesky / eigen solve T = (A + kappa I)^-1 (-B/2), the
 * leakage U/C, noise Sigma and kappa maps, and the coaddition epilogue.  Plain C, caller-allocated
 * buffers, no torch types.  Every entry point returns 0 (IMCOM_OK) or a negative imcom_status;
 * imcom_last_error() gives the message of the calling thread's last failure.
 *
 * Pointers are host or device pointers as selected by `memspace`: with IMCOM_MEM_HOST the library
 * stages through its own device workspace and the call is synchronous; with IMCOM_MEM_DEVICE the
 * work is enqueued on the context's stream (imcom_ctx_set_stream) and the call returns immediately
 * (small per-stamp arrays named "host" below are always host pointers).  One context per GPU and
 * caller thread; a context is not thread-safe.  There is NO CPU fallback anywhere in this library.
 *
 * All matrices are C-order (row-major) float64 unless stated.  N = selected input pixels of a
 * stamp, m = n2f*n2f output pixels, nv = number of kappa nodes, n_out = target PSFs (1 per call
 * here; the host mirror loops over n_out as the reference does, lakernel.py:165,291,349).
 */
#ifndef IMCOM_HIP_H
#define IMCOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMCOM_HIP_VERSION 100 /* 0.1.0 */
#define IMCOM_PAIR_SWAP (1 << 29)
#define IMCOM_PAIR_FLIP (1 << 30)

typedef enum {
    IMCOM_OK = 0,
    IMCOM_ERR_ARG = -1,         /* bad argument (null pointer, bad size, unaligned layout) */
    IMCOM_ERR_HIP = -2,         /* a HIP runtime call failed (no device, launch failure, ...) */
    IMCOM_ERR_NOMEM = -3,       /* device workspace allocation failed */
    IMCOM_ERR_UNSUPPORTED = -4, /* valid request this build cannot serve */
    IMCOM_ERR_NUMERIC = -5      /* non-finite input or a factorisation that cannot be repaired */
} imcom_status;

typedef enum { IMCOM_MEM_HOST = 0, IMCOM_MEM_DEVICE = 1 } imcom_memspace;

typedef struct imcom_ctx imcom_ctx;

/* ------------------------------------------------------------------ library / context --------- */
int imcom_version(void);
/* 1 when the library was built with the developer extras (make DEV=1: experimental kernels and cross-checks), else 0. */
int imcom_dev_build(void);
const char *imcom_last_error(void);
int imcom_device_count(int *count);
/* device: HIP device ordinal.  Creates the context's own stream; block->GPU farming uses one
 * context per GPU (one process per GPU), no inter-GPU traffic (docs/run_README.rst:81-100). */
int imcom_ctx_create(int device, imcom_ctx **ctx);
int imcom_ctx_destroy(imcom_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) for all subsequent work; NULL is the
 * legacy default stream (torch's default stream).  A fresh context runs on its own stream. */
int imcom_ctx_set_stream(imcom_ctx *ctx, void *hip_stream);
int imcom_ctx_sync(imcom_ctx *ctx);
/* Bytes of device workspace currently held by the context (grows on demand, never inside a call
 * whose sizes were seen before). */
int imcom_ctx_workspace_bytes(imcom_ctx *ctx, size_t *bytes);
/* Give the workspace back to the device (after draining the context's streams); the next call allocates what it needs.
 * For drivers that change kernel or batch size between blocks: the workspace otherwise keeps the size of the largest call. */
int imcom_ctx_workspace_release(imcom_ctx *ctx);
/* ONE OWNER for device memory.  By default a context allocates its workspace itself (hipMalloc, growing on demand).  After this
 * call it works in the caller's buffer `ptr` of `bytes` bytes (256-byte aligned; NULL, 0 = "none yet") and never allocates device
 * memory again: a call that needs more returns IMCOM_ERR_NOMEM without having queued anything, imcom_ctx_workspace_needed says how
 * much it asked for, and the caller -- who knows what else lives on the device -- provides it or plans smaller.  The buffer must
 * stay valid until the next imcom_ctx_set_workspace / imcom_ctx_destroy; imcom_ctx_workspace_release only forgets it.  The
 * reference's analogue is TEMPFILE, its "virtual memory" for A sub-blocks (psfutil.py:2056-2085): where they live is the user's call.
 * pyimcom_amd.Context takes the buffer from torch's allocator, so that torch is the only allocator on the device. */
int imcom_ctx_set_workspace(imcom_ctx *ctx, void *ptr, size_t bytes);
/* Workspace bytes the most recent call on this context asked for (whether or not it got them). */
int imcom_ctx_workspace_needed(imcom_ctx *ctx, size_t *bytes);
/* Elapsed device milliseconds spent in the named kernel family since the last reset, measured with
 * HIP events on the context's stream when profiling is enabled (bench.py's roofline leg).
 * family: "solve_gemm", "chol_gemm", "chol_diag", "build_A", "build_B", "finalize", "epilogue",
 *         "eigen", "lakernel1", ... ; launches receives the number of launches accumulated. */
int imcom_ctx_profile_enable(imcom_ctx *ctx, int on); /* on = 2: also per-launch scopes inside long stages ("symv4": the band reduction's pass over the trailing matrix) */
int imcom_ctx_profile_reset(imcom_ctx *ctx);
int imcom_ctx_profile_get(imcom_ctx *ctx, const char *family, double *ms, long *launches);
/* Diagnostic for the roofline: keeps the fp64 MFMA pipe of every SIMD busy -- and nothing else: no LDS, no memory, no
 * barriers -- for about `millis` ms and reports the rate reached [TFLOP/s]: the ceiling of this chip at the clock it holds under
 * matrix load, to set beside the guide's 78.6 TFLOP/s. */
int imcom_ctx_mfma_probe(imcom_ctx *ctx, double millis, double *tflops);
/* Diagnostic: the k loop of the tile engine as a plain batched product C[M][N] = A[M][K] (row-major) B[K][N] on pseudo-random
 * operands in workspace, `reps` launches; variant 0: the production 128 x 128 tiles (two 8-wave workgroups per CU), variant 1: 256 x 128
 * tiles by one 16-wave workgroup per CU (half the B traffic per flop).  M % 256 = N % 128 = K % 16 = 0. */
int imcom_ctx_gemm_probe(imcom_ctx *ctx, int variant, int M, int N, int K, int batch, int reps, double *tflops);

/* ------------------------------------------------------------------ native-routine seam --------
 * Replaces furry_parakeet.pyimcom_croutines.* (imported at lakernel.py:41-47, psfutil.py:37-49),
 * whose in-tree specification is routine.py.  Outputs are written in place; off-grid points of the
 * scattered interpolators leave the output element untouched (routine.py:166-167, 231-232). */

/* routine.py:29-122 iD5512C_getw: w[i*10 .. i*10+9] = the 10 taps for fh[i] (fh = frac - 1/2). */
int imcom_d5512_getw(imcom_ctx *ctx, const double *fh, long n, double *w, int memspace);
/* routine.py:125-181 iD5512C (sym=0) / 184-253 iD5512C_sym (sym=1; nout must be a square, only the
 * upper triangle is interpolated and mirrored).  infunc[nlayer][ngy][ngx], fhatout[nlayer][nout]. */
int imcom_interp_d5512(imcom_ctx *ctx, const double *infunc, int nlayer, int ngy, int ngx,
                       const double *xpos, const double *ypos, long nout, double *fhatout, int sym,
                       int memspace);
/* routine.py:256-338 gridD5512C: infunc[ngy][ngx], xpos[npi][nxo], ypos[npi][nyo],
 * fhatout[npi][nyo*nxo]; off-grid rows/columns contribute zero weight (306-323). */
int imcom_grid_d5512(imcom_ctx *ctx, const double *infunc, int ngy, int ngx, const double *xpos,
                     const double *ypos, long npi, int nxo, int nyo, double *fhatout, int memspace);
/* routine.py:341-430 lakernel1: per-output-pixel geometric bisection on kappa.
 * lam[n], mPhalf[m][n] -> kappa[m], Sigma[m], UC[m], T[m][n] (T still to be multiplied by Q^T). */
int imcom_lakernel1(imcom_ctx *ctx, const double *lam, const double *mPhalf, long m, long n,
                    double C, double targetleak, double kCmin, double kCmax, int nbis,
                    double *kappa, double *Sigma, double *UC, double *T, double smax, int memspace);
/* routine.py:487-588 build_reduced_T_wrap (with lsolve_sps 433-484 inside).
 * Nflat[m*nv*nv], Dflat[m*nv], Eflat[m*nv*nv], kappa[nv] ascending -> out_*[m], out_w[m*nv]. */
int imcom_build_reduced_T(imcom_ctx *ctx, const double *Nflat, const double *Dflat,
                          const double *Eflat, const double *kappa, int nv, long m, double ucmin,
                          double smax, double *out_kappa, double *out_Sigma, double *out_UC,
                          double *out_w, int memspace);

/* ------------------------------------------------------------------ LA-kernel seam --------------
 * Replaces lakernel.CholKernel (lakernel.py:226-394) and lakernel.EigenKernel (141-223) behind the
 * OutStamp.LAKERNEL registry (coadd.py:839-844, 1091-1093), batched over independent stamps.
 *
 *   batch        number of stamps
 *   n[batch]     HOST array: input pixels per stamp (ragged batches allowed; n[s] == 0 gives the
 *                lakernel.py:110-119 outputs UC=1, Sigma=0, kappa=1 and no T)
 *   ldn          row stride (elements) and per-stamp extent of A: A[s] is at A + s*ldn*ldn, row i at
 *                + i*ldn; likewise B rows / T rows are ldn long.  ldn >= max n.
 *   m            output pixels per stamp
 *   A            [batch][ldn][ldn] system matrices (only the leading n[s] x n[s] part is read;
 *                never modified: coadd.py keeps using outst.sysmata when save_abc is set)
 *   mBhalf       [batch][m][ldn]  -B/2 in the reference layout (lakernel.py:287)
 *   C            [batch] HOST: target normalisation (psfutil.py:1290 outovlc)
 *   kappaC[nv]   HOST: kappa/C nodes, ascending (config KAPPAC); nv==1 -> single-kappa path
 *   T            [batch][m][ldn] float32 out (reference dtype, lakernel.py:122)
 *   UC,Sigma,kappa [batch][m] float32 out
 *   info[batch]  HOST out: 0 ok; >0 = the Cholesky repair of lakernel.py:262-279 was applied
 *                (value = node index + 1 of the first repaired factorisation)
 */
int imcom_solve_chol(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A,
                     const double *mBhalf, const double *C, const double *kappaC, int nv,
                     double ucmin, double smax, float *T, float *UC, float *Sigma, float *kappa,
                     int *info, int memspace);
/* lakernel.CholKernel (lakernel.py:226-394) for SEVERAL OutStamps per call -- e.g. the four OutStamps that share the PSF overlap
 * of a 2 x 2 group (coadd.py:1091-1093 calls the kernel of one OutStamp at a time) -- each with its own HOST arrays as the
 * reference holds them: A[s] = outst.sysmata [n[s]][n[s]], mBhalf[s] = outst.mhalfb[j_out] [m][n[s]], T[s] [m][n[s]] float32,
 * UC[s] / Sigma[s] / kappa[s] [m] float32; C[nst], info[nst] as in imcom_solve_chol.  One batched factorisation and solve for
 * all stamps; -B/2 crosses PCIe behind the factorisation.  n[s] == 0 gives the lakernel.py:110-119 outputs and no T. */
int imcom_solve_chol_stamps(imcom_ctx *ctx, int nst, const int *n, int m, const double *const *A,
                            const double *const *mBhalf, const double *C, const double *kappaC, int nv, double ucmin,
                            double smax, float *const *T, float *const *UC, float *const *Sigma, float *const *kappa,
                            int *info);
/* Same contract, eigendecomposition path; nv==1: lakernel.py:154-172, nv>1: 174-223 with nbis
 * bisections (reference default 13) and the kappa *= C quirk of line 222.
 * info[s]: 0 = A + kappaC[0] C I is positive definite (the PSF-overlap matrices of this problem): solved in the band basis
 * (DESIGN.md "Eigen path"); 1 = it is NOT -- A has an eigenvalue at or below -kappaC[0] C, every pivot of the reduced matrix
 * is checked -- and the stamp was solved through the eigendecomposition itself with the reference's own formulas, which
 * divide by lam_i + kappa whatever its sign (numpy.linalg.eigh at lakernel.py:162, 201; routine.lakernel1): the kernel a
 * user falls back to when Cholesky fails serves any symmetric A.  Results are valid either way; never an error. */
int imcom_solve_eigen(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A,
                      const double *mBhalf, const double *C, const double *kappaC, int nv,
                      double ucmin, double smax, int nbis, float *T, float *UC, float *Sigma,
                      float *kappa, int *info, int memspace);
/* Bytes of device workspace imcom_solve_eigen_resident takes for `batch` stamps of these leading dimensions (a planner adds
 * them to its own buffers: pyimcom_amd.blockrun.stamp_bytes).  Pure arithmetic: no context, no device call. */
int imcom_solve_eigen_workspace(int batch, int ldn, int ldm, int m, size_t *bytes);
/* The same for imcom_solve_chol_resident / _begin / _redo (nv kappa nodes; the repair path's scratch included). */
int imcom_solve_chol_workspace(int batch, int ldn, int m, int ldm, int nv, size_t *bytes);
/* Householder reduction of symmetric matrices to band form, the basis the Eigen kernel's kappa search works in
 * (numpy.linalg.eigh at lakernel.py:162, 201 is not needed for it: DESIGN.md "Eigen path"): A = Q B Q^T, B[i][j] = 0 for
 * |i - j| > 4, Q = H_0 H_1 ... with H_r = I - tau_r v_r v_r^T, v_r zero above its pivot row r + 4 (v_r[r+4] = 1).
 *   A     [batch][ldn][ldn] symmetric, leading n[s] x n[s] used; ldn a multiple of 128, at most 4864 (the N x 4 panel lives in LDS)
 *   band  [batch][5][ldn] out: band[t][i] = B[i+t][i];  V [batch][ldn][ldn] out: row r = v_r;  tau [batch][ldn] out */
int imcom_band_reduce(imcom_ctx *ctx, int batch, const int *n_host, int ldn, const double *A, double *band, double *V,
                      double *tau, int memspace);
/* The same kernel on the resident layouts of imcom_build_A / imcom_build_B (DEVICE pointers): A [batch][ldn][ldn], Bt = -B/2
 * input-pixel-major [batch][ldn][ldm] (zero padded), output Tt [batch][ldn][ldm] float32; ldn, ldm multiples of 128.
 * EigenKernel._call_single_kappa / _call_multi_kappa (lakernel.py:154-223) without the transposes of the reference layout. */
int imcom_solve_eigen_resident(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm, const double *A,
                               const double *Bt, const double *C_host, const double *kappaC_host, int nv, double ucmin,
                               double smax, int nbis, float *Tt, float *UC, float *Sigma, float *kappa, int *info_host);

/* ---- secondary LA kernels: lakernel.IterKernel 533-744, lakernel.EmpirKernel 747-805 ----------------------
 * Geometry (all in output-pixel units, as the reference computes them at lakernel.py:617-622 / 757-761):
 *   out_yx  [batch][2][m]  outst.yx_val: y then x of every output pixel (row-major over the n2f x n2f stamp)
 *   in_y,in_x [batch][ldn] outst.iny_val / inx_val
 *   rho_acc = (cfg.instamp_pad / arcsec) / (cfg.dtheta * 3600): acceptance radius
 * n, C, kappaC are HOST arrays as for imcom_solve_chol; the other pointers follow `memspace`.
 *
 * imcom_solve_iter: per output pixel, conjugate gradients (lakernel.py:397-442: x0 = 0, stop at |r| < rtol |b| or
 * after maxiter steps) on the sub-system of the input pixels with hypot(dy, dx) < rho_acc; T is float32 and zero
 * outside the disc.  nv == 1: lines 588-654; nv > 1: 656-744 (node solutions combined by build_reduced_T_wrap).
 * exact_UC selects E = T A T^T (reference default for nv > 1) against the approximation D - kappa N (default for
 * nv == 1).  At most 4096 input pixels per acceptance disc (IMCOM_ERR_ARG beyond). */
int imcom_solve_iter(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A,
                     const double *mBhalf, const double *C, const double *kappaC, int nv, double ucmin,
                     double smax, const double *out_yx, const double *in_y, const double *in_x,
                     double rho_acc, double rtol, int maxiter, int exact_UC, float *T, float *UC,
                     float *Sigma, float *kappa, int memspace);
/* What the last imcom_solve_iter call on this context did at its LAST kappa node (the reference has no counterpart: its
 * conjugate_gradient, lakernel.py:397-442, returns x only; a parity test of recurrences that stop at maxiter needs the step
 * counts, and the bench its roofline).  stats[8] (host): [0] 4 x 4 patches solved by the blocked solver, [1] sum over the
 * patches of (union size rounded up to 16)^2 x steps the patch ran (x 2 x 16 = its flops, x 8 = the bytes of sub-matrix it
 * streamed by the full-storage kernel), [2] sum of steps, [3] sum of (union size)^2, [4] largest union of a patch, [5] 1 = blocked
 * solver, 0 = the per-pixel kernel (a union above 1024), [6] bytes of sub-matrix the patches streamed over their steps, [7] 1 = the
 * half-storage kernel (unions up to 864: only the tiles on and below the diagonal are stored and read).  steps (host, optional): CG steps used per output pixel
 * [batch][m], nsteps = batch * m of that call. */
int imcom_solve_iter_stats(imcom_ctx *ctx, double *stats, int *steps, long nsteps);
/* imcom_solve_empir: T_ai = max(rho_acc - dist_ai, 0) / sum_i max(rho_acc - dist_ai, 0) (a pixel with no input
 * pixel in range gets NaN, as in the reference); kappa = kappaC0 * C, Sigma = sum T^2, UC = 1 + (T A T^T - 2 D)/C.
 * no_qlt_ctrl != 0 (cfg.no_qlt_ctrl, coadd.py:856-858): only T is produced, A / mBhalf / C may be NULL and the
 * maps are zero (lakernel.py:774-777). */
int imcom_solve_empir(imcom_ctx *ctx, int batch, const int *n, int ldn, int m, const double *A,
                      const double *mBhalf, const double *C, double kappaC0, const double *out_yx,
                      const double *in_y, const double *in_x, double rho_acc, int no_qlt_ctrl, float *T,
                      float *UC, float *Sigma, float *kappa, int memspace);
/* Batched symmetric eigendecomposition used by the eigen path and by the Cholesky repair
 * (replaces numpy.linalg.eigh at lakernel.py:162,201,266): lam ascending [batch][ldn],
 * Q[batch][ldn][ldn] with eigenvectors in columns.  A is not modified. */
int imcom_eigh(imcom_ctx *ctx, int batch, const int *n, int ldn, const double *A, double *lam,
               double *Q, int memspace);

/* ------------------------------------------------------------------ stamp / matrix seam ---------
 * Device-resident replacement of PSFOvl._call_ii_self/_call_ii_cross (psfutil.py:1597-1732,
 * 1401-1495) + OutStamp A assembly (coadd.py:1027-1068), PSFOvl._call_io_cross (1497-1595) +
 * B assembly (coadd.py:1075-1082), and OutStamp._perform_coaddition (coadd.py:1294-1363).
 * These take DEVICE pointers only (except arrays marked HOST). */

/* Geometry shared by the table interpolations (PSFGrp.setup / PSFOvl.setup class attributes,
 * psfutil.py:568-613, 1065-1089, carried explicitly instead of process-global state). */
typedef struct {
    int nsamp;           /* PSFOvl.nsamp.  Tables handed to the builders carry the 6-pixel zero border
                            of np.pad(..., 6) (psfutil.py:1471-1473, 1580, 1696): [nsamp+12][nsamp+12] */
    double nc;           /* PSFOvl.nc, table centre */
    double dscale;       /* PSFGrp.dscale: output pixels per table sample */
    double flat_penalty; /* PSFOvl.flat_penalty */
} imcom_table_geom;

/* ---- pixel partition: the binning loop of InImage.partition_pixels coadd.py:329-358 (device pointers only) ----
 * The host visits the relevant sparse-grid cells in order and evaluates the WCS (coadd.py:335-336); what it hands
 * over, in visiting order: out_x, out_y [npix] f64 (position in output-block pixels), in_x, in_y [npix] u16 (index in
 * the input image), mask [npix] u8 (0 = masked; NULL = none), use_instamps [nst][nst] u8 (blk.use_instamps, nst =
 * n1P + 2).  A pixel is kept when pix_lower < x, y < pix_upper, unmasked, and its stamp (j_st, i_st) =
 * floor((pos - pix_lower) / n2) is in use; kept pixels are appended to their stamp in visiting order:
 * y_idx, x_idx u16 and y_val, x_val f64 [nst][nst][npixmax], pix_count u32 [nst][nst].
 * IMCOM_ERR_ARG if a stamp would receive more than npixmax pixels (the reference's arrays overflow there). */
int imcom_partition_pixels(imcom_ctx *ctx, long npix, const double *out_x, const double *out_y,
                           const unsigned short *in_x, const unsigned short *in_y, const unsigned char *mask,
                           const unsigned char *use_instamps, int nst, int n2, double pix_lower,
                           double pix_upper, int npixmax, unsigned short *y_idx, unsigned short *x_idx,
                           double *y_val, double *x_val, unsigned int *pix_count);

/* ---- input-pixel selection: OutStamp._process_input_stamps coadd.py:886-977 + InStamp.make_selection 716-749 ----
 * The block's InStamps lie back to back in a pool:
 *   pool_x, pool_y [npool] f64 (InStamp.x_val / y_val), pool_data [n_inframe][npool] f32 (InStamp.data),
 *   pool_expo [npool] i32 (exposure index of each pixel, from InStamp.pix_cumsum),
 *   inst_off [n_inst+1]: InStamp i owns pool[inst_off[i] : inst_off[i+1]].
 * Per output stamp s and neighbour idx = 0..8 (row-major over dj, di = -1..1, coadd.py:878):
 *   inst_id [batch][9] InStamp index or -1; pivot_x / pivot_y [batch][9] (NaN = None, lines 918-919);
 *   radius = rpix_search (line 912; NaN = select everything).
 * A pixel is kept when (x - px)^2 + (y - py)^2 < radius^2 (terms of missing pivot coordinates dropped), in pool
 * order.  Outputs x, y [batch][ldn], indata [batch][n_inframe][ldn], expo [batch][ldn] (zero padded) and
 * cumsum [batch][10] = inpix_cumsum; n[s] = cumsum[s][9].  IMCOM_ERR_ARG if a stamp selects more than ldn. */
int imcom_select_pixels(imcom_ctx *ctx, int batch, const double *pool_x, const double *pool_y,
                        const float *pool_data, long npool, int n_inframe, const int *pool_expo,
                        const long *inst_off, int n_inst, const int *inst_id, const double *pivot_x,
                        const double *pivot_y, double radius, int ldn, double *x, double *y, float *indata,
                        int *expo, int *cumsum, int memspace);

/* A[s][i][j] for i,j < n[s] (exactly symmetric: the element with i before j is interpolated and
 * mirrored, as the reference's sub-block assembly does, coadd.py:1038-1068), with the
 * padding rows/cols n[s] <= i < ldn set to the identity so the factorisation kernels can run on
 * whole tiles.
 *   x,y          [batch][ldn] input pixel positions in output-pixel units (coadd.py:969-972)
 *   psf          [batch][ldn] int32: stamp-local PSF index of each pixel, < npsf_max
 *   tables       [ntab][nsamp+12][nsamp+12] zero-bordered overlap tables (PSFOvl.ovl_arr entries)
 *   pair_tab     [batch][npsf_max][npsf_max] int32 code of the ordered PSF pair (p_i,p_j), i before j:
 *                bits 0..27 table index; bit 30 (IMCOM_PAIR_FLIP) = table flipped in both axes (the
 *                np.flip of psfutil.py:1658-1665); bit 29 (IMCOM_PAIR_SWAP) = the reference evaluated
 *                this block from the other stamp's side and transposed it (psfutil.py:1990-1996), i.e.
 *                interpolate at r_j - r_i; negative = no table (value 0 + penalty)
 *   pair_pen     [batch][npsf_max][npsf_max] constant added to every element of the pair's block:
 *                -flat_penalty/n_in (+flat_penalty for the same exposure), psfutil.py:1482-1486,
 *                1705-1708
 */
int imcom_build_A(imcom_ctx *ctx, int batch, const int *n_host, int ldn, const double *x,
                  const double *y, const int *psf, const double *tables, int ntab,
                  const imcom_table_geom *geom, const int *pair_tab, const double *pair_pen,
                  int npsf_max, double *A);
/* Bt[s][i][a] = -B/2 transposed (input-pixel-major, the native layout of gridD5512C's output,
 * routine.py:273) for i < n[s], a < m = n2f*n2f; rows i >= n[s] and columns a >= m up to ldm are
 * zero.  Output pixel a = iy*n2f + ix sits at (out_x0[s] + ix, out_y0[s] + iy) (coadd.py:879-882).
 *   io_tab       [batch][npsf_max] int32: input-output overlap table of each stamp-local PSF
 */
int imcom_build_B(imcom_ctx *ctx, int batch, const int *n_host, int ldn, const double *x,
                  const double *y, const int *psf, const double *tables, int ntab,
                  const imcom_table_geom *geom, const int *io_tab, int npsf_max,
                  const double *out_x0, const double *out_y0, int n2f, int ldm, double *Bt);
/* Resident single/multi-kappa Cholesky solve on the layouts produced by imcom_build_A/_B:
 * A[batch][ldn][ldn] (padding = identity), Bt[batch][ldn][ldm].  Outputs: Tt[batch][ldn][ldm]
 * float32 (input-pixel-major), UC/Sigma/kappa [batch][m] float32.  Needs ldn, ldm multiples of 128. */
int imcom_solve_chol_resident(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm,
                              const double *A, const double *Bt, const double *C_host,
                              const double *kappaC_host, int nv, double ucmin, double smax,
                              float *Tt, float *UC, float *Sigma, float *kappa, int *info_host);
/* imcom_solve_chol_resident in two halves, for a caller with host work to do while the device factors and solves
 * (the reference's loop is synchronous, lakernel.py:84-138; a block driver prepares its next pass in between).
 * _begin queues the whole first attempt and returns.  _end waits for it: every A + kappa I positive definite (the
 * normal case) -> info = 0, IMCOM_OK, outputs final; otherwise _end returns 1 with info[s] != 0 for the stamps whose
 * factorisation failed (the other stamps' outputs are final) and the caller runs imcom_solve_chol_resident_redo on those
 * (one kappa node) or imcom_solve_chol_resident on the same arguments (the eigh-shift repair of lakernel.py:262-279).  Work queued on the
 * context between the two calls runs behind the solve; a begin whose _end never came is waited for and forgotten by the next begin. */
int imcom_solve_chol_resident_begin(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm,
                                    const double *A, const double *Bt, const double *C_host,
                                    const double *kappaC_host, int nv, double ucmin, double smax,
                                    float *Tt, float *UC, float *Sigma, float *kappa);
int imcom_solve_chol_resident_end(imcom_ctx *ctx, int batch, int *info_host);
/* CholKernel._call_single_kappa (lakernel.py:281-323) for SOME stamps of a resident batch: redo_host[s] = 0 leaves stamp s and
 * its outputs untouched, 1 solves it, 2 solves it knowing that the factorisation of A + kappa I fails (what _end reported), i.e.
 * straight to _cholesky_wrapper's repair (lakernel.py:262-279: AA_ii += |w[0]| + 1e-16 with w[0] the smallest eigenvalue of A),
 * 3 the same on the caller's EXPECTATION (the stamps before it failed; a stamp whose A + kappa I is positive definite after all is
 * recognised and solved without the repair).  nv must be 1.  info as imcom_solve_chol_resident, written for the stamps that were solved. */
int imcom_solve_chol_resident_redo(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm,
                                   const double *A, const double *Bt, const double *C_host,
                                   const double *kappaC_host, int nv, double ucmin, double smax,
                                   float *Tt, float *UC, float *Sigma, float *kappa, const int *redo_host, int *info_host);
/* _cholesky_wrapper's repair (lakernel.py:262-279) needs w[0], the smallest eigenvalue of a failed stamp's A; the library finds it by
 * inverse subspace iteration, which starts from a shift sigma with A + sigma I positive definite.  A driver that
 * has just repaired neighbouring stamps knows where w[0] lies: imcom_ctx_set_repair_hint(ctx, h) with h ~ max |w[0]| of those stamps
 * makes the following Cholesky calls on this context start at h (1 + 5 %) -- one factorisation inside the iteration instead of two.  The
 * hint changes the iteration's path, not what it converges to (w[0] to 1e-11 either way; a hint that is too small for a stamp costs that
 * stamp one failed factorisation).  0 clears it.  imcom_ctx_last_repair: how many stamps the last Cholesky call repaired and the range
 * of their w[0] (count = 0: none, the range is then 0). */
int imcom_ctx_set_repair_hint(imcom_ctx *ctx, double lmin_abs);
/* The same knowledge for the host-array entries (imcom_solve_chol, imcom_solve_chol_stamps; one kappa node): expect != 0 makes the
 * following calls on this context skip the factorisation of A + kappa I that a driver has just seen fail on the neighbouring stamps and
 * go straight to the repair (lakernel.py:262-279) -- the resident path's redo code 2.  A stamp whose A + kappa I is positive definite
 * after all is recognised by the smallest-eigenvalue iteration and solved without the repair (same outputs, one wasted iteration).
 * 0 clears it. */
int imcom_ctx_set_repair_expect(imcom_ctx *ctx, int expect);
int imcom_ctx_last_repair(imcom_ctx *ctx, int *count, double *w0_min, double *w0_max);
/* coadd.py:1320-1354: fade taper of T (trapezoid, 1222-1292), per-exposure weight sums, Neff and
 * outimage = T . indata.
 *   Tt           [batch][ldn][ldm] float32 (tapered in place when fade > 0)
 *   indata       [batch][n_inframe][ldn] float32 input pixel values (coadd.py:975)
 *   expo         [batch][ldn] int32 exposure (input image) index of each pixel, < n_expo
 *   outimage     [batch][n_inframe][m] float32
 *   Tsum_stamp   [batch][n_expo] float64; Tsum_inpix, Neff [batch][m] float64 (numpy's result dtype)
 */
int imcom_coadd_epilogue(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm,
                         int n2f, int fade, int n2, float *Tt, const float *indata, int n_inframe,
                         const int *expo, int n_expo, float *outimage, double *Tsum_stamp,
                         double *Tsum_inpix, double *Neff);
/* imcom_solve_chol_resident and imcom_coadd_epilogue in ONE call, for fade == 0 (with fade > 0 the map tapers of
 * coadd.py:1118-1122 come in between: use the three calls): CholKernel (lakernel.py:281-394) on the device layouts, then
 * OutStamp._perform_coaddition (coadd.py:1294-1363) for the same stamps.  With one kappa node the coaddition's sums -- per
 * exposure sum_j T[a][j] and per input frame sum_j T[a][j] indata[j] -- are taken from the tiles of T while the backward
 * launches of the solve still hold them, so T (20 MB per stamp) is not read again; with several nodes the stand-alone
 * epilogue runs inside the call.  Arguments as in the two entries; all pointers DEVICE except n_host / C / kappaC / info. */
int imcom_solve_chol_resident_coadd(imcom_ctx *ctx, int batch, const int *n_host, int ldn, int m, int ldm,
                                    const double *A, const double *Bt, const double *C, const double *kappaC, int nv,
                                    double ucmin, double smax, float *Tt, float *UC, float *Sigma, float *kappa,
                                    int *info, int n2f, int fade, int n2, const float *indata, int n_inframe,
                                    const int *expo, int n_expo, float *outimage, double *Tsum_stamp,
                                    double *Tsum_inpix, double *Neff);
/* OutStamp.trapezoid (coadd.py:1222-1292) on [batch][n2f][n2f] float32 maps (kappa, Sigma, UC). */
int imcom_trapezoid_f32(imcom_ctx *ctx, float *maps, long nmaps, int n2f, int fade);
/* OutStamp._build_system_matrices, coadd.py:1104-1107: after the "Iterative" kernel (whose U/C and Sigma can come
 * out negative) the reference sets UC = np.maximum(UC, 1e-32), Sigma = np.maximum(Sigma, 1e-32), before the map
 * taper.  maps[i] = maps[i] < lo ? lo : maps[i] on `count` device float32 values; NaN stays NaN (np.maximum). */
int imcom_clamp_min_f32(imcom_ctx *ctx, float *maps, long count, float lo);

/* Block._output_stamp_wrapper map updates (coadd.py:1975-1993), on the device: for every stamp s of the batch,
 * dst[layer][(jst-1)*n2 + r][(ist-1)*n2 + c] += src[s][layer][r][c], r,c < n2f = n2 + 2*fade.  dst is one of the
 * block's float32 maps [nlayer][nside_pf][nside_pf] (out_map with nlayer = n_inframe; UC/Sigma/kappa/Tsum/Neff
 * with nlayer = 1); src is float32 or float64 (src_is_f64).  jst/ist are HOST arrays of 1-based OutStamp
 * indices.  Overlapping neighbours are added in four index-parity passes, so the result is deterministic. */
int imcom_block_accumulate(imcom_ctx *ctx, int batch, const int *jst_host, const int *ist_host, int n2, int fade,
                           int nlayer, const void *src, int src_is_f64, float *dst, int nside_pf);
/* The same map updates for OVERLAPPING stamps (fade > 0) without a dependence on the visiting order.  A block pixel belongs
 * to at most four stamps, one of each index parity ((jst & 1) << 1 | (ist & 1)).  imcom_block_place STORES every stamp's
 * tile, in the dtype it arrives in, into the layer of its parity: layers [4][nlayer][nside_pf][nside_pf] float32 or
 * float64 (src_is_f64), zero before the first call.  imcom_block_combine then forms dst[layer][row][col] by adding the
 * layers of a pixel in the order in which the reference's loop meets their stamps -- order = 1: coadd.py:2056-2059, cells of
 * 2 x 2 stamps from (j_st_min, i_st_min) on (coadd.py:1808-1838), row by row of cells, inside a cell dj outer, di inner;
 * order = 0: plain rows, j_st outer, i_st inner -- every addition rounded as numpy's `f32_map[window] += tile` (float32 +
 * float32 -> float32; float32 + float64 in double, rounded once): the result carries the reference's own rounding whatever
 * batches, passes or processes the stamps were dealt to (pyimcom_amd.farm shares a block's passes between GPUs).
 * dst: [nlayer][nside_pf][nside_pf] float32, nside_pf = n1P * n2 + 2 * fade. */
int imcom_block_place(imcom_ctx *ctx, int batch, const int *jst_host, const int *ist_host, int n2, int fade, int nlayer,
                      const void *src, int src_is_f64, void *layers, int nside_pf);
int imcom_block_combine(imcom_ctx *ctx, int n1P, int n2, int fade, long nlayer, const void *layers, int src_is_f64,
                        float *dst, int nside_pf, int order, int j_st_min, int i_st_min);
/* Block.build_output_file boundary recovery (coadd.py:2163-2181): OutStamp.trapezoid(maps, fade,
 * recover_mode=True, pad_widths=(b,t,l,r)) on float32 maps [nmaps][ny][nx]. */
int imcom_trapezoid_recover_f32(imcom_ctx *ctx, float *maps, long nmaps, int ny, int nx, int fade, int pad_b,
                                int pad_t, int pad_l, int pad_r);
/* Block.compress_map coadd.py:2087-2138 (device pointers): out[i] = clip(floor(coef * log10(clip(map[i], 1e-32, inf))
 * + 0.5), a_min, a_max) in float32 arithmetic, as int16 (is_unsigned = 0) or uint16 (1).  The reference's
 * coefficients (coadd.py:2249-2303): U/C -5000 uint16, Sigma -10000 int16, kappa -5000 uint16, Tsum 200000 int16,
 * Neff 50000 uint16. */
int imcom_compress_map_f32(imcom_ctx *ctx, const float *map, long count, int coef, int is_unsigned, void *out);

/* ---- PSF images -> sample grid, target PSFs --------------------------------------------------------------
 * imcom_sample_psf: PSFGrp._sample_psf psfutil.py:709-795 followed by the circular cut-out / normalisation of
 * PSFGrp.__init__ 650-656.  psf [n_psf][ny][nx]; yxco [n_psf][2][nsamp*nsamp] = y then x offsets of the sampling
 * positions from the image centre in oversampled native pixels (the host computes them through the WCS,
 * psfutil.py:751-771), or NULL for the unrotated grid PSFGrp.yxo (the target-PSF path, evaluated with gridD5512C
 * as the reference does at 786-793).  psf_arr [n_psf][nsamp][nsamp] out; samples off the image stay zero. */
int imcom_sample_psf(imcom_ctx *ctx, int n_psf, const double *psf, int ny, int nx, const double *yxco,
                     int nsamp, int psf_circ, int psf_norm, double *psf_arr, int memspace);
/* The sampling positions yxco of imcom_sample_psf from a coarse lattice: the reference evaluates outpix2world2inpix at all
 * nsamp^2 positions of a PSF group and exposure (psfutil.py:751-771: 146 689 WCS evaluations, the host half of the Block seam);
 * over the ~5" they span that map is smooth, so a caller may evaluate it on an L x L lattice of Chebyshev-Lobatto nodes
 * u_a = u_mid + u_half cos(pi a / (L - 1)) of the sample coordinate and hand over
 *   lattice [count][2][L][L]  the offsets (y plane then x plane, as yxco) at lattice point (node a along y, node b along x)
 *   W       [nsamp][L] (HOST) W[i][a] = Lagrange basis polynomial a of the nodes at sample coordinate i
 * yxco[c][k][iy][ix] = sum_a sum_b W[iy][a] W[ix][b] lattice[c][k][a][b] -- exact for maps of degree < L per axis (an affine WCS,
 * SIP distortion up to order L - 1), 2 <= L <= 33.  lattice / yxco follow `memspace`. */
int imcom_lattice_positions(imcom_ctx *ctx, int count, int L, const double *W, const double *lattice, int nsamp,
                            double *yxco, int memspace);
/* OutPSF.psf_gaussian psfutil.py:117-146 and OutPSF.psf_simple_airy 148-223 (n x n, row-major) */
int imcom_psf_gaussian(imcom_ctx *ctx, int n, double sigmax, double sigmay, double *out, int memspace);
int imcom_psf_simple_airy(imcom_ctx *ctx, int n, double ldp, double obsc, double tophat_conv, double sigma,
                          double *out, int memspace);

/* InImage.smooth_and_pad (reference src/pyimcom/coadd.py:433-474), the smearing of a raw PSF image that
 * InImage.get_psf_pos applies before the image reaches PSFGrp (coadd.py:603-640): zero-pad by
 * npad = imcom_smooth_pad_width() = ceil(tophatwidth + 6 gaussiansigma + 1) rounded up to a multiple of 4 on every
 * side, then convolve (circularly on the padded grid, as the reference's FFT product does) with a top-hat of width
 * tophatwidth and a Gaussian of sigma gaussiansigma, both in pixels of the array.
 *   in  [n][ny][nx]                       out [n][ny + 2 npad][nx + 2 npad]      (host or device, memspace) */
int imcom_smooth_pad_width(double tophatwidth, double gaussiansigma);
int imcom_smooth_and_pad(imcom_ctx *ctx, int n, const double *in, int ny, int nx, double tophatwidth, double gaussiansigma,
                         double *out, int memspace);

/* PSFGrp.accel_pad_and_rfft2 + PSFOvl._build_psfovl (psfutil.py:943-986, 1244-1294): correlation
 * tables out[p][q] = irfft2(rft(psf1[p]) * conj(rft(psf2[q]))), rolled by nc and cropped to
 * nsamp x nsamp.  psf1[n1][nsamp][nsamp], psf2[n2][nsamp][nsamp]; pairs[npairs][2] HOST lists the
 * (p,q) wanted, tables[npairs][nsamp+12][nsamp+12] (zero border included).
 * amp_penalty HOST: NULL, or {cfg.amp_penalty[0], cfg.amp_penalty[1] * oversamp}: both groups' spectra are
 * reweighted by 1 + a0 exp(-2 pi^2 |u|^2 a1^2) as PSFGrp.__init__ does (psfutil.py:661-671). */
int imcom_psf_overlap(imcom_ctx *ctx, const double *psf1, int n1, const double *psf2, int n2,
                      int nsamp, int nfft, const int *pairs_host, int npairs, const double *amp_penalty,
                      double *tables);

/* The two halves of imcom_psf_overlap, for callers that keep the forward spectra of a PSF group resident and cross
 * them with several other groups (SysMatA builds PSFOvl(grp1, grp2) for every pair of neighbouring 2x2 groups from
 * the same PSFGrp.psf_rft, psfutil.py:1904-2010, 943-986):
 *   imcom_psf_spectra_size   doubles per PSF of a spectra buffer, or 0 when nfft has no butterfly plan (other prime
 *                            factors than 2, 3, 5, or > 1024) -- then only imcom_psf_overlap (dense-DFT form) serves
 *   imcom_psf_spectra        spectra[n][size] (DEVICE) = rfft2 of the zero-padded PSFs psf[n][nsamp][nsamp] (DEVICE)
 *   imcom_psf_overlap_spectra  tables[npairs][nsamp+12][nsamp+12] (DEVICE) exactly as imcom_psf_overlap, from spectra;
 *                            pairs (HOST) index spec1 / spec2
 *   imcom_psf_overlap_spectra_win  the same with a window per pair, win (HOST) [npairs][4] = {row_lo, row_hi, col_lo, col_hi} in
 *                            window coordinates 0..nsamp (table row = 6 + window row), or NULL: only that part of a table
 *                            (and the zero border next to it) is guaranteed to be written.  For the cross tables of two
 *                            PSF groups that own disjoint ranges of InStamp cells (SysMatA.ji_st2psf, psfutil.py:1803-1824):
 *                            the separations between their pixels have one sign along every axis in which the groups
 *                            differ, so half (a quarter) of such a table is never interpolated (psfutil.py:1401-1495). */
long imcom_psf_spectra_size(int nsamp, int nfft);
int imcom_psf_spectra(imcom_ctx *ctx, const double *psf, int n, int nsamp, int nfft, double *spectra);
int imcom_psf_overlap_spectra(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp, int nfft,
                              const int *pairs, int npairs, const double *amp_penalty, double *tables);
int imcom_psf_overlap_spectra_win(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp, int nfft,
                                  const int *pairs, int npairs, const double *amp_penalty, const int *win, double *tables);
/*   imcom_psf_overlap_spectra_slots  the same into an ARENA of tables: slots (HOST) [npairs] = index of the arena table that
 *                            receives pair t's result, `tables` = the arena's first table, nslots its size (range check).
 *                            A block keeps the PSFOvl sets of the PSF groups it is working on resident and replaces the least
 *                            recently used ones (the reference's reference-counted SysMatA cache, psfutil.py:1868-1902,
 *                            2012-2092): freed tables are reused one by one, so a set need not be contiguous. */
int imcom_psf_overlap_spectra_slots(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp, int nfft,
                                    const int *pairs, int npairs, const double *amp_penalty, const int *win, const int *slots,
                                    int nslots, double *tables);

#ifdef __cplusplus
}
#endif
#endif /* IMCOM_HIP_H */
