/*
 * imcom_oracle.c -- CPU restatement of the reference's native numerics for the
 * IMCOM postage-stamp path.  TEST INFRASTRUCTURE ONLY: this file is the checker
 * for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg); nothing under pyimcom_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  Every function below is checked against golden
 * vectors produced by running the reference's own Python
 * (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle.py).
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference checkout, src/pyimcom/routine.py unless stated otherwise).  The
 * third-party C library furry_parakeet (un-pinned in requirements.txt:15, source
 * not in the reference tree) implements the same routines; the reference's
 * tests/pyimcom/test_routine.py pins routine.py == furry_parakeet to 1e-9.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so
 * the arithmetic matches the reference's interpreted float64 operations).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* IN-1  routine.py:29-122  iD5512C_getw                                      */
/* Five (even, odd) degree-4 polynomials in fh^2; taps k and 9-k are e+o, e-o */
static const double D5512_EVEN[5][5] = {
    /* fh2^4 ... fh2^0 */
    {+1.651881673372979740e-05, -3.145538007199505447e-04, +1.793518183780194427e-03,
     -2.904014557029917318e-03, +6.187591260980151433e-04},
    {-1.146756217210629335e-04, +2.883845374976550142e-03, -1.857047531896089884e-02,
     +3.147734488597204311e-02, -6.753293626461192439e-03},
    {+3.256838096371517067e-04, -9.702063770653997568e-03, +8.678848026470635524e-02,
     -1.659182651092198924e-01, +3.620560878249733799e-02},
    {-4.541830837949564726e-04, +1.494862093737218955e-02, -1.668775957435094937e-01,
     +5.879306056792649171e-01, -1.367845996704077915e-01},
    {+2.266560930061513573e-04, -7.815848920941316502e-03, +9.686607348538181506e-02,
     -4.505856722239036105e-01, +6.067135256905490381e-01},
};
static const double D5512_ODD[5][5] = {
    {-3.486978652054735998e-06, +6.753750285320532433e-05, -3.871378836550175566e-04,
     +6.279918076641771273e-04, -1.338434614116611838e-04},
    {+3.121412120355294799e-05, -8.040343683015897672e-04, +5.209574765466357636e-03,
     -8.847326408846412429e-03, +1.898674086370833597e-03},
    {-1.243658986204533102e-04, +3.804930695189636097e-03, -3.434861846914529643e-02,
     +6.581033749134083954e-02, -1.436476114189205733e-02},
    {+2.894406669584551734e-04, -9.794291009695265532e-03, +1.104231510875857830e-01,
     -3.906954914039130755e-01, +9.092432925988773451e-02},
    {-4.336085507644610966e-04, +1.537862263741893339e-02, -1.925091434770601628e-01,
     +8.993141455798455697e-01, -1.213035309579723942e+00},
};

void orc_d5512_getw(double *w, double fh)
{
    const double fh2 = fh * fh;
    for (int k = 0; k < 5; k++) {
        const double *ce = D5512_EVEN[k], *co = D5512_ODD[k];
        double e = (((ce[0] * fh2 + ce[1]) * fh2 + ce[2]) * fh2 + ce[3]) * fh2 + ce[4];
        double o = ((((co[0] * fh2 + co[1]) * fh2 + co[2]) * fh2 + co[3]) * fh2 + co[4]) * fh;
        w[k] = e + o;
        w[9 - k] = e - o;
    }
}

/* one 10x10 stencil evaluation: inner sum over x (j), outer over y (i) -- 176-180 */
static double stencil(const double *f, long ngx, int yi, int xi, const double *wx, const double *wy)
{
    double out = 0.0;
    for (int i = 0; i < 10; i++) {
        const double *row = f + (long)(yi - 4 + i) * ngx + (xi - 4);
        double strip = 0.0;
        for (int j = 0; j < 10; j++) strip += wx[j] * row[j];
        out += strip * wy[i];
    }
    return out;
}

/* IN-2  routine.py:125-181  iD5512C: scattered points, off-grid outputs untouched (166-167) */
void orc_interp_d5512(const double *infunc, int nlayer, int ngy, int ngx, const double *xpos,
                      const double *ypos, long nout, double *fhatout)
{
    /* points are independent: threads split them (bench.py's cpu_baseline uses all host cores; orc_set_threads(1)
     * gives the scalar figure).  No value depends on the thread count. */
#pragma omp parallel for schedule(static) if (nout > 4096)
    for (long p = 0; p < nout; p++) {
        double wx[10], wy[10];
        double x = xpos[p], y = ypos[p];
        int xi = (int)x, yi = (int)y; /* np.int32(x): truncation toward zero */
        if (xi < 4 || xi >= ngx - 5 || yi < 4 || yi >= ngy - 5) continue;
        orc_d5512_getw(wx, x - xi - 0.5);
        orc_d5512_getw(wy, y - yi - 0.5);
        for (int l = 0; l < nlayer; l++)
            fhatout[(long)l * nout + p] = stencil(infunc + (long)l * ngy * ngx, ngx, yi, xi, wx, wy);
    }
}

/* IN-3  routine.py:184-253  iD5512C_sym: upper triangle of a sq x sq output, then mirrored */
void orc_interp_d5512_sym(const double *infunc, int nlayer, int ngy, int ngx, const double *xpos,
                          const double *ypos, long nout, double *fhatout)
{
    double wx[10], wy[10];
    long sq = (long)sqrt((double)(nout + 1));
    for (long a = 0; a < sq; a++)
        for (long b = a; b < sq; b++) {
            long p = a * sq + b;
            double x = xpos[p], y = ypos[p];
            int xi = (int)x, yi = (int)y;
            if (xi < 4 || xi >= ngx - 5 || yi < 4 || yi >= ngy - 5) continue;
            orc_d5512_getw(wx, x - xi - 0.5);
            orc_d5512_getw(wy, y - yi - 0.5);
            for (int l = 0; l < nlayer; l++)
                fhatout[(long)l * nout + p] = stencil(infunc + (long)l * ngy * ngx, ngx, yi, xi, wx, wy);
        }
    for (long a = 1; a < sq; a++)
        for (long b = 0; b < a; b++)
            for (int l = 0; l < nlayer; l++)
                fhatout[(long)l * nout + a * sq + b] = fhatout[(long)l * nout + b * sq + a];
}

/* IN-4  routine.py:256-338  gridD5512C: separable grid; off-grid -> zero weights at tap 4 (306-323) */
void orc_grid_d5512(const double *infunc, int ngy, int ngx, const double *xpos, const double *ypos,
                    long npi, int nxo, int nyo, double *fhatout)
{
#pragma omp parallel if (npi > 64)
    {
        double *wx = (double *)malloc(sizeof(double) * 10 * (size_t)nxo);
        double *wy = (double *)malloc(sizeof(double) * 10 * (size_t)nyo);
        int *xi = (int *)malloc(sizeof(int) * (size_t)nxo);
        int *yi = (int *)malloc(sizeof(int) * (size_t)nyo);
#pragma omp for schedule(static)
        for (long p = 0; p < npi; p++) {
            for (int ix = 0; ix < nxo; ix++) {
                double x = xpos[p * nxo + ix];
                xi[ix] = (int)x;
                if (xi[ix] < 4 || xi[ix] >= ngx - 5) {
                    xi[ix] = 4;
                    memset(wx + 10 * ix, 0, 10 * sizeof(double));
                } else
                    orc_d5512_getw(wx + 10 * ix, x - xi[ix] - 0.5);
            }
            for (int iy = 0; iy < nyo; iy++) {
                double y = ypos[p * nyo + iy];
                yi[iy] = (int)y;
                if (yi[iy] < 4 || yi[iy] >= ngy - 5) {
                    yi[iy] = 4;
                    memset(wy + 10 * iy, 0, 10 * sizeof(double));
                } else
                    orc_d5512_getw(wy + 10 * iy, y - yi[iy] - 0.5);
            }
            double *o = fhatout + p * (long)nyo * nxo;
            for (int iy = 0; iy < nyo; iy++)
                for (int ix = 0; ix < nxo; ix++)
                    *o++ = stencil(infunc, ngx, yi[iy], xi[ix], wx + 10 * ix, wy + 10 * iy);
        }
        free(wx); free(wy); free(xi); free(yi);
    }
}

/* thread count of the two interpolators above (0 = OpenMP default = all cores) */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}

/* EI-2  routine.py:341-430  lakernel1: per output pixel geometric bisection on kappa */
void orc_lakernel1(const double *lam, const double *mPhalf, long m, long n, double C,
                   double targetleak, double kCmin, double kCmax, int nbis, double *kappa,
                   double *Sigma, double *UC, double *T, double smax)
{
    for (long a = 0; a < m; a++) {
        const double *p = mPhalf + a * n;
        double factor = sqrt(kCmax / kCmin);
        double kap = sqrt(kCmax * kCmin);
        for (int it = 0; it < nbis; it++) {
            double s1 = 0.0, s2 = 0.0;
            for (long i = 0; i < n; i++) {
                double v = p[i] / (lam[i] + kap);
                s2 += v * v;
                s1 += (lam[i] + 2.0 * kap) * v * v;
            }
            double udc = 1.0 - s1 / C;
            factor = sqrt(factor);
            kap *= (udc > targetleak && s2 < smax) ? 1.0 / factor : factor;
        }
        double s1 = 0.0, s2 = 0.0;
        for (long i = 0; i < n; i++) {
            double v = p[i] / (lam[i] + kap);
            T[a * n + i] = v;
            s2 += v * v;
            s1 += (lam[i] + 2.0 * kap) * v * v;
        }
        Sigma[a] = s2;
        kappa[a] = kap;
        UC[a] = 1.0 - s1 / C;
    }
}

/* routine.py:433-484  lsolve_sps: in-place lower Cholesky of A, then two substitutions */
void orc_lsolve_sps(int N, double *A, double *x, const double *b)
{
    double p1[64];
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < i; j++) {
            double s = 0.0;
            for (int k = 0; k < j; k++) s += A[i * N + k] * A[j * N + k];
            A[i * N + j] = (A[i * N + j] - s) / A[j * N + j];
        }
        double s = 0.0;
        for (int k = 0; k < i; k++) s += A[i * N + k] * A[i * N + k];
        A[i * N + i] = sqrt(A[i * N + i] - s);
    }
    double *pp = N <= 64 ? p1 : (double *)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; i++) {
        double s = 0.0;
        for (int j = 0; j < i; j++) s += A[i * N + j] * pp[j];
        pp[i] = (b[i] - s) / A[i * N + i];
    }
    for (int i = N - 1; i >= 0; i--) {
        double s = 0.0;
        for (int j = i + 1; j < N; j++) s += A[j * N + i] * x[j];
        x[i] = (pp[i] - s) / A[i * N + i];
    }
    if (pp != p1) free(pp);
}

/* CH-3 inner  routine.py:487-588  build_reduced_T_wrap */
void orc_build_reduced_T(const double *Nflat, const double *Dflat, const double *Eflat,
                         const double *kappa, int nv, long m, double ucmin, double smax,
                         double *out_kappa, double *out_Sigma, double *out_UC, double *out_w)
{
    const int nv2 = nv * nv;
    double *M = (double *)malloc(sizeof(double) * (size_t)nv2);
    double *w = (double *)malloc(sizeof(double) * (size_t)nv);
    for (long a = 0; a < m; a++) {
        const double *Na = Nflat + a * nv2, *Ea = Eflat + a * nv2, *Da = Dflat + a * nv;
        /* node interval: scan down from the top node (546-554) */
        int iv = nv - 1;
        double UC = ucmin * 10, S = smax / 10;
        while (iv > 0 && ucmin < UC && smax > S) {
            iv--;
            S = Na[iv * (nv + 1)];
            UC = 1.0 - 2.0 * Da[iv] + Ea[iv * (nv + 1)];
        }
        double kmid = sqrt(kappa[iv] * kappa[iv + 1]);
        double factor = pow(kappa[iv + 1] / kappa[iv], 0.25);
        for (int it = 0; it < 12; it++) {
            for (int r = 0; r < nv; r++)
                for (int c = 0; c <= r; c++) M[r * nv + c] = Ea[r + nv * c] + kmid * Na[r + nv * c];
            orc_lsolve_sps(nv, M, w, Da);
            for (int r = 0; r < nv; r++) out_w[a * nv + r] = w[r];
            S = 0.0;
            for (int r = 0; r < nv; r++) {
                double s = 0.0;
                for (int c = 0; c < nv; c++) s += Na[r + nv * c] * w[c];
                S += s * w[r];
            }
            UC = 1.0 - kmid * S;
            for (int r = 0; r < nv; r++) UC -= Da[r] * w[r];
            kmid *= (ucmin < UC && smax > S) ? 1.0 / factor : factor;
            factor = sqrt(factor);
        }
        out_kappa[a] = kmid;
        out_Sigma[a] = S;
        out_UC[a] = UC;
    }
    free(M); free(w);
}
