// fft_radix.h -- the in-register butterflies of the line FFTs (psf_overlap.hip): DFTs of 2, 3, 4, 5, 8 and 16 complex
// values, forward (exp(-2 pi i nk/R)) and inverse (exp(+...), unnormalised), in place, natural order in and out.
// Plain arithmetic on a two-double struct, so that the same text compiles for the host: tests/test_fft_radix.py checks
// every butterfly against the direct sum with g++.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IMCOM_FFT_HD __host__ __device__ __forceinline__
namespace imcom {
typedef double2 cplx;
#else
#define IMCOM_FFT_HD inline
namespace imcom {
struct cplx { double x, y; };
static inline cplx make_double2(double x, double y) { return cplx{x, y}; }
#endif

IMCOM_FFT_HD cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
IMCOM_FFT_HD cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
IMCOM_FFT_HD cplx cmulf(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * (-i) forward, a * (+i) inverse: the quarter-turn twiddle
template <bool INV> IMCOM_FFT_HD cplx cturn(cplx a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }
// a * exp(-+ i pi/4) and a * exp(-+ 3 i pi/4)
template <bool INV> IMCOM_FFT_HD cplx cturn8(cplx a)
{
    const double h = 0.7071067811865476;
    return INV ? make_double2(h * (a.x - a.y), h * (a.x + a.y)) : make_double2(h * (a.x + a.y), h * (a.y - a.x));
}
template <bool INV> IMCOM_FFT_HD cplx cturn38(cplx a)
{
    const double h = 0.7071067811865476;
    return INV ? make_double2(-h * (a.x + a.y), h * (a.x - a.y)) : make_double2(h * (a.y - a.x), -h * (a.x + a.y));
}

template <int R, bool INV> struct SmallDft;

template <bool INV> struct SmallDft<2, INV> {
    static IMCOM_FFT_HD void run(cplx (&v)[2]) { const cplx a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b); }
};

template <bool INV> struct SmallDft<4, INV> {
    static IMCOM_FFT_HD void run(cplx (&v)[4])
    {
        const cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cturn<INV>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = cadd(t1, t3);
        v[3] = csub(t1, t3);
    }
};

template <bool INV> struct SmallDft<3, INV> {
    static IMCOM_FFT_HD void run(cplx (&v)[3])
    {
        const double s3 = INV ? 0.8660254037844386 : -0.8660254037844386;  // sin(-+ 2 pi / 3)
        const cplx t = cadd(v[1], v[2]), u = csub(v[1], v[2]);
        const cplx m = make_double2(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y), iu = make_double2(-s3 * u.y, s3 * u.x);  // i s3 u
        v[0] = cadd(v[0], t);
        v[1] = cadd(m, iu);
        v[2] = csub(m, iu);
    }
};

template <bool INV> struct SmallDft<5, INV> {
    static IMCOM_FFT_HD void run(cplx (&v)[5])
    {
        // y_u = sum_t v_t w^(u t), w = exp(-+ 2 pi i / 5)
        const double c1 = 0.30901699437494745, c2 = -0.8090169943749475;
        const double s1 = INV ? 0.9510565162951535 : -0.9510565162951535, s2 = INV ? 0.5877852522924731 : -0.5877852522924731;
        const cplx a = cadd(v[1], v[4]), b = csub(v[1], v[4]), c = cadd(v[2], v[3]), d = csub(v[2], v[3]);
        const cplx m1 = make_double2(v[0].x + c1 * a.x + c2 * c.x, v[0].y + c1 * a.y + c2 * c.y);
        const cplx m2 = make_double2(v[0].x + c2 * a.x + c1 * c.x, v[0].y + c2 * a.y + c1 * c.y);
        const cplx n1 = make_double2(-(s1 * b.y + s2 * d.y), s1 * b.x + s2 * d.x);  // i (s1 b + s2 d)
        const cplx n2 = make_double2(-(s2 * b.y - s1 * d.y), s2 * b.x - s1 * d.x);  // i (s2 b - s1 d)
        v[0] = make_double2(v[0].x + a.x + c.x, v[0].y + a.y + c.y);
        v[1] = cadd(m1, n1);
        v[4] = csub(m1, n1);
        v[2] = cadd(m2, n2);
        v[3] = csub(m2, n2);
    }
};

// 8 = 4 x 2: n = 2 n1 + n2, k = k1 + 4 k2:  DFT4 over n1 for each n2, twiddle W8^(n2 k1), DFT2 over n2
template <bool INV> struct SmallDft<8, INV> {
    static IMCOM_FFT_HD void run(cplx (&v)[8])
    {
        cplx e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        SmallDft<4, INV>::run(e);
        SmallDft<4, INV>::run(o);
        o[1] = cturn8<INV>(o[1]);
        o[2] = cturn<INV>(o[2]);
        o[3] = cturn38<INV>(o[3]);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int k = 0; k < 4; k++) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
    }
};

// 16 = 4 x 4: n = 4 n1 + n2, k = k1 + 4 k2:  DFT4 over n1 for each n2, twiddle W16^(n2 k1), DFT4 over n2 for each k1
template <bool INV> struct SmallDft<16, INV> {
    static IMCOM_FFT_HD void run(cplx (&v)[16])
    {
        const double c1 = 0.9238795325112867, s1 = 0.3826834323650898;  // cos, sin (pi / 8)
        const double sg = INV ? 1.0 : -1.0;
        cplx y[4][4];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int n2 = 0; n2 < 4; n2++) {
            cplx t[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
            SmallDft<4, INV>::run(t);
            for (int k1 = 0; k1 < 4; k1++) y[n2][k1] = t[k1];
        }
        // W16^(n2 k1): exponents 1 2 3 / 2 4 6 / 3 6 9
        y[1][1] = cmulf(y[1][1], make_double2(c1, sg * s1));
        y[1][2] = cturn8<INV>(y[1][2]);
        y[1][3] = cmulf(y[1][3], make_double2(s1, sg * c1));
        y[2][1] = cturn8<INV>(y[2][1]);
        y[2][2] = cturn<INV>(y[2][2]);
        y[2][3] = cturn38<INV>(y[2][3]);
        y[3][1] = cmulf(y[3][1], make_double2(s1, sg * c1));
        y[3][2] = cturn38<INV>(y[3][2]);
        y[3][3] = cmulf(y[3][3], make_double2(-c1, -sg * s1));
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int k1 = 0; k1 < 4; k1++) {
            cplx t[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
            SmallDft<4, INV>::run(t);
            for (int k2 = 0; k2 < 4; k2++) v[k1 + 4 * k2] = t[k2];
        }
    }
};

}  // namespace imcom
