// d5512.h -- D5512 interpolation weights and the 10x10 stencil (shared by interp.hip and build_a.hip).
#pragma once
#include "common.h"

namespace imcom {

// routine.py:29-122: taps k and 9-k are even(fh^2) +/- odd(fh^2)*fh, Horner in fh^2.
static __constant__ double D5512_EVEN[5][5] = {
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
static __constant__ double D5512_ODD[5][5] = {
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

__device__ __forceinline__ void d5512_getw(double (&w)[10], double fh)
{
    const double fh2 = fh * fh;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double e = D5512_EVEN[k][0], o = D5512_ODD[k][0];
#pragma unroll
        for (int c = 1; c < 5; c++) {
            e = e * fh2 + D5512_EVEN[k][c];
            o = o * fh2 + D5512_ODD[k][c];
        }
        o *= fh;
        w[k] = e + o;
        w[9 - k] = e - o;
    }
}

// truncation toward zero like np.int32(x); anything absurd is treated as off-grid
__device__ __forceinline__ int to_cell(double x)
{
    return (x > -1.0e9 && x < 1.0e9) ? (int)x : -1000000;
}

// one 10x10 stencil: f points at tap (0,0); step = +1 (plain table) or -1 (table flipped in both axes,
// base at its last element): inner sum over x taps, outer over y taps (routine.py:176-180)
__device__ __forceinline__ double stencil(const double *__restrict__ f, long row_stride, int step,
                                          const double (&wx)[10], const double (&wy)[10])
{
    double out = 0.0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const double *row = f + i * row_stride;
        double strip = 0.0;
#pragma unroll
        for (int j = 0; j < 10; j++) strip += wx[j] * row[j * step];
        out += strip * wy[i];
    }
    return out;
}

}  // namespace imcom
