// Host check of pyimcom_amd/csrc/fft_radix.h: every in-register butterfly against the direct DFT sum.
// Prints "R dir maxerr" per case; tests/test_fft_radix.py compiles and runs it with g++.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "fft_radix.h"

using namespace imcom;

template <int R, bool INV> static double check()
{
    cplx v[R], ref[R], in[R];
    double worst = 0.0;
    for (int trial = 0; trial < 20; trial++) {
        for (int t = 0; t < R; t++) in[t] = v[t] = make_double2(drand48() - 0.5, drand48() - 0.5);
        for (int u = 0; u < R; u++) {
            double re = 0.0, im = 0.0;
            for (int t = 0; t < R; t++) {
                const double a = (INV ? 2.0 : -2.0) * M_PI * (double)((u * t) % R) / R;
                re += in[t].x * cos(a) - in[t].y * sin(a);
                im += in[t].x * sin(a) + in[t].y * cos(a);
            }
            ref[u] = make_double2(re, im);
        }
        SmallDft<R, INV>::run(v);
        for (int u = 0; u < R; u++) worst = fmax(worst, fmax(fabs(v[u].x - ref[u].x), fabs(v[u].y - ref[u].y)));
    }
    printf("%d %s %.3e\n", R, INV ? "inv" : "fwd", worst);
    return worst;
}

int main()
{
    srand48(7);
    double w = 0.0;
    w = fmax(w, check<2, false>());  w = fmax(w, check<2, true>());
    w = fmax(w, check<3, false>());  w = fmax(w, check<3, true>());
    w = fmax(w, check<4, false>());  w = fmax(w, check<4, true>());
    w = fmax(w, check<5, false>());  w = fmax(w, check<5, true>());
    w = fmax(w, check<8, false>());  w = fmax(w, check<8, true>());
    w = fmax(w, check<16, false>()); w = fmax(w, check<16, true>());
    return w < 1e-14 ? 0 : 1;
}
