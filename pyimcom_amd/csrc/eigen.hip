// eigen.hip -- batched symmetric eigendecomposition and the eigen LA path (placeholder: filled in below).
#include "common.h"
#include "launchers.h"
using namespace imcom;
extern "C" {
int imcom_eigh(imcom_ctx *, int, const int *, int, const double *, double *, double *, int)
{
    set_error("imcom_eigh: not built yet");
    return IMCOM_ERR_UNSUPPORTED;
}
int imcom_solve_eigen(imcom_ctx *, int, const int *, int, int, const double *, const double *, const double *,
                      const double *, int, double, double, int, float *, float *, float *, float *, int *, int)
{
    set_error("imcom_solve_eigen: not built yet");
    return IMCOM_ERR_UNSUPPORTED;
}
}
