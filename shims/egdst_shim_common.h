/*
 * egdst_shim_common.h -- what the three MEX shims share: the model object's run-time scalars -> egdst_desc, its parameter
 * values -> one draw, its solution cells -> the handle.  Counterpart of parseModel() + loadparameters() of the reference
 * (egdst_lib.c:34-62, compile.m:469-475) for a library whose model plugin is compiled in.
 *
 * The shims keep NO state between calls: every gateway call creates a handle, gives it what the reference's gateway reads
 * from the model object (parameters; for the simulator and the accessor also the cells M and D, egdst_simulator.c:61-68,
 * egdst_call.c:28-34), calls the library and destroys the handle.  Build, next to the per-model library:
 *     mex egdst_solver_hip.c    -I<repo>/include -I<repo>/shims -L<model dir> -legdst -output egdst_solver
 *     mex egdst_simulator_hip.c -I<repo>/include -I<repo>/shims -L<model dir> -legdst -output egdst_simulator
 *     mex egdst_call_hip.c      -I<repo>/include -I<repo>/shims -L<model dir> -legdst -output egdst_call
 * (the three `mex` lines of compile.m:781,793,805).  There is no MATLAB in the build image: tests/test_shims.py holds the
 * sources to a declarations-only mex.h (gcc -fsyntax-only -Wall -Werror), and the same sequence of library calls is what
 * tests/test_gpu_parity.py::test_import_* runs through ctypes.
 */
#ifndef EGDST_SHIM_COMMON_H
#define EGDST_SHIM_COMMON_H

#include "mex.h"
#include "egdst.h"

static double shim_prop(const mxArray *model, const char *name)
{
    return *mxGetPr(mxGetProperty(model, 0, name)); /* egdst_lib.c:37-48 */
}

/* run-time scalars + quadrature; `d->quadrature` points into a property copy that lives as long as the call */
static void shim_descriptor(const mxArray *model, egdst_desc *d)
{
    d->t0 = (int)shim_prop(model, "t0");
    d->T = (int)shim_prop(model, "T");
    d->ngridm = (int)shim_prop(model, "ngridm");
    d->ngridmax = (int)shim_prop(model, "ngridmax");
    d->nthrhmax = (int)shim_prop(model, "nthrhmax");
    d->ny = (int)shim_prop(model, "ny");
    d->mmax = shim_prop(model, "mmax");
    d->a0 = shim_prop(model, "a0");
    d->quadrature = mxGetPr(mxGetProperty(model, 0, "quadrature")); /* [qw qx], egdstmodel.m:1157-1160 */
}

/* a one-draw handle with the model's current parameter values (loadparameters, compile.m:469-475); NULL + message on failure */
static egdst_handle *shim_handle(const mxArray *model, const egdst_desc *d, const egdst_model_info *info)
{
    egdst_handle *h = NULL;
    double *par = (double *)mxMalloc(sizeof(double) * (size_t)(info->nparam > 0 ? info->nparam : 1));
    int i;
    for (i = 0; i < info->nparam; i++) par[i] = mxGetScalar(mxGetField(mxGetProperty(model, 0, "param"), i, "value"));
    if (egdst_create(d, 1, /*keep_history=*/1, NULL, &h)) {
        mxFree(par);
        return NULL;
    }
    if (egdst_set_params(h, par, 1)) {
        egdst_destroy(h);
        h = NULL;
    }
    mxFree(par);
    return h;
}

/* the solution of the model object -> the handle, cell by cell (cell index ist+it*nst, egdst_solver.c:922).  An empty or
 * missing cell stays unsolved (length 0), as in the reference, whose simulator then stops at it.  Returns 0 or an EGDST code. */
static int shim_upload_solution(egdst_handle *h, const mxArray *M, const mxArray *D, int nst, int nt)
{
    int it, ist, rc;
    for (it = 0; it < nt; it++)
        for (ist = 0; ist < nst; ist++) {
            const mxArray *cm = mxGetCell(M, (mwIndex)(ist + it * nst)), *cd = mxGetCell(D, (mwIndex)(ist + it * nst));
            const int len = cm ? (int)mxGetM(cm) : 0, thlen = cd ? (int)mxGetM(cd) : 0;
            if (cm && len > 0 && mxGetN(cm) != 4) return EGDST_E_ARG;
            if (cd && thlen > 0 && mxGetN(cd) != 2) return EGDST_E_ARG;
            rc = egdst_set_cell_M(h, 0, it, ist, len, len > 0 ? mxGetPr(cm) : NULL); /* (len x 4) [M C A V], column-major */
            if (rc) return rc;
            rc = egdst_set_cell_D(h, 0, it, ist, thlen, thlen > 0 ? mxGetPr(cd) : NULL); /* (thlen x 2) [D TH] */
            if (rc) return rc;
        }
    return 0;
}

#endif
