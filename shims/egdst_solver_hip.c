/*
 * egdst_solver_hip.c -- MEX gateway  [M, D, dbgout] = egdst_solver(model)  over the MI355X library.
 * Replaces @egdstmodel/egdst_solver.c:143-239 (mexFunction); called by egdstmodel.solve, egdstmodel.m:1170.
 * Same argument counts, same outputs: M and D are nst x nt cell arrays in saveoutput's layout (egdst_solver.c:917-952),
 * dbgout the (nt*nst*nd*2*nt) x 7 kink log (:178-181, 1866-1879).  Errors of the solver do not throw: warning + partially
 * filled cells (:237).
 */
#include "egdst_shim_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    const mxArray *model;
    egdst_model_info info;
    egdst_desc d;
    egdst_handle *h;
    mwSize dims[2];
    int it, ist, nt, rc, nrow;

    if (nrhs != 1) mexErrMsgTxt("Error: wrong number of inputs!");
    if (nlhs != 3) mexErrMsgTxt("Error: wrong number of outputs!"); /* DEBUGOUT build, egdst_solver.c:151 */
    model = prhs[0];
    egdst_get_model_info(&info);
    shim_descriptor(model, &d);
    nt = d.T - d.t0 + 1;
    h = shim_handle(model, &d, &info);
    if (!h) mexErrMsgTxt(egdst_last_error());
    egdst_set_dbgout(h, 1); /* keep the kink log: third output */
    rc = egdst_solve(h);

    dims[0] = (mwSize)info.nst;
    dims[1] = (mwSize)nt;
    plhs[0] = mxCreateCellArray(2, dims);
    plhs[1] = mxCreateCellArray(2, dims);
    for (it = 0; it < nt; it++)
        for (ist = 0; ist < info.nst; ist++) {
            int len = 0, thlen = 0;
            mxArray *M, *D;
            egdst_cell_dims(h, 0, it, ist, &len, &thlen);
            if (len == 0) continue; /* infeasible state, or not reached before an error: the cell stays empty */
            M = mxCreateDoubleMatrix((mwSize)len, 4, mxREAL);
            D = mxCreateDoubleMatrix((mwSize)thlen, 2, mxREAL);
            egdst_get_cell_M(h, 0, it, ist, mxGetPr(M)); /* column-major [M C A V], row 0 = the a0 row */
            egdst_get_cell_D(h, 0, it, ist, mxGetPr(D)); /* [D TH] */
            mxSetCell(plhs[0], (mwIndex)(ist + it * info.nst), M);
            mxSetCell(plhs[1], (mwIndex)(ist + it * info.nst), D);
        }
    nrow = nt * info.nst * info.nd * 2 * nt;
    plhs[2] = mxCreateDoubleMatrix((mwSize)nrow, 7, mxREAL);
    egdst_get_dbgout(h, 0, mxGetPr(plhs[2]), NULL);
    if (rc) mexWarnMsgTxt(egdst_last_error());
    egdst_destroy(h);
}
