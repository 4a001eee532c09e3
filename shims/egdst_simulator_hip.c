/*
 * egdst_simulator_hip.c -- MEX gateway  sims = egdst_simulator(model, rndtype)  over the MI355X library.
 * Replaces @egdstmodel/egdst_simulator.c:47-117 (mexFunction); called by egdstmodel.sim, egdstmodel.m:1268.
 * As the reference does, it takes the solution from the model object (properties M and D, :66-68) -- the model may have
 * been solved in another session -- and uploads it (egdst_set_cell_M / egdst_set_cell_D); nothing is kept between calls.
 * Output: nsimout x nt x nsim, nsimout = 11 + nnst + nnd + numel(eq) (:95-101), NaN where an agent has no value.
 */
#include "egdst_shim_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    const mxArray *model, *init, *rs, *M, *D;
    egdst_model_info info;
    egdst_desc d;
    egdst_handle *h;
    mwSize dims[3];
    int nsim, nt, rndtype, rc;

    if (nrhs != 2) mexErrMsgTxt("Error: wrong number of inputs!");
    if (nlhs != 1) mexErrMsgTxt("Error: wrong number of outputs!");
    model = prhs[0];
    egdst_get_model_info(&info);
    shim_descriptor(model, &d);
    nt = d.T - d.t0 + 1;
    init = mxGetProperty(model, 0, "init");
    rs = mxGetProperty(model, 0, "randstream");
    nsim = (int)mxGetM(init);
    rndtype = (int)mxGetScalar(prhs[1]);
    M = mxGetProperty(model, 0, "M");
    D = mxGetProperty(model, 0, "D");
    if (M == NULL || D == NULL) mexErrMsgTxt("Error: the model has not yet been solved!"); /* :68 */

    h = shim_handle(model, &d, &info);
    if (!h) mexErrMsgTxt(egdst_last_error());
    rc = shim_upload_solution(h, M, D, info.nst, nt);
    if (rc) {
        egdst_destroy(h);
        mexErrMsgTxt(rc == EGDST_E_ARG ? "Error: the cells of M and D do not have the layout of a solution!" : egdst_last_error());
    }
    dims[0] = (mwSize)(11 + info.nnst + info.nnd + info.neq);
    dims[1] = (mwSize)nt;
    dims[2] = (mwSize)nsim;
    plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    rc = egdst_simulate(h, 0, mxGetPr(init), nsim, mxGetPr(rs), (long long)mxGetNumberOfElements(rs), rndtype, mxGetPr(plhs[0]));
    egdst_destroy(h);
    if (rc) mexErrMsgTxt(egdst_last_error()); /* short randstream, unsolved cell: hard errors in the reference too, :54-75 */
}
