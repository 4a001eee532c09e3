/*
 * egdst_call_hip.c -- MEX gateway  res = egdst_call(model, sw, args)  over the MI355X library.
 * Replaces @egdstmodel/egdst_call.c:17-125 (mexFunction) and :127-164 (vf); called by egdstmodel.call, egdstmodel.m:1190-1200.
 * sw: 1 utility, 2 marginal utility, 3 discount, 4 budget, 5 marginal budget, 6 value function; args is narg x ncol with
 * MATLAB's 1-based it / ist / id.  The solution comes from the model object (:32-34) and is uploaded; nothing is kept
 * between calls.  Wrong argument counts only warn in the reference (:25-26); so here.
 */
#include "egdst_shim_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    const mxArray *model, *M, *D;
    egdst_model_info info;
    egdst_desc d;
    egdst_handle *h;
    int sw, narg, ncol, nt, rc;

    if (nrhs != 3) mexWarnMsgTxt("Error in call(): wrong number of inputs!");
    if (nlhs != 1) mexWarnMsgTxt("Error in call(): wrong number of outputs!");
    model = prhs[0];
    egdst_get_model_info(&info);
    shim_descriptor(model, &d);
    nt = d.T - d.t0 + 1;
    M = mxGetProperty(model, 0, "M");
    D = mxGetProperty(model, 0, "D");
    if (M == NULL || D == NULL) mexErrMsgTxt("Error: the model has not yet been solved!"); /* :34 */
    sw = (int)mxGetScalar(prhs[1]);
    narg = (int)mxGetM(prhs[2]);
    ncol = (int)mxGetN(prhs[2]);
    plhs[0] = mxCreateDoubleMatrix((mwSize)narg, 1, mxREAL); /* zeros: what a wrong column count leaves behind */

    h = shim_handle(model, &d, &info);
    if (!h) mexErrMsgTxt(egdst_last_error());
    rc = shim_upload_solution(h, M, D, info.nst, nt);
    if (!rc) rc = egdst_call(h, 0, sw, narg, ncol, mxGetPr(prhs[2]), mxGetPr(plhs[0]));
    egdst_destroy(h);
    if (rc) mexWarnMsgTxt(egdst_last_error());
}
