/*
 * DECLARATIONS ONLY -- test infrastructure (tests/test_shims.py).  The part of MATLAB's C MEX API (mex.h / matrix.h) that
 * the shim sources under shims/ use, as prototypes with no bodies, so that `gcc -fsyntax-only -Wall -Werror` can hold the shim sources to the
 * API's types: an undeclared identifier, a wrong argument count or a pointer mismatch fails the test.  Nothing can be
 * linked or run against this header; it is not a stand-in for MATLAB, and the reference is not built with it.
 * Signatures as documented by MathWorks (C Matrix API / MEX library).
 */
#ifndef EGDST_TEST_MEX_DECLS_H
#define EGDST_TEST_MEX_DECLS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS = 1, mxSTRUCT_CLASS = 2, mxLOGICAL_CLASS = 3, mxDOUBLE_CLASS = 6 } mxClassID;

double *mxGetPr(const mxArray *pa);
double mxGetScalar(const mxArray *pa);
size_t mxGetM(const mxArray *pa);
size_t mxGetN(const mxArray *pa);
size_t mxGetNumberOfElements(const mxArray *pa);
mxArray *mxGetProperty(const mxArray *pa, mwIndex index, const char *propname);
mxArray *mxGetField(const mxArray *pa, mwIndex index, const char *fieldname);
mxArray *mxGetCell(const mxArray *pa, mwIndex index);
void mxSetCell(mxArray *pa, mwIndex index, mxArray *value);
mxArray *mxCreateCellArray(mwSize ndim, const mwSize *dims);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag);
void *mxMalloc(size_t n);
void mxFree(void *ptr);
void mexErrMsgTxt(const char *error_msg);
void mexWarnMsgTxt(const char *warn_msg);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
