/*
 * mex_stub/mex.h -- DECLARATIONS ONLY, for `g++ -fsyntax-only` of aoadmm_mex.cpp in a container
 * that has no MATLAB (no mex.h / libmx).  Nothing here is linked, run or shipped; a real build uses
 * MATLAB's own mex.h (R2018a interleaved-complex API).  Signatures follow the documented C Matrix API.
 */
#ifndef AOADMM_MEX_STUB_H
#define AOADMM_MEX_STUB_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
#ifdef __cplusplus
extern "C" {
#endif
bool mxIsStruct(const mxArray*);
bool mxIsCell(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsEmpty(const mxArray*);
bool mxIsClass(const mxArray*, const char*);
mxArray* mxGetField(const mxArray*, mwIndex, const char*);
void mxSetField(mxArray*, mwIndex, const char*, mxArray*);
mxArray* mxGetCell(const mxArray*, mwIndex);
void mxSetCell(mxArray*, mwIndex, mxArray*);
mxArray* mxGetProperty(const mxArray*, mwIndex, const char*);
double mxGetScalar(const mxArray*);
double* mxGetDoubles(const mxArray*);
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
char* mxArrayToString(const mxArray*);
void mxFree(void*);
double mxGetNaN(void);
const mwSize* mxGetDimensions(const mxArray*);
mwSize mxGetNumberOfDimensions(const mxArray*);
bool mxIsChar(const mxArray*);
void* mxGetData(const mxArray*);
bool mxIsUint8(const mxArray*);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateCellMatrix(mwSize, mwSize);
mxArray* mxCreateStructMatrix(mwSize, mwSize, int, const char**);
mxArray* mxCreateString(const char*);
mxArray* mxDuplicateArray(const mxArray*);
void mexErrMsgIdAndTxt(const char*, const char*, ...);
int mexAtExit(void (*)(void));
int mexPrintf(const char*, ...);
int mexEvalString(const char*);
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#ifdef __cplusplus
}
#endif
#endif
