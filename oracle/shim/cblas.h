/* TEST INFRASTRUCTURE ONLY (oracle/): minimal CBLAS surface needed to compile the reference CPU
 * sources where they lie (/root/reference/src/math_functions_cpu.cpp:44,63,99,105 call
 * cblas_sgemm/dgemm/saxpy/daxpy; /root/reference/src/mkl_alternate.hpp:34-36 includes <cblas.h>).
 * No CBLAS header is installed in this image, so the calls are forwarded to the Fortran BLAS
 * symbols (sgemm_/dgemm_) and cblas_?axpy that libtorch_cpu.so exports (MKL, statically linked
 * into torch).  This header is written for this repo; it is not taken from any BLAS distribution. */
#ifndef ME_AMD_ORACLE_CBLAS_SHIM_H
#define ME_AMD_ORACLE_CBLAS_SHIM_H
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_ORDER;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
void sgemm_(const char *, const char *, const int *, const int *, const int *, const float *,
            const float *, const int *, const float *, const int *, const float *, float *,
            const int *);
void dgemm_(const char *, const char *, const int *, const int *, const int *, const double *,
            const double *, const int *, const double *, const int *, const double *, double *,
            const int *);
void cblas_saxpy(const int, const float, const float *, const int, float *, const int);
void cblas_daxpy(const int, const double, const double *, const int, double *, const int);
#ifdef __cplusplus
}
#endif
/* Row-major C = op(A) op(B) is the column-major product C^T = op(B)^T op(A)^T. */
static inline void cblas_sgemm(CBLAS_ORDER order, CBLAS_TRANSPOSE ta, CBLAS_TRANSPOSE tb, int M,
                               int N, int K, float alpha, const float *A, int lda,
                               const float *B, int ldb, float beta, float *C, int ldc) {
  char ca = (ta == CblasNoTrans) ? 'N' : 'T', cb = (tb == CblasNoTrans) ? 'N' : 'T';
  if (order == CblasColMajor)
    sgemm_(&ca, &cb, &M, &N, &K, &alpha, A, &lda, B, &ldb, &beta, C, &ldc);
  else
    sgemm_(&cb, &ca, &N, &M, &K, &alpha, B, &ldb, A, &lda, &beta, C, &ldc);
}
static inline void cblas_dgemm(CBLAS_ORDER order, CBLAS_TRANSPOSE ta, CBLAS_TRANSPOSE tb, int M,
                               int N, int K, double alpha, const double *A, int lda,
                               const double *B, int ldb, double beta, double *C, int ldc) {
  char ca = (ta == CblasNoTrans) ? 'N' : 'T', cb = (tb == CblasNoTrans) ? 'N' : 'T';
  if (order == CblasColMajor)
    dgemm_(&ca, &cb, &M, &N, &K, &alpha, A, &lda, B, &ldb, &beta, C, &ldc);
  else
    dgemm_(&cb, &ca, &N, &M, &K, &alpha, B, &ldb, A, &lda, &beta, C, &ldc);
}
#endif
