/*
 * gpx.h — C ABI of libgpx.so: the MI355X-native GP training hot path (exact / FITC / VFE negative
 * log marginal likelihood + hyper-parameter gradient, and predictive mean / variance).
 *
 * Plain C: double* buffers, int64_t sizes, int status (0 = OK), never throws.  Matrices are
 * column-major; inputs X are d x n column-major (one point = d contiguous doubles), exactly the
 * reference's Eigen layout (GP.cpp:18-19).  Citations are relative to the reference repository.
 */
#ifndef GPX_H
#define GPX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- device memory helpers (no reference counterpart: the reference is host-only) ---------- */
int gpx_device_count(int* n);
int gpx_set_device(int dev);
int gpx_malloc(void** p, int64_t bytes);
int gpx_free(void* p);
int gpx_memcpy_h2d(void* dst, const void* src, int64_t bytes);
int gpx_memcpy_d2h(void* dst, const void* src, int64_t bytes);
int gpx_memset_zero(void* dst, int64_t bytes);
int gpx_device_synchronize(void);

/* ---- dense building blocks on DEVICE pointers ---------------------------------------------- */
/* Replaces the Eigen dense products used throughout GP.cpp / FITC.cpp / VFE.cpp (e.g.
 * FITC.cpp:217, VFE.cpp:173).  C = alpha*op(A)*op(B)^T + beta*C; a_kc/b_kc = 1 means the operand
 * is stored K-contiguous; flags: see GEMM_* in csrc/gpx_internal.h (1 = lower tiles only). */
int gpx_dgemm(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
              int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int flags);
int gpx_dgemm_splitk(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                     int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                     int flags, double* work, int64_t work_doubles);

#ifdef __cplusplus
}
#endif
#endif
