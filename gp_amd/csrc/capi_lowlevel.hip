// capi_lowlevel.hip — device-memory helpers and the low-level dense entry points of the C ABI.
#include "gpx_internal.h"
#include "ctx.h"
#include "../../include/gpx.h"

extern "C" {

int gpx_device_count(int* n) {
    HIP_TRY(hipGetDeviceCount(n));
    return GPX_OK;
}
int gpx_set_device(int dev) {
    HIP_TRY(hipSetDevice(dev));
    return GPX_OK;
}
int gpx_malloc(void** p, int64_t bytes) {
    HIP_TRY(hipMalloc(p, (size_t)(bytes > 0 ? bytes : 1)));
    return GPX_OK;
}
int gpx_free(void* p) {
    HIP_TRY(hipFree(p));
    return GPX_OK;
}
int gpx_memcpy_h2d(void* dst, const void* src, int64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return GPX_OK;
}
int gpx_memcpy_d2h(void* dst, const void* src, int64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return GPX_OK;
}
int gpx_memset_zero(void* dst, int64_t bytes) {
    HIP_TRY(hipMemset(dst, 0, (size_t)bytes));
    return GPX_OK;
}
int gpx_device_synchronize(void) {
    HIP_TRY(hipDeviceSynchronize());
    return GPX_OK;
}

int gpx_dgemm(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
              int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int flags) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return GPX_ERR_ARG;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || ldc < M) return GPX_ERR_ARG;
    return gpx_gemm(0, a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags);
}

int gpx_dgemm_splitk(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                     int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                     int flags, double* work, int64_t work_doubles) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return GPX_ERR_ARG;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || ldc < M) return GPX_ERR_ARG;
    return gpx_gemm_splitk(0, a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags, work,
                           work_doubles);
}

int gpx_dgemm_kscale(int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                     const double* kscale, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                     int flags, double* work, int64_t work_doubles) {
    if (!A || !B || !C || !kscale || M < 0 || N < 0 || K < 0) return GPX_ERR_ARG;
    if (lda < K || ldb < (b_kc ? K : N) || ldc < M) return GPX_ERR_ARG;
    return gpx_gemm_splitk(0, 1, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags, work, work_doubles,
                           kscale);
}

int gpx_dgemm_rowdot(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, const double* B,
                     int64_t ldb, double* C, int64_t ldc, int flags, const double* E, int64_t lde, const double* evec,
                     double* R1, double* R2, int64_t ldr) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return GPX_ERR_ARG;
    if (!R1 && !(evec && R2 && !E)) return GPX_ERR_ARG;   // R1 == NULL: the operand dot R2 = A evec
    if (lda < M || ldb < N || ldc < M || (R1 && ldr < M) || (E && lde < M)) return GPX_ERR_ARG;
    GemmRowdot rd = {E, lde, evec, R1, R2, ldr};
    return gpx_gemm_rowdot(0, 0, 0, M, N, K, alpha, A, lda, B, ldb, C, ldc, flags, rd);
}

// measurement / test entry (not in gpx.h): the split-K product with BOTH operands K-contiguous and row-panelled
int gpx_dgemm_splitk_panelled(int64_t M, int64_t N, int64_t K, double alpha, const double* A, const double* B, int64_t pan, int64_t pst,
                              double beta, double* C, int64_t ldc, int flags, double* work, int64_t work_doubles, const double* kscale) {
    GemmPanels pn = {};
    pn.panA = pn.panB = pan; pn.pstA = pn.pstB = pst;
    return gpx_gemm_splitk(0, 1, 1, M, N, K, alpha, A, pan, B, pan, beta, C, ldc, flags, work, work_doubles, kscale, &pn);
}
int64_t gpx_dgemm_splitk_panel_rows(int64_t M, int64_t N, int64_t K, int flags, int64_t work_doubles) {
    return gpx_gemm_splitk_panel_rows(M, N, K, flags, work_doubles);
}

// measurement / test entry (not in gpx.h): the two products of a recursion node as gpx_potrf_inv issues them, C0 = C0 - L L^T on the
// lower tiles (L: n2 x n1) and C1 = L W (W: n1 x n1 lower triangular, K-contiguous), in one launch where gpx_gemm_pair pairs them
int gpx_dgemm_pair_node(int64_t n2, int64_t n1, const double* L, int64_t ldl, const double* W, int64_t ldw, double* C0, int64_t ldc0,
                        double* C1, int64_t ldc1) {
    if (!L || !W || !C0 || !C1 || n1 <= 0 || n2 <= 0 || ldl < n2 || ldw < n1 || ldc0 < n2 || ldc1 < n2) return GPX_ERR_ARG;
    return gpx_gemm_pair(0, n2, n2, n1, -1.0, L, ldl, L, ldl, 1.0, C0, ldc0, GEMM_LOWER, n2, n1, n1, 1.0, L, ldl, W, ldw, 0.0, C1,
                         ldc1, GEMM_KLO_N);
}

int gpx_dscal_stream(double* buf, int64_t count, double factor, void* hip_stream) {
    if (!buf || count < 0) return GPX_ERR_ARG;
    return gpx_vec_shift(static_cast<hipStream_t>(hip_stream), count, buf, 0.0, factor, buf);
}

/* measurement hooks */
int gpx_prof_gemm_begin(double min_flops, int max_launches) { return gpx_gemm_prof_begin(min_flops, max_launches); }
int gpx_prof_gemm_end(double* total_ms, double* total_flops, int64_t* launches, double* skipped_flops) {
    long l = 0;
    const int rc = gpx_gemm_prof_end(total_ms, total_flops, &l, skipped_flops);
    *launches = l;
    return rc;
}
int gpx_bench_mfma_f64_peak(double* tflops) { return gpx_mfma_peak(tflops); }
int gpx_prof_pairwise_begin(double min_bytes, int max_launches) { return gpx_pair_prof_begin(min_bytes, max_launches); }
int gpx_prof_pairwise_end(int cls, double* total_ms, double* total_bytes, int64_t* launches) {
    if (!total_ms || !total_bytes || !launches || (cls != 0 && cls != 1)) return GPX_ERR_ARG;
    long l = 0;
    const int rc = gpx_pair_prof_end(cls, total_ms, total_bytes, &l);
    *launches = l;
    return rc;
}

}  // extern "C"
