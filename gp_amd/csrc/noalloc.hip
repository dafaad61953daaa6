// noalloc.hip — the one entry point of libgpx.so that allocates nothing (no reference counterpart).
//
// gpx_dgemm_splitk_noalloc is gpx_dgemm_splitk (the short-and-wide Gram products Kux Kxu of
// FITC.cpp:217 / VFE.cpp:173: K >> M, N, split over K into partial slabs that a second kernel sums)
// with the slab workspace taken from a file-scope device array instead of a caller's buffer, for
// callers that cannot call hipMalloc: code under HIP graph capture (the launches go to the caller's
// hip_stream; NULL = the legacy stream, which cannot be captured), tight serving loops, harnesses
// that forbid allocation.  Each GPU the library runs on holds its own copy of the array until the
// process ends; its address on a device is looked up once (hipGetSymbolAddress) on the first call
// made with that device current, and only that pointer — never the array's name — is handed to the
// kernels.  Calls on one GPU share the array: calls that overlap in time on one GPU race.
#include "gpx_internal.h"
#include "../../include/gpx.h"
#include <mutex>

namespace {

// split-K slabs of the no-alloc entry point: GPX_NOALLOC_WORK_DOUBLES doubles = 64 MiB per GPU
static __device__ __align__(256) double scratch[GPX_NOALLOC_WORK_DOUBLES];

constexpr int kMaxDev = 64;
double* g_scratch_ptr[kMaxDev] = {};  // per device: each GPU holds its own copy at its own address
std::mutex g_scratch_mu;              // first calls from several host threads look the address up once

int scratch_of_current_device(double** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDev) return GPX_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return GPX_ERR_HIP;
        g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_scratch_ptr[dev];
    return GPX_OK;
}

}  // namespace

extern "C" int gpx_dgemm_splitk_noalloc(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, double alpha,
                                        const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
                                        double* C, int64_t ldc, int flags, void* hip_stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return GPX_ERR_ARG;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || ldc < M) return GPX_ERR_ARG;
    if (M > GPX_NOALLOC_MAX_MN || (M > 0 && N > GPX_NOALLOC_MAX_MN / M)) {   // M*N is never formed: no overflow
        fprintf(stderr, "gpx_dgemm_splitk_noalloc: M x N = %lld x %lld exceeds the static workspace limit of %lld output "
                        "elements (use gpx_dgemm_splitk with a caller-provided workspace)\n",
                (long long)M, (long long)N, (long long)GPX_NOALLOC_MAX_MN);
        return GPX_ERR_ARG;
    }
    double* work = nullptr;
    const int rc = scratch_of_current_device(&work);
    if (rc) return rc;
    // Nothing has to start at zero: every slab element the reduction reads was written by this call's
    // GEMM kernel (slab stride = M*N of THIS call; with GEMM_LOWER the reduction skips the tiles above the
    // diagonal, which are never written), so contents left by an earlier, larger call are never read.
    return gpx_gemm_splitk(static_cast<hipStream_t>(hip_stream), a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags, work,
                           (long)GPX_NOALLOC_WORK_DOUBLES);
}
