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
/* C = alpha * A^T diag(kscale) B' + beta*C with A stored K-contiguous (K x M, column stride lda): the
 * weighted Gram products Kux Gamma^-1 Kxu (FITC.cpp:217) and B^T diag(w) B (FITC.cpp:276-290 summed
 * over the hypers) without forming the row-scaled n x m copy the reference builds. */
int gpx_dgemm_kscale(int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                     const double* kscale, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                     int flags, double* work, int64_t work_doubles);

/* C = alpha * A * B^T (both operands M/N-contiguous) together with the row statistics of the result in 2*ceil(N/128)
 * parts of 64 columns (column n belongs to part p = 2*(n/128) + (n%64)/32: the columns one wave of the 128-tile owns;
 * callers sum over the parts; ldr >= M, every part of every row written, deterministic):
 *   R1[p*ldr + m] = sum_{n in part p} C(m,n) * E(m,n)     (E == NULL: E = C, row sums of squares)
 *   R2[p*ldr + m] = sum_{n in part p} E(m,n) * evec[n]    (E, evec, R2 all given or all NULL)
 * Replaces the separate passes diag(Kxu Kuu^-1 Kux) = rowsum(V o V) (FITC.cpp:215), rowsum(C o Kxu) and Kxu * a
 * (FITC.cpp:219-220, 262-263): the statistics leave the GEMM that forms V / C in its epilogue.
 * Operand dot (E == NULL, R1 == NULL, evec (K entries) and R2 (M entries) given; full K range, no triangular flags):
 *   R2[m] = sum_k A(m,k) * evec[k]
 * formed from the fragments of A on their way to the matrix cores: Kxu * a of VFE.cpp:176 (alpha = (r - Kxu a)/sn2)
 * without a pass of its own over Kxu, inside the product Kxu * P of VFE.cpp:205-209. */
int gpx_dgemm_rowdot(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, const double* B,
                     int64_t ldb, double* C, int64_t ldc, int flags, const double* E, int64_t lde, const double* evec,
                     double* R1, double* R2, int64_t ldr);

/* No counterpart in the reference.  gpx_dgemm_splitk for callers that cannot allocate (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the split-K slab workspace is a
 * static device array of the library, GPX_NOALLOC_WORK_DOUBLES doubles (64 MiB) on every GPU the library
 * runs on, held until the process ends.  Calls hipMalloc / hipFree never; the first call on a device looks
 * the array's address up once (hipGetSymbolAddress).  Limit: M*N <= GPX_NOALLOC_MAX_MN output elements
 * (e.g. the m = 1024 Gram product), larger sizes return 1 (GPX_ERR_ARG) with a message on stderr.  All
 * calls on one GPU share the array: calls that overlap in time on one GPU race.  The kernels are launched on
 * hip_stream (a hipStream_t; NULL = the legacy stream like gpx_dgemm_splitk, which a graph capture cannot record:
 * callers under capture pass their capturing stream, and must have made one un-captured call on that device
 * before, so that the address look-up is not part of the capture).  Bit-identical to gpx_dgemm_splitk for a
 * workspace of the same capacity. */
#define GPX_NOALLOC_WORK_DOUBLES (8L * 1024 * 1024)
#define GPX_NOALLOC_MAX_MN (1024L * 1024)
int gpx_dgemm_splitk_noalloc(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                             int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                             int flags, void* hip_stream);

/* ---- measurement hooks (no reference counterpart) ------------------------------------------ */
/* HIP-event timing of every GEMM launch with >= min_flops useful flops, on the launch stream; a
 * negative min_flops selects exactly the launches of the 128x128-tile kernel (the one bench.py's MFMA
 * roofline object and the rocprofv3 --pmc traffic figure are about). */
int gpx_prof_gemm_begin(double min_flops, int max_launches);
int gpx_prof_gemm_end(double* total_ms, double* total_flops, int64_t* launches, double* skipped_flops);
/* The same for the HBM-bound pairwise kernels: cls 0 = kernel-matrix builds (Cov::k, algorithmic bytes
 * = 8 p q written, 4 n^2 for the lower-triangle-only build), cls 1 = the fused gradient trace pass
 * (bytes = the weight matrices read); launches below min_bytes are left out. */
int gpx_prof_pairwise_begin(double min_bytes, int max_launches);
int gpx_prof_pairwise_end(int cls, double* total_ms, double* total_bytes, int64_t* launches);
/* back-to-back v_mfma_f64_16x16x4_f64 from registers on every SIMD: the matrix pipe's ceiling */
int gpx_bench_mfma_f64_peak(double* tflops);

/* ---- the hot path: objective + gradient, _setK, predict ------------------------------------- */
typedef struct gpx_ctx gpx_ctx;
enum { GPX_MODEL_EXACT = 0, GPX_MODEL_FITC = 1, GPX_MODEL_VFE = 2 };
enum { GPX_COV_SE_ARD = 0, GPX_COV_SE_ISO = 1 }; /* GP.h:17-21 CovFunc */

/* One context per GPU / per GP object (GP.h:14: one GP instance owns its data and solvers).
 * Not re-entrant, like the reference object (GP.cpp:129 mutates the shared solver). */
int gpx_ctx_create(int device, gpx_ctx** out);
int gpx_ctx_destroy(gpx_ctx* ctx);

/* GP::GP(train_in, train_out) (GP.cpp:20-37) and add_data (GP.cpp:43-55): uploads HOST X (d x n
 * column-major) and y (n).  For row-sharded FITC/VFE each rank passes its own rows. */
int gpx_set_data(gpx_ctx* ctx, const double* X, const double* y, int64_t n, int64_t d);
/* FITC::set_inducing / VFE::set_inducing (FITC.cpp:22-26, VFE.cpp:21-25): HOST U (d x m). */
int gpx_set_inducing(gpx_ctx* ctx, const double* U, int64_t m);

/* Sum-all-reduce of `count` doubles at DEVICE pointer buf, in place, across the ranks that share
 * the training rows (no reference counterpart: the reference is single-process).  Called from
 * inside gpx_nll_grad / gpx_prepare for FITC and VFE after the library's stream has been
 * synchronised; must return 0 after the reduced values are visible to later device work. */
typedef int (*gpx_allreduce_fn)(double* buf, int64_t count, void* user);
int gpx_set_allreduce(gpx_ctx* ctx, gpx_allreduce_fn fn, void* user);
/* Stream-ordered flavour of the same hook (no reference counterpart either): ENQUEUE the in-place
 * sum-all-reduce on hip_stream (a hipStream_t of the library; buf is complete in that stream's
 * order, the consumers are queued behind the call on it) and return without waiting — e.g.
 * ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, (hipStream_t)hip_stream).  The host then
 * never blocks inside an evaluation and the m x m chains stay queued ahead of the GPU.  The library
 * has two streams and hands the hook whichever one the reduced values are consumed on (FITC: the
 * m x m part of the second all-reduce goes on the side stream beside the gradient trace, the trace
 * sums follow on the main stream: three calls per evaluation; VFE: two).  Setting one flavour
 * clears the other. */
typedef int (*gpx_allreduce_stream_fn)(double* buf, int64_t count, void* hip_stream, void* user);
int gpx_set_allreduce_stream(gpx_ctx* ctx, gpx_allreduce_stream_fn fn, void* user);
/* Native flavour (no reference counterpart; north star "RCCL all-reduce over xGMI of the m x m and m-vector partials"):
 * the library itself issues ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, nccl_comm, <its stream>) for both
 * all-reduces of an evaluation and for the one of gpx_prepare, stream-ordered like the hook above.  nccl_comm is an
 * ncclComm_t created and owned by the caller (one process per GPU, ncclCommInitRank; world size 1 is valid); NULL
 * unregisters.  libgpx.so does not link librccl: ncclAllReduce is resolved at run time, first in the process (the
 * library the communicator came from), then from librccl.so.1; returns 4 (unsupported) when neither exists.
 * gpx_rccl_timing(ctx, 1) brackets every collective with two events on its stream; gpx_rccl_elapsed returns the device
 * time between them (waiting for peers included) and their count since the last call, and resets both. */
int gpx_set_allreduce_rccl(gpx_ctx* ctx, void* nccl_comm);
int gpx_rccl_timing(gpx_ctx* ctx, int on);
int gpx_rccl_elapsed(gpx_ctx* ctx, double* total_ms, int64_t* collectives);
const char* gpx_rccl_source(void);
/* buf *= factor, enqueued on hip_stream (no reference counterpart): lets a test stand in for "R ranks
 * with identical shards" inside a stream-ordered hook, where the all-reduce is a multiplication by R. */
int gpx_dscal_stream(double* buf, int64_t count, double factor, void* hip_stream);

/* virtual GP::_calcNegLogProb(hyp, g, calc_grad) (GP.cpp:120-176), FITC::_calcNegLogProb
 * (FITC.cpp:201-319), VFE::_calcNegLogProb (VFE.cpp:165-249).  hyp and grad are HOST vectors of
 * length d+3 (ARD) / 4 (iso): [log l.., log sf, log sn, mean].  jitter = _jitter_u (sparse only).
 * On numerical failure *nll = +inf and grad = +inf (exact, GP.cpp:132-134,170) or 0 (sparse,
 * FITC.cpp:232, VFE.cpp:194); the return status stays 0. */
int gpx_nll_grad(gpx_ctx* ctx, int model, int cov, const double* hyp, int calc_grad, double jitter,
                 double* nll, double* grad);

/* GP::_setK (GP.cpp:423-444), FITC::_setK (FITC.cpp:165-200), VFE::_setK (VFE.cpp:132-160):
 * factor at the final hypers and cache alpha.  hyp is in/out for the exact model (noise
 * escalation, GP.cpp:431-440), jitter is in/out for the sparse models (doubling loops). */
int gpx_prepare(gpx_ctx* ctx, int model, int cov, double* hyp, double* jitter);

/* The residual part of GP::_check_SPD (GP.cpp:455-457: inv_err = mean|I - K K^-1| < 1e-4) inside gpx_prepare of the
 * exact model, OFF by default (sampled_columns = 0: a successful Cholesky is the SPD test).  With sampled_columns > 0
 * the mean is taken over that many evenly spaced columns j of e_j - K (K^-1 e_j) (>= n: all columns, the reference's
 * exact value); a value >= 1e-4 escalates the noise like a failed pivot (GP.cpp:431-440).  Cost: two triangular solves
 * with that many right-hand sides + one n x cols x n product on rebuilt rows of K.  The eigenvalue part of _check_SPD
 * is implied by the Cholesky test. */
int gpx_set_spd_check(gpx_ctx* ctx, int sampled_columns);
int gpx_last_spd_residual(gpx_ctx* ctx, double* inv_err);

/* GP::_predict / _predict_y / _predict_s2 (GP.cpp:273-334), FITC (FITC.cpp:109-164), VFE
 * (VFE.cpp:113-131) without input gradients: HOST Xs (d x t); mu and/or s2 may be NULL. */
int gpx_predict(gpx_ctx* ctx, const double* Xs, int64_t t, double* mu, double* s2);
/* The need_g = true branches of the same functions (GP.cpp:284-296, 305-313, 324-333;
 * FITC.cpp:118-129; VFE.cpp:118-123) for t points at once: gmu, gs2 are d x t column-major
 * (d mean / d x*, d s2 / d x*), each nullable.  VFE has no variance gradient (returns 4). */
int gpx_predict_with_grad(gpx_ctx* ctx, const double* Xs, int64_t t, double* mu, double* s2, double* gmu,
                          double* gs2);

/* Cov::k (Cov/Cov.h:14; CovSEard.cpp:7-11, CovSEiso.cpp:6-11) on HOST buffers: x1 d x p, x2 d x q,
 * K p x q column-major.  hyp holds at least the kernel hypers. */
int gpx_cov_k(int cov, int64_t d, const double* hyp, const double* x1, int64_t p, const double* x2,
              int64_t q, double* K);
/* Cov::dk_dhyp (Cov/Cov.h:15-16; CovSEard.cpp:12-27, CovSEiso.cpp:12-31) on HOST buffers: dK (p x q column-major) =
 * d k(x1, x2) / d hyp[idx] from K = k(x1, x2) (p x q; NULL: the 4-argument overload, K is computed first).  idx < d: one
 * lengthscale (ARD) / idx 0: the shared one (iso); the last kernel hyper (sf): 2 K.  For API compatibility with seam (1):
 * the training path never materialises a dK (gpx_nll_grad contracts all of them in one fused pass). */
int gpx_cov_dk_dhyp(int cov, int64_t d, const double* hyp, int64_t idx, const double* x1, int64_t p, const double* x2,
                    int64_t q, const double* K, double* dK);
/* MatrixSolverLLT (MatrixSolver.cpp:6-29) on HOST buffers: A n x n column-major (lower read).
 * *spd = check_SPD(); logdet, X = solve(B) (n x nrhs, in place) and Inv = inverse() are optional. */
int gpx_chol_solve(int64_t n, const double* A, int* spd, double* logdet, double* B, int64_t nrhs,
                   double* Inv);

/* The same with the factor kept ON THE DEVICE between MatrixSolver::decomp (MatrixSolver.cpp:9-12) and
 * ::solve / ::inverse (MatrixSolver.cpp:13-16, 25-28): one O(n^3) factorisation per decomp(), not one per
 * solve.  The handle belongs to `device` (-1: the caller's current device); every call switches to it and restores
 * the caller's current device before it returns; solve / inverse staging buffers are kept in the handle. */
typedef struct gpx_chol gpx_chol;
int gpx_chol_factor(int device, int64_t n, const double* A, gpx_chol** out, int* spd, double* logdet);
int gpx_chol_solve_factored(const gpx_chol* f, double* B, int64_t nrhs);
int gpx_chol_inverse_factored(const gpx_chol* f, double* Inv);
int gpx_chol_free(gpx_chol* f);

#ifdef __cplusplus
}
#endif
#endif
