/*
 * gp_oracle.h — CPU restatement of the Brightorange90/GP hot path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or fixtures for this
 * path and cannot be built here (Eigen, NLopt and the MVMO submodule are
 * absent).  This oracle is therefore pinned only by (i) an independent NumPy/SciPy evaluation of
 * the same closed forms (tests/test_oracle.py), (ii) the reference's own analytic-vs-finite-
 * difference gradient check (GP.cpp:486-507, FITC.cpp:324-352, VFE.cpp:254-282) and (iii) the
 * reference's data files as inputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this code.
 * The product (libgpx.so) never links or loads it.
 *
 * Conventions (same as the reference): inputs X are d x n column-major (one point = d contiguous
 * doubles, GP.cpp:18-19); matrices are column-major; hyp = [log l_1..log l_d, log sf, log sn, mean]
 * for SE-ARD (CovSEard.cpp:6, GP.cpp:565-574) and [log l, log sf, log sn, mean] for SE-iso
 * (CovSEiso.cpp:5).  All functions follow the reference's literal operation order (explicit
 * inverse, per-hyper dK materialisation, direct-difference distances).
 */
#ifndef GP_ORACLE_H
#define GP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_COV_SE_ARD = 0, ORC_COV_SE_ISO = 1 };

/* number of kernel hypers: d+1 (ARD) or 2 (iso)  — CovSEard.cpp:6, CovSEiso.cpp:5 */
int orc_cov_num_hyp(int cov, int d);

/* util.cpp:64-75  D[p x q] = squared distances by direct differences */
void orc_sdist_mm(int d, const double* X, long p, const double* Y, long q, double* D);

/* CovSEard.cpp:7-11 / CovSEiso.cpp:6-11   K[p x q] */
void orc_cov_k(int cov, int d, const double* hyp, const double* x1, long p, const double* x2, long q,
               double* K);
/* CovSEard.cpp:12-22 / CovSEiso.cpp:12-28   dK[p x q] for kernel hyper idx, given K */
void orc_cov_dk_dhyp(int cov, int d, const double* hyp, int idx, const double* x1, long p,
                     const double* x2, long q, const double* K, double* dK);
/* CovSEard.cpp:28-40 / CovSEiso.cpp:34-46   dK[d x q] = d k(x1, x2_b) / d x1 for ONE point x1 */
void orc_cov_dk_dx1(int cov, int d, const double* hyp, const double* x1, const double* x2, long q,
                    double* dK);

/* util.h:32-41: sample standard deviation with n-1 */
double orc_stddev(const double* v, long n, long stride);

/* GP.cpp:85-92 (+ CovSEard.cpp:72-79 / CovSEiso.cpp:79-85): default hypers, length nk+2 */
void orc_default_hyps(int cov, int d, long n, const double* X, const double* y, double noise_lb,
                      double* hyp);
/* GP.cpp:514-534 (+ CovSEard.cpp:41-71 / CovSEiso.cpp:47-78): bounds, length nk+2 each */
void orc_hyp_range(int cov, int d, long n, const double* X, const double* y, double noise_lb,
                   double* lb, double* ub);

/* MatrixSolver.cpp:6-29 (LLT semantics). A is n x n column-major, lower triangle is read.
 * Returns 0 when SPD (all pivots > 0).  L overwrites the lower triangle. */
int orc_chol(double* A, long n);
void orc_chol_solve(const double* L, long n, double* B, long nrhs); /* in place, B n x nrhs */
double orc_chol_logdet(const double* L, long n);
void orc_chol_inverse(const double* L, long n, double* Inv);        /* solve(Identity) */

/* GP.cpp:120-176. Returns NLL (+inf on failure; grad filled with +inf).  grad length nk+2. */
double orc_exact_nll(int cov, int d, long n, const double* X, const double* y, const double* hyp,
                     int calc_grad, double* grad);

/* Exact-GP state after GP::_setK (GP.cpp:423-444) */
typedef struct orc_exact_state orc_exact_state;
/* hyp is in/out: the noise escalation loop (GP.cpp:431-440) may raise hyp[nk].
 * literal_spd_check != 0 runs the reference's full _check_SPD (eigenvalues + |I-K Kinv|);
 * 0 uses Cholesky success only. */
orc_exact_state* orc_exact_setk(int cov, int d, long n, const double* X, const double* y,
                                double* hyp, int literal_spd_check);
void orc_exact_free(orc_exact_state*);
/* GP.cpp:273-334: mu[t], s2[t] (either may be NULL); gy/gs2 [d x t] may be NULL */
void orc_exact_predict(const orc_exact_state*, const double* Xs, long t, double* mu, double* s2,
                       double* gy, double* gs2);

/* FITC.cpp:201-319.  U is d x m.  grad length nk+2 (zeros on failure, FITC.cpp:228-233). */
double orc_fitc_nll(int cov, int d, long n, const double* X, const double* y, const double* U, long m,
                    double jitter, const double* hyp, int calc_grad, double* grad);
/* VFE.cpp:165-249 */
double orc_vfe_nll(int cov, int d, long n, const double* X, const double* y, const double* U, long m,
                   double jitter, const double* hyp, int calc_grad, double* grad);

typedef struct orc_sparse_state orc_sparse_state;
/* FITC.cpp:165-200 ; jitter is in/out (doubling loop) */
orc_sparse_state* orc_fitc_setk(int cov, int d, long n, const double* X, const double* y,
                                const double* U, long m, double* jitter, const double* hyp);
/* VFE.cpp:132-160 */
orc_sparse_state* orc_vfe_setk(int cov, int d, long n, const double* X, const double* y,
                               const double* U, long m, double* jitter, const double* hyp);
void orc_sparse_free(orc_sparse_state*);
/* FITC.cpp:109-164 */
void orc_fitc_predict(const orc_sparse_state*, const double* Xs, long t, double* mu, double* s2,
                      double* gy, double* gs2);
/* VFE.cpp:113-131 (no s2 gradient in the reference) */
void orc_vfe_predict(const orc_sparse_state*, const double* Xs, long t, double* mu, double* s2,
                     double* gy);

#ifdef __cplusplus
}
#endif
#endif
