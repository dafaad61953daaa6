/*
 * gp_oracle.c — CPU restatement of the Brightorange90/GP hot path.  TEST INFRASTRUCTURE ONLY.
 * See gp_oracle.h for the header comment ("PARITY UNPINNED") and the conventions.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * The operation ORDER of the reference is kept (explicit inverses, one dK per hyper, direct
 * difference distances); only Eigen's internal blocking of products/factorisations is not
 * reproduced, so results agree with the Eigen build to rounding, not bitwise.
 */
#include "gp_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* The reference is single-threaded (CMakeLists.txt:12 has no -fopenmp) and so is libgporacle.so.  The
 * same file built with -fopenmp (libgporacle_omp.so, `make omp`) spreads the independent columns /
 * right-hand sides of the O(n^3), O(n m^2) and O(n^2 d) loops over the host cores: that build exists
 * only for bench.py's "all host cores" CPU timing; every parity test uses the serial build. */
#ifdef _OPENMP
#define OMP_PRAGMA(x) _Pragma(#x)
#define OMP_FOR OMP_PRAGMA(omp parallel for schedule(static))
#define OMP_FOR_IF(c) OMP_PRAGMA(omp parallel for schedule(static) if (c))
#else
#define OMP_FOR
#define OMP_FOR_IF(c)
#endif

static double* dnew(size_t n) {
    double* p = (double*)calloc(n ? n : 1, sizeof(double));
    if (!p) abort();
    return p;
}

int orc_cov_num_hyp(int cov, int d) { return cov == ORC_COV_SE_ARD ? d + 1 : 2; }
static int sf_index(int cov, int d) { return cov == ORC_COV_SE_ARD ? d : 1; }

/* util.cpp:64-75: dists.col(i) = sdist_vm(Y.col(i), X) = (X.colwise() - y_i).colwise().squaredNorm() */
void orc_sdist_mm(int d, const double* X, long p, const double* Y, long q, double* D) {
    OMP_FOR
    for (long b = 0; b < q; ++b) {
        const double* yb = Y + (size_t)b * d;
        double* col = D + (size_t)b * p;
        for (long a = 0; a < p; ++a) {
            const double* xa = X + (size_t)a * d;
            double s = 0.0;
            for (int k = 0; k < d; ++k) {
                const double t = xa[k] - yb[k];
                s += t * t;
            }
            col[a] = s;
        }
    }
}

/* scaled copies as in CovSEard.cpp:9-10 (multiply by exp(-hyp_k)) / CovSEiso.cpp:8-10 (divide by l) */
static double* scaled_inputs(int cov, int d, const double* hyp, const double* x, long p) {
    double* s = dnew((size_t)p * d);
    if (cov == ORC_COV_SE_ARD) {
        double* il = dnew(d);
        for (int k = 0; k < d; ++k) il[k] = exp(-1 * hyp[k]);
        for (long a = 0; a < p; ++a)
            for (int k = 0; k < d; ++k) s[(size_t)a * d + k] = il[k] * x[(size_t)a * d + k];
        free(il);
    } else {
        const double l = exp(hyp[0]);
        for (size_t e = 0; e < (size_t)p * d; ++e) s[e] = x[e] / l;
    }
    return s;
}

/* CovSEard.cpp:7-11 ; CovSEiso.cpp:6-11 */
void orc_cov_k(int cov, int d, const double* hyp, const double* x1, long p, const double* x2, long q,
               double* K) {
    double* s1 = scaled_inputs(cov, d, hyp, x1, p);
    double* s2 = scaled_inputs(cov, d, hyp, x2, q);
    orc_sdist_mm(d, s1, p, s2, q, K);
    const double sf2 = exp(2 * hyp[sf_index(cov, d)]);
    OMP_FOR
    for (size_t e = 0; e < (size_t)p * q; ++e) K[e] = sf2 * exp(-0.5 * K[e]);
    free(s1);
    free(s2);
}

/* CovSEard.cpp:12-22 ; CovSEiso.cpp:12-28 */
void orc_cov_dk_dhyp(int cov, int d, const double* hyp, int idx, const double* x1, long p,
                     const double* x2, long q, const double* K, double* dK) {
    if (cov == ORC_COV_SE_ARD) {
        if (idx < d) {
            const double l_inv = exp(-1 * hyp[idx]);
            OMP_FOR
            for (long b = 0; b < q; ++b) {
                const double xb = l_inv * x2[(size_t)b * d + idx];
                for (long a = 0; a < p; ++a) {
                    const double t = l_inv * x1[(size_t)a * d + idx] - xb;
                    dK[(size_t)b * p + a] = K[(size_t)b * p + a] * (t * t);
                }
            }
        } else {
            for (size_t e = 0; e < (size_t)p * q; ++e) dK[e] = 2 * K[e];
        }
    } else {
        if (idx == 0) {
            double* s1 = scaled_inputs(cov, d, hyp, x1, p);
            double* s2 = scaled_inputs(cov, d, hyp, x2, q);
            orc_sdist_mm(d, s1, p, s2, q, dK);
            for (size_t e = 0; e < (size_t)p * q; ++e) dK[e] = K[e] * dK[e];
            free(s1);
            free(s2);
        } else {
            for (size_t e = 0; e < (size_t)p * q; ++e) dK[e] = 2 * K[e];
        }
    }
}

/* CovSEard.cpp:28-40 ; CovSEiso.cpp:34-46 */
void orc_cov_dk_dx1(int cov, int d, const double* hyp, const double* x1, const double* x2, long q,
                    double* dK) {
    double* K = dnew(q);
    orc_cov_k(cov, d, hyp, x1, 1, x2, q, K); /* 1 x q */
    for (long b = 0; b < q; ++b)
        for (int k = 0; k < d; ++k) {
            const double il2 = exp(-2 * hyp[cov == ORC_COV_SE_ARD ? k : 0]);
            dK[(size_t)b * d + k] = K[b] * (il2 * (x2[(size_t)b * d + k] - x1[k]));
        }
    free(K);
}

/* util.h:32-41 */
double orc_stddev(const double* v, long n, long stride) {
    double s = 0;
    for (long i = 0; i < n; ++i) s += v[i * stride];
    const double mean = s / (double)n;
    double ss = 0;
    for (long i = 0; i < n; ++i) {
        const double t = v[i * stride] - mean;
        ss += t * t;
    }
    return sqrt(ss / (double)(n - 1));
}

/* GP.cpp:85-92 ; CovSEard.cpp:72-79 ; CovSEiso.cpp:79-85 */
void orc_default_hyps(int cov, int d, long n, const double* X, const double* y, double noise_lb,
                      double* hyp) {
    const int nk = orc_cov_num_hyp(cov, d);
    if (cov == ORC_COV_SE_ARD) {
        for (int i = 0; i < d; ++i) hyp[i] = log(orc_stddev(X + i, n, d));
        hyp[d] = log(orc_stddev(y, n, 1));
    } else {
        hyp[0] = 0;
        hyp[1] = log(orc_stddev(y, n, 1));
    }
    const double a = log(noise_lb), b = log(orc_stddev(y, n, 1) * 1e-3);
    hyp[nk] = a > b ? a : b;
    double s = 0;
    for (long i = 0; i < n; ++i) s += y[i];
    hyp[nk + 1] = s / (double)n;
}

/* GP.cpp:514-534 ; CovSEard.cpp:41-71 ; CovSEiso.cpp:47-78 */
void orc_hyp_range(int cov, int d, long n, const double* X, const double* y, double noise_lb,
                   double* lb, double* ub) {
    const int nk = orc_cov_num_hyp(cov, d), nh = nk + 2;
    const double big = 0.5 * log(0.5 * DBL_MAX);
    for (int i = 0; i < nh; ++i) {
        lb[i] = -INFINITY;
        ub[i] = big;
    }
    double ymin = y[0], ymax = y[0];
    for (long i = 1; i < n; ++i) {
        if (y[i] < ymin) ymin = y[i];
        if (y[i] > ymax) ymax = y[i];
    }
    for (int i = 0; i < d; ++i) {
        double xmin = X[i], xmax = X[i];
        for (long a = 1; a < n; ++a) {
            const double v = X[(size_t)a * d + i];
            if (v < xmin) xmin = v;
            if (v > xmax) xmax = v;
        }
        const double distance = xmax - xmin;
        const double magic_num = 0.05 * distance;
        const double thres = 1e-4;
        const double ub1 = 0.5 * log(0.05 * DBL_MAX);
        const double ub2 = log(distance / sqrt(-2 * log(1 - thres)));
        const double lscale_lb = log(magic_num) - 0.5 * log(-2 * log(1.5 * DBL_MIN));
        const double lscale_ub = ub1 < ub2 ? ub1 : ub2;
        if (cov == ORC_COV_SE_ARD) {
            lb[i] = lscale_lb;
            ub[i] = lscale_ub;
        } else {
            if (lscale_lb > lb[0]) lb[0] = lscale_lb;
            if (lscale_ub < ub[0]) ub[0] = lscale_ub;
        }
    }
    if (cov == ORC_COV_SE_ARD) {
        const double e = DBL_EPSILON, r = ymax - ymin;
        lb[d] = log(e > e * r ? e : e * r);
        ub[d] = log(10 * e > 10 * r ? 10 * e : 10 * r);
    } else {
        const double e = DBL_EPSILON, r = ymax - ymin;
        lb[1] = log(0.0 > e * r ? 0.0 : e * r);
        ub[1] = log(10 * r);
    }
    lb[nh - 2] = log(noise_lb);
    {
        const double a = log(10 * noise_lb), b = ub[nk - 1];
        ub[nh - 2] = a > b ? a : b;
    }
    lb[nh - 1] = ymin;
    ub[nh - 1] = ymax;
    for (int i = 0; i < nh; ++i) {
        lb[i] -= DBL_EPSILON;
        ub[i] += DBL_EPSILON;
    }
}

/* ---- MatrixSolverLLT (MatrixSolver.cpp:6-29) ---------------------------------------------- */

/* decomp + check_SPD: Eigen::LLT reports NumericalIssue on the first pivot <= 0 */
int orc_chol(double* A, long n) {
    for (long j = 0; j < n; ++j) {
        double* cj = A + (size_t)j * n;
        const double piv = cj[j];
        if (piv <= 0.0) return (int)(j + 1);
        const double ljj = sqrt(piv);
        cj[j] = ljj;
        for (long i = j + 1; i < n; ++i) cj[i] /= ljj;
        OMP_FOR_IF(n - j > 128)
        for (long k = j + 1; k < n; ++k) {
            const double lkj = cj[k];
            double* ck = A + (size_t)k * n;
            for (long i = k; i < n; ++i) ck[i] -= lkj * cj[i];
        }
    }
    return 0;
}

/* solve: L L^T x = b by forward then backward substitution, one right-hand side at a time */
void orc_chol_solve(const double* L, long n, double* B, long nrhs) {
    OMP_FOR_IF(nrhs > 1)
    for (long r = 0; r < nrhs; ++r) {
        double* b = B + (size_t)r * n;
        for (long j = 0; j < n; ++j) {
            const double* cj = L + (size_t)j * n;
            const double bj = b[j] / cj[j];
            b[j] = bj;
            for (long i = j + 1; i < n; ++i) b[i] -= bj * cj[i];
        }
        for (long j = n - 1; j >= 0; --j) {
            const double* cj = L + (size_t)j * n;
            double s = b[j];
            for (long i = j + 1; i < n; ++i) s -= cj[i] * b[i];
            b[j] = s / cj[j];
        }
    }
}

/* log_det: 2 * sum log L_ii  (MatrixSolver.cpp:21-24) */
double orc_chol_logdet(const double* L, long n) {
    double s = 0;
    for (long i = 0; i < n; ++i) s += log(L[(size_t)i * n + i]);
    return 2 * s;
}

/* inverse: solve(Identity)  (MatrixSolver.cpp:25-28) */
void orc_chol_inverse(const double* L, long n, double* Inv) {
    memset(Inv, 0, sizeof(double) * (size_t)n * n);
    for (long i = 0; i < n; ++i) Inv[(size_t)i * n + i] = 1.0;
    orc_chol_solve(L, n, Inv, n);
}

static int all_finite(const double* v, long n) {
    for (long i = 0; i < n; ++i)
        if (!isfinite(v[i])) return 0;
    return 1;
}

/* ---- exact GP (GP.cpp:120-176) ------------------------------------------------------------ */
double orc_exact_nll(int cov, int d, long n, const double* X, const double* y, const double* hyp,
                     int calc_grad, double* g) {
    const int nk = orc_cov_num_hyp(cov, d), nh = nk + 2;
    const double sn2 = exp(2 * hyp[nh - 2]);
    const double mean = hyp[nh - 1];
    double* K = dnew((size_t)n * n);
    orc_cov_k(cov, d, hyp, X, n, X, n, K);
    double* L = dnew((size_t)n * n);
    memcpy(L, K, sizeof(double) * (size_t)n * n);
    for (long i = 0; i < n; ++i) L[(size_t)i * n + i] += sn2;
    const int info = orc_chol(L, n);

    double nlp = INFINITY;
    if (calc_grad)
        for (int i = 0; i < nh; ++i) g[i] = INFINITY;
    if (info == 0) {
        double* ty = dnew(n);
        double* alpha = dnew(n);
        for (long i = 0; i < n; ++i) alpha[i] = ty[i] = y[i] - mean;
        orc_chol_solve(L, n, alpha, 1);
        double dot = 0;
        for (long i = 0; i < n; ++i) dot += ty[i] * alpha[i];
        const double data_fit = 0.5 * dot;
        const double model_complexity = 0.5 * orc_chol_logdet(L, n);
        const double norm_const = 0.5 * (double)n * log(2 * M_PI);
        nlp = data_fit + model_complexity + norm_const;
        if (!isfinite(nlp)) {
            nlp = INFINITY;
        } else if (calc_grad) {
            double* Q = dnew((size_t)n * n);
            double* dK = dnew((size_t)n * n);
            orc_chol_inverse(L, n, Q);
            for (long b = 0; b < n; ++b)
                for (long a = 0; a < n; ++a) Q[(size_t)b * n + a] -= alpha[a] * alpha[b];
            for (int i = 0; i < nk; ++i) {
                orc_cov_dk_dhyp(cov, d, hyp, i, X, n, X, n, K, dK);
                double s = 0;
                for (size_t e = 0; e < (size_t)n * n; ++e) s += Q[e] * dK[e];
                g[i] = 0.5 * s;
            }
            double tr = 0, sa = 0;
            for (long i = 0; i < n; ++i) {
                tr += Q[(size_t)i * n + i];
                sa += alpha[i];
            }
            g[nh - 2] = sn2 * tr;
            g[nh - 1] = -1 * sa;
            if (!all_finite(g, nh)) {
                nlp = INFINITY;
                for (int i = 0; i < nh; ++i) g[i] = INFINITY;
            }
            free(Q);
            free(dK);
        }
        free(ty);
        free(alpha);
    }
    free(K);
    free(L);
    return nlp;
}

/* ---- exact GP _setK / _check_SPD / predict (GP.cpp:273-334, 423-461) ---------------------- */
struct orc_exact_state {
    int cov, d;
    long n;
    double* X;
    double* hyp;
    double* L;      /* factor of K + sn2 I */
    double* invKys; /* GP.cpp:443 */
};

/* eigenvalues of a symmetric matrix by cyclic Jacobi (stand-in for
 * K.selfadjointView<Lower>().eigenvalues(), GP.cpp:451); small n only */
static void sym_eigenvalues(const double* A, long n, double* w) {
    double* M = dnew((size_t)n * n);
    for (long j = 0; j < n; ++j)
        for (long i = 0; i < n; ++i) M[(size_t)j * n + i] = i >= j ? A[(size_t)j * n + i] : A[(size_t)i * n + j];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (long j = 0; j < n; ++j)
            for (long i = 0; i < n; ++i) {
                if (i == j) diag += M[(size_t)j * n + i] * M[(size_t)j * n + i];
                else off += M[(size_t)j * n + i] * M[(size_t)j * n + i];
            }
        if (off <= 1e-30 * diag) break;
        for (long p = 0; p < n - 1; ++p)
            for (long q = p + 1; q < n; ++q) {
                const double apq = M[(size_t)q * n + p];
                if (apq == 0.0) continue;
                const double app = M[(size_t)p * n + p], aqq = M[(size_t)q * n + q];
                const double theta = (aqq - app) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (long k = 0; k < n; ++k) { /* columns p,q */
                    const double akp = M[(size_t)p * n + k], akq = M[(size_t)q * n + k];
                    M[(size_t)p * n + k] = c * akp - s * akq;
                    M[(size_t)q * n + k] = s * akp + c * akq;
                }
                for (long k = 0; k < n; ++k) { /* rows p,q */
                    const double apk = M[(size_t)k * n + p], aqk = M[(size_t)k * n + q];
                    M[(size_t)k * n + p] = c * apk - s * aqk;
                    M[(size_t)k * n + q] = s * apk + c * aqk;
                }
            }
    }
    for (long i = 0; i < n; ++i) w[i] = M[(size_t)i * n + i];
    free(M);
}

/* GP.cpp:445-461.  Leaves the factor of K in L (as the reference leaves it in _matrix_solver). */
static int check_spd(const double* K, long n, double* L, int literal) {
    memcpy(L, K, sizeof(double) * (size_t)n * n);
    const int info = orc_chol(L, n);
    if (!literal) return info == 0;
    int ok = info == 0;
    double* w = dnew(n);
    sym_eigenvalues(K, n, w);
    for (long i = 0; i < n; ++i)
        if (!(w[i] >= 0)) ok = 0;
    free(w);
    if (info == 0) {
        double* Inv = dnew((size_t)n * n);
        orc_chol_inverse(L, n, Inv);
        double err = 0;
        for (long j = 0; j < n; ++j)
            for (long i = 0; i < n; ++i) {
                double s = 0;
                for (long k = 0; k < n; ++k) s += K[(size_t)k * n + i] * Inv[(size_t)j * n + k];
                err += fabs((i == j ? 1.0 : 0.0) - s);
            }
        err /= (double)n * (double)n;
        if (!(err < 1e-4)) ok = 0;
        free(Inv);
    }
    return ok;
}

orc_exact_state* orc_exact_setk(int cov, int d, long n, const double* X, const double* y,
                                double* hyp, int literal) {
    const int nk = orc_cov_num_hyp(cov, d), nh = nk + 2;
    orc_exact_state* st = (orc_exact_state*)calloc(1, sizeof(*st));
    st->cov = cov; st->d = d; st->n = n;
    st->X = dnew((size_t)n * d);
    memcpy(st->X, X, sizeof(double) * (size_t)n * d);
    double* Kcov = dnew((size_t)n * n);
    double* K = dnew((size_t)n * n);
    st->L = dnew((size_t)n * n);
    orc_cov_k(cov, d, hyp, X, n, X, n, Kcov);
    double sn2 = exp(2 * hyp[nh - 2]);
    memcpy(K, Kcov, sizeof(double) * (size_t)n * n);
    for (long i = 0; i < n; ++i) K[(size_t)i * n + i] = sn2 + Kcov[(size_t)i * n + i];
    int spd = check_spd(K, n, st->L, literal);
    while (!spd) { /* GP.cpp:431-440 */
        hyp[nh - 2] = isinf(hyp[nh - 2]) ? log(DBL_EPSILON) : hyp[nh - 2] + log(sqrt(10));
        sn2 = exp(2 * hyp[nh - 2]);
        for (long i = 0; i < n; ++i) K[(size_t)i * n + i] = sn2 + Kcov[(size_t)i * n + i];
        spd = check_spd(K, n, st->L, literal);
    }
    memcpy(st->L, K, sizeof(double) * (size_t)n * n);
    orc_chol(st->L, n);
    st->invKys = dnew(n);
    for (long i = 0; i < n; ++i) st->invKys[i] = y[i] - hyp[nh - 1];
    orc_chol_solve(st->L, n, st->invKys, 1);
    st->hyp = dnew(nh);
    memcpy(st->hyp, hyp, sizeof(double) * nh);
    free(Kcov);
    free(K);
    return st;
}

void orc_exact_free(orc_exact_state* st) {
    if (!st) return;
    free(st->X); free(st->hyp); free(st->L); free(st->invKys); free(st);
}

/* GP.cpp:273-334 */
void orc_exact_predict(const orc_exact_state* st, const double* Xs, long t, double* mu, double* s2,
                       double* gy, double* gs2) {
    const int cov = st->cov, d = st->d;
    const long n = st->n;
    const int nk = orc_cov_num_hyp(cov, d), nh = nk + 2;
    const double sf2 = exp(2 * st->hyp[sf_index(cov, d)]);
    const double sn2 = exp(2 * st->hyp[nh - 2]);
    const double mean = st->hyp[nh - 1];
    double* k_test = dnew((size_t)t * n); /* t x n */
    orc_cov_k(cov, d, st->hyp, Xs, t, st->X, n, k_test);
    double* kks = NULL; /* n x t = solve(k_test^T) */
    if (s2 || gs2) {
        kks = dnew((size_t)n * t);
        for (long i = 0; i < t; ++i)
            for (long a = 0; a < n; ++a) kks[(size_t)i * n + a] = k_test[(size_t)a * t + i];
        orc_chol_solve(st->L, n, kks, t);
    }
    for (long i = 0; i < t; ++i) {
        if (mu) {
            double s = 0;
            for (long a = 0; a < n; ++a) s += k_test[(size_t)a * t + i] * st->invKys[a];
            mu[i] = mean + s;
        }
        if (s2) {
            double s = 0;
            for (long a = 0; a < n; ++a) s += k_test[(size_t)a * t + i] * kks[(size_t)i * n + a];
            double v = sf2 - s;
            if (v < 0) v = 0;
            s2[i] = v + sn2;
        }
    }
    if (gy || gs2) {
        double* gk = dnew((size_t)d * n);
        for (long i = 0; i < t; ++i) {
            orc_cov_dk_dx1(cov, d, st->hyp, Xs + (size_t)i * d, st->X, n, gk);
            for (int k = 0; k < d; ++k) {
                double a1 = 0, a2 = 0;
                for (long a = 0; a < n; ++a) {
                    a1 += gk[(size_t)a * d + k] * st->invKys[a];
                    if (gs2) a2 += gk[(size_t)a * d + k] * kks[(size_t)i * n + a];
                }
                if (gy) gy[(size_t)i * d + k] = a1;
                if (gs2) gs2[(size_t)i * d + k] = 0.0 - 2 * a2; /* diag_dk_dx1 == 0 */
            }
        }
        free(gk);
    }
    free(k_test);
    free(kks);
}

/* ---- small dense helpers for the sparse models -------------------------------------------- */
/* C[m x n] = A[m x k] * B[k x n] */
static void gemm_nn(long m, long n, long k, const double* A, const double* B, double* C) {
    memset(C, 0, sizeof(double) * (size_t)m * n);
    OMP_FOR_IF(n > 1)
    for (long j = 0; j < n; ++j)
        for (long l = 0; l < k; ++l) {
            const double b = B[(size_t)j * k + l];
            const double* a = A + (size_t)l * m;
            double* c = C + (size_t)j * m;
            for (long i = 0; i < m; ++i) c[i] += a[i] * b;
        }
}
static void transpose(long m, long n, const double* A, double* At) { /* A m x n -> At n x m */
    for (long j = 0; j < n; ++j)
        for (long i = 0; i < m; ++i) At[(size_t)i * n + j] = A[(size_t)j * m + i];
}
static double dotn(long n, const double* a, const double* b) {
    double s = 0;
    for (long i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
static double sum_prod(size_t n, const double* a, const double* b) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* diag_dk_dhyp(hyp, X).transpose().col(i): CovSEard.cpp:85-96.  For SE-iso the reference writes
 * row _dim (CovSEiso.cpp:95,102 — out of range for d >= 4); the oracle writes the sf row (1). */
static double dkn_diag_value(int cov, int d, const double* hyp, int i) {
    return i == sf_index(cov, d) ? 2 * exp(2 * hyp[sf_index(cov, d)]) : 0.0;
}

/* ---- FITC objective (FITC.cpp:201-319) ----------------------------------------------------- */
double orc_fitc_nll(int cov, int d, long n, const double* X, const double* y, const double* U, long m,
                    double jitter, const double* hyp, int calc_grad, double* g) {
    const int nk = orc_cov_num_hyp(cov, d), nh = nk + 2;
    const double sn2 = exp(2 * hyp[nh - 2]);
    const double sf2 = exp(2 * hyp[sf_index(cov, d)]);
    double* Kuu = dnew((size_t)m * m);
    double* Kxu = dnew((size_t)n * m);
    double* Kux = dnew((size_t)m * n);
    double* ty = dnew(n);
    orc_cov_k(cov, d, hyp, U, m, U, m, Kuu);
    for (long i = 0; i < m; ++i) Kuu[(size_t)i * m + i] += jitter;
    orc_cov_k(cov, d, hyp, X, n, U, m, Kxu);
    transpose(n, m, Kxu, Kux);
    for (long i = 0; i < n; ++i) ty[i] = y[i] - hyp[nh - 1];

    double* Lu = dnew((size_t)m * m);
    memcpy(Lu, Kuu, sizeof(double) * (size_t)m * m);
    const int info_u = orc_chol(Lu, m);
    double nlz = INFINITY;
    if (info_u != 0) {
        if (calc_grad) memset(g, 0, sizeof(double) * nh);
        free(Kuu); free(Kxu); free(Kux); free(ty); free(Lu);
        return nlz;
    }
    double* KuuinvKux = dnew((size_t)m * n);
    memcpy(KuuinvKux, Kux, sizeof(double) * (size_t)m * n);
    orc_chol_solve(Lu, m, KuuinvKux, n);
    double* Gamma = dnew(n);
    double* invG = dnew(n);
    for (long a = 0; a < n; ++a) {
        double s = 0;
        for (long b = 0; b < m; ++b) s += Kxu[(size_t)b * n + a] * KuuinvKux[(size_t)a * m + b];
        Gamma[a] = (sn2 + sf2 - s) / sn2;
        invG[a] = 1.0 / Gamma[a];
    }
    /* A = sn2*Kuu + Kux * invG.asDiagonal() * Kxu   (FITC.cpp:217) */
    double* KuxG = dnew((size_t)m * n); /* Kux * diag(invG) */
    for (long a = 0; a < n; ++a)
        for (long b = 0; b < m; ++b) KuxG[(size_t)a * m + b] = Kux[(size_t)a * m + b] * invG[a];
    double* A = dnew((size_t)m * m);
    gemm_nn(m, m, n, KuxG, Kxu, A);
    for (size_t e = 0; e < (size_t)m * m; ++e) A[e] = sn2 * Kuu[e] + A[e];
    double* LA = dnew((size_t)m * m);
    memcpy(LA, A, sizeof(double) * (size_t)m * m);
    const int info_A = orc_chol(LA, m);

    double* l2_tmp1 = dnew(m);
    double* l2_tmp4 = dnew(n);
    double data_fit = NAN, model_complexity = NAN;
    if (info_A == 0) {
        double* igy = dnew(n);
        for (long a = 0; a < n; ++a) igy[a] = invG[a] * ty[a];
        gemm_nn(m, 1, n, Kux, igy, l2_tmp1);
        double* sol = dnew(m);
        memcpy(sol, l2_tmp1, sizeof(double) * m);
        orc_chol_solve(LA, m, sol, 1);
        double* Ks = dnew(n);
        gemm_nn(n, 1, m, Kxu, sol, Ks);
        for (long a = 0; a < n; ++a) l2_tmp4[a] = invG[a] * (ty[a] - Ks[a]);
        data_fit = dotn(n, ty, l2_tmp4) / sn2;
        double slg = 0;
        for (long a = 0; a < n; ++a) slg += log(Gamma[a]);
        model_complexity = orc_chol_logdet(LA, m) - orc_chol_logdet(Lu, m) + slg + (double)(n - m) * log(sn2);
        nlz = 0.5 * (data_fit + model_complexity + (double)n * log(2 * M_PI));
        free(igy); free(sol); free(Ks);
    }
    if (!(info_A == 0 && isfinite(data_fit) && isfinite(model_complexity))) {
        nlz = INFINITY;
        if (calc_grad) memset(g, 0, sizeof(double) * nh);
    }
    if (calc_grad && isfinite(nlz)) {
        double* Kuu_inv = dnew((size_t)m * m);
        double* A_inv = dnew((size_t)m * m);
        orc_chol_inverse(Lu, m, Kuu_inv);
        orc_chol_inverse(LA, m, A_inv);
        memset(g, 0, sizeof(double) * nh);
        double* iGK = dnew((size_t)n * m); /* inv_Gamma_Kxu = (Kux*diag(invG))^T */
        transpose(m, n, KuxG, iGK);
        double* KxuKuuinv = dnew((size_t)n * m);
        transpose(m, n, KuuinvKux, KxuKuuinv);
        double* igy = dnew(n);
        for (long a = 0; a < n; ++a) igy[a] = invG[a] * ty[a];
        double* F = dnew(m);
        gemm_nn(m, 1, n, Kux, igy, F);
        double* AinvF = dnew(m);
        memcpy(AinvF, F, sizeof(double) * m);
        orc_chol_solve(LA, m, AinvF, 1);
        double* iGKAinv = dnew((size_t)n * m);
        gemm_nn(n, m, m, iGK, A_inv, iGKAinv);
        double* c1 = dnew(n); /* (iGKAinv * Kux_invGamma).diagonal() */
        for (long a = 0; a < n; ++a) {
            double s = 0;
            for (long b = 0; b < m; ++b) s += iGKAinv[(size_t)b * n + a] * KuxG[(size_t)a * m + b];
            c1[a] = s;
        }
        double* iGKAinvF = dnew(n);
        gemm_nn(n, 1, m, iGKAinv, F, iGKAinvF);
        double* gg = dnew(n);
        double* ggy = dnew(n);
        for (long a = 0; a < n; ++a) {
            gg[a] = invG[a] * invG[a];
            ggy[a] = invG[a] * igy[a];
        }
        double* KxuAinvF = dnew(n);
        gemm_nn(n, 1, m, Kxu, AinvF, KxuAinvF);

        double* dKuu = dnew((size_t)m * m);
        double* dKxu = dnew((size_t)n * m);
        double* dKux = dnew((size_t)m * n);
        double* P = dnew((size_t)n * m); /* Kxu_Kuu_inv * dKuu */
        double* dGamma = dnew(n);
        double* t1 = dnew(n);
        double* t2 = dnew(n);
        double* dF = dnew(m);
        double* tmpm = dnew(m);
        double* tmpm2 = dnew(m);
        double* v_tmp = dnew(m);
        double* tn = dnew(n);
        double* iGKAinvT = dnew((size_t)m * n);
        transpose(n, m, iGKAinv, iGKAinvT);
        for (int i = 0; i < nk; ++i) {
            orc_cov_dk_dhyp(cov, d, hyp, i, U, m, U, m, Kuu, dKuu);
            orc_cov_dk_dhyp(cov, d, hyp, i, X, n, U, m, Kxu, dKxu);
            transpose(n, m, dKxu, dKux);
            /* FITC.cpp:276 — the reference's flagged O(N m^2) bottleneck */
            gemm_nn(n, m, m, KxuKuuinv, dKuu, P);
            const double dkn = dkn_diag_value(cov, d, hyp, i);
            for (long a = 0; a < n; ++a) {
                double s = 0;
                for (long b = 0; b < m; ++b)
                    s += KxuKuuinv[(size_t)b * n + a] * (2 * dKxu[(size_t)b * n + a] - P[(size_t)b * n + a]);
                dGamma[a] = (dkn - s) / sn2;
            }
            const double g_logKuu = sum_prod((size_t)m * m, dKuu, Kuu_inv);
            const double g_logA = sn2 * sum_prod((size_t)m * m, dKuu, A_inv) +
                                  2 * sum_prod((size_t)m * n, dKux, iGKAinvT) - dotn(n, dGamma, c1);
            const double g_logGamma = dotn(n, dGamma, invG);
            const double g_mc = g_logA - g_logKuu + g_logGamma;
            for (long a = 0; a < n; ++a) {
                t1[a] = gg[a] * dGamma[a];
                t2[a] = ggy[a] * dGamma[a];
            }
            gemm_nn(m, 1, n, dKux, igy, tmpm);
            gemm_nn(m, 1, n, Kux, t2, tmpm2);
            for (long b = 0; b < m; ++b) dF[b] = tmpm[b] - tmpm2[b];
            /* v_tmp = sn2*dKuu*AinvF + 2*dKux*iGKAinvF - Kux*(t1 .* KxuAinvF) */
            gemm_nn(m, 1, m, dKuu, AinvF, v_tmp);
            gemm_nn(m, 1, n, dKux, iGKAinvF, tmpm);
            for (long a = 0; a < n; ++a) tn[a] = t1[a] * KxuAinvF[a];
            gemm_nn(m, 1, n, Kux, tn, tmpm2);
            for (long b = 0; b < m; ++b) v_tmp[b] = sn2 * v_tmp[b] + 2 * tmpm[b] - tmpm2[b];
            const double g_datafit_1 = -1 * dotn(n, ty, t2);
            gemm_nn(m, 1, m, A_inv, dF, tmpm);
            const double g_datafit_2_1 = 2 * dotn(m, F, tmpm);
            const double g_datafit_2_2 = dotn(m, AinvF, v_tmp);
            const double g_datafit = (g_datafit_1 - (g_datafit_2_1 - g_datafit_2_2)) / sn2;
            g[i] = 0.5 * (g_mc + g_datafit);
        }
        /* noise and mean, FITC.cpp:311-314 */
        double s_ig = 0, s_c1 = 0, sq = 0, sl = 0;
        for (long a = 0; a < n; ++a) {
            s_ig += invG[a];
            s_c1 += c1[a];
            sq += l2_tmp4[a] * l2_tmp4[a];
            sl += l2_tmp4[a] / sn2;
        }
        const double dl_noise_1 = 0.5 * (s_ig - s_c1) / sn2;
        const double dl_noise_2 = -0.5 * sq / pow(sn2, 2);
        g[nh - 2] = (dl_noise_1 + dl_noise_2) * (2 * sn2);
        g[nh - 1] = -1 * sl;
        free(Kuu_inv); free(A_inv); free(iGK); free(KxuKuuinv); free(igy); free(F); free(AinvF);
        free(iGKAinv); free(c1); free(iGKAinvF); free(gg); free(ggy); free(KxuAinvF); free(dKuu);
        free(dKxu); free(dKux); free(P); free(dGamma); free(t1); free(t2); free(dF); free(tmpm);
        free(tmpm2); free(v_tmp); free(tn); free(iGKAinvT);
    }
    free(Kuu); free(Kxu); free(Kux); free(ty); free(Lu); free(KuuinvKux); free(Gamma); free(invG);
    free(KuxG); free(A); free(LA); free(l2_tmp1); free(l2_tmp4);
    return nlz;
}

/* ---- VFE objective (VFE.cpp:165-249) ------------------------------------------------------- */
double orc_vfe_nll(int cov, int d, long n, const double* X, const double* yin, const double* U, long m,
                   double jitter, const double* hyp, int calc_grad, double* g) {
    const int nk = orc_cov_num_hyp(cov, d), nh = nk + 2;
    const double sn2 = exp(2 * hyp[nh - 2]);
    const double mean = hyp[nh - 1];
    const double sf2 = exp(2 * hyp[sf_index(cov, d)]);
    double* y = dnew(n);
    for (long i = 0; i < n; ++i) y[i] = yin[i] - mean;
    double* Kuu = dnew((size_t)m * m);
    double* Kxu = dnew((size_t)n * m);
    double* Kux = dnew((size_t)m * n);
    orc_cov_k(cov, d, hyp, U, m, U, m, Kuu);
    for (long i = 0; i < m; ++i) Kuu[(size_t)i * m + i] += jitter;
    orc_cov_k(cov, d, hyp, X, n, U, m, Kxu);
    transpose(n, m, Kxu, Kux);
    double* Kuxxu = dnew((size_t)m * m);
    gemm_nn(m, m, n, Kux, Kxu, Kuxxu);
    double* A = dnew((size_t)m * m);
    for (size_t e = 0; e < (size_t)m * m; ++e) A[e] = sn2 * Kuu[e] + Kuxxu[e];
    double* Lu = dnew((size_t)m * m);
    double* LA = dnew((size_t)m * m);
    memcpy(Lu, Kuu, sizeof(double) * (size_t)m * m);
    memcpy(LA, A, sizeof(double) * (size_t)m * m);
    const int info_u = orc_chol(Lu, m);
    const int info_A = orc_chol(LA, m);
    double nlz = INFINITY;
    if (info_u != 0 || info_A != 0) {
        if (calc_grad) memset(g, 0, sizeof(double) * nh);
        free(y); free(Kuu); free(Kxu); free(Kux); free(Kuxxu); free(A); free(Lu); free(LA);
        return nlz;
    }
    double* Kuu_inv = dnew((size_t)m * m);
    double* A_inv = dnew((size_t)m * m);
    orc_chol_inverse(Lu, m, Kuu_inv);
    orc_chol_inverse(LA, m, A_inv);
    double* Kuxy = dnew(m);
    gemm_nn(m, 1, n, Kux, y, Kuxy);
    double* sol = dnew(m);
    memcpy(sol, Kuxy, sizeof(double) * m);
    orc_chol_solve(LA, m, sol, 1);
    double* alpha = dnew(n);
    gemm_nn(n, 1, m, Kxu, sol, alpha);
    for (long a = 0; a < n; ++a) alpha[a] = (y[a] - alpha[a]) / sn2;

    const double f0 = 0.5 * (double)n * log(2 * M_PI);
    const double f_mc = 0.5 * (orc_chol_logdet(LA, m) - orc_chol_logdet(Lu, m) + (double)(n - m) * log(sn2));
    const double f_df = 0.5 * dotn(n, y, alpha);
    double tr = 0; /* (Kuu_inv * Kuxxu).trace() */
    for (long i = 0; i < m; ++i)
        for (long k = 0; k < m; ++k) tr += Kuu_inv[(size_t)k * m + i] * Kuxxu[(size_t)i * m + k];
    const double f_tr = 0.5 * ((double)n * sf2 - tr) / sn2;
    nlz = f0 + f_mc + f_df + f_tr;
    if (!isfinite(nlz)) {
        nlz = INFINITY;
        if (calc_grad) memset(g, 0, sizeof(double) * nh);
    }
    if (calc_grad && isfinite(nlz)) {
        memset(g, 0, sizeof(double) * nh);
        double* AinvKux = dnew((size_t)m * n);
        gemm_nn(m, n, m, A_inv, Kux, AinvKux);
        double* AinvF = dnew(m);
        gemm_nn(m, 1, n, AinvKux, y, AinvF);
        double* M1 = dnew((size_t)m * m); /* AinvF AinvF^T */
        for (long j = 0; j < m; ++j)
            for (long i = 0; i < m; ++i) M1[(size_t)j * m + i] = AinvF[i] * AinvF[j];
        double* KM1 = dnew((size_t)n * m);
        gemm_nn(n, m, m, Kxu, M1, KM1);
        double* KM1T = dnew((size_t)m * n);
        transpose(n, m, KM1, KM1T);
        double* AinvFyt = dnew((size_t)m * n);
        for (long a = 0; a < n; ++a)
            for (long b = 0; b < m; ++b) AinvFyt[(size_t)a * m + b] = AinvF[b] * y[a];
        double* KuuinvKux = dnew((size_t)m * n);
        gemm_nn(m, n, m, Kuu_inv, Kux, KuuinvKux);
        double* KuuinvKuxT = dnew((size_t)n * m);
        transpose(m, n, KuuinvKux, KuuinvKuxT);
        double* KKKK = dnew((size_t)m * m);
        gemm_nn(m, m, n, KuuinvKux, KuuinvKuxT, KKKK);
        double* dKuu = dnew((size_t)m * m);
        double* dKxu = dnew((size_t)n * m);
        double* dKux = dnew((size_t)m * n);
        for (int i = 0; i < nk; ++i) {
            orc_cov_dk_dhyp(cov, d, hyp, i, U, m, U, m, Kuu, dKuu);
            orc_cov_dk_dhyp(cov, d, hyp, i, X, n, U, m, Kxu, dKxu);
            transpose(n, m, dKxu, dKux);
            const double g_df1 = -1 * (sn2 * sum_prod((size_t)m * m, dKuu, M1) +
                                       2 * sum_prod((size_t)m * n, dKux, KM1T));
            const double g_df2 = 2 * sum_prod((size_t)m * n, dKux, AinvFyt);
            const double g_df = -0.5 * (g_df1 + g_df2) / sn2;
            const double d_logA = sn2 * sum_prod((size_t)m * m, A_inv, dKuu) +
                                  2 * sum_prod((size_t)m * n, AinvKux, dKux);
            const double d_logKuu = sum_prod((size_t)m * m, Kuu_inv, dKuu);
            const double g_mc = 0.5 * (d_logA - d_logKuu);
            const double d_diag_Kxx = (double)n * dkn_diag_value(cov, d, hyp, i);
            const double d_diag_Q = 2 * sum_prod((size_t)n * m, dKxu, KuuinvKuxT) -
                                    sum_prod((size_t)m * m, dKuu, KKKK);
            const double g_tr = 0.5 * (d_diag_Kxx - d_diag_Q) / sn2;
            g[i] = g_df + g_mc + g_tr;
        }
        double* AinvKuxT = dnew((size_t)n * m);
        transpose(m, n, AinvKux, AinvKuxT);
        double sa = 0;
        for (long a = 0; a < n; ++a) sa += alpha[a];
        g[nh - 2] = -1 * sn2 * dotn(n, alpha, alpha) +
                    ((double)n - sum_prod((size_t)n * m, Kxu, AinvKuxT)) - 2 * f_tr;
        g[nh - 1] = -1 * sa;
        free(AinvKux); free(AinvF); free(M1); free(KM1); free(KM1T); free(AinvFyt); free(KuuinvKux);
        free(KuuinvKuxT); free(KKKK); free(dKuu); free(dKxu); free(dKux); free(AinvKuxT);
    }
    free(y); free(Kuu); free(Kxu); free(Kux); free(Kuxxu); free(A); free(Lu); free(LA);
    free(Kuu_inv); free(A_inv); free(Kuxy); free(sol); free(alpha);
    return nlz;
}

/* ---- sparse _setK / predict ---------------------------------------------------------------- */
struct orc_sparse_state {
    int cov, d;
    long m;
    double* U;
    double* hyp;
    double* Lu; /* _u_solver */
    double* LA; /* _A_solver */
    double* alpha;
};

static orc_sparse_state* sparse_new(int cov, int d, const double* U, long m, const double* hyp) {
    const int nh = orc_cov_num_hyp(cov, d) + 2;
    orc_sparse_state* st = (orc_sparse_state*)calloc(1, sizeof(*st));
    st->cov = cov; st->d = d; st->m = m;
    st->U = dnew((size_t)m * d);
    memcpy(st->U, U, sizeof(double) * (size_t)m * d);
    st->hyp = dnew(nh);
    memcpy(st->hyp, hyp, sizeof(double) * nh);
    st->Lu = dnew((size_t)m * m);
    st->LA = dnew((size_t)m * m);
    st->alpha = dnew(m);
    return st;
}
void orc_sparse_free(orc_sparse_state* st) {
    if (!st) return;
    free(st->U); free(st->hyp); free(st->Lu); free(st->LA); free(st->alpha); free(st);
}

/* FITC.cpp:165-200 */
orc_sparse_state* orc_fitc_setk(int cov, int d, long n, const double* X, const double* y,
                                const double* U, long m, double* jitter, const double* hyp) {
    const int nh = orc_cov_num_hyp(cov, d) + 2;
    orc_sparse_state* st = sparse_new(cov, d, U, m, hyp);
    const double sn2 = exp(2 * hyp[nh - 2]);
    const double sf2 = exp(2 * hyp[sf_index(cov, d)]);
    double* r = dnew(n);
    for (long i = 0; i < n; ++i) r[i] = y[i] - hyp[nh - 1];
    double* Kxu = dnew((size_t)n * m);
    double* Kux = dnew((size_t)m * n);
    double* Kuu = dnew((size_t)m * m);
    orc_cov_k(cov, d, hyp, X, n, U, m, Kxu);
    transpose(n, m, Kxu, Kux);
    orc_cov_k(cov, d, hyp, U, m, U, m, Kuu);
    double* KuuinvKux = dnew((size_t)m * n);
    double* invG = dnew(n);
    double* KuxG = dnew((size_t)m * n);
    double* A = dnew((size_t)m * m);
    int first = 1, spd = 0;
    while (!spd) {
        if (!first) *jitter *= 2;
        memcpy(st->Lu, Kuu, sizeof(double) * (size_t)m * m);
        for (long i = 0; i < m; ++i) st->Lu[(size_t)i * m + i] += *jitter;
        const int iu = orc_chol(st->Lu, m);
        int iA = 1;
        if (iu == 0) {
            memcpy(KuuinvKux, Kux, sizeof(double) * (size_t)m * n);
            orc_chol_solve(st->Lu, m, KuuinvKux, n);
            for (long a = 0; a < n; ++a) {
                double s = 0;
                for (long b = 0; b < m; ++b) s += Kxu[(size_t)b * n + a] * KuuinvKux[(size_t)a * m + b];
                invG[a] = 1.0 / ((sn2 + sf2 - s) / sn2);
            }
            for (long a = 0; a < n; ++a)
                for (long b = 0; b < m; ++b) KuxG[(size_t)a * m + b] = Kux[(size_t)a * m + b] * invG[a];
            gemm_nn(m, m, n, KuxG, Kxu, A);
            for (size_t e = 0; e < (size_t)m * m; ++e) A[e] = sn2 * Kuu[e] + A[e];
            memcpy(st->LA, A, sizeof(double) * (size_t)m * m);
            if (!first) /* FITC.cpp:196: A + jitter*I only inside the retry loop */
                for (long i = 0; i < m; ++i) st->LA[(size_t)i * m + i] += *jitter;
            iA = orc_chol(st->LA, m);
        }
        spd = (iu == 0 && iA == 0);
        first = 0;
    }
    double* ig_r = dnew(n);
    for (long a = 0; a < n; ++a) ig_r[a] = invG[a] * r[a];
    gemm_nn(m, 1, n, Kux, ig_r, st->alpha);
    orc_chol_solve(st->LA, m, st->alpha, 1);
    free(r); free(Kxu); free(Kux); free(Kuu); free(KuuinvKux); free(invG); free(KuxG); free(A); free(ig_r);
    return st;
}

/* VFE.cpp:132-160 */
orc_sparse_state* orc_vfe_setk(int cov, int d, long n, const double* X, const double* y,
                               const double* U, long m, double* jitter, const double* hyp) {
    const int nh = orc_cov_num_hyp(cov, d) + 2;
    orc_sparse_state* st = sparse_new(cov, d, U, m, hyp);
    const double sn2 = exp(2 * hyp[nh - 2]);
    double* r = dnew(n);
    for (long i = 0; i < n; ++i) r[i] = y[i] - hyp[nh - 1];
    double* Kxu = dnew((size_t)n * m);
    double* Kux = dnew((size_t)m * n);
    double* Kuu = dnew((size_t)m * m);
    double* S = dnew((size_t)m * m);
    orc_cov_k(cov, d, hyp, X, n, U, m, Kxu);
    transpose(n, m, Kxu, Kux);
    orc_cov_k(cov, d, hyp, U, m, U, m, Kuu);
    gemm_nn(m, m, n, Kux, Kxu, S);
    int first = 1, spd = 0;
    while (!spd) {
        if (!first)
            for (long i = 0; i < m; ++i) Kuu[(size_t)i * m + i] += *jitter; /* cumulative, VFE.cpp:149 */
        memcpy(st->Lu, Kuu, sizeof(double) * (size_t)m * m);
        for (size_t e = 0; e < (size_t)m * m; ++e) st->LA[e] = Kuu[e] + S[e] / sn2;
        const int iu = orc_chol(st->Lu, m);
        const int iA = orc_chol(st->LA, m);
        if (!first) *jitter *= 2; /* VFE.cpp:156 */
        spd = (iu == 0 && iA == 0);
        first = 0;
    }
    gemm_nn(m, 1, n, Kux, r, st->alpha);
    orc_chol_solve(st->LA, m, st->alpha, 1);
    for (long b = 0; b < m; ++b) st->alpha[b] /= sn2;
    free(r); free(Kxu); free(Kux); free(Kuu); free(S);
    return st;
}

/* FITC.cpp:109-164 */
void orc_fitc_predict(const orc_sparse_state* st, const double* Xs, long t, double* mu, double* s2,
                      double* gy, double* gs2) {
    const int cov = st->cov, d = st->d;
    const long m = st->m;
    const int nh = orc_cov_num_hyp(cov, d) + 2;
    const double sf2 = exp(2 * st->hyp[sf_index(cov, d)]);
    const double sn2 = exp(2 * st->hyp[nh - 2]);
    const double mean = st->hyp[nh - 1];
    double* Ks = dnew((size_t)t * m); /* t x m */
    orc_cov_k(cov, d, st->hyp, Xs, t, st->U, m, Ks);
    double* KsT = dnew((size_t)m * t);
    transpose(t, m, Ks, KsT);
    double* KinvK = NULL;
    if (s2 || gs2) {
        KinvK = dnew((size_t)m * t);
        double* tmp = dnew((size_t)m * t);
        memcpy(KinvK, KsT, sizeof(double) * (size_t)m * t);
        memcpy(tmp, KsT, sizeof(double) * (size_t)m * t);
        orc_chol_solve(st->Lu, m, KinvK, t);
        orc_chol_solve(st->LA, m, tmp, t);
        for (size_t e = 0; e < (size_t)m * t; ++e) KinvK[e] = KinvK[e] - sn2 * tmp[e];
        free(tmp);
    }
    for (long i = 0; i < t; ++i) {
        if (mu) mu[i] = dotn(m, KsT + (size_t)i * m, st->alpha) + mean;
        if (s2) {
            const double v = sn2 + (sf2 - dotn(m, KsT + (size_t)i * m, KinvK + (size_t)i * m));
            s2[i] = v > sn2 ? v : sn2;
        }
    }
    if (gy || gs2) {
        double* gk = dnew((size_t)d * m);
        for (long i = 0; i < t; ++i) {
            orc_cov_dk_dx1(cov, d, st->hyp, Xs + (size_t)i * d, st->U, m, gk);
            for (int k = 0; k < d; ++k) {
                double a1 = 0, a2 = 0;
                for (long b = 0; b < m; ++b) {
                    a1 += gk[(size_t)b * d + k] * st->alpha[b];
                    if (gs2) a2 += gk[(size_t)b * d + k] * KinvK[(size_t)i * m + b];
                }
                if (gy) gy[(size_t)i * d + k] = a1;
                if (gs2) gs2[(size_t)i * d + k] = 0.0 - 2 * a2;
            }
        }
        free(gk);
    }
    free(Ks); free(KsT); free(KinvK);
}

/* VFE.cpp:113-131 */
void orc_vfe_predict(const orc_sparse_state* st, const double* Xs, long t, double* mu, double* s2,
                     double* gy) {
    const int cov = st->cov, d = st->d;
    const long m = st->m;
    const int nh = orc_cov_num_hyp(cov, d) + 2;
    const double sf2 = exp(2 * st->hyp[sf_index(cov, d)]);
    const double mean = st->hyp[nh - 1];
    double* Ks = dnew((size_t)t * m);
    orc_cov_k(cov, d, st->hyp, Xs, t, st->U, m, Ks);
    double* KsT = dnew((size_t)m * t);
    transpose(t, m, Ks, KsT);
    if (mu)
        for (long i = 0; i < t; ++i) mu[i] = dotn(m, KsT + (size_t)i * m, st->alpha) + mean;
    if (s2) {
        double* KinvK = dnew((size_t)m * t);
        double* tmp = dnew((size_t)m * t);
        memcpy(KinvK, KsT, sizeof(double) * (size_t)m * t);
        memcpy(tmp, KsT, sizeof(double) * (size_t)m * t);
        orc_chol_solve(st->Lu, m, KinvK, t);
        orc_chol_solve(st->LA, m, tmp, t);
        for (long i = 0; i < t; ++i) {
            double s = 0;
            for (long b = 0; b < m; ++b)
                s += KsT[(size_t)i * m + b] * (KinvK[(size_t)i * m + b] - tmp[(size_t)i * m + b]);
            const double v = sf2 - s;
            s2[i] = v > 0 ? v : 0;
        }
        free(KinvK); free(tmp);
    }
    if (gy) {
        double* gk = dnew((size_t)d * m);
        for (long i = 0; i < t; ++i) {
            orc_cov_dk_dx1(cov, d, st->hyp, Xs + (size_t)i * d, st->U, m, gk);
            for (int k = 0; k < d; ++k) {
                double a1 = 0;
                for (long b = 0; b < m; ++b) a1 += gk[(size_t)b * d + k] * st->alpha[b];
                gy[(size_t)i * d + k] = a1;
            }
        }
        free(gk);
    }
    free(Ks); free(KsT);
}
