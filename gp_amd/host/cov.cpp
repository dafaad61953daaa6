#include "cov.hpp"
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include "../../include/gpx.h"

namespace gpx {

double stddev(const double* v, std::size_t n, std::size_t stride) {
    double s = 0;
    for (std::size_t i = 0; i < n; ++i) s += v[i * stride];
    const double mean = s / (double)n;
    double ss = 0;
    for (std::size_t i = 0; i < n; ++i) ss += (v[i * stride] - mean) * (v[i * stride] - mean);
    return std::sqrt(ss / (double)(n - 1));
}

MatrixXd Cov::k(const VectorXd& hyp, const MatrixXd& x1, const MatrixXd& x2) const {
    if (x1.rows() != _dim || x2.rows() != _dim || hyp.size() < num_hyp()) throw std::invalid_argument("Cov::k: shape");
    MatrixXd K(x1.cols(), x2.cols());
    const int rc = gpx_cov_k(abi_id(), (int64_t)_dim, hyp.data(), x1.data(), (int64_t)x1.cols(), x2.data(),
                             (int64_t)x2.cols(), K.data());
    if (rc) throw std::runtime_error("gpx_cov_k failed with status " + std::to_string(rc));
    return K;
}
MatrixXd Cov::dk_dhyp_device(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2, const MatrixXd& K) const {
    if (x1.rows() != _dim || x2.rows() != _dim || K.rows() != x1.cols() || K.cols() != x2.cols())
        throw std::invalid_argument("Cov::dk_dhyp: shapes");
    MatrixXd dK(K.rows(), K.cols());
    const int rc = gpx_cov_dk_dhyp(abi_id(), (int64_t)_dim, hyp.data(), (int64_t)idx, x1.data(), (int64_t)x1.cols(), x2.data(),
                                   (int64_t)x2.cols(), K.data(), dK.data());
    if (rc) throw std::runtime_error("gpx: cov_dk_dhyp failed with status " + std::to_string(rc));
    return dK;
}
MatrixXd Cov::dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2) const {
    return dk_dhyp(hyp, idx, x1, x2, k(hyp, x1, x2));
}
MatrixXd Cov::dk_dx1(const VectorXd& hyp, const VectorXd& x1, const MatrixXd& x2) const {
    MatrixXd one(_dim, 1);
    for (std::size_t i = 0; i < _dim; ++i) one(i, 0) = x1[i];
    return dk_dx1(hyp, x1, x2, k(hyp, one, x2));  // CovSEard.cpp:28-32
}
MatrixXd Cov::dk_dx1(const VectorXd& hyp, const VectorXd& x1, const MatrixXd& x2, const RowVectorXd& K) const {
    if (K.rows() != 1 || K.cols() != x2.cols() || x1.size() != _dim) throw std::invalid_argument("Cov::dk_dx1: shape");
    MatrixXd dK(_dim, x2.cols());  // CovSEard.cpp:33-40
    for (std::size_t b = 0; b < x2.cols(); ++b)
        for (std::size_t i = 0; i < _dim; ++i) dK(i, b) = K(0, b) * (inv_lscale2(hyp, i) * (x2(i, b) - x1[i]));
    return dK;
}
VectorXd Cov::diag_k(const VectorXd& hyp, const MatrixXd& x) const {
    return VectorXd::Constant(x.cols(), std::exp(2 * hyp[sf_index()]));
}
MatrixXd Cov::diag_dk_dhyp(const VectorXd& hyp, const MatrixXd& x) const {
    // The reference's SE-iso writes row _dim (CovSEiso.cpp:95,102: out of range for d >= 4);
    // here the row of the sf hyper is written for both kernels.
    MatrixXd g(hyp.size(), x.cols());
    const double v = 2 * std::exp(2 * hyp[sf_index()]);
    for (std::size_t j = 0; j < x.cols(); ++j) g(sf_index(), j) = v;
    return g;
}
MatrixXd Cov::diag_dk_dhyp(const VectorXd& hyp, const MatrixXd& x, const MatrixXd&) const { return diag_dk_dhyp(hyp, x); }
MatrixXd Cov::diag_dk_dx1(const VectorXd&, const VectorXd&) const { return MatrixXd(_dim, 1); }
MatrixXd Cov::diag_dk_dx1(const VectorXd& hyp, const VectorXd& x1, const RowVectorXd&) const { return diag_dk_dx1(hyp, x1); }

std::pair<double, double> Cov::lscale_range(const MatrixXd& xs, std::size_t i) const {
    double lo = xs(i, 0), hi = xs(i, 0);
    for (std::size_t a = 1; a < xs.cols(); ++a) {
        lo = std::min(lo, xs(i, a));
        hi = std::max(hi, xs(i, a));
    }
    const double distance = hi - lo, thres = 1e-4;
    const double lb = std::log(0.05 * distance) - 0.5 * std::log(-2 * std::log(1.5 * DBL_MIN));
    const double ub = std::min(0.5 * std::log(0.05 * DBL_MAX), std::log(distance / std::sqrt(-2 * std::log(1 - thres))));
    return {lb, ub};
}

static void y_range(const VectorXd& ys, double& lo, double& hi) {
    lo = hi = ys[0];
    for (std::size_t i = 1; i < ys.size(); ++i) {
        lo = std::min(lo, ys[i]);
        hi = std::max(hi, ys[i]);
    }
}

// ---- SE-ARD ---------------------------------------------------------------------------------
int CovSEard::abi_id() const { return GPX_COV_SE_ARD; }
double CovSEard::inv_lscale2(const VectorXd& hyp, std::size_t k) const { return std::exp(-2 * hyp[k]); }
MatrixXd CovSEard::dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2,
                           const MatrixXd& K) const {
    return dk_dhyp_device(hyp, idx, x1, x2, K);   // CovSEard.cpp:12-22 on the device
}
std::pair<VectorXd, VectorXd> CovSEard::cov_hyp_range(const MatrixXd& xs, const VectorXd& ys) const {
    VectorXd lb(num_hyp()), ub(num_hyp());
    for (std::size_t i = 0; i < _dim; ++i) {
        const auto r = lscale_range(xs, i);
        lb[i] = r.first;
        ub[i] = r.second;
    }
    double lo, hi;
    y_range(ys, lo, hi);
    const double e = std::numeric_limits<double>::epsilon();
    lb[_dim] = std::log(std::max(e, e * (hi - lo)));
    ub[_dim] = std::log(std::max(10 * e, 10 * (hi - lo)));
    return {lb, ub};
}
VectorXd CovSEard::default_hyp(const MatrixXd& xs, const VectorXd& ys) const {
    VectorXd h(num_hyp());
    for (std::size_t i = 0; i < _dim; ++i) h[i] = std::log(stddev(xs.data() + i, xs.cols(), _dim));
    h[_dim] = std::log(stddev(ys.data(), ys.size()));
    return h;
}

// ---- SE-iso ---------------------------------------------------------------------------------
int CovSEiso::abi_id() const { return GPX_COV_SE_ISO; }
double CovSEiso::inv_lscale2(const VectorXd& hyp, std::size_t) const { return std::exp(-2 * hyp[0]); }
MatrixXd CovSEiso::dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2,
                           const MatrixXd& K) const {
    if (idx > 1) {
        std::fprintf(stderr, "Wrong index for dk_dhyp\n");  // CovSEiso.cpp:24-26
        std::exit(EXIT_FAILURE);
    }
    return dk_dhyp_device(hyp, idx, x1, x2, K);   // CovSEiso.cpp:12-23 on the device
}
std::pair<VectorXd, VectorXd> CovSEiso::cov_hyp_range(const MatrixXd& xs, const VectorXd& ys) const {
    VectorXd lb(2, -std::numeric_limits<double>::infinity()), ub(2, 0.5 * std::log(0.5 * DBL_MAX));
    for (std::size_t i = 0; i < _dim; ++i) {
        const auto r = lscale_range(xs, i);
        lb[0] = std::max(lb[0], r.first);
        ub[0] = std::min(ub[0], r.second);
    }
    double lo, hi;
    y_range(ys, lo, hi);
    lb[1] = std::log(std::max(0.0, std::numeric_limits<double>::epsilon() * (hi - lo)));
    ub[1] = std::log(10 * (hi - lo));
    return {lb, ub};
}
VectorXd CovSEiso::default_hyp(const MatrixXd&, const VectorXd& ys) const {
    VectorXd h(2);
    h[0] = 0;
    h[1] = std::log(stddev(ys.data(), ys.size()));
    return h;
}

}  // namespace gpx
