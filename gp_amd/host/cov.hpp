// cov.hpp — the kernel plug-in seam (Cov/Cov.h:5-28, CovSEard.h, CovSEiso.h) over the C ABI.
//
// Same class names and method meanings as the reference.  k() runs the HIP kernel-matrix build
// (gpx_cov_k); the O(1)/O(n) members (num_hyp, default_hyp, cov_hyp_range, diag_k, diag_dk_dhyp)
// are host code.  dk_dhyp materialises one dK ON THE DEVICE (gpx_cov_dk_dhyp) from a K that is already there
// — it exists for API compatibility only: the training path never calls it (the fused trace kernel
// contracts all dK at once).
#pragma once
#include <utility>
#include "dense.hpp"

namespace gpx {

class Cov {
protected:
    std::size_t _dim;

public:
    explicit Cov(std::size_t d) : _dim(d) {}
    virtual ~Cov() {}
    virtual int abi_id() const = 0;  // GPX_COV_SE_ARD / GPX_COV_SE_ISO
    virtual std::size_t num_hyp() const = 0;
    virtual MatrixXd k(const VectorXd& hyp, const MatrixXd& x1, const MatrixXd& x2) const;
    virtual MatrixXd dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2) const;
    virtual MatrixXd dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2,
                             const MatrixXd& K) const = 0;
    virtual MatrixXd dk_dx1(const VectorXd& hyp, const VectorXd& x1, const MatrixXd& x2) const;
    // Cov/Cov.h:18 — the same from a kernel row K = k(x1, x2) (1 x q) that the caller already holds
    virtual MatrixXd dk_dx1(const VectorXd& hyp, const VectorXd& x1, const MatrixXd& x2, const RowVectorXd& K) const;
    virtual std::pair<VectorXd, VectorXd> cov_hyp_range(const MatrixXd& xs, const VectorXd& ys) const = 0;
    virtual VectorXd default_hyp(const MatrixXd& xs, const VectorXd& ys) const = 0;
    virtual VectorXd diag_k(const VectorXd& hyp, const MatrixXd& x) const;
    virtual MatrixXd diag_dk_dhyp(const VectorXd& hyp, const MatrixXd& x) const;
    // Cov/Cov.h:25,27 — overloads that take an already computed K; the SE kernels ignore it
    // (CovSEard.cpp:85-103: the diagonal is sf2 whatever K holds)
    virtual MatrixXd diag_dk_dhyp(const VectorXd& hyp, const MatrixXd& x, const MatrixXd& K) const;
    virtual MatrixXd diag_dk_dx1(const VectorXd& hyp, const VectorXd& x1) const;
    virtual MatrixXd diag_dk_dx1(const VectorXd& hyp, const VectorXd& x1, const RowVectorXd& K) const;

protected:
    // one dK on the device (gpx_cov_dk_dhyp): host matrices in, host matrix out, as seam (1) demands
    MatrixXd dk_dhyp_device(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2, const MatrixXd& K) const;
    virtual std::size_t sf_index() const = 0;
    virtual double inv_lscale2(const VectorXd& hyp, std::size_t k) const = 0;
    // [lower, upper] of log lengthscale for input dimension i (CovSEard.cpp:46-66)
    std::pair<double, double> lscale_range(const MatrixXd& xs, std::size_t i) const;
};

class CovSEard : public Cov {
public:
    explicit CovSEard(std::size_t d) : Cov(d) {}
    int abi_id() const override;
    std::size_t num_hyp() const override { return _dim + 1; }
    using Cov::dk_dhyp;
    using Cov::dk_dx1;
    MatrixXd dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2,
                     const MatrixXd& K) const override;
    std::pair<VectorXd, VectorXd> cov_hyp_range(const MatrixXd& xs, const VectorXd& ys) const override;
    VectorXd default_hyp(const MatrixXd& xs, const VectorXd& ys) const override;

protected:
    std::size_t sf_index() const override { return _dim; }
    double inv_lscale2(const VectorXd& hyp, std::size_t k) const override;
};

class CovSEiso : public Cov {
public:
    explicit CovSEiso(std::size_t d) : Cov(d) {}
    int abi_id() const override;
    std::size_t num_hyp() const override { return 2; }
    using Cov::dk_dhyp;
    using Cov::dk_dx1;
    MatrixXd dk_dhyp(const VectorXd& hyp, std::size_t idx, const MatrixXd& x1, const MatrixXd& x2,
                     const MatrixXd& K) const override;
    std::pair<VectorXd, VectorXd> cov_hyp_range(const MatrixXd& xs, const VectorXd& ys) const override;
    VectorXd default_hyp(const MatrixXd& xs, const VectorXd& ys) const override;

protected:
    std::size_t sf_index() const override { return 1; }
    double inv_lscale2(const VectorXd& hyp, std::size_t k) const override;
};

// sample standard deviation with n-1 (util.h:32-41)
double stddev(const double* v, std::size_t n, std::size_t stride = 1);

}  // namespace gpx
