// dense.hpp — minimal column-major dense types for the host facade.
//
// The reference's API is typed on Eigen::MatrixXd / VectorXd (GP.h:80-121).  Eigen is not a
// dependency of this library; these two classes keep the same memory layout (column-major,
// contiguous, data()/rows()/cols()/size()) so that an Eigen object converts without a copy of
// semantics: any type with data(), rows(), cols() is accepted by the converting constructors.
#pragma once
#include <cstddef>
#include <initializer_list>
#include <vector>

namespace gpx {

class VectorXd {
    std::vector<double> v_;

public:
    VectorXd() = default;
    explicit VectorXd(std::size_t n, double fill = 0.0) : v_(n, fill) {}
    VectorXd(std::initializer_list<double> l) : v_(l) {}
    VectorXd(const std::vector<double>& s) : v_(s) {}
    template <class E, class = decltype(std::declval<const E&>().data()), class = decltype(std::declval<const E&>().size()),
              class = typename std::enable_if<!std::is_same<E, VectorXd>::value && !std::is_same<E, std::vector<double>>::value>::type>
    VectorXd(const E& e) : v_(e.data(), e.data() + e.size()) {}
    std::size_t size() const { return v_.size(); }
    std::size_t rows() const { return v_.size(); }
    std::size_t cols() const { return 1; }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    double& operator()(std::size_t i) { return v_[i]; }
    double operator()(std::size_t i) const { return v_[i]; }
    double& operator[](std::size_t i) { return v_[i]; }
    double operator[](std::size_t i) const { return v_[i]; }
    const std::vector<double>& std() const { return v_; }
    static VectorXd Constant(std::size_t n, double c) { return VectorXd(n, c); }
};

class MatrixXd {
    std::size_t r_ = 0, c_ = 0;
    std::vector<double> v_;

public:
    MatrixXd() = default;
    MatrixXd(std::size_t r, std::size_t c, double fill = 0.0) : r_(r), c_(c), v_(r * c, fill) {}
    MatrixXd(const VectorXd& x) : r_(x.size()), c_(1), v_(x.data(), x.data() + x.size()) {}
    template <class E, class = decltype(std::declval<const E&>().data()), class = decltype(std::declval<const E&>().rows()),
              class = decltype(std::declval<const E&>().cols()),
              class = typename std::enable_if<!std::is_same<E, MatrixXd>::value && !std::is_same<E, VectorXd>::value>::type>
    MatrixXd(const E& e) : r_(e.rows()), c_(e.cols()), v_(e.data(), e.data() + e.rows() * e.cols()) {}
    std::size_t rows() const { return r_; }
    std::size_t cols() const { return c_; }
    std::size_t size() const { return v_.size(); }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    double& operator()(std::size_t i, std::size_t j) { return v_[i + j * r_]; }
    double operator()(std::size_t i, std::size_t j) const { return v_[i + j * r_]; }
    VectorXd col(std::size_t j) const {
        VectorXd x(r_);
        for (std::size_t i = 0; i < r_; ++i) x[i] = (*this)(i, j);
        return x;
    }
    MatrixXd leftCols(std::size_t k) const { return middleCols(0, k); }
    MatrixXd rightCols(std::size_t k) const { return middleCols(c_ - k, k); }
    MatrixXd middleCols(std::size_t j0, std::size_t k) const {
        MatrixXd m(r_, k);
        for (std::size_t e = 0; e < r_ * k; ++e) m.v_[e] = v_[j0 * r_ + e];
        return m;
    }
    MatrixXd transpose() const {
        MatrixXd t(c_, r_);
        for (std::size_t j = 0; j < c_; ++j)
            for (std::size_t i = 0; i < r_; ++i) t(j, i) = (*this)(i, j);
        return t;
    }
    void appendCols(const MatrixXd& o) { v_.insert(v_.end(), o.v_.begin(), o.v_.end()); c_ += o.c_; if (!r_) r_ = o.r_; }
};

// Eigen::RowVectorXd in the reference's Cov interface (Cov/Cov.h:18,27): a 1 x q matrix here
typedef MatrixXd RowVectorXd;

}  // namespace gpx
