// matrix_solver.hpp — MatrixSolver / MatrixSolverLLT / MatrixSolverQR (MatrixSolver.h:3-35).
//
// Seam (2) of the reference moves whole host matrices across the boundary; it is kept as an
// API-compatible wrapper over the blocked MFMA Cholesky on the device: decomp() uploads and factors
// ONCE (gpx_chol_factor), the factor stays in HBM, solve() / inverse() reuse it.  The QR class maps
// to the same Cholesky: for the SPD matrices this library factors, log|det|, solve and inverse agree
// with column-pivoted QR to rounding — deliberate.
#pragma once
#include "dense.hpp"

struct gpx_chol;

namespace gpx {

class MatrixSolver {
public:
    MatrixSolver() {}
    virtual ~MatrixSolver() {}
    virtual void decomp(const MatrixXd&) = 0;
    virtual bool check_SPD() const = 0;
    virtual double log_det() const = 0;
    virtual MatrixXd solve(const MatrixXd& ys) const = 0;
    virtual MatrixXd inverse() const = 0;
};

class MatrixSolverLLT : public MatrixSolver {
    gpx_chol* _f = nullptr;   // device-resident factor
    std::size_t _n = 0;
    bool _spd = false;
    double _logdet = 0;
    int _device = -1;         // -1: the caller's current device at decomp()

public:
    MatrixSolverLLT() {}
    explicit MatrixSolverLLT(int device) : _device(device) {}   // GP::device() of the object that owns the solver
    ~MatrixSolverLLT() override;
    MatrixSolverLLT(const MatrixSolverLLT&) = delete;
    MatrixSolverLLT& operator=(const MatrixSolverLLT&) = delete;
    void decomp(const MatrixXd&) override;
    bool check_SPD() const override { return _spd; }
    double log_det() const override { return _logdet; }
    MatrixXd solve(const MatrixXd& ys) const override;
    MatrixXd inverse() const override;
};

class MatrixSolverQR : public MatrixSolverLLT {};

}  // namespace gpx
