#include "matrix_solver.hpp"
#include <limits>
#include <stdexcept>
#include <string>
#include "../../include/gpx.h"

namespace gpx {

static void must(int rc, const char* what) {
    if (rc) throw std::runtime_error(std::string("gpx: ") + what + " failed with status " + std::to_string(rc));
}

MatrixSolverLLT::~MatrixSolverLLT() { gpx_chol_free(_f); }

void MatrixSolverLLT::decomp(const MatrixXd& x) {
    if (x.rows() != x.cols()) throw std::invalid_argument("MatrixSolver::decomp: not square");
    gpx_chol_free(_f);
    _f = nullptr;
    _n = x.rows();
    int spd = 0;
    _logdet = std::numeric_limits<double>::quiet_NaN();
    must(gpx_chol_factor(_device, (int64_t)_n, x.data(), &_f, &spd, &_logdet), "chol_factor");
    _spd = spd != 0;
}
MatrixXd MatrixSolverLLT::solve(const MatrixXd& ys) const {
    if (!_f || ys.rows() != _n) throw std::invalid_argument("MatrixSolver::solve: decomp() first, rows must match");
    MatrixXd x = ys;
    must(gpx_chol_solve_factored(_f, x.data(), (int64_t)x.cols()), "chol_solve");
    return x;
}
MatrixXd MatrixSolverLLT::inverse() const {
    if (!_f) throw std::invalid_argument("MatrixSolver::inverse: decomp() first");
    MatrixXd inv(_n, _n);
    must(gpx_chol_inverse_factored(_f, inv.data()), "chol_inverse");
    return inv;
}

}  // namespace gpx
