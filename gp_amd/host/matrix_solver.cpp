#include "matrix_solver.hpp"
#include <limits>
#include <stdexcept>
#include <string>
#include "../../include/gpx.h"

namespace gpx {

static void run(const MatrixXd& a, int* spd, double* ld, double* B, int64_t nrhs, double* Inv) {
    const int rc = gpx_chol_solve((int64_t)a.rows(), a.data(), spd, ld, B, nrhs, Inv);
    if (rc) throw std::runtime_error("gpx_chol_solve failed with status " + std::to_string(rc));
}

void MatrixSolverLLT::decomp(const MatrixXd& x) {
    if (x.rows() != x.cols()) throw std::invalid_argument("MatrixSolver::decomp: not square");
    _a = x;
    int spd = 0;
    _logdet = std::numeric_limits<double>::quiet_NaN();
    run(_a, &spd, &_logdet, nullptr, 0, nullptr);
    _spd = spd != 0;
}
MatrixXd MatrixSolverLLT::solve(const MatrixXd& ys) const {
    MatrixXd x = ys;
    int spd = 0;
    run(_a, &spd, nullptr, x.data(), (int64_t)x.cols(), nullptr);
    return x;
}
MatrixXd MatrixSolverLLT::inverse() const {
    MatrixXd inv(_a.rows(), _a.cols());
    int spd = 0;
    run(_a, &spd, nullptr, nullptr, 0, inv.data());
    return inv;
}

}  // namespace gpx
