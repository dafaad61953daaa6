// bounds_check.cpp — host-only checks of the hyper bookkeeping of the facade (GP.cpp:85-92, 514-574;
// CovSEard.cpp:41-79; CovSEiso.cpp:47-85): the optimiser's box, default hypers and the noise-free packing.
// No GPU call is made (the device context is created at the first evaluation only), so this runs on CPU.
// Reads train_x / train_y (text, rows = points) from the current directory; prints `key v...` lines.
#include <cstdio>
#include "../../gp_amd/host/gp.hpp"
using namespace gpx;

static void pv(const char* k, const VectorXd& v) {
    std::printf("%s", k);
    for (std::size_t i = 0; i < v.size(); ++i) std::printf(" %.17g", v[i]);
    std::printf("\n");
}
static void pvec(const char* k, const std::vector<double>& v) { pv(k, VectorXd(v)); }

int main() {
    const MatrixXd X = read_matrix("train_x").transpose();
    const MatrixXd y = read_matrix("train_y");
    for (int cf = 0; cf < 2; ++cf) {
        GP gp(X, y, cf == 0 ? GP::COV_SE_ARD : GP::COV_SE_ISO, GP::QR);
        const char* t = cf == 0 ? "ard" : "iso";
        char key[64];
        std::snprintf(key, sizeof key, "%s_lb", t); pv(key, gp.hyp_lower_bounds());
        std::snprintf(key, sizeof key, "%s_ub", t); pv(key, gp.hyp_upper_bounds());
        std::snprintf(key, sizeof key, "%s_default", t); pv(key, gp.get_default_hyps());
        VectorXd h = gp.get_default_hyps();
        std::snprintf(key, sizeof key, "%s_pack", t); pvec(key, gp.pack_hyp(h));
        gp.set_noise_free(true);   // GP.cpp:79-84: drops the noise slot from the optimiser's vector
        std::snprintf(key, sizeof key, "%s_default_noise_free", t); pv(key, gp.get_default_hyps());
        const std::vector<double> packed = gp.pack_hyp(h);
        std::snprintf(key, sizeof key, "%s_pack_noise_free", t); pvec(key, packed);
        std::snprintf(key, sizeof key, "%s_unpack_noise_free", t); pv(key, gp.unpack_hyp(packed));
        gp.set_noise_free(false);
        gp.set_noise_lower_bound(1e-2);   // GP.cpp:62-78; takes effect in the bounds at the next _init
        std::snprintf(key, sizeof key, "%s_default_nlb", t); pv(key, gp.get_default_hyps());
    }
    // Cov overloads that take an already computed K (Cov/Cov.h:25,27): the SE kernels ignore it
    CovSEard ard(X.rows());
    VectorXd hh(X.rows() + 3, 0.25);
    const MatrixXd d1 = ard.diag_dk_dhyp(hh, X.leftCols(3)), d2 = ard.diag_dk_dhyp(hh, X.leftCols(3), MatrixXd(3, 3, 7.0));
    double diff = 0;
    for (std::size_t e = 0; e < d1.size(); ++e) diff += d1.data()[e] - d2.data()[e];
    std::printf("diag_dk_dhyp_rows %zu\ndiag_dk_dhyp_sf %.17g\ndiag_overload_diff %.17g\n", d1.rows(), d1(X.rows(), 1), diff);
    const MatrixXd z = ard.diag_dk_dx1(hh, X.col(0), MatrixXd(1, 1, 3.0));
    std::printf("diag_dk_dx1_shape %zu %zu\n", z.rows(), z.cols());
    return 0;
}
