// facade_check.cpp — exercises the C++ facade (GP / FITC / VFE / Cov / MatrixSolver) on text data
// in the current directory and prints `key value...` lines that tests/test_gpu_facade.py compares
// with the oracle.  usage: facade_check <num_train> <num_inducing> <fixed 0|1>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include "../../gp_amd/host/gp.hpp"
#include "../../gp_amd/host/optimizer.hpp"
using namespace gpx;

static void pv(const char* k, const VectorXd& v) {
    std::printf("%s", k);
    for (std::size_t i = 0; i < v.size(); ++i) std::printf(" %.17g", v[i]);
    std::printf("\n");
}

template <class M>
static void run(const char* tag, M& gp, const VectorXd& init, const MatrixXd& test_x, bool fixed) {
    char key[64];
    VectorXd g;
    const double nll0 = gp.nll(init, g);
    std::snprintf(key, sizeof key, "%s_nll0", tag); std::printf("%s %.17g\n", key, nll0);
    std::snprintf(key, sizeof key, "%s_grad0", tag); pv(key, g);
    std::snprintf(key, sizeof key, "%s_fdrel", tag); std::printf("%s %.17g\n", key, gp.check_gradient(init, 1e-4));
    gp.set_fixed(fixed);
    if (const char* e = std::getenv("GPX_MAXEVAL")) gp.set_max_eval(std::atoi(e));   // tests: a budget inside which the run converges
    const double nll = gp.train(init);
    std::snprintf(key, sizeof key, "%s_nll", tag); std::printf("%s %.17g\n", key, nll);
    std::fprintf(stderr, "%s optimizer status: %s\n", tag, gp.last_optimizer_status().c_str());
    std::snprintf(key, sizeof key, "%s_evals", tag); std::printf("%s %d\n", key, gp.last_optimizer_evals());
    std::snprintf(key, sizeof key, "%s_hyp", tag); pv(key, gp.get_hyp());
    std::snprintf(key, sizeof key, "%s_mu", tag); pv(key, gp.batch_predict_y(test_x));
    std::snprintf(key, sizeof key, "%s_s2", tag); pv(key, gp.batch_predict_s2(test_x));
    double y1, s1;
    gp.predict(test_x.col(0), y1, s1);
    std::snprintf(key, sizeof key, "%s_single", tag); std::printf("%s %.17g %.17g\n", key, y1, s1);
    // input gradients at the first test point (GP.h:106-115); VFE has the mean gradient only
    const VectorXd x0 = test_x.col(0);
    VectorXd gy;
    const double yg = gp.predict_y_with_grad(x0, gy);
    std::snprintf(key, sizeof key, "%s_gy", tag); pv(key, gy);
    std::snprintf(key, sizeof key, "%s_yg", tag); std::printf("%s %.17g\n", key, yg);
    try {
        const auto all = gp.predict_with_grad(x0);
        std::snprintf(key, sizeof key, "%s_gs2", tag); pv(key, std::get<3>(all));
        std::snprintf(key, sizeof key, "%s_gy2", tag); pv(key, std::get<2>(all));
        VectorXd gs;
        const double sg = gp.predict_s2_with_grad(x0, gs);
        std::snprintf(key, sizeof key, "%s_sg", tag); std::printf("%s %.17g\n", key, sg);
    } catch (const std::logic_error& e) {
        std::snprintf(key, sizeof key, "%s_gs2_unsupported", tag); std::printf("%s 1\n", key);
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 200, m = argc > 2 ? atol(argv[2]) : 32;
    const bool fixed = argc > 3 ? atoi(argv[3]) != 0 : true;
    const MatrixXd train_x = read_matrix("train_x").transpose();
    const MatrixXd train_y = read_matrix("train_y");
    const MatrixXd test_x = read_matrix("test_x").transpose();
    MatrixXd y(n, 1);
    for (long i = 0; i < n; ++i) y(i, 0) = train_y(i, 0);
    const MatrixXd X = train_x.leftCols(n);
    {
        GP gp(X, y, GP::COV_SE_ARD, GP::QR);
        VectorXd init = gp.get_default_hyps();
        pv("default_hyp", init);
        init[init.size() - 2] = std::log(0.1 * stddev(y.data(), n));
        run("gp", gp, init, test_x, fixed);
    }
    {
        FITC gp(X, y);
        gp.set_inducing(train_x.rightCols(m));
        VectorXd init = gp.get_default_hyps();
        init[init.size() - 2] = std::log(0.1 * stddev(y.data(), n));
        run("fitc", gp, init, test_x, fixed);
    }
    {
        VFE gp(X, y);
        gp.set_inducing(train_x.rightCols(m));
        VectorXd init = gp.get_default_hyps();
        init[init.size() - 2] = std::log(0.1 * stddev(y.data(), n));
        run("vfe", gp, init, test_x, fixed);
    }
    {   // VFE::train when the optimiser itself fails (VFE.cpp:94-101): the start search is re-run with num_hyp * 10
        // evaluations from the point reached; GP / FITC keep that point (GP.cpp:260-265)
        struct Failing : IOptimizer {
            const char* name() const override { return "failing test optimiser"; }
            void set_maxeval(int) override {}
            void set_min_objective(ObjectiveFn, void*) override {}
            void set_lower_bounds(const std::vector<double>&) override {}
            void set_upper_bounds(const std::vector<double>&) override {}
            OptResult optimize(std::vector<double>&, double&) override { OptResult r; r.status = "FAILURE"; return r; }
        };
        auto factory = [](std::size_t) -> IOptimizer* { return new Failing(); };
        VFE vfe(X, y);
        vfe.set_inducing(train_x.rightCols(m));
        VectorXd init = vfe.get_default_hyps();
        init[init.size() - 2] = std::log(0.1 * stddev(y.data(), n));
        vfe.set_optimizer_factory(factory);
        vfe.train(init);
        const VectorXd want = vfe.select_init_hyp(vfe.num_hyp() * 10, init);
        bool same = true, moved = false;
        for (std::size_t i = 0; i < want.size(); ++i) { same = same && want[i] == vfe.get_hyp()[i]; moved = moved || want[i] != init[i]; }
        FITC fitc(X, y);
        fitc.set_inducing(train_x.rightCols(m));
        fitc.set_optimizer_factory(factory);
        fitc.train(init);
        bool kept = true;
        for (std::size_t i = 0; i < init.size(); ++i) kept = kept && fitc.get_hyp()[i] == init[i];
        std::printf("vfe_failure_fallback %d %d\nfitc_failure_keeps_point %d\n", (int)same, (int)moved, (int)kept);
    }
    {   // seams (1) and (2)
        CovSEiso cov(X.rows());
        VectorXd h = {0.3, 0.1};
        const MatrixXd K = cov.k(h, X.leftCols(5), X.leftCols(7));
        VectorXd flat(std::vector<double>(K.data(), K.data() + K.size()));
        pv("iso_k", flat);
        const MatrixXd dKi = cov.dk_dhyp(h, 0, X.leftCols(5), X.leftCols(7));   // Cov/Cov.h:15, on the device
        VectorXd dflat(std::vector<double>(dKi.data(), dKi.data() + dKi.size()));
        pv("iso_dk0", dflat);
        CovSEard ard(X.rows());
        VectorXd ha(X.rows() + 1, 0.5);
        MatrixXd Kxx = ard.k(ha, X.leftCols(40), X.leftCols(40));
        for (std::size_t i = 0; i < 40; ++i) Kxx(i, i) += 0.01;
        MatrixSolverLLT llt;
        llt.decomp(Kxx);
        std::printf("llt_spd %d\nllt_logdet %.17g\n", (int)llt.check_SPD(), llt.log_det());
        const MatrixXd inv = llt.inverse();
        double err = 0;
        for (std::size_t i = 0; i < 40; ++i)
            for (std::size_t j = 0; j < 40; ++j) {
                double s = 0;
                for (std::size_t k = 0; k < 40; ++k) s += Kxx(i, k) * inv(k, j);
                err = std::fmax(err, std::fabs(s - (i == j ? 1.0 : 0.0)));
            }
        std::printf("llt_inv_err %.3g\n", err);
    }
    return 0;
}
