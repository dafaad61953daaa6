// mle_oracle_check.cpp — GP::train's driver logic (GP.cpp:177-272: start clamped into the box, maxeval
// 160 / 130 / 150, best point returned) run on the CPU ORACLE's objective with the SAME optimiser object the
// facade uses (gpx::make_optimizer).  tests/test_gpu_mle_parity.py runs the facade's train() on the GPU on the
// same data and start point and compares the final NLL and the evaluation count.  Test infrastructure only.
// usage: mle_oracle_check <n> <m>   (train_x, train_y as text in the current directory; U = the last m rows)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "../../gp_amd/host/gp.hpp"
#include "../../gp_amd/host/optimizer.hpp"
extern "C" {
#include "../../oracle/gp_oracle.h"
}
using namespace gpx;

struct Problem {
    int model, d;
    long n, m;
    const double *X, *y, *U;
    double jitter;
    int calls = 0;
};
static double objective(const std::vector<double>& x, std::vector<double>& g, void* data) {
    Problem* p = static_cast<Problem*>(data);
    ++p->calls;
    std::vector<double> grad(x.size());
    double v;
    if (p->model == 0) v = orc_exact_nll(ORC_COV_SE_ARD, p->d, p->n, p->X, p->y, x.data(), g.empty() ? 0 : 1, grad.data());
    else if (p->model == 1) v = orc_fitc_nll(ORC_COV_SE_ARD, p->d, p->n, p->X, p->y, p->U, p->m, p->jitter, x.data(), g.empty() ? 0 : 1, grad.data());
    else v = orc_vfe_nll(ORC_COV_SE_ARD, p->d, p->n, p->X, p->y, p->U, p->m, p->jitter, x.data(), g.empty() ? 0 : 1, grad.data());
    if (!g.empty()) g = grad;
    return v;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 256, m = argc > 2 ? atol(argv[2]) : 32;
    const MatrixXd tx = read_matrix("train_x"), ty = read_matrix("train_y");   // rows = points
    const int d = (int)tx.cols();
    std::vector<double> X((size_t)n * d), y(n), U((size_t)m * d);
    for (long a = 0; a < n; ++a) {
        for (int i = 0; i < d; ++i) X[(size_t)a * d + i] = tx(a, i);
        y[a] = ty(a, 0);
    }
    for (long b = 0; b < m; ++b)
        for (int i = 0; i < d; ++i) U[(size_t)b * d + i] = tx(tx.rows() - m + b, i);
    const int nh = d + 3;
    std::vector<double> lb(nh), ub(nh), h0(nh);
    const double noise_lb = 1e-3;
    orc_hyp_range(ORC_COV_SE_ARD, d, n, X.data(), y.data(), noise_lb, lb.data(), ub.data());
    orc_default_hyps(ORC_COV_SE_ARD, d, n, X.data(), y.data(), noise_lb, h0.data());
    h0[nh - 2] = std::log(0.1 * orc_stddev(y.data(), n, 1));   // the start point tests/cpp/facade_check.cpp uses
    const char* tags[3] = {"gp", "fitc", "vfe"};
    const int maxeval[3] = {160, 130, 150};   // GP.cpp:232, FITC.cpp:75, VFE.cpp:74
    for (int model = 0; model < 3; ++model) {
        Problem p{model, d, n, m, X.data(), y.data(), U.data(), std::pow(0.1 * noise_lb, 2)};
        std::vector<double> none, x = h0;
        const double nll0 = objective(x, none, &p);   // GP.cpp:196
        std::unique_ptr<IOptimizer> opt(make_optimizer(nh));
        const char* me = std::getenv("GPX_MAXEVAL");   // tests: a budget inside which the run converges, on both sides
        opt->set_maxeval(me ? std::atoi(me) : maxeval[model]);
        opt->set_min_objective(objective, &p);
        opt->set_lower_bounds(lb);
        opt->set_upper_bounds(ub);
        double f;
        const OptResult r = opt->optimize(x, f);
        std::printf("%s_nll0 %.17g\n%s_nll %.17g\n%s_evals %d\n%s_status %s\n", tags[model], nll0, tags[model], f, tags[model],
                    r.evals, tags[model], r.status.c_str());
        std::printf("%s_hyp", tags[model]);
        for (double v : x) std::printf(" %.17g", v);
        std::printf("\n");
    }
    return 0;
}
