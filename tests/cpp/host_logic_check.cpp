// host_logic_check.cpp — CPU-only checks of the host logic that needs no GPU: the bounded L-BFGS that
// stands in for NLopt's LD_SLSQP behind the same callback shape (GP.cpp:186-191, 231-265) and the
// Eigen-layout stand-ins of dense.hpp.  Prints `key value` lines; tests/test_cpu_host_logic.py asserts.
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../gp_amd/host/dense.hpp"
#include "../../gp_amd/host/optimizer.hpp"
using namespace gpx;

static int g_calls = 0;

static double quad(const std::vector<double>& x, std::vector<double>& g, void*) {
    // minimum at (3, -2, 0.5); the box cuts the first coordinate at 1
    const double c[3] = {3.0, -2.0, 0.5}, w[3] = {1.0, 10.0, 0.1};
    double f = 0;
    ++g_calls;
    for (int i = 0; i < 3; ++i) {
        f += w[i] * (x[i] - c[i]) * (x[i] - c[i]);
        if (!g.empty()) g[i] = 2 * w[i] * (x[i] - c[i]);
    }
    return f;
}
static double rosen(const std::vector<double>& x, std::vector<double>& g, void*) {
    ++g_calls;
    const double a = 1 - x[0], b = x[1] - x[0] * x[0];
    if (!g.empty()) {
        g[0] = -2 * a - 400 * x[0] * b;
        g[1] = 200 * b;
    }
    return a * a + 100 * b * b;
}
static double stopper(const std::vector<double>& x, std::vector<double>& g, void* data) {
    if (++g_calls >= *static_cast<int*>(data)) throw ForcedStop();
    return quad(x, g, nullptr);
}
static double nan_after(const std::vector<double>& x, std::vector<double>& g, void*) {
    // +inf away from the start region, as the GP objective returns on a non-SPD K (GP.cpp:132-134)
    if (x[0] > 0.5) {
        ++g_calls;
        for (auto& v : g) v = INFINITY;
        return INFINITY;
    }
    return quad(x, g, nullptr);
}

int main() {
    {
        BoundedLBFGS o(3);
        o.set_lower_bounds({-5, -5, -5});
        o.set_upper_bounds({1, 5, 5});
        o.set_min_objective(quad, nullptr);
        std::vector<double> x = {-4, 4, 4};
        double f;
        g_calls = 0;
        OptResult r = o.optimize(x, f);
        std::printf("quad_x %.12g %.12g %.12g\nquad_f %.12g\nquad_evals %d\nquad_calls %d\n", x[0], x[1], x[2], f, r.evals, g_calls);
        std::printf("quad_status_ok %d\n", r.status != "FAILURE");
    }
    {
        BoundedLBFGS o(3);   // start outside the box: clamped first (GP.cpp:200-210)
        o.set_lower_bounds({-5, -5, -5});
        o.set_upper_bounds({1, 5, 5});
        o.set_min_objective(quad, nullptr);
        std::vector<double> x = {40, -40, 0};
        double f;
        o.optimize(x, f);
        std::printf("clamp_x %.12g %.12g %.12g\n", x[0], x[1], x[2]);
    }
    {
        BoundedLBFGS o(2);
        o.set_lower_bounds({-2, -2});
        o.set_upper_bounds({2, 2});
        o.set_maxeval(400);
        o.set_min_objective(rosen, nullptr);
        std::vector<double> x = {-1.2, 1.0};
        double f;
        OptResult r = o.optimize(x, f);
        std::printf("rosen_x %.12g %.12g\nrosen_f %.12g\nrosen_evals %d\n", x[0], x[1], f, r.evals);
    }
    {
        BoundedLBFGS o(2);
        o.set_lower_bounds({-2, -2});
        o.set_upper_bounds({2, 2});
        o.set_maxeval(7);
        o.set_min_objective(rosen, nullptr);
        std::vector<double> x = {-1.2, 1.0};
        double f;
        g_calls = 0;
        OptResult r = o.optimize(x, f);
        std::vector<double> none;
        std::printf("budget_evals %d\nbudget_calls %d\nbudget_status %s\nbudget_best_is_returned %d\n", r.evals, g_calls,
                    r.status.c_str(), std::fabs(rosen(x, none, nullptr) - f) < 1e-15);
    }
    {
        BoundedLBFGS o(3);
        int limit = 3;
        o.set_min_objective(stopper, &limit);
        std::vector<double> x = {0, 0, 0};
        double f;
        g_calls = 0;
        OptResult r = o.optimize(x, f);
        std::printf("forced_status %s\nforced_calls %d\nforced_f_finite %d\n", r.status.c_str(), g_calls, std::isfinite(f));
    }
    {
        BoundedLBFGS o(3);
        o.set_lower_bounds({-5, -5, -5});
        o.set_upper_bounds({5, 5, 5});
        o.set_min_objective(nan_after, nullptr);
        std::vector<double> x = {-1, 1, 1};
        double f;
        OptResult r = o.optimize(x, f);
        std::printf("inf_region_x0 %.12g\ninf_region_f_finite %d\ninf_region_status %s\n", x[0], std::isfinite(f), r.status.c_str());
    }
    {
        MatrixXd m(2, 3);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) m(i, j) = 10 * i + j;
        const MatrixXd t = m.transpose();
        const VectorXd c = m.col(2);
        std::printf("dense_colmajor %g %g %g\n", m.data()[1], m.data()[2], m.data()[5]);  // (1,0), (0,1), (1,2)
        std::printf("dense_transpose %zu %zu %g\n", (size_t)t.rows(), (size_t)t.cols(), t(2, 1));
        std::printf("dense_col %g %g\n", c[0], c[1]);
        const MatrixXd l = m.leftCols(2), r = m.rightCols(1);
        std::printf("dense_blocks %g %g\n", l(1, 1), r(1, 0));
    }
    return 0;
}
