// optimizer.cpp — projected L-BFGS with Armijo backtracking along the projection arc.
#include "optimizer.hpp"
#include <algorithm>
#include <cmath>
#include <deque>
#include <limits>

namespace gpx {
namespace {
double dot(const std::vector<double>& a, const std::vector<double>& b) {
    double s = 0;
    for (std::size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
    return s;
}
}  // namespace

OptResult BoundedLBFGS::optimize(std::vector<double>& x, double& fmin) {
    OptResult res;
    const std::size_t n = n_;
    auto clamp = [&](std::vector<double>& v) {
        for (std::size_t i = 0; i < n; ++i) v[i] = std::min(std::max(v[i], lb_[i]), ub_[i]);
    };
    clamp(x);
    std::vector<double> g(n), xbest = x, gn(n), xn(n), d(n);
    double fbest = std::numeric_limits<double>::infinity();
    int evals = 0;
    auto eval = [&](const std::vector<double>& p, std::vector<double>& grad) {
        ++evals;
        const double v = f_(p, grad, data_);
        if (v < fbest) { fbest = v; xbest = p; }
        return v;
    };
    const int mem = 10;
    std::deque<std::vector<double>> S, Y;
    std::deque<double> R;
    try {
        double f = eval(x, g);
        if (!std::isfinite(f)) res.status = "FAILURE";
        while (res.status.empty() && evals < maxeval_) {
            // free variables: not pinned at a bound by the gradient
            std::vector<char> fr(n, 1);
            double pgnorm = 0;
            for (std::size_t i = 0; i < n; ++i) {
                if ((x[i] <= lb_[i] && g[i] > 0) || (x[i] >= ub_[i] && g[i] < 0)) fr[i] = 0;
                else pgnorm = std::max(pgnorm, std::fabs(g[i]));
            }
            if (pgnorm <= 1e-12 * std::max(1.0, std::fabs(f))) { res.status = "XTOL_REACHED"; break; }
            // two-loop recursion on the free subspace
            for (std::size_t i = 0; i < n; ++i) d[i] = fr[i] ? g[i] : 0.0;
            std::vector<double> al(S.size());
            for (int k = (int)S.size() - 1; k >= 0; --k) {
                al[k] = R[k] * dot(S[k], d);
                for (std::size_t i = 0; i < n; ++i) d[i] -= al[k] * Y[k][i];
            }
            if (!S.empty()) {
                const double gam = dot(S.back(), Y.back()) / dot(Y.back(), Y.back());
                for (std::size_t i = 0; i < n; ++i) d[i] *= gam;
            } else {
                const double sc = 1.0 / std::max(1.0, std::sqrt(dot(d, d)));
                for (std::size_t i = 0; i < n; ++i) d[i] *= sc;
            }
            for (std::size_t k = 0; k < S.size(); ++k) {
                const double be = R[k] * dot(Y[k], d);
                for (std::size_t i = 0; i < n; ++i) d[i] += (al[k] - be) * S[k][i];
            }
            double slope = 0;
            for (std::size_t i = 0; i < n; ++i) { d[i] = fr[i] ? -d[i] : 0.0; slope += g[i] * d[i]; }
            if (!(slope < 0)) {  // not a descent direction: restart from steepest descent
                S.clear(); Y.clear(); R.clear();
                slope = 0;
                for (std::size_t i = 0; i < n; ++i) { d[i] = fr[i] ? -g[i] : 0.0; slope += g[i] * d[i]; }
            }
            double t = 1.0, fn = f;
            bool ok = false;
            while (evals < maxeval_) {
                for (std::size_t i = 0; i < n; ++i) xn[i] = x[i] + t * d[i];
                clamp(xn);
                fn = eval(xn, gn);
                double dec = 0;
                for (std::size_t i = 0; i < n; ++i) dec += g[i] * (xn[i] - x[i]);
                if (std::isfinite(fn) && fn <= f + 1e-4 * dec) { ok = true; break; }
                t *= 0.5;
                if (t < 1e-12) break;
            }
            if (!ok) { res.status = evals >= maxeval_ ? "MAXEVAL_REACHED" : "XTOL_REACHED"; break; }
            std::vector<double> s(n), yv(n);
            for (std::size_t i = 0; i < n; ++i) { s[i] = xn[i] - x[i]; yv[i] = gn[i] - g[i]; }
            const double sy = dot(s, yv);
            if (sy > 1e-12 * std::sqrt(dot(s, s) * dot(yv, yv))) {
                S.push_back(s); Y.push_back(yv); R.push_back(1.0 / sy);
                if ((int)S.size() > mem) { S.pop_front(); Y.pop_front(); R.pop_front(); }
            }
            const double fold = f;
            x = xn; g = gn; f = fn;
            if (std::fabs(fold - f) <= ftol_rel_ * std::max(std::fabs(f), 1e-300)) { res.status = "FTOL_REACHED"; break; }
        }
        if (res.status.empty()) res.status = "MAXEVAL_REACHED";
    } catch (const ForcedStop&) {
        res.status = "FORCED_STOP";
    }
    x = xbest;
    fmin = fbest;
    res.fbest = fbest;
    res.evals = evals;
    return res;
}

}  // namespace gpx
