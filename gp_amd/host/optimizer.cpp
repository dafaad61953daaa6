// optimizer.cpp — projected L-BFGS with Armijo backtracking along the projection arc, and the NLopt adaptor.
#include "optimizer.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#if defined(__has_include)
#if __has_include(<nlopt.hpp>)
#include <nlopt.hpp>
#define GPX_HAVE_NLOPT 1
#endif
#endif

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

#ifdef GPX_HAVE_NLOPT
namespace {
class NloptSLSQP : public IOptimizer {
    nlopt::opt opt_;
    ObjectiveFn f_ = nullptr;
    void* data_ = nullptr;
    int evals_ = 0;
    double fbest_ = std::numeric_limits<double>::infinity();
    std::vector<double> xbest_;
    static double tramp(const std::vector<double>& x, std::vector<double>& g, void* self) {
        NloptSLSQP* o = static_cast<NloptSLSQP*>(self);
        ++o->evals_;
        double v;
        try {
            v = o->f_(x, g, o->data_);
        } catch (const ForcedStop&) {
            throw nlopt::forced_stop();  // GP.cpp:188-189
        }
        if (v < o->fbest_) { o->fbest_ = v; o->xbest_ = x; }
        return v;
    }

public:
    explicit NloptSLSQP(std::size_t n) : opt_(nlopt::LD_SLSQP, (unsigned)n) {}   // GP.cpp:231
    const char* name() const override { return "NLopt LD_SLSQP"; }
    void set_maxeval(int m) override { opt_.set_maxeval(m); }                   // GP.cpp:232
    void set_min_objective(ObjectiveFn f, void* data) override {
        f_ = f;
        data_ = data;
        opt_.set_min_objective(&NloptSLSQP::tramp, this);                       // GP.cpp:233
    }
    void set_lower_bounds(const std::vector<double>& lb) override { opt_.set_lower_bounds(lb); }   // GP.cpp:234
    void set_upper_bounds(const std::vector<double>& ub) override { opt_.set_upper_bounds(ub); }   // GP.cpp:235
    OptResult optimize(std::vector<double>& x, double& fmin) override {
        OptResult r;
        const std::vector<double> lb = opt_.get_lower_bounds(), ub = opt_.get_upper_bounds();
        for (std::size_t i = 0; i < x.size(); ++i) x[i] = std::min(std::max(x[i], lb[i]), ub[i]);   // GP.cpp:208-215
        try {
            const nlopt::result res = opt_.optimize(x, fmin);                    // GP.cpp:257
            r.status = res == nlopt::FTOL_REACHED ? "FTOL_REACHED" : res == nlopt::XTOL_REACHED ? "XTOL_REACHED"
                     : res == nlopt::MAXEVAL_REACHED ? "MAXEVAL_REACHED" : res == nlopt::SUCCESS ? "SUCCESS" : "STOPVAL_OR_TIME";
        } catch (const nlopt::forced_stop&) {
            r.status = "FORCED_STOP";
        } catch (const std::runtime_error&) {                                    // swallowed as GP.cpp:260-265
            r.status = "FAILURE";
        }
        if (!xbest_.empty()) { x = xbest_; fmin = fbest_; }
        r.fbest = fmin;
        r.evals = evals_;
        return r;
    }
};
}  // namespace
bool nlopt_available() { return true; }
#else
bool nlopt_available() { return false; }
#endif

IOptimizer* make_optimizer(std::size_t n) {
#ifdef GPX_HAVE_NLOPT
    const char* e = std::getenv("GPX_OPTIMIZER");
    if (!(e && std::strcmp(e, "lbfgs") == 0)) return new NloptSLSQP(n);
#endif
    return new BoundedLBFGS(n);
}

}  // namespace gpx
