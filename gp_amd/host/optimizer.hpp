// optimizer.hpp — the optimiser seam of GP::train (GP.cpp:231-265).
//
// The reference drives its objective with NLopt's LD_SLSQP through the C-style callback
//     double f(const std::vector<double>& x, std::vector<double>& g, void* data)
// (g.empty() => no gradient wanted; GP.cpp:186-191).  NLopt is not a dependency here.  The same
// callback shape is kept and a built-in bound-constrained projected L-BFGS (with the reference's
// maxeval budget and "best point seen" result) stands in; an NLopt adaptor can be plugged behind
// the same interface by a user who has NLopt.
#pragma once
#include <string>
#include <vector>

namespace gpx {

typedef double (*ObjectiveFn)(const std::vector<double>& x, std::vector<double>& g, void* data);

struct OptResult {
    double fbest = 0;
    int evals = 0;
    std::string status;  // "FTOL_REACHED", "XTOL_REACHED", "MAXEVAL_REACHED", "FORCED_STOP", "FAILURE"
};

struct ForcedStop {};  // thrown by a callback to stop the optimiser (nlopt::forced_stop's role, GP.cpp:188-189)

class BoundedLBFGS {
public:
    explicit BoundedLBFGS(std::size_t n) : n_(n), lb_(n, -1e300), ub_(n, 1e300) {}
    void set_maxeval(int m) { maxeval_ = m; }
    void set_min_objective(ObjectiveFn f, void* data) { f_ = f; data_ = data; }
    void set_lower_bounds(const std::vector<double>& lb) { lb_ = lb; }
    void set_upper_bounds(const std::vector<double>& ub) { ub_ = ub; }
    void set_ftol_rel(double t) { ftol_rel_ = t; }
    // minimises from x (clamped into the bounds); on return x is the best point evaluated
    OptResult optimize(std::vector<double>& x, double& fmin);

private:
    std::size_t n_;
    std::vector<double> lb_, ub_;
    ObjectiveFn f_ = nullptr;
    void* data_ = nullptr;
    int maxeval_ = 160;
    double ftol_rel_ = 1e-10;
};

}  // namespace gpx
