// optimizer.hpp — the optimiser seam of GP::train (GP.cpp:231-265).
//
// The reference drives its objective with NLopt's LD_SLSQP through the C-style callback
//     double f(const std::vector<double>& x, std::vector<double>& g, void* data)
// (g.empty() => no gradient wanted; GP.cpp:186-191).  NLopt is not a dependency here.  The same
// callback shape is kept and a built-in bound-constrained projected L-BFGS (with the reference's
// maxeval budget and "best point seen" result) stands in behind the IOptimizer seam; where <nlopt.hpp>
// is present at build time an NLopt LD_SLSQP adaptor is compiled behind the same seam and preferred.
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

// The optimiser seam: what GP::train needs from nlopt::opt (GP.cpp:231-265).
class IOptimizer {
public:
    virtual ~IOptimizer() {}
    virtual const char* name() const = 0;
    virtual void set_maxeval(int m) = 0;
    virtual void set_min_objective(ObjectiveFn f, void* data) = 0;
    virtual void set_lower_bounds(const std::vector<double>& lb) = 0;
    virtual void set_upper_bounds(const std::vector<double>& ub) = 0;
    // minimises from x (clamped into the bounds); on return x is the best point evaluated
    virtual OptResult optimize(std::vector<double>& x, double& fmin) = 0;
};

class BoundedLBFGS : public IOptimizer {
public:
    explicit BoundedLBFGS(std::size_t n) : n_(n), lb_(n, -1e300), ub_(n, 1e300) {}
    const char* name() const override { return "built-in bounded L-BFGS"; }
    void set_maxeval(int m) override { maxeval_ = m; }
    void set_min_objective(ObjectiveFn f, void* data) override { f_ = f; data_ = data; }
    void set_lower_bounds(const std::vector<double>& lb) override { lb_ = lb; }
    void set_upper_bounds(const std::vector<double>& ub) override { ub_ = ub; }
    void set_ftol_rel(double t) { ftol_rel_ = t; }
    OptResult optimize(std::vector<double>& x, double& fmin) override;

private:
    std::size_t n_;
    std::vector<double> lb_, ub_;
    ObjectiveFn f_ = nullptr;
    void* data_ = nullptr;
    int maxeval_ = 160;
    double ftol_rel_ = 1e-10;
};

// nlopt::opt(nlopt::LD_SLSQP, n) exactly as GP.cpp:231-238 sets it up; compiled only where <nlopt.hpp> exists
// (link with -lnlopt).  A callback that throws ForcedStop is translated to nlopt::forced_stop and back.
bool nlopt_available();
// The optimiser GP::train uses: NLopt LD_SLSQP when this library was built with NLopt (and the environment
// variable GPX_OPTIMIZER is not "lbfgs"), else the built-in bounded L-BFGS.  Caller owns the result.
IOptimizer* make_optimizer(std::size_t n);

}  // namespace gpx
