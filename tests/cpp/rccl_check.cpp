// rccl_check.cpp — the native RCCL path behind the C++ facade at world size 1 (one GPU is what the test box has):
// a FITC / VFE object with a registered ncclComm_t must reproduce the communicator-less NLL and gradient BIT FOR BIT
// (a one-rank sum all-reduce is the identity), the collectives must actually have been issued (count > 0), and _setK +
// predict go through the same path.  Prints `key value...` lines for tests/test_gpu_rccl.py.
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include "../../gp_amd/host/gp.hpp"
#include "../../gp_amd/host/synth.hpp"
#include "../../include/gpx.h"
using namespace gpx;

template <class M>
static int run(const char* tag, ncclComm_t comm, long n, long m, int d) {
    const uint64_t seed = 20240601ULL + 7;
    const std::vector<double> hyp = synth::eval_hyps(d);
    const std::vector<double> Xv = synth::make_inputs(seed, n, d), yv = synth::make_targets(seed, Xv, n, d, hyp);
    const std::vector<double> Xs = synth::make_inputs(seed, 33, d, n);
    MatrixXd X((std::size_t)d, (std::size_t)n), y((std::size_t)n, 1), U((std::size_t)d, (std::size_t)m), T((std::size_t)d, 33);
    std::memcpy(X.data(), Xv.data(), Xv.size() * 8);
    std::memcpy(y.data(), yv.data(), yv.size() * 8);
    std::memcpy(U.data(), Xv.data(), (std::size_t)m * d * 8);
    std::memcpy(T.data(), Xs.data(), Xs.size() * 8);
    VectorXd h(hyp), g0, g1;
    double f0, f1, p0, p1;
    long ncoll = 0;
    {
        M gp(X, y);
        gp.set_inducing(U);
        f0 = gp.nll(h, g0);
        gp.set_fixed(true);
        gp.train(h);
        p0 = gp.batch_predict_y(T)[7] + gp.batch_predict_s2(T)[7];
    }
    {
        M gp(X, y);
        gp.set_inducing(U);
        gp.set_communicator(comm, 0, 1);
        gp.time_collectives(true);
        f1 = gp.nll(h, g1);
        gp.collective_ms(&ncoll);
        gp.set_fixed(true);
        gp.train(h);
        p1 = gp.batch_predict_y(T)[7] + gp.batch_predict_s2(T)[7];
    }
    bool same = f0 == f1 && g0.size() == g1.size() && p0 == p1;
    for (std::size_t i = 0; same && i < g0.size(); ++i) same = g0[i] == g1[i];
    std::printf("%s_nll %.17g %.17g\n%s_bit_identical %d\n%s_collectives %ld\n", tag, f0, f1, tag, (int)same, tag, ncoll);
    return same ? 0 : 1;
}

int main() {
    ncclUniqueId id;
    ncclComm_t comm = nullptr;
    if (gpx_set_device(0)) return 2;
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) {
        std::fprintf(stderr, "rccl_check: cannot create a one-rank communicator\n");
        return 3;
    }
    int rc = run<FITC>("fitc", comm, 5000, 200, 6);
    rc |= run<VFE>("vfe", comm, 5000, 200, 6);
    std::printf("rccl_source %s\n", gpx_rccl_source());
    ncclCommDestroy(comm);
    return rc;
}
