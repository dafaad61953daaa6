// dist_runner.cpp — row-sharded FITC / VFE through the C++ facade and RCCL, one process per GPU, no Python
// (north star: "Host code stays C++ ... sharded by training rows across the 8 GPUs of one node, with an RCCL all-reduce
// over xGMI").  No reference counterpart: the reference is single-process.
//
//   gpx_dist --gpus N [--model fitc|vfe] [--n rows] [--m inducing] [--d dims] [--steps K] [--warmup W] [--no-t1]
//
// The parent never touches the GPU: it starts N children of its own image (posix_spawn, each told its rank through
// the environment) BEFORE any HIP call, waits for them and returns their worst exit status.  Rank 0 creates the
// ncclUniqueId and hands it over in a file; every rank binds to GPU rank % device_count, builds ITS rows of the
// synthetic problem of SURVEY.md 8(d) (gp_amd/host/synth.hpp = gp_amd/synth.py), registers the communicator with the
// facade object (SparseGP::set_communicator) and times K NLL+gradient evaluations between two barriers.  Rank 0
// prints one JSON line: per-rank ms, device ms inside the collectives per evaluation, and — with t_1 measured in the
// same run on rank 0's GPU alone — the parallel efficiency t_1 / (N t_N).
#include <rccl/rccl.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/gpx.h"
#include "../gp.hpp"
#include "../synth.hpp"

extern char** environ;
using namespace gpx;
using clk = std::chrono::steady_clock;

struct Args {
    int gpus = 1, d = 20, steps = 5, warmup = 1;
    long n = 1L << 20, m = 1024;
    std::string model = "fitc";
    bool t1 = true;
};

static Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (k == "--gpus") a.gpus = atoi(val());
        else if (k == "--model") a.model = val();
        else if (k == "--n") a.n = atol(val());
        else if (k == "--m") a.m = atol(val());
        else if (k == "--d") a.d = atoi(val());
        else if (k == "--steps") a.steps = atoi(val());
        else if (k == "--warmup") a.warmup = atoi(val());
        else if (k == "--no-t1") a.t1 = false;
        else { fprintf(stderr, "gpx_dist: unknown argument %s\n", k.c_str()); exit(2); }
    }
    if (a.model == "vfe" && a.n == (1L << 20) && a.m == 1024 && a.d == 20) { a.n = 1L << 22; a.m = 2048; a.d = 50; }   // BASELINE configs[4]
    return a;
}

static int parent(const Args& a, char** argv) {
    const std::string idfile = "/tmp/gpx_dist_id_" + std::to_string((long)getpid());
    std::vector<pid_t> pids;
    for (int r = 0; r < a.gpus; ++r) {
        std::vector<std::string> env;
        for (char** e = environ; *e; ++e) env.push_back(*e);
        env.push_back("GPX_DIST_RANK=" + std::to_string(r));
        env.push_back("GPX_DIST_WORLD=" + std::to_string(a.gpus));
        env.push_back("GPX_DIST_IDFILE=" + idfile);
        env.push_back("HSA_ENABLE_IPC_MODE_LEGACY=0");
        std::vector<char*> envp;
        for (auto& s : env) envp.push_back(const_cast<char*>(s.c_str()));
        envp.push_back(nullptr);
        pid_t pid = 0;
        if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, argv, envp.data()) != 0) {
            perror("gpx_dist: posix_spawn");
            return 1;
        }
        pids.push_back(pid);
    }
    int worst = 0;
    for (pid_t p : pids) {
        int st = 0;
        waitpid(p, &st, 0);
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        worst = std::max(worst, rc);
    }
    unlink(idfile.c_str());
    return worst;
}

#define NCCL_MUST(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "gpx_dist rank %d: %s -> %s\n", rank, #x, ncclGetErrorString(r_)); return 3; } } while (0)

struct Problem {
    std::vector<double> X, y, U, hyp;
    long n;
};
static Problem make(const Args& a, long r0, long r1) {
    const uint64_t seed = 20240601ULL + (a.model == "fitc" ? 3 : 4);
    Problem p;
    p.n = r1 - r0;
    p.hyp = synth::eval_hyps(a.d);
    p.X = synth::make_inputs(seed, p.n, a.d, r0);
    p.y = synth::make_targets(seed, p.X, p.n, a.d, p.hyp, r0);
    p.U = synth::make_inputs(seed, a.m, a.d, 0);   // inducing points: the first m training points
    return p;
}
template <class M>
static double evaluate(M& gp, const Problem& p, VectorXd& g) {
    VectorXd h(p.hyp);
    return gp.nll(h, g);
}

template <class M>
static int worker_model(const Args& a, int rank, int world, int device, ncclComm_t comm, double* bar) {
    const auto rr = shard_range(a.n, rank, world);
    Problem p = make(a, rr.first, rr.second);
    MatrixXd X((std::size_t)a.d, (std::size_t)p.n), y((std::size_t)p.n, 1), U((std::size_t)a.d, (std::size_t)a.m);
    std::memcpy(X.data(), p.X.data(), p.X.size() * sizeof(double));
    std::memcpy(y.data(), p.y.data(), p.y.size() * sizeof(double));
    std::memcpy(U.data(), p.U.data(), p.U.size() * sizeof(double));
    auto barrier = [&]() {   // a one-double all-reduce on the NULL stream + a device synchronisation
        ncclAllReduce(bar, bar, 1, ncclDouble, ncclSum, comm, nullptr);
        gpx_device_synchronize();
    };
    double nll = 0, ms = 0, coll_ms = 0;
    long ncoll = 0;
    VectorXd g;
    {
        M gp(X, y);
        gp.set_device(device);
        gp.set_inducing(U);
        gp.set_communicator(comm, rank, world);
        for (int i = 0; i < std::max(a.warmup, 1); ++i) nll = evaluate(gp, p, g);
        gp.time_collectives(true);
        barrier();
        const auto t0 = clk::now();
        for (int i = 0; i < a.steps; ++i) nll = evaluate(gp, p, g);
        gpx_device_synchronize();
        ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count() / a.steps;
        barrier();
        coll_ms = gp.collective_ms(&ncoll) / a.steps;
    }
    // per-rank times: every rank fills its slot of a zeroed vector, one sum-all-reduce
    std::vector<double> slots(2 * world, 0.0);
    slots[rank] = ms;
    slots[world + rank] = coll_ms;
    gpx_memcpy_h2d(bar + 8, slots.data(), (int64_t)slots.size() * 8);
    ncclAllReduce(bar + 8, bar + 8, slots.size(), ncclDouble, ncclSum, comm, nullptr);
    gpx_device_synchronize();
    gpx_memcpy_d2h(slots.data(), bar + 8, (int64_t)slots.size() * 8);
    double t1 = 0, nll1 = 0;
    if (rank == 0 && world > 1 && a.t1) {   // the whole problem on this GPU alone, no communicator, same run
        Problem f = make(a, 0, a.n);
        MatrixXd Xf((std::size_t)a.d, (std::size_t)f.n), yf((std::size_t)f.n, 1);
        std::memcpy(Xf.data(), f.X.data(), f.X.size() * sizeof(double));
        std::memcpy(yf.data(), f.y.data(), f.y.size() * sizeof(double));
        M gp(Xf, yf);
        gp.set_device(device);
        gp.set_inducing(U);
        VectorXd g1;
        nll1 = evaluate(gp, f, g1);
        gpx_device_synchronize();
        const auto t0 = clk::now();
        for (int i = 0; i < a.steps; ++i) nll1 = evaluate(gp, f, g1);
        gpx_device_synchronize();
        t1 = std::chrono::duration<double, std::milli>(clk::now() - t0).count() / a.steps;
    }
    barrier();
    if (rank == 0) {
        const double tN = *std::max_element(slots.begin(), slots.begin() + world);
        const double flops = (a.model == "fitc" ? 6.0 : 3.0) * (double)a.n * (double)a.m * (double)a.m;
        printf("{\"runner\": \"gpx_dist (C++ facade + RCCL, one process per GPU)\", \"model\": \"%s\", \"n\": %ld, \"m\": %ld, \"d\": %d, "
               "\"world\": %d, \"steps\": %d, \"nll\": %.17g, \"grad0\": %.17g, \"ms_per_eval\": %.4f, \"evals_per_s\": %.4f, "
               "\"algorithmic_tflops_total\": %.2f, \"rccl\": \"%s\", \"collectives_per_eval\": %.1f, \"per_rank_ms\": [",
               a.model.c_str(), a.n, a.m, a.d, world, a.steps, nll, g[0], tN, 1e3 / tN, flops / (tN * 1e-3) / 1e12, gpx_rccl_source(),
               (double)ncoll / a.steps);
        for (int r = 0; r < world; ++r) printf("%s%.4f", r ? ", " : "", slots[r]);
        printf("], \"collective_ms_per_eval_per_rank\": [");
        for (int r = 0; r < world; ++r) printf("%s%.4f", r ? ", " : "", slots[world + r]);
        printf("]");
        if (t1 > 0)
            printf(", \"t1_ms_single_gpu_same_run\": %.4f, \"parallel_efficiency\": %.4f, \"nll_single_gpu\": %.17g, \"nll_rel_diff\": %.3g",
                   t1, t1 / (world * tN), nll1, std::fabs(nll - nll1) / std::fabs(nll1));
        printf("}\n");
        fflush(stdout);
    }
    return 0;
}

static int worker(const Args& a, int rank, int world, const char* idfile) {
    int ndev = 0;
    if (gpx_device_count(&ndev) || ndev < 1) { fprintf(stderr, "gpx_dist rank %d: no GPU\n", rank); return 2; }
    const int device = rank % ndev;
    if (gpx_set_device(device)) return 2;
    ncclUniqueId id;
    if (rank == 0) {
        NCCL_MUST(ncclGetUniqueId(&id));
        if (world > 1) {
            const std::string tmp = std::string(idfile) + ".tmp";
            std::ofstream f(tmp, std::ios::binary);
            f.write(id.internal, sizeof(id.internal));
            f.close();
            rename(tmp.c_str(), idfile);
        }
    } else {
        bool got = false;
        for (int i = 0; i < 1200 && !got; ++i) {   // up to 120 s
            std::ifstream f(idfile, std::ios::binary);
            if (f.good() && f.read(id.internal, sizeof(id.internal))) got = true;
            else std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) { fprintf(stderr, "gpx_dist rank %d: no ncclUniqueId in %s\n", rank, idfile); return 3; }
    }
    ncclComm_t comm = nullptr;
    NCCL_MUST(ncclCommInitRank(&comm, world, id, rank));
    void* bar = nullptr;
    if (gpx_malloc(&bar, (8 + 2 * world) * 8) || gpx_memset_zero(bar, (8 + 2 * world) * 8)) return 2;
    int rc;
    try {
        rc = a.model == "fitc" ? worker_model<FITC>(a, rank, world, device, comm, (double*)bar)
                               : worker_model<VFE>(a, rank, world, device, comm, (double*)bar);
    } catch (const std::exception& e) {
        fprintf(stderr, "gpx_dist rank %d: %s\n", rank, e.what());
        rc = 4;
    }
    gpx_free(bar);
    ncclCommDestroy(comm);
    return rc;
}

int main(int argc, char** argv) {
    const Args a = parse(argc, argv);
    const char* er = getenv("GPX_DIST_RANK");
    if (!er && a.gpus > 1) return parent(a, argv);   // no HIP call has been made in this process
    const int rank = er ? atoi(er) : 0, world = er ? atoi(getenv("GPX_DIST_WORLD")) : 1;
    return worker(a, rank, world, er ? getenv("GPX_DIST_IDFILE") : "");
}
