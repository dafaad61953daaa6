// gp.hpp — GP / FITC / VFE with the reference's class API (GP.h:14-122, FITC.h, VFE.h), host C++
// over the C ABI of libgpx.so.  Training data lives on the GPU from construction; the virtual
// _calcNegLogProb / _setK / _predict* hooks (seam (3)) each make ONE library call.
//
// Deviations from the reference, all deliberate:
//  * MatrixDecomp::QR selects the same Cholesky as ::Cholesky (SPD inputs).
//  * _check_SPD's eigenvalue / |I - K K^-1| test (GP.cpp:445-461) is replaced by "every Cholesky
//    pivot > 0"; the noise / jitter escalation loops are reproduced as coded.
//  * train() goes through the IOptimizer seam (optimizer.hpp): NLopt LD_SLSQP where <nlopt.hpp> exists at
//    build time, else the built-in bounded L-BFGS (NLopt is not in this image), and
//    select_init_hyp() (MVMO, an un-vendored submodule of the reference) falls back to a
//    deterministic scan over the bounds' interior.  Same budgets (160/130/150 evaluations).
//  * VFE::predict_with_grad / predict_s2_with_grad throw std::logic_error where the reference
//    has assert(not need_g) / leaves the gradient unset (VFE.cpp:109-112, 125-131).
#pragma once
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "cov.hpp"
#include "dense.hpp"
#include "matrix_solver.hpp"

struct gpx_ctx;

namespace gpx {

class GP {
public:
    enum CovFunc { COV_SE_ARD, COV_SE_ISO };
    enum MatrixDecomp { Cholesky, QR };

protected:
    MatrixXd _train_in;   // dim x num_data
    VectorXd _train_out;  // num_data
    CovFunc _cf;
    MatrixDecomp _md;
    Cov* _cov;
    std::size_t _num_train;
    double _noise_lb;
    const std::size_t _dim;
    const std::size_t _num_hyp;
    VectorXd _hyps, _hyps_lb, _hyps_ub;
    bool _fixhyps = false;
    bool _noise_free = false;
    bool _trained;
    int _max_eval = 160;       // GP.cpp:232
    std::string _last_status;  // optimiser status of the last train()
    int _last_evals = 0;
    gpx_ctx* _ctx = nullptr;   // created on _device at the first evaluation
    int _device = 0;
    bool _uploaded = false;
    int _spd_cols = 0;
    class IOptimizer* (*_opt_factory)(std::size_t) = nullptr;

    virtual int _model() const;  // GPX_MODEL_*
    virtual double _jitter() const { return 0.0; }
    virtual void _init();
    virtual void _setK();
    // what train() does when the optimiser itself fails (nlopt throws std::runtime_error): GP and FITC swallow it and
    // keep the point reached (GP.cpp:260-265, FITC.cpp:94-100), VFE re-runs the start search with num_hyp * 10
    // evaluations from that point (VFE.cpp:94-101)
    virtual VectorXd _after_optimizer_failure(const VectorXd& reached) { return reached; }
    void _upload();
    double _calcNegLogProb(const std::vector<double>&, std::vector<double>& g) const;
    double _calcNegLogProb(const VectorXd& hyp) const;
    double _calcNegLogProb(const VectorXd& hyp, VectorXd&) const;
    virtual double _calcNegLogProb(const VectorXd& hyp, VectorXd& g, bool calc_grad) const;
    virtual void _predict(const MatrixXd& x, VectorXd* y, VectorXd* s2) const;
    // with input gradients (need_g = true in GP.h:62-64): gy / gs2 are dim x num_test
    virtual void _predict_g(const MatrixXd& x, VectorXd* y, VectorXd* s2, MatrixXd* gy, MatrixXd* gs2) const;
    void _set_hyp_range();
    VectorXd vec2hyp(const std::vector<double>& vx) const;
    std::vector<double> hyp2vec(const VectorXd& vx) const;
    double _hyp_sn2(const VectorXd& hyp) const;
    double _hyp_mean(const VectorXd& hyp) const;
    double _likelihood_gradient_checking(const VectorXd& hyp, const VectorXd& grad);
    double _gradient_error(const VectorXd& hyp, const VectorXd& grad, double eps);
    static double _objective(const std::vector<double>& x, std::vector<double>& g, void* data);

public:
    GP(const MatrixXd& train_in, const MatrixXd& train_out, CovFunc cf = COV_SE_ARD, MatrixDecomp md = Cholesky);
    virtual ~GP();
    GP(const GP&) = delete;
    GP& operator=(const GP&) = delete;
    void add_data(const MatrixXd& train_in, const MatrixXd& train_out);

    std::size_t dim() const noexcept { return _dim; }
    std::size_t num_hyp() const noexcept { return _num_hyp; }
    std::size_t num_train() const noexcept { return _num_train; }
    VectorXd get_default_hyps() const noexcept;
    const MatrixXd& train_in() const noexcept { return _train_in; }
    const VectorXd& train_out() const noexcept { return _train_out; }
    bool trained() const noexcept { return _trained; }
    bool noise_free() const { return _noise_free; }
    void set_noise_lower_bound(double) noexcept;
    void set_fixed(bool f) { _fixhyps = f; }
    // evaluation budget of train() (defaults: 160 / 130 / 150 as GP.cpp:232, FITC.cpp:75, VFE.cpp:74) and the optimiser
    // behind the seam (default: make_optimizer; tests inject their own)
    void set_max_eval(int m) { _max_eval = m; }
    void set_optimizer_factory(class IOptimizer* (*factory)(std::size_t)) { _opt_factory = factory; }
    void set_noise_free(bool flag);
    const std::string& last_optimizer_status() const { return _last_status; }
    int last_optimizer_evals() const { return _last_evals; }
    // GPU of this object (default 0, or the environment variable GPX_DEVICE): one rank per GPU sets its
    // LOCAL_RANK here before the first evaluation
    void set_device(int device);
    int device() const { return _device; }
    // residual part of _check_SPD (GP.cpp:455-457) inside _setK of the exact GP: mean|I - K K^-1| over `cols` sampled
    // columns must be < 1e-4, else the noise escalates; 0 (default) = Cholesky success only (see the header comment)
    void set_spd_check(int cols) { _spd_cols = cols; }
    double last_spd_residual() const;
    // the optimiser's box (GP.cpp:514-534) and the packing of the hyper vector it sees (GP.cpp:535-564:
    // a noise-free GP drops the noise slot)
    const VectorXd& hyp_lower_bounds() const { return _hyps_lb; }
    const VectorXd& hyp_upper_bounds() const { return _hyps_ub; }
    std::vector<double> pack_hyp(const VectorXd& hyp) const { return hyp2vec(hyp); }
    VectorXd unpack_hyp(const std::vector<double>& v) const { return vec2hyp(v); }

    virtual double train(const VectorXd& init_hyps);
    double train();
    const VectorXd& get_hyp() const noexcept { return _hyps; }
    // analytic NLL + gradient at hyp (what the optimiser sees) and the relative error of the
    // gradient against central differences (GP.cpp:486-507, FITC::test_obj, VFE::test_obj)
    double nll(const VectorXd& hyp, VectorXd& grad) const { return _calcNegLogProb(hyp, grad, true); }
    double check_gradient(const VectorXd& hyp, double eps = 1e-3);

    double predict_y(const VectorXd& xs) const;
    double predict_s2(const VectorXd& xs) const;
    double predict_y_with_grad(const VectorXd& xs, VectorXd& g) const;
    double predict_s2_with_grad(const VectorXd& xs, VectorXd& g) const;
    std::pair<double, VectorXd> predict_y_with_grad(const VectorXd& xs) const;
    std::pair<double, VectorXd> predict_s2_with_grad(const VectorXd& xs) const;
    void predict(const VectorXd& xs, double& y, double& s2) const;
    void predict_with_grad(const VectorXd& xs, double& y, double& s2, VectorXd& gy, VectorXd& gs2) const;
    std::tuple<double, double> predict(const VectorXd& xs) const;
    std::tuple<double, double, VectorXd, VectorXd> predict_with_grad(const VectorXd& xs) const;
    VectorXd batch_predict_y(const MatrixXd& xs) const;
    VectorXd batch_predict_s2(const MatrixXd& xs) const;
    void batch_predict(const MatrixXd& xs, VectorXd& y, VectorXd& s2) const;
    VectorXd select_init_hyp(std::size_t max_eval, const VectorXd& def_hyp);
};

class SparseGP : public GP {
protected:
    MatrixXd _inducing;
    std::size_t _num_inducing;
    double _jitter_u = 1e-8;
    bool _inducing_uploaded = false;
    void* _comm = nullptr;   // ncclComm_t of the row-sharded run (set_communicator)
    int _rank = 0, _world = 1;
    bool _comm_registered = false;
    double _jitter() const override { return _jitter_u; }
    void _init() override;
    void _setK() override;
    void _upload_inducing();
    double _calcNegLogProb(const VectorXd& hyp, VectorXd& g, bool calc_grad) const override;
    SparseGP(const MatrixXd& train_in, const MatrixXd& train_out, CovFunc cf, MatrixDecomp md);

public:
    void set_inducing(const MatrixXd& u);
    // Row-sharded evaluation, one process per GPU (no reference counterpart: the reference is single-process; north star
    // "Kuf.Kfu and diag corrections sharded by training rows ... RCCL all-reduce over xGMI").  THIS object holds the rows
    // shard_range(n, rank, world) of the training set (the caller constructs it from its shard), the inducing points and
    // the hypers are the same on every rank; nccl_comm is the caller's ncclComm_t (ncclCommInitRank, one rank per GPU).
    // The library then issues the sum-all-reduces of every evaluation (VFE two; FITC three: the m x m part of the second
    // one runs beside the trace kernel) and the one of _setK itself, on its own streams; NLL and gradient come out identical on every rank.  Call before the first evaluation or any time after.
    void set_communicator(void* nccl_comm, int rank, int world);
    int rank() const { return _rank; }
    int world() const { return _world; }
    // device time inside the collectives since the last call (two events around each; enable with time_collectives(true))
    void time_collectives(bool on);
    double collective_ms(long* count = nullptr);
    double train(const VectorXd& init_hyps) override;
    using GP::train;
    void test_obj(const VectorXd& hyp);  // prints [hyp, grad, grad_fd] and both timings
};

class FITC : public SparseGP {
    int _model() const override;

public:
    FITC(const MatrixXd& train_in, const MatrixXd& train_out, CovFunc cf = COV_SE_ARD, MatrixDecomp md = QR)
        : SparseGP(train_in, train_out, cf, md) { _max_eval = 130; }  // FITC.cpp:75
};

class VFE : public SparseGP {
    int _model() const override;
    VectorXd _after_optimizer_failure(const VectorXd& reached) override { return select_init_hyp(_num_hyp * 10, reached); }   // VFE.cpp:99

public:
    VFE(const MatrixXd& train_in, const MatrixXd& train_out, CovFunc cf = COV_SE_ARD, MatrixDecomp md = QR)
        : SparseGP(train_in, train_out, cf, md) { _max_eval = 150; }  // VFE.cpp:74
};

// rows [first, last) of rank's block of n training rows split over world ranks (the same split as gp_amd/sharding.py)
std::pair<long, long> shard_range(long n, int rank, int world);

// whitespace text matrix IO of the demos (util.cpp:14-62, mygp.cpp:32-36)
MatrixXd read_matrix(const std::string& path);
void write_pred(const std::string& path, const VectorXd& mean, const VectorXd& s2);

}  // namespace gpx
