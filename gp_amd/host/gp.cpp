#include "gp.hpp"
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include "../../include/gpx.h"
#include "optimizer.hpp"

namespace gpx {
namespace {
const double INF = std::numeric_limits<double>::infinity();
void must(int rc, const char* what) {
    if (rc) throw std::runtime_error(std::string("gpx: ") + what + " failed with status " + std::to_string(rc));
}
Cov* make_cov(std::size_t d, GP::CovFunc cf) {
    switch (cf) {
        case GP::COV_SE_ARD: return new CovSEard(d);
        case GP::COV_SE_ISO: return new CovSEiso(d);
    }
    std::cerr << "Unknown cov: " << cf << std::endl;  // GP.cpp:583-585
    std::exit(EXIT_FAILURE);
}
}  // namespace

GP::GP(const MatrixXd& train_in, const MatrixXd& train_out, CovFunc cf, MatrixDecomp md)
    : _train_in(train_in), _train_out(std::vector<double>(train_out.data(), train_out.data() + train_out.rows())),
      _cf(cf), _md(md), _cov(make_cov(train_in.rows(), cf)), _num_train(train_in.cols()), _noise_lb(1e-3),
      _dim(train_in.rows()), _num_hyp(_cov->num_hyp() + 2), _hyps_lb(_num_hyp), _hyps_ub(_num_hyp), _trained(false) {
    if (_num_train != train_out.rows()) throw std::invalid_argument("GP: train_in.cols() != train_out.rows()");
    if (const char* e = std::getenv("GPX_DEVICE")) _device = std::atoi(e);   // one rank per GPU: e.g. LOCAL_RANK
    _set_hyp_range();   // host only: the device context is created at the first evaluation (_upload)
}
GP::~GP() {
    delete _cov;
    gpx_ctx_destroy(_ctx);
}
int GP::_model() const { return GPX_MODEL_EXACT; }

void GP::set_device(int device) {
    if (_ctx) throw std::logic_error("GP::set_device: the device context already exists (call before the first evaluation)");
    _device = device;
}
void GP::_upload() {
    if (!_ctx) must(gpx_ctx_create(_device, &_ctx), "ctx_create");
    if (_uploaded) return;
    must(gpx_set_data(_ctx, _train_in.data(), _train_out.data(), (int64_t)_num_train, (int64_t)_dim), "set_data");
    _uploaded = true;
}
void GP::add_data(const MatrixXd& x, const MatrixXd& y) {
    if (x.rows() != _dim || x.cols() != y.rows()) throw std::invalid_argument("add_data: shape");
    _train_in.appendCols(x);
    std::vector<double> o(_train_out.std());
    o.insert(o.end(), y.data(), y.data() + y.rows());
    _train_out = VectorXd(o);
    _num_train += x.cols();
    _trained = false;
    _uploaded = false;
}
void GP::set_noise_lower_bound(double nlb) noexcept {
    if (nlb < 0) {
        std::cerr << "Can't set noise lower bound: lower bound must be positive" << std::endl;
        return;
    }
    _noise_lb = nlb;
    if (_noise_free) std::cerr << "Noise-free GP, ignore the noise lower bound" << std::endl;
    else if (_noise_lb == 0) _noise_lb += std::numeric_limits<double>::epsilon();
}
void GP::set_noise_free(bool flag) {
    _noise_free = flag;
    if (_noise_free) _noise_lb = 0;
}
VectorXd GP::get_default_hyps() const noexcept {
    VectorXd hyp(_num_hyp);
    const VectorXd ch = _cov->default_hyp(_train_in, _train_out);
    for (std::size_t i = 0; i < ch.size(); ++i) hyp[i] = ch[i];
    const double sd = stddev(_train_out.data(), _train_out.size());
    hyp[_num_hyp - 2] = _noise_free ? -INF : std::max(std::log(_noise_lb), std::log(sd * 1e-3));
    double s = 0;
    for (std::size_t i = 0; i < _num_train; ++i) s += _train_out[i];
    hyp[_num_hyp - 1] = s / (double)_num_train;
    return hyp;
}
void GP::_init() {
    _set_hyp_range();
    _trained = false;
}
void GP::_set_hyp_range() {
    const double big = 0.5 * std::log(0.5 * std::numeric_limits<double>::max());
    _hyps_lb = VectorXd::Constant(_num_hyp, -INF);
    _hyps_ub = VectorXd::Constant(_num_hyp, big);
    const auto cr = _cov->cov_hyp_range(_train_in, _train_out);
    const std::size_t nk = _cov->num_hyp();
    for (std::size_t i = 0; i < nk; ++i) {
        _hyps_lb[i] = cr.first[i];
        _hyps_ub[i] = cr.second[i];
    }
    _hyps_lb[_num_hyp - 2] = std::log(_noise_lb);
    _hyps_ub[_num_hyp - 2] = std::max(std::log(10 * _noise_lb), _hyps_ub[nk - 1]);
    double lo = _train_out[0], hi = _train_out[0];
    for (std::size_t i = 1; i < _num_train; ++i) {
        lo = std::min(lo, _train_out[i]);
        hi = std::max(hi, _train_out[i]);
    }
    _hyps_lb[_num_hyp - 1] = lo;
    _hyps_ub[_num_hyp - 1] = hi;
    for (std::size_t i = 0; i < _num_hyp; ++i) {
        _hyps_lb[i] -= std::numeric_limits<double>::epsilon();
        _hyps_ub[i] += std::numeric_limits<double>::epsilon();
    }
}
VectorXd GP::vec2hyp(const std::vector<double>& vx) const {
    VectorXd hyp(_num_hyp);
    if (!_noise_free) {
        for (std::size_t i = 0; i < _num_hyp; ++i) hyp[i] = vx[i];
    } else {
        for (std::size_t i = 0; i < _cov->num_hyp(); ++i) hyp[i] = vx[i];
        hyp[_num_hyp - 2] = -INF;
        hyp[_num_hyp - 1] = vx.back();
    }
    return hyp;
}
std::vector<double> GP::hyp2vec(const VectorXd& hyp) const {
    std::vector<double> vx(_noise_free ? _num_hyp - 1 : _num_hyp);
    for (std::size_t i = 0; i < _cov->num_hyp(); ++i) vx[i] = hyp[i];
    if (_noise_free) vx[_num_hyp - 2] = hyp[_num_hyp - 1];
    else {
        vx[_num_hyp - 2] = hyp[_num_hyp - 2];
        vx[_num_hyp - 1] = hyp[_num_hyp - 1];
    }
    return vx;
}
double GP::_hyp_sn2(const VectorXd& hyp) const { return std::exp(2 * hyp[_num_hyp - 2]); }
double GP::_hyp_mean(const VectorXd& hyp) const { return hyp[_num_hyp - 1]; }

double GP::_calcNegLogProb(const std::vector<double>& x, std::vector<double>& g) const {
    const bool need_g = !g.empty();
    VectorXd eg;
    const double nlz = _calcNegLogProb(vec2hyp(x), eg, need_g);
    if (need_g) g = hyp2vec(eg);
    return nlz;
}
double GP::_calcNegLogProb(const VectorXd& hyp) const {
    VectorXd dummy;
    return _calcNegLogProb(hyp, dummy, false);
}
double GP::_calcNegLogProb(const VectorXd& hyp, VectorXd& grad) const { return _calcNegLogProb(hyp, grad, true); }
double GP::_calcNegLogProb(const VectorXd& hyp, VectorXd& g, bool calc_grad) const {
    const_cast<GP*>(this)->_upload();
    double nlz = INF;
    if (calc_grad) g = VectorXd(_num_hyp);
    must(gpx_nll_grad(_ctx, _model(), _cov->abi_id(), hyp.data(), calc_grad ? 1 : 0, _jitter(), &nlz,
                      calc_grad ? g.data() : nullptr), "nll_grad");
    return nlz;
}
double GP::_objective(const std::vector<double>& x, std::vector<double>& g, void* data) {
    GP* gp = reinterpret_cast<GP*>(data);
    for (double v : x)
        if (std::isnan(v)) throw ForcedStop();  // GP.cpp:188-189
    return gp->_calcNegLogProb(x, g);
}

double GP::train() { return train(get_default_hyps()); }
double GP::train(const VectorXd& init_hyps) {
    _init();
    _hyps = init_hyps;
    if (_noise_free) _hyps[_num_hyp - 2] = -INF;
    double nlz = _calcNegLogProb(_hyps);
    if (!std::isfinite(nlz)) _hyps = select_init_hyp(_num_hyp * 50, _hyps);
    if (!_fixhyps) {
        std::vector<double> lb = hyp2vec(_hyps_lb), ub = hyp2vec(_hyps_ub), hyp0 = hyp2vec(_hyps);
        std::unique_ptr<IOptimizer> opt(_opt_factory ? _opt_factory(hyp0.size()) : make_optimizer(hyp0.size()));   // NLopt LD_SLSQP where built in (GP.cpp:231)
        opt->set_maxeval(_max_eval);
        opt->set_min_objective(&GP::_objective, this);
        opt->set_lower_bounds(lb);
        opt->set_upper_bounds(ub);
        const OptResult r = opt->optimize(hyp0, nlz);
        _last_evals = r.evals;
        _last_status = r.status + " after " + std::to_string(r.evals) + " evaluations (" + opt->name() + ")";
        _hyps = vec2hyp(hyp0);
        if (r.status == "FAILURE") _hyps = _after_optimizer_failure(_hyps);   // the catch blocks of GP.cpp:260-265 / VFE.cpp:94-101
    }
    _setK();
    _trained = true;
    return _calcNegLogProb(_hyps);
}
double GP::last_spd_residual() const {
    double v = 0;
    if (_ctx) must(gpx_last_spd_residual(_ctx, &v), "last_spd_residual");
    return v;
}
void GP::_setK() {
    _upload();
    must(gpx_set_spd_check(_ctx, _spd_cols), "set_spd_check");
    double jit = 0.0;
    must(gpx_prepare(_ctx, GPX_MODEL_EXACT, _cov->abi_id(), _hyps.data(), &jit), "prepare");
}
VectorXd GP::select_init_hyp(std::size_t max_eval, const VectorXd& def_hyp) {
    // MVMO (GP.cpp:463-485) is an un-vendored dependency of the reference.  Stand-in: a
    // deterministic low-discrepancy scan of the box around def_hyp, NLL only, keeping the best
    // finite point and the reference's "sn2 > sf2 => reject" rule.
    VectorXd best = def_hyp;
    double fbest = INF;
    const std::size_t nk = _cov->num_hyp();
    for (std::size_t e = 0; e < max_eval; ++e) {
        VectorXd h = def_hyp;
        for (std::size_t i = 0; i < _num_hyp; ++i) {
            if (_noise_free && i == _num_hyp - 2) continue;
            double lo = std::max(_hyps_lb[i], def_hyp[i] - 3.0), hi = std::min(_hyps_ub[i], def_hyp[i] + 3.0);
            if (i == _num_hyp - 1) { lo = _hyps_lb[i]; hi = _hyps_ub[i]; }
            const double u = std::fmod(0.5 + (double)(e + 1) * 0.6180339887498949 * (double)(i + 1), 1.0);
            h[i] = lo + u * (hi - lo);
        }
        if (_hyp_sn2(h) > std::exp(2 * h[nk - 1])) continue;
        const double f = _calcNegLogProb(h);
        if (f < fbest) { fbest = f; best = h; }
    }
    return best;
}
// GP.cpp:486-507: relative L2 error of the caller's analytic gradient against central differences (eps 1e-3)
double GP::_likelihood_gradient_checking(const VectorXd& hyp, const VectorXd& grad) { return _gradient_error(hyp, grad, 1e-3); }
double GP::check_gradient(const VectorXd& hyp, double epsi) {
    VectorXd g;
    _calcNegLogProb(hyp, g, true);
    return _gradient_error(hyp, g, epsi);
}
double GP::_gradient_error(const VectorXd& hyp, const VectorXd& g, double epsi) {
    double num = 0, den = 0;
    for (std::size_t i = 0; i < _num_hyp; ++i) {
        VectorXd h = hyp;
        h[i] = hyp[i] + epsi;
        const double f1 = _calcNegLogProb(h);
        h[i] = hyp[i] - epsi;
        const double f2 = _calcNegLogProb(h);
        const double fd = (f1 - f2) / (2 * epsi);
        num += (g[i] - fd) * (g[i] - fd);
        den += g[i] * g[i];
    }
    return std::sqrt(num) / std::sqrt(den);
}

void GP::_predict(const MatrixXd& x, VectorXd* y, VectorXd* s2) const {
    assert(_trained);
    if (x.rows() != _dim) throw std::invalid_argument("predict: dimension");
    if (y) *y = VectorXd(x.cols());
    if (s2) *s2 = VectorXd(x.cols());
    must(gpx_predict(_ctx, x.data(), (int64_t)x.cols(), y ? y->data() : nullptr, s2 ? s2->data() : nullptr), "predict");
}
// GP.cpp:284-296, 305-313, 324-333 / FITC.cpp:118-129 / VFE.cpp:118-123, all test points in one call
void GP::_predict_g(const MatrixXd& x, VectorXd* y, VectorXd* s2, MatrixXd* gy, MatrixXd* gs2) const {
    assert(_trained);
    if (x.rows() != _dim) throw std::invalid_argument("predict: dimension");
    if (gs2 && _model() == GPX_MODEL_VFE)
        throw std::logic_error("VFE has no predictive-variance input gradient (VFE.cpp:109-112, 125-131)");
    VectorXd yy(x.cols()), ss(x.cols());
    if (gy) *gy = MatrixXd(_dim, x.cols());
    if (gs2) *gs2 = MatrixXd(_dim, x.cols());
    must(gpx_predict_with_grad(_ctx, x.data(), (int64_t)x.cols(), yy.data(), ss.data(), gy ? gy->data() : nullptr,
                               gs2 ? gs2->data() : nullptr),
         "predict_with_grad");
    if (y) *y = yy;
    if (s2) *s2 = ss;
}
static MatrixXd as_col(const VectorXd& v) { return MatrixXd(v); }
static VectorXd col0(const MatrixXd& m) {
    VectorXd v(m.rows());
    for (long i = 0; i < (long)m.rows(); ++i) v[i] = m(i, 0);
    return v;
}
double GP::predict_y_with_grad(const VectorXd& xs, VectorXd& g) const {
    VectorXd y;
    MatrixXd gg;
    _predict_g(as_col(xs), &y, nullptr, &gg, nullptr);
    g = col0(gg);
    return y[0];
}
double GP::predict_s2_with_grad(const VectorXd& xs, VectorXd& g) const {
    VectorXd s;
    MatrixXd gg;
    _predict_g(as_col(xs), nullptr, &s, nullptr, &gg);
    g = col0(gg);
    return s[0];
}
std::pair<double, VectorXd> GP::predict_y_with_grad(const VectorXd& xs) const {
    VectorXd g;
    const double y = predict_y_with_grad(xs, g);
    return {y, g};
}
std::pair<double, VectorXd> GP::predict_s2_with_grad(const VectorXd& xs) const {
    VectorXd g;
    const double s = predict_s2_with_grad(xs, g);
    return {s, g};
}
void GP::predict_with_grad(const VectorXd& xs, double& y, double& s2, VectorXd& gy, VectorXd& gs2) const {
    VectorXd yy, ss;
    MatrixXd g1, g2;
    _predict_g(as_col(xs), &yy, &ss, &g1, &g2);
    y = yy[0];
    s2 = ss[0];
    gy = col0(g1);
    gs2 = col0(g2);
}
std::tuple<double, double, VectorXd, VectorXd> GP::predict_with_grad(const VectorXd& xs) const {
    double y, s2;
    VectorXd gy, gs2;
    predict_with_grad(xs, y, s2, gy, gs2);
    return std::tuple<double, double, VectorXd, VectorXd>(y, s2, gy, gs2);
}
double GP::predict_y(const VectorXd& xs) const { VectorXd y; _predict(as_col(xs), &y, nullptr); return y[0]; }
double GP::predict_s2(const VectorXd& xs) const { VectorXd s; _predict(as_col(xs), nullptr, &s); return s[0]; }
void GP::predict(const VectorXd& xs, double& y, double& s2) const {
    VectorXd yy, ss;
    _predict(as_col(xs), &yy, &ss);
    y = yy[0];
    s2 = ss[0];
}
std::tuple<double, double> GP::predict(const VectorXd& xs) const {
    double y, s2;
    predict(xs, y, s2);
    return std::tuple<double, double>(y, s2);
}
VectorXd GP::batch_predict_y(const MatrixXd& xs) const { VectorXd y; _predict(xs, &y, nullptr); return y; }
VectorXd GP::batch_predict_s2(const MatrixXd& xs) const { VectorXd s; _predict(xs, nullptr, &s); return s; }
void GP::batch_predict(const MatrixXd& xs, VectorXd& y, VectorXd& s2) const { _predict(xs, &y, &s2); }

// ---- FITC / VFE -------------------------------------------------------------------------------
SparseGP::SparseGP(const MatrixXd& train_in, const MatrixXd& train_out, CovFunc cf, MatrixDecomp md)
    : GP(train_in, train_out, cf, md), _inducing(train_in), _num_inducing(train_in.cols()) {}
int FITC::_model() const { return GPX_MODEL_FITC; }
int VFE::_model() const { return GPX_MODEL_VFE; }
void SparseGP::set_inducing(const MatrixXd& u) {
    if (u.rows() != _dim) throw std::invalid_argument("set_inducing: dimension");
    _inducing = u;
    _num_inducing = u.cols();
    _inducing_uploaded = false;
}
void SparseGP::set_communicator(void* nccl_comm, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("set_communicator: rank/world");
    _comm = nccl_comm;
    _rank = rank;
    _world = world;
    _comm_registered = false;
}
void SparseGP::time_collectives(bool on) {
    _upload_inducing();
    must(gpx_rccl_timing(_ctx, on ? 1 : 0), "rccl_timing");
}
double SparseGP::collective_ms(long* count) {
    _upload_inducing();
    double ms = 0;
    int64_t c = 0;
    must(gpx_rccl_elapsed(_ctx, &ms, &c), "rccl_elapsed");
    if (count) *count = (long)c;
    return ms;
}
std::pair<long, long> shard_range(long n, int rank, int world) {
    if (rank < 0 || rank >= world) throw std::invalid_argument("shard_range: rank out of range");
    return {(long)((long long)n * rank / world), (long)((long long)n * (rank + 1) / world)};
}
void SparseGP::_upload_inducing() {
    _upload();
    if (!_comm_registered) {
        must(gpx_set_allreduce_rccl(_ctx, _comm), "set_allreduce_rccl");
        _comm_registered = true;
    }
    if (_inducing_uploaded) return;
    must(gpx_set_inducing(_ctx, _inducing.data(), (int64_t)_num_inducing), "set_inducing");
    _inducing_uploaded = true;
}
void SparseGP::_init() {
    GP::_init();
    _jitter_u = std::pow(1e-1 * _noise_lb, 2);  // FITC.cpp:30, VFE.cpp:29
}
double SparseGP::_calcNegLogProb(const VectorXd& hyp, VectorXd& g, bool calc_grad) const {
    const_cast<SparseGP*>(this)->_upload_inducing();
    return GP::_calcNegLogProb(hyp, g, calc_grad);
}
void SparseGP::_setK() {
    _upload_inducing();
    must(gpx_prepare(_ctx, _model(), _cov->abi_id(), _hyps.data(), &_jitter_u), "prepare");
}
double SparseGP::train(const VectorXd& hyp) {
    if (_noise_free) {
        std::cerr << "FITC/VFE can't be noise free" << std::endl;  // FITC.cpp:36-40
        set_noise_free(false);
    }
    const double nlz = GP::train(hyp);
    std::cout << _last_status << std::endl;  // explain_nlopt(r), FITC.cpp:102
    return nlz;
}
void SparseGP::test_obj(const VectorXd& hyp) {
    using clk = std::chrono::high_resolution_clock;
    VectorXd grad;
    const auto t1 = clk::now();
    const double nlz = _calcNegLogProb(hyp, grad, true);
    const auto t2 = clk::now();
    const double epsi = _model() == GPX_MODEL_FITC ? 1e-3 : 1e-6;  // FITC.cpp:334, VFE.cpp:264
    VectorXd fd(_num_hyp);
    const auto t3 = clk::now();
    for (std::size_t i = 0; i < _num_hyp; ++i) {
        VectorXd h1 = hyp, h2 = hyp;
        h1[i] += epsi;
        h2[i] -= epsi;
        fd[i] = (GP::_calcNegLogProb(h1) - GP::_calcNegLogProb(h2)) / (2 * epsi);
    }
    const auto t4 = clk::now();
    std::cout << "---------------------------------------" << std::endl;
    for (std::size_t i = 0; i < _num_hyp; ++i) std::cout << hyp[i] << " " << grad[i] << " " << fd[i] << std::endl;
    std::cout << "nlz: " << nlz << std::endl;
    std::cout << "Time grad: " << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() << " ms" << std::endl;
    std::cout << "Time grad_diff: " << std::chrono::duration_cast<std::chrono::milliseconds>(t4 - t3).count() << " ms" << std::endl;
}

// ---- text IO ------------------------------------------------------------------------------------
MatrixXd read_matrix(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("read_matrix: cannot open " + path);
    std::vector<std::vector<double>> rows;
    std::string line;
    while (std::getline(f, line)) {
        std::stringstream s(line);
        std::vector<double> r;
        std::string tok;
        while (s >> tok) r.push_back(std::stod(tok));
        if (r.empty() && rows.empty()) continue;
        if (!rows.empty() && r.size() != rows[0].size()) {
            if (r.empty()) continue;
            std::cerr << "Invalid vector read from file, dim not match" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        rows.push_back(r);
    }
    if (rows.empty()) {
        std::cerr << "Empty matrix in file " << path << std::endl;
        std::exit(EXIT_FAILURE);
    }
    MatrixXd m(rows.size(), rows[0].size());
    for (std::size_t i = 0; i < rows.size(); ++i)
        for (std::size_t j = 0; j < rows[0].size(); ++j) m(i, j) = rows[i][j];
    return m;
}
void write_pred(const std::string& path, const VectorXd& mean, const VectorXd& s2) {
    std::ofstream f(path);   // default stream precision (6 significant digits), as `f << rec` in mygp.cpp:32-36
    for (std::size_t i = 0; i < mean.size(); ++i) f << mean[i] << " " << std::sqrt(s2[i]) << "\n";
}

}  // namespace gpx
