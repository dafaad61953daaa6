// demo.cpp — the three command-line programs of the reference (mygp.cpp, test_fitc.cpp,
// test_vfe.cpp) against this library: `gp [num_train]`, `fitc [num_train [num_inducing]]`,
// `vfe [num_train [num_inducing]]`.  They read the whitespace text matrices train_x, train_y and
// test_x from the current directory and write `pred` (two columns: mean, std).
// Built three times with -DDEMO_MODEL=0|1|2.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include "../gp.hpp"
using namespace gpx;

int main(int argc, char** argv) {
    std::cout << std::setprecision(9);
    const MatrixXd train_x = read_matrix("train_x").transpose();
    const MatrixXd train_y = read_matrix("train_y");
    const MatrixXd test_x = read_matrix("test_x").transpose();
    long num_train = (long)train_x.cols(), num_inducing = (long)train_x.cols();
    if (argc > 1) num_train = atol(argv[1]);
    if (argc > 2) num_inducing = atol(argv[2]);
    if (num_train > (long)train_x.cols() || num_inducing > (long)train_x.cols()) {
        std::cerr << "Num train > " << train_x.cols() << std::endl;
        return EXIT_FAILURE;
    }
    MatrixXd y(num_train, 1);
    for (long i = 0; i < num_train; ++i) y(i, 0) = train_y(i, 0);
    // GPX_DEMO_FIXED=1 (not in the reference's CLI): set_fixed(true), i.e. train() only runs _setK at the start
    // hypers, so that tests can pin the `pred` file at known hypers
    const bool fixed = std::getenv("GPX_DEMO_FIXED") && std::getenv("GPX_DEMO_FIXED")[0] == '1';
    const auto t1 = std::chrono::high_resolution_clock::now();
#if DEMO_MODEL == 0
    GP gp(train_x.leftCols(num_train), y, GP::COV_SE_ARD, GP::QR);
    gp.set_fixed(fixed);
    const double nlz = gp.train();
#else
#if DEMO_MODEL == 1
    FITC gp(train_x.leftCols(num_train), y, GP::COV_SE_ARD, GP::QR);
    VectorXd init = gp.get_default_hyps();
#else
    VFE gp(train_x.leftCols(num_train), y, GP::COV_SE_ARD, GP::QR);
    VectorXd init = gp.get_default_hyps();
    init[init.size() - 2] = std::log(stddev(train_y.data(), train_y.rows()));  // test_vfe.cpp:38-39: std of the WHOLE train_y
#endif
    gp.set_inducing(train_x.rightCols(num_inducing));
    gp.set_fixed(fixed);
    const double nlz = gp.train(init);
#endif
    const auto t2 = std::chrono::high_resolution_clock::now();
    std::cerr << "optimizer: " << gp.last_optimizer_status() << std::endl;   // not in the reference's output: stderr
    std::cout << "Training time: " << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() * 1e-3
              << " seconds" << std::endl;
    std::cout << "Negative log likelihood: " << nlz << std::endl;
    std::cout << "Optimized hyperparameters:" << std::endl;
    for (std::size_t i = 0; i < gp.get_hyp().size(); ++i) std::cout << gp.get_hyp()[i] << std::endl;
#if DEMO_MODEL != 0
    gp.test_obj(gp.get_hyp());
#endif
    const auto t3 = std::chrono::high_resolution_clock::now();
    const VectorXd predy = gp.batch_predict_y(test_x);
    const VectorXd preds2 = gp.batch_predict_s2(test_x);
    const auto t4 = std::chrono::high_resolution_clock::now();
    std::cout << "Prediction time: " << std::chrono::duration_cast<std::chrono::microseconds>(t4 - t3).count() * 1e-6
              << " seconds" << std::endl;
    write_pred("pred", predy, preds2);
    return EXIT_SUCCESS;
}
