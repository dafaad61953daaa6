// synth.hpp — the deterministic synthetic SE-ARD data of SURVEY.md section 8(d) in C++ (the same counter-based
// splitmix64 streams as gp_amd/synth.py, so a C++ host program and bench.py evaluate the same problem; values agree
// with the Python generator to the last bits of cos / log, not bit for bit).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace gpx {
namespace synth {

inline double uniform_at(uint64_t seed, uint64_t pos) {   // u in [0,1) at stream position pos (0-based)
    uint64_t z = seed + (pos + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
inline double normal_at(uint64_t seed, uint64_t start, uint64_t i) {   // Box-Muller on uniforms start + 2i, start + 2i + 1
    const double u1 = std::fmax(uniform_at(seed, start + 2 * i), 1e-300), u2 = uniform_at(seed, start + 2 * i + 1);
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}
// log l_k = 0.5 log d + 0.1 sin k, log sf = 0, log sn = log 0.1, mean = 0
inline std::vector<double> eval_hyps(int d) {
    std::vector<double> h(d + 3);
    for (int k = 0; k < d; ++k) h[k] = 0.5 * std::log((double)d) + 0.1 * std::sin((double)k);
    h[d] = 0.0;
    h[d + 1] = std::log(0.1);
    h[d + 2] = 0.0;
    return h;
}
// points start_point .. start_point + n - 1 of the stream, d x n column-major (one point = d contiguous doubles)
inline std::vector<double> make_inputs(uint64_t seed, long n, int d, long start_point = 0) {
    std::vector<double> X((size_t)n * d);
    const uint64_t base = (uint64_t)start_point * d;
#pragma omp parallel for
    for (long i = 0; i < n * d; ++i) X[i] = std::sqrt(3.0) * (2.0 * uniform_at(seed, base + i) - 1.0);
    return X;
}
// y_a = mean + sf sqrt(2/D) sum_j cos(w_j . x_a + b_j) + sn N(0,1), D = 256 random Fourier features
inline std::vector<double> make_targets(uint64_t seed, const std::vector<double>& X, long n, int d, const std::vector<double>& hyp,
                                        long start_point = 0, int D = 256) {
    const uint64_t fseed = seed ^ 0x5EEDULL;
    std::vector<double> w((size_t)D * d), b(D), y(n);
    for (long i = 0; i < (long)D * d; ++i) w[i] = normal_at(fseed, 0, i) / std::exp(hyp[i % d]);
    for (int j = 0; j < D; ++j) b[j] = 2.0 * M_PI * uniform_at(fseed, 2ULL * D * d + j);
    const double sf = std::exp(hyp[d]), sn = std::exp(hyp[d + 1]), mean = hyp[d + 2];
#pragma omp parallel for
    for (long a = 0; a < n; ++a) {
        double s = 0;
        for (int j = 0; j < D; ++j) {
            double t = b[j];
            for (int k = 0; k < d; ++k) t += X[(size_t)a * d + k] * w[(size_t)j * d + k];
            s += std::cos(t);
        }
        y[a] = mean + sf * std::sqrt(2.0 / D) * s + sn * normal_at(seed ^ 0xA11CEULL, 2ULL * start_point, a);
    }
    return y;
}

}  // namespace synth
}  // namespace gpx
