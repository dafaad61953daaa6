// gemm_cached_probe.hip — is the fp64 GEMM limited by operand fetch latency or by issue?
// Runs gpx_gemm on M=N=K=n once with real leading dimensions and once with lda=ldb=0 (every K
// column aliases the first one: operands stay L1/L2-resident, same instruction stream).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gp_amd/csrc -I include -c tools/probe/gemm_cached_probe.hip -o /tmp/p.o && hipcc --offload-arch=gfx950 /tmp/p.o gp_amd/csrc/obj/gemm_f64.o -o tools/probe/gemm_cached_probe
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpx_internal.h"
#include "kernels.h"

static double run(hipStream_t st, long n, long ld, const double* A, const double* B, double* C, int akc, int bkc, int flags) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    gpx_gemm(st, akc, bkc, n, n, n, 1.0, A, ld, B, ld, 0.0, C, n, flags);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 3; ++i) gpx_gemm(st, akc, bkc, n, n, n, 1.0, A, ld, B, ld, 0.0, C, n, flags);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * n * n * n * 3 / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 16384;
    const long ld = n + 16;
    double *A, *B, *C;
    if (hipMalloc(&A, ld * n * 8) || hipMalloc(&B, ld * n * 8) || hipMalloc(&C, n * n * 8)) return 1;
    if (argc > 2) {  // random operands: the data-dependent power draw of a real evaluation
        std::vector<double> h((size_t)ld * n);
        unsigned long long x = 88172645463325252ULL;
        for (auto& v : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v = (double)(x >> 11) / 9007199254740992.0 - 0.5;
        }
        hipMemcpy(A, h.data(), ld * n * 8, hipMemcpyHostToDevice);
        for (auto& v : h) v = -v * 0.7;
        hipMemcpy(B, h.data(), ld * n * 8, hipMemcpyHostToDevice);
    } else {
        hipMemset(A, 0x3c, ld * n * 8);  // constant finite doubles
        hipMemset(B, 0x3d, ld * n * 8);
    }
    hipStream_t st;
    hipStreamCreate(&st);
    for (int v = 0; v < 4; ++v) {
        const int akc = v >> 1, bkc = v & 1;
        printf("akc=%d bkc=%d  real ld: %6.1f TF   ld=0 (cached operands): %6.1f TF   register-staged loop: %6.1f TF\n", akc, bkc,
               run(st, n, ld, A, B, C, akc, bkc, 0), run(st, n, 0, A, B, C, akc, bkc, 0),
               run(st, n, ld, A, B, C, akc, bkc, GEMM_NO_DLDS));
        fflush(stdout);
    }
    return 0;
}
