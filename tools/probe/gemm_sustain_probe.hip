// gemm_sustain_probe.hip — does the fp64 GEMM rate decay with the LENGTH of the run (power / clock management)
// rather than with the operand footprint?  Runs M=N=K=n back to back for `secs` seconds and prints the rate of
// every group of launches, for real leading dimensions and for lda=ldb=0 (L2-resident operands).
// build: as gemm_cached_probe.hip
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpx_internal.h"
#include "kernels.h"

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 16384;
    const double secs = argc > 2 ? atof(argv[2]) : 4.0;
    const long ld = n + 16;
    double *A, *B, *C;
    if (hipMalloc(&A, ld * n * 8) || hipMalloc(&B, ld * n * 8) || hipMalloc(&C, n * n * 8)) return 1;
    {
        std::vector<double> h((size_t)ld * n);
        unsigned long long x = 88172645463325252ULL;
        for (auto& v : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v = (double)(x >> 11) / 9007199254740992.0 - 0.5;
        }
        hipMemcpy(A, h.data(), ld * n * 8, hipMemcpyHostToDevice);
        for (auto& v : h) v = -v * 0.7;
        hipMemcpy(B, h.data(), ld * n * 8, hipMemcpyHostToDevice);
    }
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double fl = 2.0 * n * n * n;
    const int group = fl < 1e13 ? 4 : 1;
    for (int pass = 0; pass < 2; ++pass) {
        const long l = pass ? 0 : ld;
        gpx_gemm(st, 0, 0, n, n, n, 1.0, A, l, B, l, 0.0, C, n, 0);
        hipStreamSynchronize(st);
        printf("n=%ld %s:", n, pass ? "ld=0 " : "real ");
        double t = 0;
        while (t < secs) {
            hipEventRecord(e0, st);
            for (int i = 0; i < group; ++i) gpx_gemm(st, 0, 0, n, n, n, 1.0, A, l, B, l, 0.0, C, n, 0);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            t += ms * 1e-3;
            printf(" %.1f", fl * group / (ms * 1e-3) / 1e12);
        }
        printf(" TF\n");
        fflush(stdout);
    }
    return 0;
}
