// gemm_shape_probe.hip — which of (tile count, K length, column stride, footprint) costs the fp64 GEMM its
// last 6 % at n = 32768?  args: groups of "M N K ld [akc bkc flags splitk_work_slabs]" separated by commas, e.g.
//   gemm_shape_probe 16384 16384 16384 16400 , 16384 16384 16384 32784
// build: as gemm_cached_probe.hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gpx_internal.h"
#include "kernels.h"

__global__ void fill_kernel(double* p, long count, unsigned long long seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ULL + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        p[i] = (double)(x >> 11) / 9007199254740992.0 - 0.5;
    }
}

int main(int argc, char** argv) {
    std::vector<std::vector<long>> cases(1);
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], ",")) cases.emplace_back();
        else cases.back().push_back(atol(argv[i]));
    }
    long maxel = 0, maxc = 0;
    for (auto& c : cases) {
        if (c.size() < 4) return 2;
        const bool akc = c.size() > 4 && c[4], bkc = c.size() > 5 && c[5];
        const long ea = c[3] * (akc ? c[0] : c[2]), eb = c[3] * (bkc ? c[1] : c[2]);
        const long el = ea > eb ? ea : eb;
        if (el > maxel) maxel = el;
        if (c[0] * c[1] > maxc) maxc = c[0] * c[1];
    }
    double *A, *B, *C;
    if (hipMalloc(&A, maxel * 8) || hipMalloc(&B, maxel * 8) || hipMalloc(&C, maxc * 8)) return 1;
    fill_kernel<<<4096, 256>>>(A, maxel, 1);
    fill_kernel<<<4096, 256>>>(B, maxel, 2);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (auto& c : cases) {
        const long M = c[0], N = c[1], K = c[2], ld = c[3];
        const int akc = c.size() > 4 ? (int)c[4] : 0, bkc = c.size() > 5 ? (int)c[5] : 0, flags = c.size() > 6 ? (int)c[6] : 0;
        const double fl = 2.0 * M * N * K;
        const int reps = fl < 2e13 ? 6 : 2;
        const long slabs = c.size() > 7 ? c[7] : 0;   // > 0: split-K with a workspace of that many M x N slabs
        double* work = nullptr;
        if (slabs > 0 && hipMalloc(&work, slabs * M * N * 8)) return 1;
        auto run = [&]() {
            if (slabs > 0) gpx_gemm_splitk(st, akc, bkc, M, N, K, 1.0, A, ld, B, ld, 0.0, C, M, flags, work, slabs * M * N);
            else gpx_gemm(st, akc, bkc, M, N, K, 1.0, A, ld, B, ld, 0.0, C, M, flags);
        };
        run();
        (void)hipStreamSynchronize(st);
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) run();
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("M=%6ld N=%6ld K=%6ld ld=%6ld akc=%d bkc=%d flags=%3d  %8.2f ms  %5.1f TF (naive flops)\n", M, N, K, ld, akc, bkc,
               flags, ms / reps, fl * reps / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
