// base_probe.hip — phase timing of the 128-row factor-and-invert base kernel (chol.hip) on one random SPD block.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPX_BASE_TIMING -I gp_amd/csrc tools/probe/base_probe.hip gp_amd/csrc/gemm_f64.hip gp_amd/csrc/vecops.hip -o tools/probe/base_probe
#include "../../gp_amd/csrc/chol.hip"
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 128, ld = 144;
    std::vector<double> B(nb * nb), A((size_t)ld * nb, 0.0);
    srand(1);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            double s = i == j ? nb : 0.0;
            for (int k = 0; k < nb; ++k) s += B[i + k * nb] * B[j + k * nb];
            A[i + (size_t)j * ld] = s;
        }
    double *dA, *dW;
    int* dinfo;
    hipMalloc(&dA, sizeof(double) * ld * nb);
    hipMalloc(&dW, sizeof(double) * ld * nb);
    hipMalloc(&dinfo, sizeof(int));
    hipMemset(dinfo, 0, sizeof(int));
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), sizeof(double) * ld * nb, hipMemcpyHostToDevice);
        hipMemset(dW, 0, sizeof(double) * ld * nb);
        hipLaunchKernelGGL(potrf_inv_base_kernel<false>, dim3(1), dim3(512), 0, 0, dA, (long)ld, nb, dW, (long)ld, dinfo);
        hipDeviceSynchronize();
        long long t[64] = {0};
#ifdef GPX_BASE_TIMING
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_base_t), sizeof(t));
#endif
        auto us = [&](int a, int b) { return (t[b] - t[a]) * 0.01; };
        printf("nb %d: load %.2f  factor %.2f  diagonal inverses %.2f  block inverse | L store %.2f  W store %.2f  total %.2f us\n", nb,
               us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 5));
        if (rep == 2) {
            printf("  block inverse + W store of waves 0-7 done after");
            for (int w = 0; w < 8; ++w) printf(" %.2f", us(3, 40 + w));
            printf(" us; SIMD of waves 0-7:");
            for (int w = 0; w < 8; ++w) printf(" %lld", (t[50 + w] >> 4) & 3);
            printf("\n");
        }
        if (rep == 2)
            for (int k = 0; k < (nb + 15) / 16; ++k)   // per sub-block column: panel | wave 0: diagonal update + factorisation | wait for the other waves
                printf("  k %d: panel %.2f  wave-0 update+chol %.2f  wait %.2f us\n", k, us(8 + 3 * k, 9 + 3 * k), us(9 + 3 * k, 10 + 3 * k),
                       k + 1 < (nb + 15) / 16 ? us(10 + 3 * k, 11 + 3 * k) : 0.0);
    }
    if (argc > 2) {   // accuracy on an ill-conditioned kernel matrix (nb <= 64): new and old base kernels against long double
        const int n = nb <= 64 ? nb : 64;
        std::vector<double> x(n), K((size_t)ld * n, 0.0);
        for (auto& v : x) v = rand() / (double)RAND_MAX;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) K[i + (size_t)j * ld] = exp(-0.5 * (x[i] - x[j]) * (x[i] - x[j]) / 0.09) + (i == j ? atof(argv[2]) : 0.0);
        // long double reference: L and W = L^-1
        std::vector<long double> Lr(n * n, 0.0L), Wr(n * n, 0.0L);
        for (int j = 0; j < n; ++j) {
            long double d = K[j + (size_t)j * ld];
            for (int k = 0; k < j; ++k) d -= Lr[j + k * n] * Lr[j + k * n];
            Lr[j + j * n] = sqrtl(d);
            for (int i = j + 1; i < n; ++i) {
                long double v = K[i + (size_t)j * ld];
                for (int k = 0; k < j; ++k) v -= Lr[i + k * n] * Lr[j + k * n];
                Lr[i + j * n] = v / Lr[j + j * n];
            }
        }
        for (int c = 0; c < n; ++c)
            for (int i = c; i < n; ++i) {
                long double v = i == c ? 1.0L : 0.0L;
                for (int k = c; k < i; ++k) v -= Lr[i + k * n] * Wr[k + c * n];
                Wr[i + c * n] = v / Lr[i + i * n];
            }
        for (int which = 0; which < 2; ++which) {
            hipMemcpy(dA, K.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice);
            hipMemset(dW, 0, sizeof(double) * ld * nb);
            if (which == 0) hipLaunchKernelGGL(potrf_inv_base_kernel<false>, dim3(1), dim3(512), 0, 0, dA, (long)ld, n, dW, (long)ld, dinfo);
            else hipLaunchKernelGGL(potrf_base_tile_kernel, dim3(1), dim3(256), 0, 0, dA, (long)ld, n, dW, (long)ld, 0, dinfo);
            hipDeviceSynchronize();
            std::vector<double> L((size_t)ld * n), W((size_t)ld * n);
            hipMemcpy(L.data(), dA, sizeof(double) * ld * n, hipMemcpyDeviceToHost);
            hipMemcpy(W.data(), dW, sizeof(double) * ld * n, hipMemcpyDeviceToHost);
            long double el = 0, ew = 0, nl = 0, nw = 0;
            for (int j = 0; j < n; ++j)
                for (int i = j; i < n; ++i) {
                    el = fmaxl(el, fabsl(L[i + (size_t)j * ld] - Lr[i + j * n]));
                    nl = fmaxl(nl, fabsl(Lr[i + j * n]));
                    ew = fmaxl(ew, fabsl(W[i + (size_t)j * ld] - Wr[i + j * n]));
                    nw = fmaxl(nw, fabsl(Wr[i + j * n]));
                }
            printf("%s kernel: max|L - Lref|/max|L| = %.3Lg   max|W - Wref|/max|W| = %.3Lg\n", which ? "old 64-row" : "new", el / nl, ew / nw);
        }
    }
    // check W L = I
    std::vector<double> L((size_t)ld * nb), W((size_t)ld * nb);
    hipMemcpy(L.data(), dA, sizeof(double) * ld * nb, hipMemcpyDeviceToHost);
    hipMemcpy(W.data(), dW, sizeof(double) * ld * nb, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = j; k <= i; ++k) s += W[i + (size_t)k * ld] * L[k + (size_t)j * ld];
            err = fmax(err, fabs(s - (i == j ? 1.0 : 0.0)));
        }
    printf("max |W L - I| = %.3g\n", err);
    return 0;
}
