// mfma_probe.hip — issue rate of v_mfma_f64_16x16x4_f64 by waves per SIMD and independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void probe(double* out, long long* cyc, int iters) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[(long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd, int cus) {
    const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd);  // 256 threads = 1 wave per SIMD
    const int blocks_per_cu = waves_per_simd > 4 ? waves_per_simd / 4 : 1;
    const int blocks = cus * blocks_per_cu;
    const int iters = 16384 / NACC;
    double* out;
    long long* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NACC><<<blocks, threads>>>(out, cyc, 64);
    hipEventRecord(e0);
    probe<NACC><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= h.size();
    const double nm = (double)iters * NACC;
    const double tf = (double)h.size() * nm * 2048 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d  acc %2d  CUs %3d : %7.1f cyc/MFMA per wave -> %6.1f cyc/MFMA per SIMD, %6.1f TFLOP/s, %.3f ms\n",
           waves_per_simd, NACC, cus, mean / nm, mean / nm / waves_per_simd, tf, ms);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int cus : {1, 256}) {
        for (int w : {1, 2, 4, 8}) {
            run<1>(w, cus);
            run<4>(w, cus);
            run<16>(w, cus);
        }
    }
    return 0;
}
