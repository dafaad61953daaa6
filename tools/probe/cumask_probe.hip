// Probe: does hipExtStreamCreateWithCUMask work on this runtime?  Creates a stream restricted to the first 32 CUs, runs a
// kernel that records which CU (HW_ID) each workgroup ran on, and prints the number of distinct (XCC, SE, CU) ids seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where_kernel(unsigned* out, unsigned* xcc) {
    if (threadIdx.x == 0) {
        unsigned hw, xc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
        out[blockIdx.x] = hw;
        xcc[blockIdx.x] = xc;
    }
    // stay a little so that workgroups spread
    double s = 0;
    for (int i = 0; i < 20000; ++i) s = fma(s, 1.0000001, 1e-9);
    if (s == 12345.0) out[0] = 0;
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    std::printf("CUs %d\n", p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    for (int mode = 0; mode < 2; ++mode) {
        hipStream_t st;
        hipError_t e;
        if (mode == 0) e = hipStreamCreate(&st);
        else {
            std::vector<uint32_t> mask(words, 0u);
            mask[0] = 0xffffffffu;   // 32 CUs
            e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data());
        }
        std::printf("mode %d create: %s\n", mode, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        unsigned *d, *x;
        const int nb = 4096;
        hipMalloc(&d, nb * 4);
        hipMalloc(&x, nb * 4);
        hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, st, d, x);
        e = hipStreamSynchronize(st);
        std::printf("mode %d sync: %s\n", mode, hipGetErrorString(e));
        std::vector<unsigned> h(nb), hx(nb);
        hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), x, nb * 4, hipMemcpyDeviceToHost);
        std::set<unsigned long> ids;
        for (int i = 0; i < nb; ++i) ids.insert(((unsigned long)(hx[i] & 0xf) << 32) | (h[i] & 0x0000ff00u) | ((h[i] >> 13) & 0x7) << 16);
        std::printf("mode %d distinct (xcc, se, cu) ids: %zu\n", mode, ids.size());
        hipStreamDestroy(st);
        hipFree(d);
        hipFree(x);
    }
    return 0;
}
