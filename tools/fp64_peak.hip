// tools/fp64_peak.hip -- measured FP64 vector FMA rate of the device: the COMPUTE roof the butterfly arithmetic of the axis
// passes is priced against in bench.py (roofline.compute).  The in-image guide lists no FP64 figure (MI355X_MICROARCH.md); the
// public specification is 78.6 TFLOP/s FP64 vector = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz, so it is measured here:
// every thread runs 16 independent chains of fused multiply-adds from registers, 2 flops per FMA and lane.
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_peak.hip -o tools/fp64_peak && tools/fp64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <typename T, int CHAINS, int ITERS> __global__ __launch_bounds__(256) void fma_chains(T *out, T a, T b)
{
    T v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = (T)(threadIdx.x + c);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) v[c] = __builtin_fma(v[c], a, b);
    }
    T s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    if (s == (T)12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;      // never true: keeps the chains alive
}

template <typename T> static int measure(const char *name)
{
    constexpr int CHAINS = 16, ITERS = 4096;
    int cus = 0;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * 8 * 4;
    T *out;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(T)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    fma_chains<T, CHAINS, ITERS><<<blocks, 256>>>(out, (T)0.999, (T)0.001);
    CHK(hipDeviceSynchronize());
    double best = 0;
    for (int rep = 0; rep < 5; rep++) {
        CHK(hipEventRecord(e0, nullptr));
        fma_chains<T, CHAINS, ITERS><<<blocks, 256>>>(out, (T)0.999, (T)0.001);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double tf = 2.0 * CHAINS * ITERS * 256.0 * blocks / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    printf("%s vector FMA: %.1f TFLOP/s (%d CUs, %d workgroups of 256 threads, %d chains x %d FMAs per thread, best of 5)\n", name, best, cus, blocks, CHAINS, ITERS);
    CHK(hipFree(out));
    return 0;
}

int main()
{
    CHK(hipSetDevice(0));
    int clk = 0;
    CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("device clock rate attribute: %.2f GHz\n", clk / 1e6);
    return measure<double>("FP64");
}
