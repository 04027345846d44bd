// tools/placeprobe.hip -- does a cheap LOCAL probe of a buffer predict how fast the plan's scatter passes run on it?
// The passes that scatter 128-byte runs are bimodal per buffer (5.7 vs 6.1-6.5 ms at 1024^3 fp64, profiles/r4_placement_spread_1gib.txt);
// dfft_tune_placement finds good buffers by executing the plan on candidates, which is collective on a multi-rank plan.  A probe that
// needs nothing but the buffer could run inside the allocator, on every rank by itself.  This tool allocates K candidate `out` buffers
// (dfft_malloc, default backing), times two probe kernels on each -- a scatter of 128-byte runs with the x pass's geometry, and a
// plain streaming write -- and then the plan's own x pass with that buffer as its target.
// usage: placeprobe [K = 8]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>

#include "../include/dfft_c.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define DCHK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, dfft_last_error()); exit(3); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// the x pass's store: workgroup (a, b) writes 1024 rows of 128 B (8 lanes x 16 B) at row stride S; 512 threads, 16 stores each
__global__ __launch_bounds__(512) void scatter_probe(v2d *dst, uint32_t nb, uint64_t row_stride, uint64_t a_stride)
{
    const uint32_t w = blockIdx.x, a = w / nb, b = w % nb;
    const int tid = threadIdx.x, l = tid & 7, t = tid >> 3;
    v2d v; v.x = (double)tid; v.y = (double)w;
    v2d *p = dst + (uint64_t)a * a_stride + (uint64_t)b * 8 + l;
#pragma unroll
    for (int c = 0; c < 16; c++) p[(uint64_t)(t + 64 * c) * row_stride] = v;
}
__global__ __launch_bounds__(512) void stream_probe(v2d *dst, size_t n)
{
    v2d v; v.x = 1.0; v.y = 2.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

static float timed(hipEvent_t e0, hipEvent_t e1, const std::function<void()> &f)
{
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        HIPCHK(hipEventRecord(e0));
        f();
        HIPCHK(hipEventRecord(e1));
        HIPCHK(hipEventSynchronize(e1));
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    const size_t N = 1024, n = N * N * N, bytes = n * 16;
    dfft_plan *plan;
    DCHK(dfft_plan_create(&plan, DFFT_PENCIL_OPT1, DFFT_F64, nullptr, nullptr, 0, -1));
    DCHK(dfft_init(plan, N, N, N, 1, 1, 1, 1));
    void *in;
    HIPCHK(hipMalloc(&in, bytes));
    HIPCHK(hipMemset(in, 0, bytes));
    DCHK(dfft_enable_phase_timing(plan, 1));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    std::vector<void *> cand(K);
    printf("candidate  scatter probe ms   stream probe ms   plan: y-FFT ms  x-FFT ms (x writes this buffer)\n");
    for (int c = 0; c < K; c++) DCHK(dfft_malloc(dfft_domain_size(plan), DFFT_CHUNK_DEFAULT, &cand[c]));      // all alive together
    for (int c = 0; c < K; c++) {
        v2d *d = (v2d *)cand[c];
        // rows of 1024*1024 points, 128 columns of 8 points per row block, 1024 row blocks (a): the API layout [kx][ky][kz]
        const float ts = timed(e0, e1, [&] { hipLaunchKernelGGL(scatter_probe, dim3(128 * 1024), dim3(512), 0, 0, d, 128u, (uint64_t)N * N, (uint64_t)N); });
        const float tw = timed(e0, e1, [&] { hipLaunchKernelGGL(stream_probe, dim3(4096), dim3(512), 0, 0, d, n); });
        float ph[8] = {0}, y = 1e30f, x = 1e30f;
        for (int r = 0; r < 4; r++) {
            DCHK(dfft_exec_c2c(plan, cand[c], in, DFFT_FORWARD));
            dfft_get_phase_times(plan, ph, 8);
            if (r) { if (ph[2] < y) y = ph[2]; if (ph[4] < x) x = ph[4]; }
        }
        printf("%9d  %16.3f  %16.3f  %14.3f  %8.3f\n", c, ts, tw, y, x);
    }
    return 0;
}
