// tools/chunkquality.hip -- is the "good / bad scatter target" property of a buffer (tools/placeprobe.hip: a bad buffer even STREAMS
// writes 20 % slower) a property of its physical chunks?  Creates up to K physical chunks of C MiB through the HIP virtual-memory API,
// all alive together, maps each by itself and times a streaming write and a streaming read on it; prints the rates in creation order.
// usage: chunkquality [chunk MiB = 1024] [K = 200]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void wr(v2d *d, size_t n)
{
    v2d v; v.x = 1.0; v.y = 2.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = v;
}
__global__ __launch_bounds__(512) void wr_group(v2d **tab, int G, size_t n)
{
    v2d v; v.x = 1.0; v.y = 2.0;
    v2d *d = tab[blockIdx.x % G];
    const size_t wg = blockIdx.x / G, nwg = gridDim.x / G;
    for (size_t i = wg * (size_t)blockDim.x + threadIdx.x; i < n; i += nwg * blockDim.x) d[i] = v;
}
__global__ __launch_bounds__(512) void rd(const v2d *d, size_t n, double *sink)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const v2d v = __builtin_nontemporal_load(d + i); s += v.x + v.y; }
    if (s == 12345.678) *sink = s;
}
int main(int argc, char **argv)
{
    const size_t cm = argc > 1 ? (size_t)atoll(argv[1]) : 1024, chunk = cm << 20;
    const int K = argc > 2 ? atoi(argv[2]) : 200;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<void *> va;
    double *sink; HIPCHK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int k = 0; k < K; k++) {
        size_t fr = 0, tot = 0; HIPCHK(hipMemGetInfo(&fr, &tot));
        if (fr < chunk + (4ull << 30)) break;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) break;
        void *p = nullptr;
        HIPCHK(hipMemAddressReserve(&p, chunk, chunk, nullptr, 0));
        HIPCHK(hipMemMap(p, chunk, 0, h, 0)); HIPCHK(hipMemRelease(h)); HIPCHK(hipMemSetAccess(p, chunk, &acc, 1));
        va.push_back(p);
    }
    printf("%zu chunks of %zu MiB alive; per chunk: write TB/s, read TB/s (best of 3)\n", va.size(), cm);
    const size_t n = chunk / 16;
    for (size_t k = 0; k < va.size(); k++) {
        float bw = 1e30f, br = 1e30f;
        for (int r = 0; r < 4; r++) {
            float ms;
            HIPCHK(hipEventRecord(e0)); hipLaunchKernelGGL(wr, dim3(2048), dim3(512), 0, 0, (v2d *)va[k], n); HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < bw) bw = ms;
            HIPCHK(hipEventRecord(e0)); hipLaunchKernelGGL(rd, dim3(2048), dim3(512), 0, 0, (const v2d *)va[k], n, sink); HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < br) br = ms;
        }
        printf("%3zu  %5.2f  %5.2f%s", k, chunk / bw / 1e9, chunk / br / 1e9, (k % 4 == 3) ? "\n" : "   |  ");
    }
    printf("\n");
    // pairs / groups of chunks written at the same time (workgroup w writes chunk w % G): does spreading the write front over distant
    // physical chunks lift the rate above what one contiguous chunk gives?
    if (va.size() >= 200) {
        v2d **tab; HIPCHK(hipMalloc(&tab, 64 * sizeof(v2d *)));
        const int dist[] = {1, 2, 8, 32, 64, 128};
        for (int G : {2, 4, 16}) {
            for (int d : dist) {
                if ((size_t)(G - 1) * d + 1 > va.size()) continue;
                std::vector<v2d *> h(G);
                for (int g = 0; g < G; g++) h[g] = (v2d *)va[(size_t)g * d];
                HIPCHK(hipMemcpy(tab, h.data(), G * sizeof(v2d *), hipMemcpyHostToDevice));
                float best = 1e30f;
                for (int r = 0; r < 4; r++) {
                    float ms;
                    HIPCHK(hipEventRecord(e0)); hipLaunchKernelGGL(wr_group, dim3(4096), dim3(512), 0, 0, tab, G, n); HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
                    HIPCHK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
                }
                printf("write %2d chunks at once, %3d chunks apart: %5.2f TB/s\n", G, d, (double)G * chunk / best / 1e9);
            }
        }
    }
    return 0;
}
