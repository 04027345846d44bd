// tools/mallprobe.hip -- can the 256 MiB Infinity Cache carry the intermediate of two consecutive axis passes?
// A 3-D transform on one GPU writes the z pass's output (16 GiB at 1024^3 fp64) and reads it back in the y pass.  If both
// passes are cut into chunks of a few x planes and run back to back (z(c), y(c), z(c+1), ...), the intermediate of a chunk
// may still sit in the memory-side cache when the second pass reads it.  This probe measures exactly that with plain
// copies:   stage 1: IN[c] -> A[c]     stage 2: A[c] -> B[c]
// over a total of `GiB` per buffer, for several chunk sizes, on one stream and on two streams (stage 2 of chunk c
// overlapping stage 1 of chunk c+1), against the unchunked order (all of stage 1, then all of stage 2).
// usage: mallprobe [GiB per buffer = 4] [repeats = 3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// 512 threads x 16 points of 16 B = 128 KiB per workgroup, contiguous (the footprint of one fp64 1024-point tile)
template <int NT_LOAD, int NT_STORE> __global__ __launch_bounds__(512) void copy_kernel(const v2d *__restrict__ src, v2d *__restrict__ dst)
{
    const v2d *s = src + (uint64_t)blockIdx.x * 8192 + threadIdx.x;
    v2d *d = dst + (uint64_t)blockIdx.x * 8192 + threadIdx.x;
    v2d v[16];
#pragma unroll
    for (int c = 0; c < 16; c++) v[c] = NT_LOAD ? __builtin_nontemporal_load(s + 512 * c) : s[512 * c];
#pragma unroll
    for (int c = 0; c < 16; c++) { if (NT_STORE) __builtin_nontemporal_store(v[c], d + 512 * c); else d[512 * c] = v[c]; }
}

using kern_t = void (*)(const v2d *, v2d *);

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 4;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t bytes = gib << 30;
    char *in, *A, *B;
    HIPCHK(hipMalloc(&in, bytes)); HIPCHK(hipMalloc(&A, bytes)); HIPCHK(hipMalloc(&B, bytes));
    HIPCHK(hipMemset(in, 1, bytes)); HIPCHK(hipMemset(A, 0, bytes)); HIPCHK(hipMemset(B, 0, bytes));
    hipStream_t s1, s2;
    HIPCHK(hipStreamCreate(&s1)); HIPCHK(hipStreamCreate(&s2));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    std::vector<hipEvent_t> ev(4096);
    for (auto &e : ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    struct Mode { const char *name; kern_t k1, k2; };
    // stage 1 reads a stream it never needs again (nt load); its store is what stage 2 wants to find on chip
    const Mode modes[] = {
        {"plain/plain      ", copy_kernel<0, 0>, copy_kernel<0, 0>},
        {"ntload / ntstore ", copy_kernel<1, 0>, copy_kernel<0, 1>},
        {"nt all           ", copy_kernel<1, 1>, copy_kernel<1, 1>},
    };
    printf("%zu GiB per buffer; time of stage 1 (IN -> A) + stage 2 (A -> B); 4 x %zu GiB of traffic if nothing is reused\n", gib, gib);
    for (const Mode &m : modes) {
        for (size_t chunk_mib : {(size_t)0, (size_t)16, (size_t)32, (size_t)64, (size_t)128, (size_t)256, (size_t)1024}) {
            const size_t chunk = chunk_mib ? chunk_mib << 20 : bytes;
            if (chunk > bytes) continue;
            const size_t nchunk = bytes / chunk;
            const unsigned wg = (unsigned)(chunk / (128 << 10));
            for (int two = 0; two < 2; two++) {
                float best = 1e30f;
                for (int r = 0; r < reps + 1; r++) {
                    HIPCHK(hipDeviceSynchronize());
                    HIPCHK(hipEventRecord(e0, s1));
                    for (size_t c = 0; c < nchunk; c++) {
                        const v2d *pi = (const v2d *)(in + c * chunk);
                        v2d *pa = (v2d *)(A + c * chunk), *pb = (v2d *)(B + c * chunk);
                        hipLaunchKernelGGL(m.k1, dim3(wg), dim3(512), 0, s1, pi, pa);
                        if (two) {
                            HIPCHK(hipEventRecord(ev[c % ev.size()], s1));
                            HIPCHK(hipStreamWaitEvent(s2, ev[c % ev.size()], 0));
                            hipLaunchKernelGGL(m.k2, dim3(wg), dim3(512), 0, s2, (const v2d *)pa, pb);
                        } else {
                            hipLaunchKernelGGL(m.k2, dim3(wg), dim3(512), 0, s1, (const v2d *)pa, pb);
                        }
                    }
                    if (two) { HIPCHK(hipEventRecord(ev[0], s2)); HIPCHK(hipStreamWaitEvent(s1, ev[0], 0)); }
                    HIPCHK(hipEventRecord(e1, s1));
                    HIPCHK(hipEventSynchronize(e1));
                    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                    if (r > 0 && ms < best) best = ms;
                }
                printf("%s chunk %5zu MiB  %s  %8.3f ms   %7.1f GB/s algorithmic (4 x buffer)\n", m.name, chunk_mib ? chunk_mib : gib * 1024,
                       two ? "two streams" : "one stream ", best, 4.0 * bytes / best * 1e-6);
                if (!chunk_mib) break;
            }
        }
    }
    return 0;
}
