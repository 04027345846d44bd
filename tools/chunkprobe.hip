// tools/chunkprobe.hip -- what bounds the transposed-tile store / tiled passes of the fp32 2048-point lines (C5)?
// As pure copies those passes run at 4.1-5.2 TB/s (profiles/r4_f32_2048_tiled_candidates.txt) although every byte moves in 2 KiB
// chunks.  This probe separates the two suspects with a copy kernel that reads one contiguous 256 KiB chunk per workgroup (a tile of
// 16 lines x 2048 points x 8 B, as the tiled load does) and writes it back as R runs of S bytes, run r of workgroup w at
//   a*A + r*(S*NB) + b*S          (w = a*NB + b: the workgroups of one row interleave their runs, as the tiled stores do)
//   * run size S: 128 B ... 32 KiB (the transposed-tile store writes TL x TL points = 2 KiB at fp32)
//   * the shape of one wave's store instruction: "coalesced" = 64 lanes x 16 B contiguous; "pieces" = 16 pieces of 32 B, one in each
//     of 16 consecutive 128-byte lines (what a line-fastest fp32 wave -- 16 lines x 4 points -- does; four waves complete a line)
//   * one or two workgroups per CU (dynamic LDS request)
// usage: chunkprobe [GiB per buffer = 4]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));      // one fp32 complex point
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WG_BYTES = 256 * 1024, THREADS = 512;

// PIECES = 0: thread t, register c moves 16 B at chunk offset (c*512 + t)*16 -- a wave covers 1 KiB contiguous on both sides
// PIECES = 1: the chunk is [point n (2048)][line l (16)] of 8 B; lane = l + 16*t4, wave v covers points 4v .. 4v+3 (+ 128*c): loads
//             are 512 B contiguous per wave instruction; the store puts point n of line l at run-relative (l*16 + n%16)*8 within the
//             2 KiB block of tile n/16 -- i.e. transposed inside TL x TL blocks, 32-byte pieces per line
template <int PIECES, int NT>
__global__ __launch_bounds__(THREADS) void copy_kernel(const char *__restrict__ src, char *__restrict__ dst, uint32_t NB, uint64_t S, uint64_t A)
{
    extern __shared__ char lds[];
    (void)lds;
    const uint32_t w = blockIdx.x, a = w / NB, b = w % NB;
    const char *chunk = src + (uint64_t)w * WG_BYTES;
    char *base = dst + (uint64_t)a * A + (uint64_t)b * S;
    const uint64_t stride = S * NB;
    const int tid = threadIdx.x;
    if (!PIECES) {
        v4f v[32];
#pragma unroll
        for (int c = 0; c < 32; c++) v[c] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(chunk + ((uint64_t)c * THREADS + tid) * 16));
#pragma unroll
        for (int c = 0; c < 32; c++) {
            const uint64_t off = ((uint64_t)c * THREADS + tid) * 16;          // offset in the workgroup's 256 KiB
            v4f *p = reinterpret_cast<v4f *>(base + (off / S) * stride + off % S);
            if (NT) __builtin_nontemporal_store(v[c], p); else *p = v[c];
        }
    } else {
        const int l = tid & 15, t = tid >> 4;                                  // 32 threads per line, 64 points each
        v2f v[64];
#pragma unroll
        for (int c = 0; c < 64; c++) v[c] = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(chunk + (((uint64_t)(t + 32 * c)) * 16 + l) * 8));
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const uint32_t n = t + 32 * c;                                     // point
            const uint64_t off = (uint64_t)(n >> 4) * 2048 + ((uint64_t)l * 16 + (n & 15)) * 8;      // transposed inside the 16 x 16 block
            v2f *p = reinterpret_cast<v2f *>(base + (off / S) * stride + off % S);
            if (NT) __builtin_nontemporal_store(v[c], p); else *p = v[c];
        }
    }
}

template <int PIECES, int NT> static float run(const char *src, char *dst, uint32_t nwg, uint32_t NB, uint64_t S, uint64_t A, size_t lds, int reps)
{
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&copy_kernel<PIECES, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        HIPCHK(hipEventRecord(e0));
        hipLaunchKernelGGL((copy_kernel<PIECES, NT>), dim3(nwg), dim3(THREADS), lds, 0, src, dst, NB, S, A);
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
    const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 4;
    const size_t bytes = gib << 30;
    char *src, *dst;
    HIPCHK(hipMalloc(&src, bytes)); HIPCHK(hipMalloc(&dst, bytes));
    HIPCHK(hipMemset(src, 1, bytes)); HIPCHK(hipMemset(dst, 0, bytes));
    const uint32_t nwg = (uint32_t)(bytes / WG_BYTES), NB = 32;      // 32 workgroups per row (zs = 512 lines = 32 tiles of 16)
    const uint64_t A = (uint64_t)NB * WG_BYTES;                       // one row of workgroups = 8 MiB
    printf("chunkprobe: %zu GiB per buffer, %u workgroups of 256 KiB, %u per row; GB/s = (read + written bytes) / best of 3\n", gib, nwg, NB);
    printf("%-9s %-10s %-6s %10s %10s\n", "run S", "wave", "WG/CU", "plain GB/s", "nt GB/s");
    const uint64_t sizes[] = {128, 512, 2048, 4096, 8192, 32768, (uint64_t)WG_BYTES};
    for (int per_cu = 1; per_cu <= 2; per_cu++) {
        const size_t lds = per_cu == 1 ? 100 * 1024 : 64 * 1024;
        for (uint64_t S : sizes) {
            const float a0 = run<0, 0>(src, dst, nwg, NB, S, A, lds, 3), a1 = run<0, 1>(src, dst, nwg, NB, S, A, lds, 3);
            printf("%-9llu %-10s %-6d %10.0f %10.0f\n", (unsigned long long)S, "coalesced", per_cu, 2.0 * bytes / a0 / 1e6, 2.0 * bytes / a1 / 1e6);
            if (S >= 2048) {
                const float b0 = run<1, 0>(src, dst, nwg, NB, S, A, lds, 3), b1 = run<1, 1>(src, dst, nwg, NB, S, A, lds, 3);
                printf("%-9llu %-10s %-6d %10.0f %10.0f\n", (unsigned long long)S, "pieces", per_cu, 2.0 * bytes / b0 / 1e6, 2.0 * bytes / b1 / 1e6);
            }
        }
    }
    return 0;
}
