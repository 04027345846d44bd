// tools/memprobe.hip -- does the PHYSICAL backing of a buffer decide how fast the strided 128-byte-run accesses of the
// y / x passes run?  Round 2 saw the same plan run a pass at 5.4 ms in one process and 6.4 ms in the next, per buffer.
// This probe allocates destination buffers in several ways (hipMalloc; the virtual-memory API with physical chunks of
// 2 MiB ... 16 GiB) and times two micro-kernels with the access patterns of the x pass on each:
//   scatter: a workgroup reads a contiguous 128 KiB chunk and writes 1024 rows of 128 B, row stride = rows_stride
//   gather : the reverse (the multi-rank inverse x pass)
// usage: memprobe [GiB per buffer = 16] [repeats = 3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// 512 threads, 16 points of 16 B per thread = 128 KiB per workgroup (the fp64 1024-point tile); launched with
// 64 KiB of dynamic LDS so that two workgroups share a CU like the FFT kernel
template <int NT_STORE, int SCATTER> __global__ __launch_bounds__(512) void pattern_kernel(const v2d *__restrict__ src, v2d *__restrict__ dst,
                                                                                          uint32_t nb, uint64_t row_stride, uint64_t a_stride)
{
    // workgroup w -> (a, b): b fastest; strided side: row k at k*row_stride + a*a_stride + b*8 + l   (elements of 16 B)
    const uint32_t w = blockIdx.x, a = w / nb, b = w % nb;
    const int tid = threadIdx.x, l = tid & 7, t = tid >> 3;          // lane = line + 8*t, 64 threads per line
    const v2d *chunk = src + (uint64_t)w * 8192;                       // contiguous side: [k][l], 1024 x 8 points
    v2d *cchunk = dst + (uint64_t)w * 8192;
    const uint64_t col = (uint64_t)a * a_stride + (uint64_t)b * 8 + l;
    v2d v[16];
    if (SCATTER) {
#pragma unroll
        for (int c = 0; c < 16; c++) v[c] = __builtin_nontemporal_load(chunk + (uint64_t)(t + 64 * c) * 8 + l);
#pragma unroll
        for (int c = 0; c < 16; c++) {
            v2d *p = dst + (uint64_t)(t + 64 * c) * row_stride + col;
            if (NT_STORE) __builtin_nontemporal_store(v[c], p); else *p = v[c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const v2d *p = src + (uint64_t)(t + 64 * c) * row_stride + col;
            v[c] = NT_STORE ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int c = 0; c < 16; c++) __builtin_nontemporal_store(v[c], cchunk + (uint64_t)(t + 64 * c) * 8 + l);
    }
}

struct Buf { std::string name; char *ptr; };

static char *vmm_alloc(size_t bytes, size_t chunk)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void *va = nullptr;
    HIPCHK(hipMemAddressReserve(&va, bytes, chunk, nullptr, 0));
    for (size_t off = 0; off < bytes; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        HIPCHK(hipMemCreate(&h, chunk, &prop, 0));
        HIPCHK(hipMemMap((char *)va + off, chunk, 0, h, 0));
        HIPCHK(hipMemRelease(h));       // the mapping keeps the memory alive
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    HIPCHK(hipMemSetAccess(va, bytes, &acc, 1));
    return (char *)va;
}

// stride sweep on ONE contiguous hipMalloc buffer: which row strides / paddings does the memory system like?
static int sweep(int reps)
{
    const size_t bytes = 16ull << 30, extra = 3ull << 30;
    char *src, *dst;
    HIPCHK(hipMalloc(&src, bytes + extra));
    HIPCHK(hipMalloc(&dst, bytes + extra));
    HIPCHK(hipMemset(src, 1, bytes + extra));
    HIPCHK(hipMemset(dst, 0, bytes + extra));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        float best = 1e30f;
        launch();
        HIPCHK(hipDeviceSynchronize());
        for (int r = 0; r < reps; r++) {
            HIPCHK(hipEventRecord(e0, nullptr));
            launch();
            HIPCHK(hipEventRecord(e1, nullptr));
            HIPCHK(hipEventSynchronize(e1));
            float ms;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        return best;
    };
    const uint32_t ntiles = (uint32_t)(bytes / (128 << 10));
    struct Case { const char *name; uint64_t row_stride_bytes, a_stride_bytes; uint32_t nb; };
    const uint64_t M = 1ull << 20;
    std::vector<Case> cases = {
        {"rows 16 MiB apart (x pass, API layout)", 16 * M, 16384, 128},
        {"rows 16 MiB + 128 B", 16 * M + 128, 16384, 128},
        {"rows 16 MiB + 256 B", 16 * M + 256, 16384, 128},
        {"rows 16 MiB + 384 B", 16 * M + 384, 16384, 128},
        {"rows 16 MiB + 640 B", 16 * M + 640, 16384, 128},
        {"rows 16 MiB + 1 KiB", 16 * M + 1024, 16384, 128},
        {"rows 16 MiB + 1 KiB + 128 B", 16 * M + 1152, 16384, 128},
        {"rows 16 MiB + 2 KiB + 128 B", 16 * M + 2176, 16384, 128},
        {"rows 16 MiB + 4 KiB", 16 * M + 4096, 16384, 128},
        {"rows 16 MiB + 4 KiB + 128 B", 16 * M + 4224, 16384, 128},
        {"rows 16 MiB + 16 KiB", 16 * M + 16384, 16384, 128},
        {"rows 16 MiB + 16 KiB + 128 B", 16 * M + 16512, 16384, 128},
        {"rows 16 MiB + 64 KiB", 16 * M + 65536, 16384, 128},
        {"rows 16 MiB + 64 KiB + 128 B", 16 * M + 65664, 16384, 128},
        {"rows 16 MiB + 1 MiB", 17 * M, 16384, 128},
        {"rows 16 MiB + 2 MiB + 4 KiB", 18 * M + 4096, 16384, 128},
        {"rows 16 KiB apart inside 16 MiB planes (y last)", 16384, 16 * M, 128},
        {"rows 16 KiB + 128 B apart inside 16 MiB + 128 KiB planes", 16384 + 128, 16 * M + 131072, 128},
        {"rows 128 KiB apart, planes of 128 MiB (tile-outer)", 131072, 128 * M, 1024},
        {"rows 128 KiB + 128 B apart, planes of 128 MiB + 128 KiB", 131072 + 128, 128 * M + 131072, 1024},
        {"rows 1 MiB apart, planes of 1 GiB", M, 1024 * M, 8192},
    };
    printf("%-52s %10s %10s %10s %10s   (ms for 16 GiB each way)\n", "strided side", "scatter", "scatter-nt", "gather", "gather-nt");
    for (auto &c : cases) {
        const uint64_t rs = c.row_stride_bytes / 16, as = c.a_stride_bytes / 16;
        const v2d *s = (const v2d *)src;
        v2d *d = (v2d *)dst;
        const float t0 = time_it([&] { pattern_kernel<0, 1><<<ntiles, 512, 65536>>>(s, d, c.nb, rs, as); });
        const float t1 = time_it([&] { pattern_kernel<1, 1><<<ntiles, 512, 65536>>>(s, d, c.nb, rs, as); });
        const float t2 = time_it([&] { pattern_kernel<0, 0><<<ntiles, 512, 65536>>>((const v2d *)dst, (v2d *)src, c.nb, rs, as); });
        const float t3 = time_it([&] { pattern_kernel<1, 0><<<ntiles, 512, 65536>>>((const v2d *)dst, (v2d *)src, c.nb, rs, as); });
        printf("%-52s %10.3f %10.3f %10.3f %10.3f\n", c.name, t0, t1, t2, t3);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "sweep") return sweep(argc > 2 ? atoi(argv[2]) : 3);
    const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 16;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t bytes = gib << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    size_t gmin = 0, grec = 0;
    HIPCHK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    HIPCHK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity: minimum %zu, recommended %zu\n", gmin, grec);

    std::vector<Buf> bufs;
    char *src;
    HIPCHK(hipMalloc(&src, bytes));
    HIPCHK(hipMemset(src, 1, bytes));
    for (int i = 0; i < 3; i++) { char *p; HIPCHK(hipMalloc(&p, bytes)); bufs.push_back({"hipMalloc#" + std::to_string(i), p}); }
    bufs.push_back({"vmm chunk=whole", vmm_alloc(bytes, bytes)});
    bufs.push_back({"vmm chunk=1GiB", vmm_alloc(bytes, 1ull << 30)});
    bufs.push_back({"vmm chunk=64MiB", vmm_alloc(bytes, 64ull << 20)});
    bufs.push_back({"vmm chunk=2MiB", vmm_alloc(bytes, 2ull << 20)});
    { char *p; HIPCHK(hipMalloc(&p, bytes)); bufs.push_back({"hipMalloc#3 (after vmm)", p}); }
    for (auto &b : bufs) HIPCHK(hipMemset(b.ptr, 0, bytes));
    HIPCHK(hipDeviceSynchronize());

    // geometry of the 1024^3 fp64 x pass scaled to the buffer: rows = 1024, row stride = bytes/1024
    const uint64_t row_stride = bytes / 16 / 1024;      // elements
    const uint32_t ntiles = (uint32_t)(bytes / (128 << 10)), nb = 128;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        float best = 1e30f;
        launch();
        HIPCHK(hipDeviceSynchronize());
        for (int r = 0; r < reps; r++) {
            HIPCHK(hipEventRecord(e0, nullptr));
            launch();
            HIPCHK(hipEventRecord(e1, nullptr));
            HIPCHK(hipEventSynchronize(e1));
            float ms;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        return best;
    };
    printf("%-26s %14s %14s %14s %14s %14s   (ms, %zu GiB moved each way; lower is better)\n", "buffer", "scatter", "scatter-nt", "gather", "gather-nt",
           "contig-copy", gib);
    for (int round = 0; round < 2; round++)
        for (auto &b : bufs) {
            v2d *d = (v2d *)b.ptr;
            const v2d *s = (const v2d *)src;
            const float t0 = time_it([&] { pattern_kernel<0, 1><<<ntiles, 512, 65536>>>(s, d, nb, row_stride, (uint64_t)nb * 8); });
            const float t1 = time_it([&] { pattern_kernel<1, 1><<<ntiles, 512, 65536>>>(s, d, nb, row_stride, (uint64_t)nb * 8); });
            const float t2 = time_it([&] { pattern_kernel<0, 0><<<ntiles, 512, 65536>>>((const v2d *)b.ptr, (v2d *)src, nb, row_stride, (uint64_t)nb * 8); });
            const float t3 = time_it([&] { pattern_kernel<1, 0><<<ntiles, 512, 65536>>>((const v2d *)b.ptr, (v2d *)src, nb, row_stride, (uint64_t)nb * 8); });
            const float t4 = time_it([&] { HIPCHK(hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyDeviceToDevice, nullptr)); });
            printf("%-26s %14.3f %14.3f %14.3f %14.3f %14.3f\n", b.name.c_str(), t0, t1, t2, t3, t4);
        }
    return 0;
}
