// tools/kbench.hip -- kernel-level benchmark harness on the C ABI (include/dfft_c.h), no Python:
// per-pass device times of a single-GPU plan with named options, plus two self-checks that need no
// oracle (plane-wave known answer and round trip).  Parity proper is tests/ (-m gpu) against oracle/.
//
//   tools/kbench --size 1024 --prec f64 --mode c2c --iters 5 --opt variant_fy=5 --opt pipeline_chunks=8 --check
//   tools/kbench --line 2048 --batch 65536 --prec f32 --variant 3         (one pass on natural lines)
//
// Build: make -C tools   (hipcc, links ../distributedfft_amd/libdfft_amd.so)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../include/dfft_c.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define DCHK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, dfft_last_error()); exit(3); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// uniform [0, 255) like the reference's scaled cuRAND input (tests/src/pencil/base.cu:45-53)
template <typename R> __device__ __forceinline__ R synth(uint64_t idx) { return (R)((double)(mix64(idx) >> 11) * (255.0 / 9007199254740992.0)); }

template <typename R> __global__ void fill_random(R *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = synth<R>(i);
}
// superposition of three plane waves with integer frequencies: the transform is N^3 at three points
struct Waves { int kx[3], ky[3], kz[3]; };
template <typename R> __global__ void fill_waves(R *p, size_t Nx, size_t Ny, size_t Nz, Waves w)
{
    const size_t n = Nx * Ny * Nz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t z = i % Nz, y = (i / Nz) % Ny, x = i / (Nz * Ny);
        double re = 0, im = 0;
        for (int q = 0; q < 3; q++) {
            const double ph = 2.0 * M_PI * ((double)((x * w.kx[q]) % Nx) / Nx + (double)((y * w.ky[q]) % Ny) / Ny + (double)((z * w.kz[q]) % Nz) / Nz);
            re += cos(ph); im += sin(ph);
        }
        p[2 * i] = (R)re; p[2 * i + 1] = (R)im;
    }
}
// max |got - want| over the spectrum, want = N^3 at the three frequencies, 0 elsewhere; block partial maxima
template <typename R> __global__ void check_waves(const R *p, size_t Nx, size_t Ny, size_t Nz, Waves w, double *partial)
{
    const size_t n = Nx * Ny * Nz;
    const double peak = (double)n;
    double m = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t z = i % Nz, y = (i / Nz) % Ny, x = i / (Nz * Ny);
        double want = 0;
        for (int q = 0; q < 3; q++) if ((int)x == w.kx[q] && (int)y == w.ky[q] && (int)z == w.kz[q]) want += peak;
        const double dr = (double)p[2 * i] - want, di = (double)p[2 * i + 1];
        m = fmax(m, fmax(fabs(dr), fabs(di)));
    }
    __shared__ double sm[256];
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
template <typename R> __global__ void check_random(const R *p, size_t n, double scale, double *partial)
{
    double m = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmax(m, fabs((double)p[i] * scale - (double)synth<R>(i)));
    __shared__ double sm[256];
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// block partial maxima of |a - b| (first gridDim entries) and |b| (next gridDim entries)
template <typename R> __global__ void diff_kernel(const R *a, const R *b, size_t n, double *partial)
{
    double md = 0, mr = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        md = fmax(md, fabs((double)a[i] - (double)b[i]));
        mr = fmax(mr, fabs((double)b[i]));
    }
    __shared__ double s0[256], s1[256];
    s0[threadIdx.x] = md; s1[threadIdx.x] = mr;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s0[threadIdx.x] = fmax(s0[threadIdx.x], s0[threadIdx.x + s]); s1[threadIdx.x] = fmax(s1[threadIdx.x], s1[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[blockIdx.x] = s0[0]; partial[gridDim.x + blockIdx.x] = s1[0]; }
}

static double reduce_partials(double *d_part, int nblk)
{
    std::vector<double> h(nblk);
    HIPCHK(hipMemcpy(h.data(), d_part, nblk * sizeof(double), hipMemcpyDeviceToHost));
    double m = 0;
    for (double v : h) m = fmax(m, v);
    return m;
}

struct Args {
    size_t Nx = 1024, Ny = 1024, Nz = 1024;
    std::string prec = "f64", mode = "c2c", label;
    int iters = 5, check = 0;
    int wall_only = 0;    // --wall-only: no phase timing (its events serialise nothing but cost host time): `iters` forward + inverse pairs
                          // enqueued back to back, one device synchronisation at the end -- the figure to compare pipeline depths and
                          // compute_streams by, because spans of chunks that overlap on two streams cannot be summed
    int latency = 0;      // --latency: no phase timing; host wall clock of the blocking exec calls (what a caller waits for)
    std::vector<std::pair<std::string, long>> opts;
    size_t line = 0, batch = 0;
    int variant = 0, debug = 0;
    // --slab: all four buffers come from ONE hipMalloc, placed in the order of --perm (letters i o w b = in, out,
    // work, back) at multiples of (buffer size + --delta bytes): probes how relative placement affects a pass
    int slab = 0;
    size_t delta = 0;
    std::string perm = "wiob";
    // --vmm MiB: every buffer comes from the virtual-memory API in physical chunks of that size (0 = off);
    // --shuffle: the chunks are mapped in a pseudo-random order
    size_t vmm_mib = 0;
    int shuffle = 0;
    // --ranks P1xP2 [--rank r]: the plan of rank r of a P1 x P2 grid with the exchange stubbed out (a callback transport that
    // moves nothing): the kernels run with that rank's real descriptors (1/P of the volume, its segment tables, its
    // pipeline chunks) on garbage, so their times are what one GPU of the multi-GPU run computes.  No --check.
    int P1 = 1, P2 = 1, rank = 0;
    // --sweep "k=v,k2=v2;k=v3": option sets measured one after the other IN ONE PROCESS ON THE SAME BUFFERS (the 128-byte-run
    // passes depend on the physical placement of a buffer, profiles/r2_placement_probe.txt: A/B across processes is noise)
    std::vector<std::vector<std::pair<std::string, long>>> sweep;
    // --tune K: out / back / the work area come from dfft_tune_placement with K physical backings per buffer
    int tune = 0;
    // --tune-variants: dfft_tune_variants on the run's buffers before the timed iterations
    int tune_variants = 0;
    int lib_buffers = 0;
};

static int dry_exchange(void *, const void *, const size_t *, const size_t *, void *, const size_t *, const size_t *, const int *, int, int, void *)
{
    return 0;
}

// --vmm-spread K (A/B, round 4): K times as many physical chunks are created as the buffer needs and every K-th is kept, the others
// released -- whatever the allocator hands out in a row, the buffer's chunks are not physical neighbours
static int g_vmm_spread = 1;
static char *vmm_alloc(size_t bytes, size_t chunk, bool shuffle)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    bytes = (bytes + chunk - 1) / chunk * chunk;
    void *va = nullptr;
    HIPCHK(hipMemAddressReserve(&va, bytes, chunk, nullptr, 0));
    const size_t n = bytes / chunk;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    if (g_vmm_spread > 1) {
        std::vector<hipMemGenericAllocationHandle_t> all(n * g_vmm_spread);
        for (size_t i = 0; i < all.size(); i++) HIPCHK(hipMemCreate(&all[i], chunk, &prop, 0));
        for (size_t i = 0; i < all.size(); i++) {
            if (i % g_vmm_spread == 0) h[i / g_vmm_spread] = all[i];
            else HIPCHK(hipMemRelease(all[i]));
        }
    } else
    for (size_t i = 0; i < n; i++) HIPCHK(hipMemCreate(&h[i], chunk, &prop, 0));
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    if (shuffle) {
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (size_t i = n - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(order[i], order[s % (i + 1)]); }
    }
    for (size_t i = 0; i < n; i++) {
        HIPCHK(hipMemMap((char *)va + i * chunk, chunk, 0, h[order[i]], 0));
        HIPCHK(hipMemRelease(h[order[i]]));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    HIPCHK(hipMemSetAccess(va, bytes, &acc, 1));
    return (char *)va;
}

static Args parse(int argc, char **argv)
{
    Args a;
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(1); } return argv[++i]; };
        if (k == "--size") {
            const char *v = next();
            size_t x = 0, y = 0, z = 0;
            if (sscanf(v, "%zux%zux%zu", &x, &y, &z) == 3) { a.Nx = x; a.Ny = y; a.Nz = z; }
            else { a.Nx = a.Ny = a.Nz = (size_t)atoll(v); }
        } else if (k == "--prec") a.prec = next();
        else if (k == "--mode") a.mode = next();
        else if (k == "--iters") a.iters = atoi(next());
        else if (k == "--check") a.check = 1;
        else if (k == "--latency") a.latency = 1;
        else if (k == "--wall-only") a.wall_only = 1;
        else if (k == "--label") a.label = next();
        else if (k == "--line") a.line = (size_t)atoll(next());
        else if (k == "--batch") a.batch = (size_t)atoll(next());
        else if (k == "--variant") a.variant = atoi(next());
        else if (k == "--debug") a.debug = atoi(next());
        else if (k == "--slab") a.slab = 1;
        else if (k == "--delta") { a.slab = 1; a.delta = (size_t)atoll(next()); }
        else if (k == "--perm") { a.slab = 1; a.perm = next(); }
        else if (k == "--vmm") a.vmm_mib = (size_t)atoll(next());
        else if (k == "--shuffle") a.shuffle = 1;
        else if (k == "--vmm-spread") g_vmm_spread = atoi(next());
        else if (k == "--ranks") { if (sscanf(next(), "%dx%d", &a.P1, &a.P2) != 2) { fprintf(stderr, "--ranks P1xP2\n"); exit(1); } }
        else if (k == "--rank") a.rank = atoi(next());
        else if (k == "--tune") a.tune = atoi(next());
        else if (k == "--tune-variants") a.tune_variants = 1;
        else if (k == "--lib-buffers") a.lib_buffers = 1;
        else if (k == "--sweep") {
            std::string all = next();
            size_t pos = 0;
            while (pos <= all.size()) {
                const size_t semi = std::min(all.find(';', pos), all.size());
                std::vector<std::pair<std::string, long>> set;
                size_t q = pos;
                while (q < semi) {
                    const size_t comma = std::min(all.find(',', q), semi);
                    const std::string kv = all.substr(q, comma - q);
                    const size_t eq = kv.find('=');
                    if (eq != std::string::npos) set.emplace_back(kv.substr(0, eq), atol(kv.c_str() + eq + 1));
                    q = comma + 1;
                }
                a.sweep.push_back(set);
                pos = semi + 1;
            }
        }
        else if (k == "--opt") {
            std::string kv = next();
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) { fprintf(stderr, "--opt key=value\n"); exit(1); }
            a.opts.emplace_back(kv.substr(0, eq), atol(kv.c_str() + eq + 1));
        } else { fprintf(stderr, "unknown argument %s\n", k.c_str()); exit(1); }
    }
    return a;
}

template <typename R> static int run_line(const Args &a)
{
    const int prec = sizeof(R) == 8 ? DFFT_F64 : DFFT_F32;
    const size_t n = a.line * a.batch;
    R *in, *out;
    HIPCHK(hipMalloc(&in, n * 2 * sizeof(R)));
    HIPCHK(hipMalloc(&out, n * 2 * sizeof(R)));
    fill_random<R><<<4096, 256>>>(in, 2 * n);
    HIPCHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    DCHK(dfft_fft1d_batched_ex(prec, a.line, a.batch, out, in, DFFT_FORWARD, nullptr, a.variant, a.debug));
    HIPCHK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    for (int it = 0; it < a.iters; it++) {
        HIPCHK(hipEventRecord(e0, nullptr));
        DCHK(dfft_fft1d_batched_ex(prec, a.line, a.batch, out, in, DFFT_FORWARD, nullptr, a.variant, a.debug));
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        best = fmin(best, ms); sum += ms;
    }
    const double bytes = 2.0 * n * 2 * sizeof(R);
    double rt = -1, dev = -1;
    if (a.check && !a.debug && a.variant != 0) {      // forward result against the default configuration's
        R *ref;
        HIPCHK(hipMalloc(&ref, n * 2 * sizeof(R)));
        DCHK(dfft_fft1d_batched_ex(prec, a.line, a.batch, ref, in, DFFT_FORWARD, nullptr, 0, 0));
        double *part;
        HIPCHK(hipMalloc(&part, 2 * 1024 * sizeof(double)));
        diff_kernel<R><<<1024, 256>>>(out, ref, 2 * n, part);
        std::vector<double> h(2048);
        HIPCHK(hipMemcpy(h.data(), part, 2048 * sizeof(double), hipMemcpyDeviceToHost));
        double md = 0, mr = 0;
        for (int i = 0; i < 1024; i++) { md = fmax(md, h[i]); mr = fmax(mr, h[1024 + i]); }
        dev = md / mr;
        HIPCHK(hipFree(ref)); HIPCHK(hipFree(part));
    }
    if (a.check && !a.debug) {      // inverse of the forward result must give N * input
        DCHK(dfft_fft1d_batched_ex(prec, a.line, a.batch, in, out, DFFT_INVERSE, nullptr, a.variant, 0));
        double *part;
        HIPCHK(hipMalloc(&part, 1024 * sizeof(double)));
        check_random<R><<<1024, 256>>>(in, 2 * n, 1.0 / (double)a.line, part);
        rt = reduce_partials(part, 1024) / 255.0;
    }
    printf("LINE %s N=%zu batch=%zu %s variant=%d debug=%d avg %.4f ms min %.4f ms  %.1f GB/s (min)  roundtrip %.2e  vs-default %.2e\n", a.label.c_str(),
           a.line, a.batch, a.prec.c_str(), a.variant, a.debug, sum / a.iters, best, bytes / best / 1e6, rt, dev);
    return 0;
}

template <typename R> static int run_plan(const Args &a)
{
    const int prec = sizeof(R) == 8 ? DFFT_F64 : DFFT_F32;
    const bool c2c = a.mode == "c2c";
    const size_t esz = 2 * sizeof(R);
    dfft_plan *plan;
    const int nranks = a.P1 * a.P2;
    dfft_comm *comm = nullptr;
    if (nranks > 1) DCHK(dfft_comm_create_callback(nranks, a.rank, dry_exchange, nullptr, &comm));
    std::vector<std::vector<std::pair<std::string, long>>> sets = a.sweep;
    if (sets.empty()) sets.emplace_back();
    const bool sweeping = sets.size() > 1;
    if (sweeping && (a.slab || a.vmm_mib)) { fprintf(stderr, "--sweep does not combine with --slab / --vmm\n"); exit(1); }
    char *in = nullptr, *out = nullptr, *back = nullptr, *slab = nullptr, *work = nullptr;
    size_t work_cap = 0;
    bool alias_back = false;
    double *part = nullptr;
    const int nblk = 4096;
  for (size_t si = 0; si < sets.size(); si++) {
    DCHK(dfft_plan_create(&plan, a.P2 == 1 && nranks > 1 ? DFFT_SLAB_OPT1 : DFFT_PENCIL_OPT1, prec, nullptr, comm, a.rank, -1));
    for (auto &kv : a.opts) DCHK(dfft_set_option(plan, kv.first.c_str(), kv.second));
    for (auto &kv : sets[si]) DCHK(dfft_set_option(plan, kv.first.c_str(), kv.second));
    DCHK(dfft_init(plan, a.Nx, a.Ny, a.Nz, a.P1, a.P2, c2c ? 1 : 0, (a.slab || a.vmm_mib || sweeping) ? 0 : 1));
    size_t isz[3];
    DCHK(dfft_get_in_size(plan, isz));
    const size_t n = nranks > 1 ? isz[0] * isz[1] * isz[2] : a.Nx * a.Ny * a.Nz;      // points of this rank's input block
    const size_t in_bytes = c2c ? n * esz : n * sizeof(R);
    const size_t dom = dfft_domain_size(plan);
    if (sweeping) {      // one work area for every option set (grown if a set needs more)
        const size_t ws = dfft_work_size_device(plan);
        // --tune: the shared work area comes from the virtual-memory API as well (64 MiB chunks: what the tuner picks most often)
        if (ws > work_cap) { if (work) DCHK(dfft_free(work)); DCHK(dfft_malloc(ws, a.tune > 1 ? 64 : 0, (void **)&work)); work_cap = ws; }
        DCHK(dfft_set_work_area(plan, work, nullptr));
    }
    if (si > 0) {
        /* buffers of the first set are reused */
    } else if (a.slab) {
        const size_t slot = ((std::max(dom, dfft_work_size_device(plan)) + 255) & ~(size_t)255) + a.delta;
        HIPCHK(hipMalloc(&slab, 4 * slot + 256));
        char *pos[256] = {nullptr};
        for (size_t i = 0; i < a.perm.size() && i < 4; i++) pos[(unsigned char)a.perm[i]] = slab + i * slot;
        in = pos['i']; out = pos['o']; back = pos['b'];
        if (!in || !out || !back || !pos['w']) { fprintf(stderr, "--perm needs the letters i o w b\n"); exit(1); }
        DCHK(dfft_set_work_area(plan, pos['w'], nullptr));
    } else if (a.vmm_mib) {
        const size_t chunk = a.vmm_mib << 20;
        char *w = vmm_alloc(dfft_work_size_device(plan), chunk, a.shuffle);
        DCHK(dfft_set_work_area(plan, w, nullptr));
        in = vmm_alloc(in_bytes, chunk, a.shuffle);
        out = vmm_alloc(dom, chunk, a.shuffle);
        back = vmm_alloc(in_bytes, chunk, a.shuffle);
    } else if (a.tune > 1) {
        HIPCHK(hipMalloc(&in, in_bytes));
        fill_random<R><<<nblk, 256>>>((R *)in, in_bytes / sizeof(R));
        HIPCHK(hipDeviceSynchronize());
        float rep[64];
        int nrep = 0;
        DCHK(dfft_tune_placement(plan, in, a.tune, (void **)&out, (void **)&back, rep, 64, &nrep));
        printf("TUNE %d backings per buffer, FFT ms (fwd + inv) per trial:", a.tune);
        for (int i = 0; i < nrep; i++) printf(" %.3f", rep[i]);
        printf("\n");
    } else {
        // --lib-buffers: out / back from the library's allocator (dfft_malloc(DFFT_CHUNK_DEFAULT): probed candidates), what a drop-in
        // caller is advised to use; default: plain hipMalloc, the reference's contract as it stands
        HIPCHK(hipMalloc(&in, in_bytes));
        if (a.lib_buffers) DCHK(dfft_malloc(dom, DFFT_CHUNK_DEFAULT, (void **)&out)); else HIPCHK(hipMalloc(&out, dom));
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        alias_back = free_b < in_bytes + (1ull << 30);       // 2048^3: the inverse writes over the input buffer
        if (alias_back) back = in;
        else if (a.lib_buffers) DCHK(dfft_malloc(in_bytes, DFFT_CHUNK_DEFAULT, (void **)&back));
        else HIPCHK(hipMalloc(&back, in_bytes));
    }
    if (!part) HIPCHK(hipMalloc(&part, nblk * sizeof(double)));
    const size_t nreal = in_bytes / sizeof(R);
    double wave_err = -1, rt_err = -1;
    auto fwd = [&]() { if (c2c) DCHK(dfft_exec_c2c(plan, out, in, DFFT_FORWARD)); else DCHK(dfft_exec_r2c(plan, out, in)); };
    auto inv = [&]() { if (c2c) DCHK(dfft_exec_c2c(plan, back, out, DFFT_INVERSE)); else DCHK(dfft_exec_c2r(plan, back, out)); };
    const bool dbg = (dfft_get_option(plan, "debug_skip") & 1) != 0;      // bit 0: copies, nothing to check (bit 1 = old address forms: checked)
    if (nranks > 1 && a.check) { fprintf(stderr, "--check is meaningless with a stubbed exchange\n"); exit(1); }
    if (a.check && c2c && !dbg) {
        Waves w = {{1, (int)a.Nx / 2 + 3, (int)a.Nx - 1}, {0, 5 % (int)a.Ny, (int)a.Ny - 2}, {(int)a.Nz - 1, 7 % (int)a.Nz, (int)a.Nz / 2}};
        fill_waves<R><<<nblk, 256>>>((R *)in, a.Nx, a.Ny, a.Nz, w);
        HIPCHK(hipDeviceSynchronize());
        fwd();
        check_waves<R><<<nblk, 256>>>((const R *)out, a.Nx, a.Ny, a.Nz, w, part);
        wave_err = reduce_partials(part, nblk) / (double)n;      // relative to the peak N^3
    }
    fill_random<R><<<nblk, 256>>>((R *)in, nreal);
    HIPCHK(hipDeviceSynchronize());
    if (a.tune_variants) {
        float rep[32];
        int nrep = 0;
        DCHK(dfft_tune_variants(plan, in, out, alias_back ? nullptr : back, rep, 32, &nrep));
        printf("TUNE-VARIANTS FFT ms per trial (first = rule-based configurations):");
        for (int i = 0; i < nrep; i++) printf(" %.3f", rep[i]);
        printf("\n");
        if (alias_back) { fill_random<R><<<nblk, 256>>>((R *)in, nreal); HIPCHK(hipDeviceSynchronize()); }
    }
    {
        int var[6], ord[6], a64[6];
        static const char *const names[6] = {"fz", "fy", "fx", "ix", "iy", "iz"};
        DCHK(dfft_get_pass_choices(plan, var, ord, a64));
        printf("CHOICES (variant/order/addr64)");
        for (int k = 0; k < 6; k++) printf(" %s=%d/%d/%d", names[k], var[k], ord[k], a64[k]);
        printf("\n");
    }
    fwd(); inv();                                   // warm-up
    if (a.check && !dbg) {
        check_random<R><<<nblk, 256>>>((const R *)back, nreal, 1.0 / (double)n, part);
        rt_err = reduce_partials(part, nblk) / 255.0;
    }
    if (a.latency) {
        // blocking execs as a caller issues them; two more warm-up rounds so that the plan's launch graph (option
        // "graph") is captured before the clock starts
        fwd(); inv(); fwd(); inv();
        double tf = 0, tb = 0;
        for (int it = 0; it < a.iters; it++) {
            const auto t0 = std::chrono::steady_clock::now();
            fwd();
            const auto t1 = std::chrono::steady_clock::now();
            inv();
            const auto t2 = std::chrono::steady_clock::now();
            tf += std::chrono::duration<double, std::micro>(t1 - t0).count();
            tb += std::chrono::duration<double, std::micro>(t2 - t1).count();
        }
        std::string optstr;
        for (auto &kv : a.opts) optstr += " " + kv.first + "=" + std::to_string(kv.second);
        printf("LATENCY %s %zux%zux%zu %s %s%s | forward %.1f us  inverse %.1f us (host wall clock of the blocking exec, %d iterations)\n",
               a.label.c_str(), a.Nx, a.Ny, a.Nz, a.prec.c_str(), a.mode.c_str(), optstr.c_str(), tf / a.iters, tb / a.iters, a.iters);
        return 0;
    }
    if (a.wall_only) {
        fwd(); inv();
        HIPCHK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < a.iters; it++) { fwd(); inv(); }
        HIPCHK(hipDeviceSynchronize());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / a.iters;
        std::string optstr;
        for (auto &kv : a.opts) optstr += " " + kv.first + "=" + std::to_string(kv.second);
        printf("WALL %s %zux%zux%zu %s %s%s rank %d of %dx%d chunks=%d | %.3f ms per forward + inverse (%d pairs back to back, no phase events)\n",
               a.label.c_str(), a.Nx, a.Ny, a.Nz, a.prec.c_str(), a.mode.c_str(), optstr.c_str(), a.rank, a.P1, a.P2, dfft_get_pipeline_chunks(plan), ms, a.iters);
        return 0;
    }
    DCHK(dfft_enable_phase_timing(plan, 1));
    double accf[8] = {0}, accb[8] = {0}, minf[8], minb[8];
    for (int i = 0; i < 8; i++) minf[i] = minb[i] = 1e30;
    int nf = 0, nb = 0;
    double wall = 0;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int it = 0; it < a.iters; it++) {
        if (alias_back) { fill_random<R><<<nblk, 256>>>((R *)in, nreal); HIPCHK(hipDeviceSynchronize()); }
        float ms[8];
        HIPCHK(hipEventRecord(e0, nullptr));
        fwd();
        nf = dfft_get_phase_times(plan, ms, 8);
        for (int i = 0; i < nf; i++) { accf[i] += ms[i]; minf[i] = fmin(minf[i], ms[i]); }
        inv();
        nb = dfft_get_phase_times(plan, ms, 8);
        for (int i = 0; i < nb; i++) { accb[i] += ms[i]; minb[i] = fmin(minb[i], ms[i]); }
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float w;
        HIPCHK(hipEventElapsedTime(&w, e0, e1));
        wall += w;
    }
    std::string optstr;
    for (auto &kv : a.opts) optstr += " " + kv.first + "=" + std::to_string(kv.second);
    for (auto &kv : sets[si]) optstr += " " + kv.first + "=" + std::to_string(kv.second);
    if (sweeping) optstr += " [set " + std::to_string(si) + ", shared buffers]";
    if (a.vmm_mib) optstr += " vmm=" + std::to_string(a.vmm_mib) + "MiB" + (a.shuffle ? " shuffled" : "");
    if (a.lib_buffers) optstr += " library buffers";
    if (a.tune > 1) optstr += " tuned placement (" + std::to_string(a.tune) + ")";
    if (a.slab) optstr += " slab perm=" + a.perm + " delta=" + std::to_string(a.delta);
    if (nranks > 1) optstr += " rank " + std::to_string(a.rank) + " of " + std::to_string(a.P1) + "x" + std::to_string(a.P2) + " (exchange stubbed), chunks=" + std::to_string(dfft_get_pipeline_chunks(plan));
    printf("PLAN %s %zux%zux%zu %s %s%s | wave_err %.2e roundtrip %.2e | wall %.3f ms/step\n", a.label.c_str(), a.Nx, a.Ny, a.Nz,
           a.prec.c_str(), a.mode.c_str(), optstr.c_str(), wave_err, rt_err, wall / a.iters);
    const size_t Nzc = c2c ? a.Nz : a.Nz / 2 + 1;
    const double half = (double)a.Nx * a.Ny * Nzc * esz / nranks, real_b = (double)in_bytes;
    double tot = 0;
    for (int dir = 0; dir < 2; dir++) {
        const int np = dir == 0 ? nf : nb;
        for (int i = 0; i < np; i++) {
            const double avg = (dir == 0 ? accf[i] : accb[i]) / a.iters, mn = dir == 0 ? minf[i] : minb[i];
            if (avg <= 0) continue;
            const char *name = dfft_phase_name(i, dir == 0 ? DFFT_FORWARD : DFFT_INVERSE);
            const bool zpass = name[0] == 'z';
            const double bytes = zpass ? real_b + half : 2 * half;
            printf("  %-10s avg %8.3f ms  min %8.3f ms  %8.1f GB/s\n", name, avg, mn, bytes / avg / 1e6);
            tot += avg;
        }
    }
    printf("  total passes %.3f ms\n", tot);
    DCHK(dfft_plan_destroy(plan));
  }
    if (work) DCHK(dfft_free(work));
    if (a.slab) HIPCHK(hipFree(slab));
    else if (a.vmm_mib) { /* process exit unmaps */ }
    else if (a.tune > 1) { HIPCHK(hipFree(in)); DCHK(dfft_free(out)); DCHK(dfft_free(back)); }
    else {
        HIPCHK(hipFree(in)); DCHK(dfft_free(out));
        if (!alias_back) DCHK(dfft_free(back));
    }
    HIPCHK(hipFree(part));
    return 0;
}

int main(int argc, char **argv)
{
    Args a = parse(argc, argv);
    if (a.line) return a.prec == "f64" ? run_line<double>(a) : run_line<float>(a);
    return a.prec == "f64" ? run_plan<double>(a) : run_plan<float>(a);
}
