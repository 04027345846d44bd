// tools/vmm_reuse_repro.hip -- standalone (no libdfft): does a virtual-memory ADDRESS RANGE that is handed out again corrupt
// runtime copies?  Round 5 saw the relay's staging -- hipMemCreate / hipMemMap memory that is only ever touched by the transport's
// hipMemcpyAsync -- deliver wrong bytes after it had been unmapped, hipMemAddressFree'd and re-created while the other virtual
// ranks' host threads kept enqueueing (profiles/r5_relay_stress.txt); keeping the reservation made it go away.  This program
// separates the candidates without the library:
//
//   cycler thread:  reserve -> hipMemCreate + hipMemMap (chunks) -> hipMemSetAccess -> [use] -> sync -> hipMemUnmap -> release
//                   --mode free    then hipMemAddressFree           (the runtime may hand the same address out again)
//                   --mode keep    the reservation is kept and the NEXT cycle maps fresh physical chunks into it
//                   --mode retire  the reservation is kept and never used again (a new range per cycle)
//                   --mode hint    hipMemAddressFree like `free`, but every reservation asks for an address that was never used before
//                                  (the `addr` argument of hipMemAddressReserve, advancing from --hint-base-tib): memory goes back to the
//                                  device AND no address is mapped twice -- if the runtime honours the hint
//                   --mode arena   one large reservation (--arena-gib), every cycle maps into the next unused part of it: no
//                                  address is used twice and the runtime sees ONE reservation (what the library's allocator does)
//   --report N  prints the mean seconds per cycle of every N cycles (does the cost of a cycle grow with the ranges retired so far?)
//       [use] = the pattern of the cycle is written into a hipMalloc source by a kernel, copied source -> range -> sink with
//               hipMemcpyAsync (--use copy, like the staging) or by kernels (--use kernel), and the sink is checked by a kernel
//   N worker threads: each on its own stream and its own hipMalloc buffers: fill kernel, hipMemcpyAsync, check kernel, forever
//                   (--pull 1: workers also hipMemcpyAsync OUT OF the cycler's current range between its [use] and its unmap,
//                    like the peers of an in-process world that pull from a rank's staging)
//
// prints, per run: cycles, how many reservations came back at an address seen before, mismatching words seen by the cycler and
// by the workers.  Exit code 0 = no mismatch, 1 = mismatches, 2 = a HIP call failed.
//   hipcc -O2 --offload-arch=gfx950 tools/vmm_reuse_repro.hip -o tools/vmm_reuse_repro -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); g_hip_failed = true; return; } } while (0)
static std::atomic<bool> g_hip_failed{false}, g_stop{false};
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void fill(uint64_t *p, size_t n, uint64_t tag)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag * 0x9E3779B97F4A7C15ull + i;
}
__global__ void copyk(uint64_t *d, const uint64_t *s, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void check(const uint64_t *p, size_t n, uint64_t tag, unsigned long long *bad)
{
    unsigned long long b = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != tag * 0x9E3779B97F4A7C15ull + i;
    if (b) atomicAdd(bad, b);
}

struct Opt {
    std::string mode = "free", use = "copy";
    int threads = 8, pull = 0, grow = 1;
    double seconds = 20;
    size_t mib = 64, chunk_mib = 2, arena_gib = 256;
    int report = 0;
    size_t hint_base_tib = 32;
    int release_late = 0;      // --release-late 1: hipMemRelease of a chunk's handle AFTER its hipMemUnmap instead of right after hipMemMap
};
static Opt O;

// the range the cycler currently offers to the pullers: (pointer, words, tag), valid while `busy` readers are counted in
static std::mutex g_mu;
static uint64_t *g_cur = nullptr;
static size_t g_cur_n = 0;
static uint64_t g_cur_tag = 0;
static std::atomic<unsigned long long> g_worker_bad{0}, g_worker_iters{0}, g_pulls{0};

static void worker(int id)
{
    CHK(hipSetDevice(0));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t n = (size_t)(8 + id) << 17;      // 8..16 MiB of uint64
    uint64_t *a, *b, *c;
    unsigned long long *bad;
    CHK(hipMalloc(&a, n * 8)); CHK(hipMalloc(&b, n * 8)); CHK(hipMalloc(&c, (O.mib << 20) * 2)); CHK(hipMalloc(&bad, 8));
    CHK(hipMemset(bad, 0, 8));
    for (uint64_t it = 1; !g_stop && !g_hip_failed; it++) {
        const uint64_t tag = it * 131 + id;
        fill<<<256, 256, 0, s>>>(a, n, tag);
        CHK(hipMemcpyAsync(b, a, n * 8, hipMemcpyDeviceToDevice, s));
        check<<<256, 256, 0, s>>>(b, n, tag, bad);
        if (O.pull) {      // pull out of the cycler's range while it is offered (the lock covers the ENQUEUE + completion: the cycler unmaps after)
            std::lock_guard<std::mutex> lk(g_mu);
            if (g_cur) {
                CHK(hipMemcpyAsync(c, g_cur, g_cur_n * 8, hipMemcpyDeviceToDevice, s));
                check<<<256, 256, 0, s>>>(c, g_cur_n, g_cur_tag, bad);
                CHK(hipStreamSynchronize(s));
                g_pulls++;
            }
        }
        if (it % 8 == 0) CHK(hipStreamSynchronize(s));
        g_worker_iters++;
    }
    CHK(hipStreamSynchronize(s));
    unsigned long long h = 0;
    CHK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    g_worker_bad += h;
}

struct CyclerResult { unsigned long long cycles = 0, reused = 0, bad = 0, first_bad_cycle = 0; };
static CyclerResult g_res;
static unsigned long long g_hint_missed = 0;

static void cycler()
{
    CHK(hipSetDevice(0));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t chunk = O.chunk_mib << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t max_bytes = (O.mib << 20) * 2;
    uint64_t *src, *sink;
    unsigned long long *bad;
    CHK(hipMalloc(&src, max_bytes)); CHK(hipMalloc(&sink, max_bytes)); CHK(hipMalloc(&bad, 8));
    CHK(hipMemset(bad, 0, 8));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::set<void *> seen;
    void *kept = nullptr;
    size_t kept_bytes = 0;
    char *arena = nullptr;
    size_t arena_used = 0;
    const size_t arena_bytes = O.arena_gib << 30;
    if (O.mode == "arena") { void *a = nullptr; CHK(hipMemAddressReserve(&a, arena_bytes, chunk, nullptr, 0)); arena = (char *)a; }
    const double t_end = now() + O.seconds;
    double t_rep = now();
    for (uint64_t cyc = 1; now() < t_end && !g_hip_failed; cyc++) {
        if (O.report && cyc % O.report == 0) { const double t = now(); printf("  cycles %llu..%llu: %.3f ms per cycle\n", (unsigned long long)(cyc - O.report), (unsigned long long)cyc, (t - t_rep) / O.report * 1e3); t_rep = t; }
        // sizes grow and shrink like a staging buffer that meets a larger exchange table (grow = 0: one size)
        size_t bytes = (O.mib << 20) * (O.grow ? 1 + cyc % 2 : 1);
        bytes = (bytes + chunk - 1) / chunk * chunk;
        void *va = nullptr;
        if (O.mode == "arena") {
            if (arena_used + bytes > arena_bytes) break;      // the arena is used up
            va = arena + arena_used;
            arena_used += bytes;
        } else if (O.mode == "hint") {
            static char *next_hint = nullptr;
            if (!next_hint) next_hint = reinterpret_cast<char *>(O.hint_base_tib << 40);
            CHK(hipMemAddressReserve(&va, bytes, chunk, next_hint, 0));
            if (va != next_hint) g_hint_missed++;
            next_hint += (bytes + ((size_t)1 << 30) - 1) >> 30 << 30;      // the next hint: a fresh GiB-aligned address
        } else if (O.mode == "keep" && kept && kept_bytes >= bytes) va = kept;
        else {
            CHK(hipMemAddressReserve(&va, O.mode == "keep" ? max_bytes / chunk * chunk + chunk : bytes, chunk, nullptr, 0));
            if (O.mode == "keep") { kept = va; kept_bytes = max_bytes / chunk * chunk + chunk; }
        }
        if (!seen.insert(va).second) g_res.reused++;
        std::vector<hipMemGenericAllocationHandle_t> handles;
        for (size_t off = 0; off < bytes; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            CHK(hipMemCreate(&h, chunk, &prop, 0));
            CHK(hipMemMap((char *)va + off, chunk, 0, h, 0));
            if (O.release_late) handles.push_back(h); else CHK(hipMemRelease(h));
        }
        CHK(hipMemSetAccess(va, bytes, &acc, 1));
        const size_t n = bytes / 8;
        const uint64_t tag = cyc * 7919;
        fill<<<512, 256, 0, s>>>(src, n, tag);
        if (O.use == "copy") {
            CHK(hipMemcpyAsync(va, src, bytes, hipMemcpyDeviceToDevice, s));
            CHK(hipMemcpyAsync(sink, va, bytes, hipMemcpyDeviceToDevice, s));
        } else {
            copyk<<<512, 256, 0, s>>>((uint64_t *)va, src, n);
            copyk<<<512, 256, 0, s>>>(sink, (const uint64_t *)va, n);
        }
        check<<<512, 256, 0, s>>>(sink, n, tag, bad);
        CHK(hipStreamSynchronize(s));
        if (O.pull) {
            { std::lock_guard<std::mutex> lk(g_mu); g_cur = (uint64_t *)va; g_cur_n = n; g_cur_tag = tag; }
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            { std::lock_guard<std::mutex> lk(g_mu); g_cur = nullptr; }      // every pull that saw it has completed (lock held across its sync)
        }
        unsigned long long h = 0;
        CHK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        if (h && !g_res.first_bad_cycle) g_res.first_bad_cycle = cyc;
        g_res.bad = h;
        CHK(hipDeviceSynchronize());      // like dfft_free: nothing in flight anywhere may lose its mapping
        for (size_t off = 0; off < bytes; off += chunk) CHK(hipMemUnmap((char *)va + off, chunk));
        for (auto h : handles) CHK(hipMemRelease(h));
        if (O.mode == "free" || O.mode == "hint") CHK(hipMemAddressFree(va, bytes));
        if (O.report && cyc % O.report == 0) { size_t fr = 0, tot = 0; CHK(hipMemGetInfo(&fr, &tot)); printf("  after cycle %llu: %.1f GiB of %.1f free\n", (unsigned long long)cyc, fr / 1073741824.0, tot / 1073741824.0); }
        g_res.cycles = cyc;
    }
}

int main(int argc, char **argv)
{
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--mode") O.mode = v; else if (k == "--use") O.use = v; else if (k == "--threads") O.threads = atoi(v.c_str());
        else if (k == "--pull") O.pull = atoi(v.c_str()); else if (k == "--grow") O.grow = atoi(v.c_str());
        else if (k == "--seconds") O.seconds = atof(v.c_str()); else if (k == "--mib") O.mib = (size_t)atol(v.c_str());
        else if (k == "--chunk-mib") O.chunk_mib = (size_t)atol(v.c_str());
        else if (k == "--arena-gib") O.arena_gib = (size_t)atol(v.c_str());
        else if (k == "--report") O.report = atoi(v.c_str());
        else if (k == "--release-late") O.release_late = atoi(v.c_str());
        else if (k == "--hint-base-tib") O.hint_base_tib = (size_t)atol(v.c_str());
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    std::vector<std::thread> th;
    for (int i = 0; i < O.threads; i++) th.emplace_back(worker, i);
    std::thread c(cycler);
    c.join();
    g_stop = true;
    for (auto &t : th) t.join();
    printf("mode=%-6s use=%-6s threads=%d pull=%d grow=%d %zu MiB in %zu MiB chunks, %.0f s: %llu cycles, %llu reservations at an address seen before, "
           "cycler mismatches %llu (first in cycle %llu), worker iterations %llu pulls %llu mismatches %llu%s\n",
           O.mode.c_str(), O.use.c_str(), O.threads, O.pull, O.grow, O.mib, O.chunk_mib, O.seconds, g_res.cycles, g_res.reused, g_res.bad, g_res.first_bad_cycle,
           (unsigned long long)g_worker_iters, (unsigned long long)g_pulls, (unsigned long long)g_worker_bad, g_hip_failed ? "  [A HIP CALL FAILED]" : "");
    if (O.mode == "hint") printf("   hints not honoured: %llu of %llu\n", g_hint_missed, g_res.cycles);
    if (g_hip_failed) return 2;
    return g_res.bad || g_worker_bad ? 1 : 0;
}
