// alloc.hip -- placement-aware device memory of libdfft_amd.so: dfft_malloc / dfft_free / dfft_last_placement_info of include/dfft_c.h
// and the work areas the library owns.  (Its own translation unit since round 6: csrc/dfft.hip keeps the plan, the pass descriptors
// and the execution chains.)  No counterpart in the reference, whose buffers are plain cudaMalloc.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "alloc.hpp"
#include "host_common.hpp"

namespace dfft {

// ------------------------------------------------------------------------------------------
// Placement-aware device memory (dfft_malloc / dfft_free / dfft_tune_placement, no counterpart in the reference: its
// buffers are plain cudaMalloc, src/pencil/mpicufft_pencil_opt1.cpp:344-365).  The passes that scatter 128-byte runs
// (y and x) run 5-10 % faster or slower depending on the PHYSICAL backing of the buffer they write to, per buffer and
// repeatably for the life of the allocation (profiles/r2_placement_probe.txt, profiles/r3_placement.txt); what the
// driver hands out differs from allocation to allocation.  So buffers can be backed through the virtual-memory API in
// physical chunks of a chosen size, and a plan can try several backings and keep the one its own passes run fastest on.
struct DevAlloc { size_t bytes, chunk; int device; };     // chunk == 0: plain hipMalloc; device = the GPU that owns the memory
static std::mutex g_alloc_mu;
static std::map<void *, DevAlloc> g_allocs;

// A virtual address must not be MAPPED TWICE on this ROCm (7.2, gfx950).  A range that was unmapped and whose address is mapped
// again -- after hipMemAddressFree and a later hipMemAddressReserve that returns the same address (what the runtime does by itself),
// or by mapping fresh chunks into a reservation that was kept -- reads and writes wrong bytes, through kernels and through runtime
// copies alike, with or without other threads enqueueing: tools/vmm_reuse_repro.hip reproduces it without this library in 2-90 map /
// unmap cycles (profiles/r6_vmm_reuse_repro.txt; round 5 met it as wrong round trips of the relay, profiles/r5_relay_stress.txt).
// Keeping every reservation forever ("retiring", round 5) is no way out either: the physical memory of an unmapped range only
// returns to the device with hipMemAddressFree (profiles/r6_vmm_cost.txt: 1000 cycles of 64 MiB cost 62.5 GiB).  What works
// (the reproducer's mode `hint`, profiles/r6_vmm_hint.txt: 10^4 cycles, memory level, no mismatch in any configuration): ranges ARE
// returned, and every reservation names the address it wants -- the next one of a region of the address space that this library
// walks through once, [DFFT_VMM_BASE_TIB = 4 TiB, 80 TiB): between the heap and the mmap area of an x86-64 process, 76 TiB = 4800
// buffers of 16 GiB.  The runtime honours the hint (an address it does not is given back and the next one is tried); when the region
// is used up the reservation fails and the default backing falls back to hipMalloc: slower scatter target, correct bytes.
static std::atomic<uintptr_t> g_next_va{0};
static hipError_t reserve_fresh(void **va, size_t total, size_t align)
{
    static const uintptr_t base = [] { const char *e = getenv("DFFT_VMM_BASE_TIB"); const long v = e ? atol(e) : 4; return (uintptr_t)(v < 1 ? 1 : v > 64 ? 64 : v) << 40; }();
    const uintptr_t end = (uintptr_t)80 << 40;
    const uintptr_t al = align > ((uintptr_t)2 << 20) ? (uintptr_t)align : ((uintptr_t)2 << 20);      // a power of two (the chunk, or the granularity)
    const uintptr_t span = ((uintptr_t)total + al - 1) / al * al;
    uintptr_t zero = 0;
    g_next_va.compare_exchange_strong(zero, base);
    for (int attempt = 0; attempt < 16; attempt++) {
        const uintptr_t a = (g_next_va.fetch_add(span + al) + al - 1) / al * al;      // (one alignment unit of slack: a gap between neighbours)
        if (a + span > end) return hipErrorOutOfMemory;
        *va = nullptr;
        const hipError_t e = hipMemAddressReserve(va, total, align, reinterpret_cast<void *>(a), 0);
        if (e != hipSuccess) return e;
        if (*va == reinterpret_cast<void *>(a)) return hipSuccess;
        (void)hipMemAddressFree(*va, total);             // the runtime chose another address (something lives at the hint): not ours to trust
    }
    *va = nullptr;
    return hipErrorOutOfMemory;
}

int dev_free(void *ptr)
{
    if (!ptr) return 0;
    DevAlloc rec{0, 0, -1};
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_allocs.find(ptr);
        if (it != g_allocs.end()) { rec = it->second; g_allocs.erase(it); }
    }
    if (!rec.chunk) { HIP_TRY(hipFree(ptr)); return 0; }
    // hipFree synchronises implicitly, hipMemUnmap does not: work still in flight on ANY stream (the plan's, torch's, a peer
    // device's pull in an in-process world) must not lose its mapping under it.  The device to drain is the one that OWNS the
    // range, which need not be the calling thread's current one (an in-process world over several GPUs).
    int cur = -1;
    (void)hipGetDevice(&cur);
    const bool hop = rec.device >= 0 && cur >= 0 && rec.device != cur;
    if (hop) HIP_TRY(hipSetDevice(rec.device));
    hipError_t e = hipDeviceSynchronize();
    for (size_t off = 0; off < rec.bytes && e == hipSuccess; off += rec.chunk) e = hipMemUnmap(static_cast<char *>(ptr) + off, rec.chunk);
    // The range goes back to the runtime (only hipMemAddressFree returns the PHYSICAL memory of an unmapped range on this ROCm: a
    // range that is unmapped but kept costs its full size until the process ends, profiles/r6_vmm_cost.txt) -- and its addresses are
    // never used again by this library: every reservation asks for a fresh address (reserve_fresh below).
    if (e == hipSuccess) e = hipMemAddressFree(ptr, rec.bytes);
    if (hop) (void)hipSetDevice(cur);
    if (e != hipSuccess) { set_error(std::string("dfft_free: ") + hipGetErrorString(e)); return (int)e; }
    return 0;
}
// chunk_mib == 0: hipMalloc.  Otherwise one virtual range backed by physical allocations of chunk_mib MiB each.
// spread > 1: `spread` times as many physical chunks are created as the buffer needs, every spread-th is mapped and the others are
// released afterwards, so that the buffer's chunks lie `spread` chunks apart in the order the driver hands them out (see
// dev_alloc_default: a buffer whose chunks are spread over the physical space is a good scatter target)
// where the last dev_alloc of this thread spent its time (the placement report: dfft_last_placement_info)
struct AllocTimes { double create = 0, map = 0, release = 0; };
static thread_local AllocTimes g_alloc_times;
static double seconds_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

int dev_alloc(size_t bytes, size_t chunk_mib, void **out, int spread)
{
    *out = nullptr;
    g_alloc_times = AllocTimes();
    if (!bytes) return fail(ERR_ARG, "zero-sized allocation");
    if (!chunk_mib) {
        int dev0 = -1;
        (void)hipGetDevice(&dev0);
        HIP_TRY(hipMalloc(out, bytes));
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        g_allocs[*out] = DevAlloc{bytes, 0, dev0};
        return 0;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (!gran) gran = 4096;
    size_t chunk = chunk_mib << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    // a buffer smaller than the chunk gets one chunk of its own (granularity-rounded) size: a 1 GiB recipe must not turn a
    // buffer of a few MiB into 1 GiB of physical memory
    const size_t whole = (bytes + gran - 1) / gran * gran;
    if (chunk > whole) chunk = whole;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void *va = nullptr;
    // alignment: the chunk when it is a power of two (physical chunks then sit on their natural boundaries), else the granularity
    const size_t align = (chunk & (chunk - 1)) == 0 ? chunk : gran;
    HIP_TRY(reserve_fresh(&va, total, align));
    size_t mapped = 0;
    hipError_t err = hipSuccess;
    if (spread > 1) {
        const size_t n = total / chunk;
        std::vector<hipMemGenericAllocationHandle_t> all;
        all.reserve(n * (size_t)spread);
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n * (size_t)spread && err == hipSuccess; i++) {
            hipMemGenericAllocationHandle_t h;
            err = hipMemCreate(&h, chunk, &prop, 0);
            if (err == hipSuccess) all.push_back(h);
        }
        g_alloc_times.create = seconds_since(t0);
        t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n && err == hipSuccess; i++, mapped += chunk)
            err = hipMemMap(static_cast<char *>(va) + i * chunk, chunk, 0, all[i * (size_t)spread], 0);
        g_alloc_times.map = seconds_since(t0);
        t0 = std::chrono::steady_clock::now();
        for (auto &h : all) (void)hipMemRelease(h);      // (a mapping keeps its physical memory alive; the chunks in between go back)
        g_alloc_times.release = seconds_since(t0);
    } else
    for (; mapped < total && err == hipSuccess; mapped += chunk) {
        hipMemGenericAllocationHandle_t h;
        err = hipMemCreate(&h, chunk, &prop, 0);
        if (err != hipSuccess) break;
        err = hipMemMap(static_cast<char *>(va) + mapped, chunk, 0, h, 0);
        (void)hipMemRelease(h);      // the mapping keeps the physical memory alive
        if (err != hipSuccess) break;
    }
    if (err == hipSuccess) {
        // read/write for this device and for every device that can reach it: the virtual ranks of an in-process world on several
        // GPUs pull from each other's buffers with device-to-device copies (hipMalloc memory is peer-visible once peer access is
        // enabled; a virtual-memory range needs the grant per device)
        std::vector<hipMemAccessDesc> acc;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = dev + 1;
        for (int d = 0; d < ndev; d++) {
            int can = d == dev;
            if (d != dev && hipDeviceCanAccessPeer(&can, d, dev) != hipSuccess) can = 0;
            if (!can) continue;
            hipMemAccessDesc a = {};
            a.location.type = hipMemLocationTypeDevice;
            a.location.id = d;
            a.flags = hipMemAccessFlagsProtReadWrite;
            acc.push_back(a);
        }
        err = hipMemSetAccess(va, total, acc.data(), acc.size());
        if (err != hipSuccess && acc.size() > 1) {      // a peer grant the driver refuses must not cost the local mapping
            hipMemAccessDesc own = {};
            own.location = prop.location;
            own.flags = hipMemAccessFlagsProtReadWrite;
            (void)hipGetLastError();
            err = hipMemSetAccess(va, total, &own, 1);
        }
    }
    if (err != hipSuccess) {
        for (size_t off = 0; off < mapped; off += chunk) (void)hipMemUnmap(static_cast<char *>(va) + off, chunk);
        (void)hipMemAddressFree(va, total);
        set_error(std::string("virtual-memory allocation failed: ") + hipGetErrorString(err));
        return (int)err;
    }
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        g_allocs[va] = DevAlloc{total, chunk, dev};
    }
    *out = va;
    return 0;
}

// The default backing of library-owned device memory (work areas, dfft_malloc(DFFT_CHUNK_DEFAULT)): the virtual-memory API with
// 1 GiB physical chunks.  Buffers backed this way are on average better targets for the passes that scatter 128-byte runs than
// hipMalloc buffers (1024^3 fp64 C2C on one GPU, fresh processes: hipMalloc 36.9-37.5 ms per forward + inverse in every one;
// 1 GiB chunks 34.9 / 35.3 / 35.0 / 35.6 but 37.2 right after 64 GiB buffers were freed; 2 MiB 35.0-36.0; 16, 64, 256 MiB and
// shuffled mapping orders the same 34.2-36.8 scatter; the search of dfft_tune_placement 33.6-34.4; profiles/r4_fixed_recipes.txt,
// r4_placement_shuffle.txt).  Neither the chunk size nor the order the chunks are mapped in controls which case one gets -- it is
// the physical pages -- so the recipe is the one with the fewest mappings, and the search stays the way to a guaranteed result.  Falls back to smaller chunks
// (fragmented memory) and finally to hipMalloc, so it never fails where hipMalloc would succeed.  DFFT_DEFAULT_CHUNK_MIB overrides
// (0 = plain hipMalloc, what rounds 1-3 used for work areas).
size_t default_chunk_mib()
{
    static const size_t v = [] {
        const char *e = getenv("DFFT_DEFAULT_CHUNK_MIB");
        return e ? (size_t)atol(e) : (size_t)1024;
    }();
    return v;
}
static int dev_alloc_recipe(size_t bytes, void **out)
{
    for (size_t c = default_chunk_mib(); c >= 2; c /= 8) {
        if (dev_alloc(bytes, c, out) == 0) return 0;
        (void)hipGetLastError();
    }
    return dev_alloc(bytes, 0, out);
}

// Streaming-write rate of a buffer in bytes per second (best of two passes after a warm-up pass).  It is the cheapest thing that
// tells a good target of the scatter passes from a bad one: ten 16 GiB buffers from the same recipe in one process streamed
// writes at 6.5 TB/s (three of them) or 5.2 TB/s (seven), and the plan's x pass ran 5.63-5.66 ms on exactly the former and
// 6.47-6.55 ms on the latter (tools/placeprobe.hip, profiles/r4_placement_probe.txt).  Local to the device: no plan, no collective.
__global__ __launch_bounds__(512) void placement_probe_kernel(double2 *d, size_t n)
{
    const double2 v = make_double2(1.0, 2.0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = v;
}
static double placement_probe(void *buf, size_t bytes)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return 0.0; }
    float best = 0;
    for (int r = 0; r < 3; r++) {
        float ms = 0;
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(placement_probe_kernel, dim3(4096), dim3(512), 0, nullptr, static_cast<double2 *>(buf), bytes / sizeof(double2));
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { best = 0; break; }
        if (r && (best == 0 || ms < best)) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    return best > 0 ? (double)bytes / (best * 1e-3) : 0.0;
}

// Default backing, placement-aware.  A buffer of 1 GiB and more is BUILT from chunks that lie far apart (dev_alloc_default: every K-th
// of K times as many, as far as the memory budget below allows) and probed with a streaming write (3 passes: 8 ms for 16 GiB): the good
// class by construction; the first one of a process gives the device its yardstick.  A built buffer at >= 0.95 x the yardstick is kept;
// where there is no room for a pool, or the built buffer falls short, plain candidates are drawn, up to DFFT_PLACEMENT_TRIES (default 6)
// -- every candidate stays alive while the next is tried: a freed candidate's physical pages would simply be handed out again -- and
// the fastest of everything probed is kept.
// Nothing is absolute: the yardstick is measured on the device at hand, and a physically contiguous buffer (hipMalloc, 2 GiB: the slow
// case by construction, profiles/r4_placement_probe.txt) is probed once to check that the built buffer really is of another class
// (>= 1.2 x: MI355X 4.5 - 4.6 TB/s against 6.5 - 7.0); DFFT_PLACEMENT_GOOD_TBPS sets an absolute threshold instead.
// Bounded: everything alive during the search -- the spread pool or the drawn candidates -- stays within HALF of the free memory
// divided by DFFT_RANKS_PER_DEVICE (processes that share the GPU all see the same free figure), and whatever fails on the way
// (a racing process took the memory) ends in the plain recipe and finally in hipMalloc: the call never fails where hipMalloc succeeds.
// Purely local: every rank of a multi-rank plan does it by itself.  DFFT_PLACEMENT_TRIES=1 switches the probe off,
// DFFT_PLACEMENT_SPREAD (default 5, 1 = off, <= 8) is K.  dfft_last_placement_info says what the last call did.
// Measured (1024^3 fp64, fresh processes): profiles/bench_r4d*.json (K = 8, pool always), profiles/r5_allocator.txt, profiles/bench_r5*.json.
struct PlacementInfo {
    size_t bytes = 0;
    int spread = 0, drawn = 0, fallback = 0;
    double rate = 0, ref_rate = 0, threshold = 0, seconds = 0;
    double ref_s = 0, create_s = 0, map_s = 0, release_s = 0, probe_s = 0;      // where the seconds went: reference probe, pool create / map / release, probes
    const char *kept = "none";
};
static std::mutex g_place_mu;
static PlacementInfo g_place_last;
static std::map<int, double> g_place_ref;      // device -> streaming rate of a contiguous buffer

static double placement_reference_rate()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0.0;
    {
        std::lock_guard<std::mutex> lk(g_place_mu);
        auto it = g_place_ref.find(dev);
        if (it != g_place_ref.end()) return it->second;
    }
    double rate = 0.0;
    void *ref = nullptr;
    const size_t rb = (size_t)2 << 30;
    if (hipMalloc(&ref, rb) == hipSuccess) {
        rate = placement_probe(ref, rb);
        (void)hipFree(ref);
    } else (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_place_mu);
    g_place_ref[dev] = rate;
    return rate;
}

static std::map<int, double> g_place_good;     // device -> streaming rate of a buffer BUILT from chunks far apart (the good class, learned once)

int dev_alloc_default(size_t bytes, void **out)
{
    if (bytes < ((size_t)32 << 20)) return dev_alloc(bytes, 0, out);      // small buffers live in the caches: nothing to place
    static const int tries = [] { const char *e = getenv("DFFT_PLACEMENT_TRIES"); const int v = e ? atoi(e) : 6; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    static const int spread_want = [] { const char *e = getenv("DFFT_PLACEMENT_SPREAD"); const int v = e ? atoi(e) : 5; return v < 1 ? 1 : v > 8 ? 8 : v; }();
    static const int sharers = [] { const char *e = getenv("DFFT_RANKS_PER_DEVICE"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    static const double abs_good = [] { const char *e = getenv("DFFT_PLACEMENT_GOOD_TBPS"); return e ? atof(e) * 1e12 : 0.0; }();
    if (bytes < ((size_t)1 << 30) || tries == 1 || default_chunk_mib() == 0) return dev_alloc_recipe(bytes, out);
    const auto t0 = std::chrono::steady_clock::now();
    int dev = 0;
    (void)hipGetDevice(&dev);
    PlacementInfo info;
    info.bytes = bytes;
    info.ref_rate = abs_good > 0 ? 0.0 : placement_reference_rate();
    info.ref_s = seconds_since(t0);
    // The yardstick is measured on THIS device: the rate of a buffer built from chunks far apart (the good class by construction),
    // learned by the first large allocation of the process; a candidate is good at >= 0.95 x it (measured: bad 5.2 - 5.9 TB/s, built
    // buffers 6.4 - 7.0; a small buffer that probes a few % low draws a few more candidates, milliseconds each).  The contiguous
    // reference only tells whether the built buffer really is of another class (>= 1.2 x), i.e. whether the yardstick means anything.
    double yard = 0.0;
    {
        std::lock_guard<std::mutex> lk(g_place_mu);
        auto it = g_place_good.find(dev);
        if (it != g_place_good.end()) yard = it->second;
    }
    auto threshold = [&]() { return abs_good > 0 ? abs_good : 0.95 * yard; };      // 0: nothing known yet
    auto good = [&](double rate) { return rate == 0.0 || (threshold() > 0 && rate >= threshold()); };
    // what this call may hold alive at any time, this buffer included
    auto budget = [&]() -> size_t {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return free_b / 2 / (size_t)sharers;
    };
    auto done = [&](void *ptr, const char *kept, double rate) {
        info.kept = kept; info.rate = rate; info.threshold = threshold();
        info.seconds = seconds_since(t0);
        std::lock_guard<std::mutex> lk(g_place_mu);
        g_place_last = info;
        *out = ptr;
        return 0;
    };
    void *best = nullptr;
    double best_rate = -1.0;
    std::vector<void *> losers;
    auto probe = [&](void *cand) {
        const auto tp = std::chrono::steady_clock::now();
        const double r = placement_probe(cand, bytes);
        info.probe_s += seconds_since(tp);
        return r;
    };
    auto take = [&](void *cand, double rate) {
        if (!best || rate > best_rate) { if (best) losers.push_back(best); best = cand; best_rate = rate; }
        else losers.push_back(cand);
    };
    auto room_for_one_more = [&]() { return (size_t)(losers.size() + (best ? 1 : 0) + 1) * bytes <= std::max(budget(), bytes); };
    auto finish = [&](const char *kept) {
        for (void *l : losers) (void)dev_free(l);
        return done(best, kept, best_rate);
    };
    // A buffer BUILT to be good: 1 GiB chunks taken every K-th from K times as many.  16 chunks written at once stream at 6.0 TB/s when
    // they are physical neighbours and 6.6 TB/s when 8 GiB apart; buffers built with K = 8, 5, 4 and 3 were good in every run
    // (tools/kbench --vmm-spread, profiles/r4_placement_probe.txt, profiles/r5_allocator.txt).  Creating the pool costs ~30 ms per GiB
    // inside hipMemCreate (the driver clears fresh memory: tools/vmm_cycle): 2 - 4 s for a 16 GiB buffer.
    auto build = [&]() -> bool {
        const size_t bud = budget(), held = (size_t)(losers.size() + (best ? 1 : 0)) * bytes;
        const int K = default_chunk_mib() >= 256 && bud > held ? (int)std::min<size_t>((size_t)spread_want, (bud - held) / bytes) : 1;
        void *cand = nullptr;
        if (K < 3 || dev_alloc(bytes, default_chunk_mib(), &cand, K) != 0) { (void)hipGetLastError(); return false; }
        info.spread = K;
        info.create_s = g_alloc_times.create; info.map_s = g_alloc_times.map; info.release_s = g_alloc_times.release;
        const double rate = probe(cand);
        // a built buffer that is clearly of another class than the contiguous reference is (and raises) the device's yardstick
        if (rate > 0 && (info.ref_rate == 0.0 || rate >= 1.2 * info.ref_rate) && rate > yard) {
            yard = rate;
            std::lock_guard<std::mutex> lk(g_place_mu);
            g_place_good[dev] = rate;
        }
        take(cand, rate);
        return true;
    };
    // 1. build the buffer (2 - 4 s per 16 GiB, see above): the first one of a process also gives the device its yardstick.  Building
    //    comes first because it gave the best scatter passes in every measurement (1024^3 fp64: 33.4 - 33.8 ms per forward + inverse
    //    with every buffer built, 34.1 - 34.6 with plain candidates accepted first: profiles/bench_r4d*.json, profiles/r5_allocator.txt);
    //    DFFT_PLACEMENT_SPREAD=1 skips it (milliseconds, plain candidates only)
    const bool first = threshold() == 0.0;
    const bool built = build();
    // (a first built buffer that is not clearly of another class than the contiguous reference -- 1.3 x -- is no yardstick: candidates follow)
    if (built && (first ? (info.ref_rate == 0.0 || best_rate >= 1.3 * info.ref_rate) : good(best_rate)))
        return finish(first ? "built from chunks K apart (first large allocation: the device's yardstick)" : "built from chunks K apart");
    // 2. no room for a pool, or the built buffer falls short: plain candidates (milliseconds each), all alive (a freed candidate's pages
    //    would simply be handed out again), until one is good; the fastest of everything probed is kept
    for (int t = 0; t < tries; t++) {
        if (best && !room_for_one_more()) break;
        void *cand = nullptr;
        if (dev_alloc_recipe(bytes, &cand) != 0) { (void)hipGetLastError(); break; }
        info.drawn++;
        take(cand, probe(cand));
        if (threshold() > 0 ? good(best_rate) : !built) break;      // (no yardstick and no room to build one: the first candidate is it)
    }
    if (best && threshold() == 0.0 && best_rate > 0 && (info.ref_rate == 0.0 || best_rate >= 1.2 * info.ref_rate)) {
        // nothing built was fit to be the yardstick: the best buffer seen is
        yard = best_rate;
        std::lock_guard<std::mutex> lk(g_place_mu);
        g_place_good[dev] = best_rate;
    }
    if (!best) {
        // nothing could be created with the search's footprint (another process took the memory in between, a tight device): the plain
        // recipe once more on its own, then hipMalloc -- where that succeeds, so does this call
        info.fallback = 1;
        void *cand = nullptr;
        int rc = dev_alloc_recipe(bytes, &cand);
        if (rc != 0) { (void)hipGetLastError(); rc = dev_alloc(bytes, 0, &cand); }
        if (rc != 0) return rc;
        return done(cand, "plain (the search found no room)", 0.0);
    }
    if (first && !built) return finish("the first plain candidate (no room to build a yardstick)");
    return finish(threshold() > 0 && good(best_rate) ? "the fastest candidate" : "the fastest candidate (none reached the threshold)");
}

int placement_info_json(char *buf, size_t capacity)
{
    PlacementInfo i;
    {
        std::lock_guard<std::mutex> lk(g_place_mu);
        i = g_place_last;
    }
    snprintf(buf, capacity, "{\"bytes\": %zu, \"spread_K\": %d, \"candidates_drawn\": %d, \"fallback\": %d, \"probe_TBps\": %.3f, "
                            "\"contiguous_reference_TBps\": %.3f, \"good_threshold_TBps\": %.3f, \"seconds\": %.3f, "
                            "\"seconds_reference_probe\": %.3f, \"seconds_pool_create\": %.3f, \"seconds_pool_map\": %.3f, \"seconds_pool_release\": %.3f, "
                            "\"seconds_probes\": %.3f, \"kept\": \"%s\"}",
             i.bytes, i.spread, i.drawn, i.fallback, i.rate / 1e12, i.ref_rate / 1e12, i.threshold / 1e12, i.seconds, i.ref_s, i.create_s, i.map_s,
             i.release_s, i.probe_s, i.kept);
    return 0;
}
}  // namespace dfft
