// tools/vmm_cycle.hip -- what creating and releasing many 1 GiB physical chunks (hipMemCreate / hipMemRelease) costs, and when: the
// placement-aware allocator (csrc/dfft.hip dev_alloc_default) builds a buffer from every K-th of K times as many chunks and releases
// the others; round 5 saw one such call take 6 s inside hipMemCreate right after another had released 64 GiB.
//   tools/vmm_cycle PATTERN      a: pool 80 keep 16, pool 80 keep 16     b: the same after a 1-chunk create/release
//                                c: the same with a 3 s pause between     d: pool 48 keep 16 twice (K = 3)
//                                e: two pools of 80 created before anything is released
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <chrono>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pool { std::vector<hipMemGenericAllocationHandle_t> h; };
static hipMemAllocationProp prop;
static const size_t CH = (size_t)1 << 30;

static int create(Pool &p, int n, const char *what)
{
    const double t0 = now();
    for (int i = 0; i < n; i++) { hipMemGenericAllocationHandle_t h; CHK(hipMemCreate(&h, CH, &prop, 0)); p.h.push_back(h); }
    printf("  %-34s create %3d chunks: %7.3f s\n", what, n, now() - t0);
    return 0;
}
// keeps every K-th chunk MAPPED (into va), releases every handle (a mapping keeps its memory)
static int keep_and_release(Pool &p, int K, void **va_out, const char *what)
{
    const int n = (int)p.h.size() / K;
    void *va = nullptr;
    CHK(hipMemAddressReserve(&va, n * CH, CH, nullptr, 0));
    double t0 = now();
    for (int i = 0; i < n; i++) CHK(hipMemMap((char *)va + i * CH, CH, 0, p.h[(size_t)i * K], 0));
    const double tm = now() - t0;
    t0 = now();
    for (auto &h : p.h) CHK(hipMemRelease(h));
    printf("  %-34s map %d: %.3f s, release %zu handles: %.3f s\n", what, n, tm, p.h.size(), now() - t0);
    p.h.clear();
    *va_out = va;
    return 0;
}
static int unmap(void *va, int n, const char *what)
{
    const double t0 = now();
    for (int i = 0; i < n; i++) CHK(hipMemUnmap((char *)va + i * CH, CH));
    CHK(hipMemAddressFree(va, n * CH));
    printf("  %-34s unmap %d: %.3f s\n", what, n, now() - t0);
    return 0;
}

int main(int argc, char **argv)
{
    const char pat = argc > 1 ? argv[1][0] : 'a';
    CHK(hipSetDevice(0));
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    printf("pattern %c\n", pat);
    Pool p, q;
    void *v1 = nullptr, *v2 = nullptr;
    const int K = pat == 'd' ? 3 : 5, n = 16;
    if (pat == 'b') { Pool w; if (create(w, 1, "warm-up")) return 2; void *vw; if (keep_and_release(w, 1, &vw, "warm-up")) return 2; if (unmap(vw, 1, "warm-up")) return 2; }
    if (pat == 'e') {
        if (create(p, n * K, "pool 1") || create(q, n * K, "pool 2")) return 2;
        if (keep_and_release(p, K, &v1, "pool 1") || keep_and_release(q, K, &v2, "pool 2")) return 2;
    } else {
        if (create(p, n * K, "pool 1") || keep_and_release(p, K, &v1, "pool 1")) return 2;
        if (pat == 'c') { sleep(3); printf("  (3 s pause)\n"); }
        if (create(q, n * K, "pool 2") || keep_and_release(q, K, &v2, "pool 2")) return 2;
    }
    Pool r;
    void *v3 = nullptr;
    if (create(r, n * K, "pool 3") || keep_and_release(r, K, &v3, "pool 3")) return 2;
    if (unmap(v1, n, "buffer 1") || unmap(v2, n, "buffer 2") || unmap(v3, n, "buffer 3")) return 2;
    Pool s;
    if (create(s, n * K, "pool 4 (after the frees)")) return 2;
    for (auto &h : s.h) CHK(hipMemRelease(h));
    return 0;
}
