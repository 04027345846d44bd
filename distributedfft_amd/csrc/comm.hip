// comm.hip -- the three exchange transports (see comm.hpp).
#include "comm.hpp"
#include "dfft_internal.hpp"
#include "../../include/dfft_c.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace dfft {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return (int)e_;                                                                    \
        }                                                                                      \
    } while (0)

// ------------------------------------------------------------------------------------------
// LocalWorld: P virtual ranks in one process on one device, one host thread per rank.
// The all-to-all is "pull": every rank publishes its send buffer + an event recorded after
// its producer kernel, ranks meet in a host barrier, then each rank enqueues one D2D copy per
// peer on its own stream.  Stands in for MPI ranks sharing a GPU
// (tests/src/pencil/random_dist_3D.cu:175-177, cudaSetDevice(rank % dev_count)).
// ------------------------------------------------------------------------------------------
struct LocalWorld : dfft_comm {
    struct Slot {
        const char *send = nullptr;
        const size_t *sdispl = nullptr;
        const int *group = nullptr;
        int ngroup = 0;
        hipEvent_t ready = nullptr, done = nullptr;
    };
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long generation = 0;

    explicit LocalWorld(int n) : slots(n) { nranks = n; }
    ~LocalWorld() override
    {
        for (auto &s : slots) {
            if (s.ready) (void)hipEventDestroy(s.ready);
            if (s.done) (void)hipEventDestroy(s.done);
        }
    }
    void barrier(int) override
    {
        std::unique_lock<std::mutex> lk(mu);
        unsigned long gen = generation;
        if (++waiting == nranks) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
    int alltoallv(int myrank, const void *send, const size_t *scount, const size_t *sdispl, void *recv,
                  const size_t *rcount, const size_t *rdispl, const int *group, int ngroup, int me,
                  hipStream_t stream, int /*channel*/) override
    {
        (void)scount;
        // A failing rank must still meet the others at both barriers (they would wait forever otherwise):
        // remember the first error, skip the device work after it, return it at the end.
        int err = 0;
        auto note = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && !err) { err = (int)e; set_error(std::string(what) + ": " + hipGetErrorString(e)); }
        };
        Slot &mine = slots[myrank];
        if (!mine.ready) {
            note(hipEventCreateWithFlags(&mine.ready, hipEventDisableTiming), "hipEventCreate");
            if (!err) note(hipEventCreateWithFlags(&mine.done, hipEventDisableTiming), "hipEventCreate");
        }
        mine.send = static_cast<const char *>(send);
        mine.sdispl = sdispl;
        mine.group = group;
        mine.ngroup = ngroup;
        if (!err) note(hipEventRecord(mine.ready, stream), "hipEventRecord");
        barrier(myrank);   // everyone has published
        for (int q = 0; q < ngroup && !err; q++) {
            const Slot &peer = slots[group[q]];
            if (group[q] != myrank && peer.ready) note(hipStreamWaitEvent(stream, peer.ready, 0), "hipStreamWaitEvent");
            if (rcount[q] && !err)
                note(hipMemcpyAsync(static_cast<char *>(recv) + rdispl[q], peer.send + peer.sdispl[me], rcount[q],
                                    hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
        }
        if (!err) note(hipEventRecord(mine.done, stream), "hipEventRecord");
        barrier(myrank);   // everyone has enqueued its pulls
        // my send buffer may be overwritten by my next kernel only after all peers pulled it
        for (int q = 0; q < ngroup && !err; q++)
            if (group[q] != myrank && slots[group[q]].done) note(hipStreamWaitEvent(stream, slots[group[q]].done, 0), "hipStreamWaitEvent");
        return err;
    }
};

// ------------------------------------------------------------------------------------------
// RCCL over xGMI: one process per GPU.  librccl is dlopen'ed so that the library loads (and
// every CPU-side test runs) on hosts without it.  One grouped ncclSend/ncclRecv batch per
// exchange on the plan's stream = the reference's All2All-Sync mode with cuda_aware = true.
// ------------------------------------------------------------------------------------------
struct Id128 { char b[128]; };
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /*ncclUniqueId by value: 128 bytes*/ Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommSplit)(void *, int, int, void **, void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.h) break;
        }
        if (!api.h) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))dlsym(api.h, "ncclCommCount");
        api.CommSplit = (decltype(api.CommSplit))dlsym(api.h, "ncclCommSplit");
        api.Send = (decltype(api.Send))dlsym(api.h, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(api.h, "ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.h, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.h, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
    });
    if (!api.h || !api.GetUniqueId || !api.CommInitRank || !api.Send || !api.Recv || !api.GroupStart ||
        !api.GroupEnd)
        return nullptr;
    return &api;
}

#define NCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        int r_ = (expr);                                                                       \
        if (r_ != 0) {                                                                         \
            set_error(std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(r_) : "rccl error")); \
            return 1000 + r_;                                                                  \
        }                                                                                      \
    } while (0)

struct RcclComm : dfft_comm {
    void *comm = nullptr;
    void *comm2 = nullptr;     // duplicate communicator (ncclCommSplit, colour 0) for channel 1
    int rank = 0;
    bool self_send = false;    // testing: the self block goes through ncclSend / ncclRecv to the own rank instead of a device copy
                               // (lets a 1-GPU box hand caller buffers -- virtual-memory ranges included -- to RCCL)
    int fixed_rank() const override { return rank; }
    // one ncclComm serialises its operations: only a duplicated communicator allows overlap
    bool concurrent_channels() const override { return comm2 != nullptr; }
    ~RcclComm() override
    {
        RcclApi *R = rccl();
        if (R && comm2 && R->CommDestroy) R->CommDestroy(comm2);
        if (R && comm && R->CommDestroy) R->CommDestroy(comm);
    }
    int transport_nranks() const override
    {
        RcclApi *R = rccl();
        int n = 0;
        if (!R || !R->CommCount || !comm || R->CommCount(comm, &n) != 0) return 0;
        return n;
    }
    // "dup_channel" = 1: second communicator over the same ranks (ncclCommSplit, colour 0) for channel 1.  Collective: every
    // rank calls it at the same point.  Without it both exchanges of a pencil plan share one communicator and RCCL
    // serialises them.
    int set_option(const char *key, long value) override
    {
        if (std::string(key ? key : "") == "self_send") { self_send = value != 0; return 0; }
        if (std::string(key ? key : "") != "dup_channel") return 1;
        RcclApi *R = rccl();
        if (!R || !comm) { set_error("librccl not available"); return 1; }
        if (value == 0) {
            if (comm2 && R->CommDestroy) R->CommDestroy(comm2);
            comm2 = nullptr;
            return 0;
        }
        if (comm2) return 0;
        if (!R->CommSplit) { set_error("this librccl has no ncclCommSplit"); return 1; }
        NCCL_TRY(R->CommSplit(comm, 0, rank, &comm2, nullptr));
        return 0;
    }
    int alltoallv(int myrank, const void *send, const size_t *scount, const size_t *sdispl, void *recv,
                  const size_t *rcount, const size_t *rdispl, const int *group, int ngroup, int me,
                  hipStream_t stream, int channel) override
    {
        RcclApi *R = rccl();
        if (!R) { set_error("librccl not available"); return 1; }
        const char *s = static_cast<const char *>(send);
        char *r = static_cast<char *>(recv);
        void *use = (channel == 1 && comm2) ? comm2 : comm;
        // self block: plain device copy, never goes through RCCL
        // (the reference skips self in its send tables, src/pencil/mpicufft_pencil.cpp:282-289)
        if (rcount[me] && !self_send)
            HIP_TRY(hipMemcpyAsync(r + rdispl[me], s + sdispl[me], rcount[me], hipMemcpyDeviceToDevice, stream));
        NCCL_TRY(R->GroupStart());
        if (rcount[me] && self_send) {
            int e1 = R->Send(s + sdispl[me], scount[me], 0, group[me], use, stream);
            int e2 = e1 ? 0 : R->Recv(r + rdispl[me], rcount[me], 0, group[me], use, stream);
            if (e1 || e2) { (void)R->GroupEnd(); set_error("ncclSend/ncclRecv to self failed"); return 1000 + (e1 ? e1 : e2); }
        }
        // a failing call inside the group must not leave the group open: remember the first error, stop
        // queueing, always close the group
        int err = 0;
        std::string what;
        for (int i = 1; i < ngroup && !err; i++) {
            // ring order (me+i)%P like the reference's comm_order (mpicufft_pencil_opt1.cpp:107-113)
            const int to = (me + i) % ngroup, from = (me - i + ngroup) % ngroup;
            if (scount[to] && (err = R->Send(s + sdispl[to], scount[to], /*ncclInt8*/ 0, group[to], use, stream)) != 0) { what = "ncclSend"; break; }
            if (rcount[from] && (err = R->Recv(r + rdispl[from], rcount[from], 0, group[from], use, stream)) != 0) { what = "ncclRecv"; break; }
        }
        const int end = R->GroupEnd();
        if (!err && end) { err = end; what = "ncclGroupEnd"; }
        if (err) {
            set_error(what + ": " + (R->GetErrorString ? R->GetErrorString(err) : "rccl error"));
            return 1000 + err;
        }
        (void)myrank;
        return 0;
    }
};

int rccl_unique_id(void *id128)
{
    RcclApi *R = rccl();
    if (!R) { set_error("librccl not available"); return 1; }
    NCCL_TRY(R->GetUniqueId(id128));
    return 0;
}

dfft_comm *make_rccl_comm(const void *id128, int nranks, int rank)
{
    RcclApi *R = rccl();
    if (!R) { set_error("librccl not available"); return nullptr; }
    RcclComm *c = new RcclComm;
    c->nranks = nranks;
    c->rank = rank;
    Id128 id;
    memcpy(id.b, id128, 128);
    int r = R->CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
        set_error(std::string("ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(r) : "error"));
        c->comm = nullptr;
        delete c;
        return nullptr;
    }
    // (a second communicator for the second exchange of pencil plans is an explicit, collective option: set_option)
    return c;
}

// ------------------------------------------------------------------------------------------
// Callback transport: the caller owns the exchange (torch.distributed.all_to_all_single over
// RCCL in bench.py, gloo in the CPU tests, MPI_Alltoallv in an MPI host).
// ------------------------------------------------------------------------------------------
struct CallbackComm : dfft_comm {
    dfft_alltoallv_fn fn = nullptr;
    void *user = nullptr;
    int rank = 0;
    int fixed_rank() const override { return rank; }
    int alltoallv(int myrank, const void *send, const size_t *scount, const size_t *sdispl, void *recv,
                  const size_t *rcount, const size_t *rdispl, const int *group, int ngroup, int me,
                  hipStream_t stream, int /*channel*/) override
    {
        (void)myrank;
        int r = fn(user, send, scount, sdispl, recv, rcount, rdispl, group, ngroup, me, (void *)stream);
        if (r != 0) set_error("all-to-all callback failed with code " + std::to_string(r));
        return r;
    }
};

// ------------------------------------------------------------------------------------------
// Two-hop relay (see comm.hpp).  Per partner round k = 1 .. ngroup - 1 (rank `me` of a group sends to member me + k and
// receives from member me - k):
//   phase A  world all-to-all: part 0 of my message straight to the partner, part 2 + h to helper h (every rank outside the
//            pair, ascending); I receive part 0 of my source's message in place and, as a helper of every other pair, one
//            part each into the staging buffer
//   phase B  world all-to-all: part 1 straight to the partner, every staged part on to its destination; I receive parts
//            1 .. nranks - 1 of my source's message in place
// ------------------------------------------------------------------------------------------
void relay_part(size_t S, int nranks, int p, size_t *off, size_t *len)
{
    const size_t parts = (size_t)(nranks < 2 ? 2 : nranks);      // K + 2 with K = nranks - 2 helpers
    size_t q = (S + parts - 1) / parts;
    q = (q + 255) & ~(size_t)255;                                 // parts start on 256-byte boundaries
    const size_t a = std::min(S, (size_t)p * q), b = std::min(S, (size_t)(p + 1) * q);
    *off = a;
    *len = b - a;
}

struct RelayMeta {
    int R = 0;                        // rounds = group size - 1
    std::vector<int> partner;         // [y * R + k - 1]: the rank y sends to in round k
    std::vector<size_t> bytes;        // ... and the size of that message
};
struct RelayBuf { char *p = nullptr; size_t cap = 0; bool device = false; };
struct RelayCache {
    std::map<uint64_t, RelayMeta> meta;
    std::map<std::pair<void *, int>, RelayBuf> staging;      // per (stream, channel): calls on one stream are ordered
    RelayBuf msend, mrecv;
};
RelayCache *relay_cache_new() { return new RelayCache; }
static void relay_buf_free(RelayBuf &b)
{
    if (!b.p) return;
    if (b.device) (void)hipFree(b.p); else free(b.p);
    b = RelayBuf();
}
void relay_cache_free(RelayCache *c)
{
    if (!c) return;
    for (auto &kv : c->staging) relay_buf_free(kv.second);
    relay_buf_free(c->msend);
    relay_buf_free(c->mrecv);
    delete c;
}
// is this a device pointer?  (the CPU tests drive the exchange with host tensors through the callback transport)
static bool relay_on_device(const void *ptr)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeUnified || at.type == hipMemoryTypeManaged;
}
static int relay_reserve(RelayBuf &b, size_t bytes, bool device, hipStream_t stream)
{
    if (b.p && b.cap >= bytes && b.device == device) return 0;
    if (b.p && b.device) HIP_TRY(hipStreamSynchronize(stream));      // earlier exchanges on this stream may still read it
    relay_buf_free(b);
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (device) HIP_TRY(hipMalloc((void **)&b.p, bytes));
    else if (!(b.p = (char *)malloc(bytes))) { set_error("relay: out of host memory"); return 1; }
    b.cap = bytes;
    b.device = device;
    return 0;
}

static int relay_gather_meta(dfft_comm *comm, RelayCache *cache, RelayMeta &M, int myrank, const size_t *scount, const int *group,
                             int ngroup, int me, bool device, hipStream_t stream, int channel)
{
    const int n = comm->nranks, R = ngroup - 1;
    const size_t rec = (size_t)R * 2 * sizeof(uint64_t);
    std::vector<uint64_t> mine((size_t)R * 2), all((size_t)n * R * 2);
    for (int k = 1; k <= R; k++) {
        const int ti = (me + k) % ngroup;
        mine[2 * (k - 1)] = (uint64_t)group[ti];
        mine[2 * (k - 1) + 1] = (uint64_t)scount[ti];
    }
    if (int r = relay_reserve(cache->msend, rec * n, device, stream)) return r;
    if (int r = relay_reserve(cache->mrecv, rec * n, device, stream)) return r;
    std::vector<uint64_t> rep((size_t)n * R * 2);
    for (int y = 0; y < n; y++) memcpy(&rep[(size_t)y * R * 2], mine.data(), rec);
    std::vector<size_t> cnt(n, rec), dsp(n);
    std::vector<int> world(n);
    for (int y = 0; y < n; y++) { dsp[y] = (size_t)y * rec; world[y] = y; }
    if (device) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(cache->msend.p, rep.data(), rec * n, hipMemcpyHostToDevice));
    } else memcpy(cache->msend.p, rep.data(), rec * n);
    if (int r = comm->alltoallv(myrank, cache->msend.p, cnt.data(), dsp.data(), cache->mrecv.p, cnt.data(), dsp.data(), world.data(), n,
                                myrank, stream, channel)) return r;
    if (device) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(all.data(), cache->mrecv.p, rec * n, hipMemcpyDeviceToHost));
    } else memcpy(all.data(), cache->mrecv.p, rec * n);
    M.R = R;
    M.partner.resize((size_t)n * R);
    M.bytes.resize((size_t)n * R);
    for (int y = 0; y < n; y++)
        for (int k = 0; k < R; k++) {
            const uint64_t t = all[((size_t)y * R + k) * 2];
            if (t >= (uint64_t)n || (int)t == y) { set_error("relay: a rank reported an invalid partner (do all ranks run the same exchange?)"); return 1; }
            M.partner[(size_t)y * R + k] = (int)t;
            M.bytes[(size_t)y * R + k] = (size_t)all[((size_t)y * R + k) * 2 + 1];
        }
    // every rank must be the partner of exactly one rank per round (the groups partition the world)
    for (int k = 0; k < R; k++) {
        std::vector<int> seen(n, 0);
        for (int y = 0; y < n; y++) seen[M.partner[(size_t)y * R + k]]++;
        for (int y = 0; y < n; y++)
            if (seen[y] != 1) { set_error("relay: the groups of this exchange do not partition the world"); return 1; }
    }
    return 0;
}

int relay_alltoallv(dfft_comm *comm, RelayCache *cache, uint64_t tag, int myrank, const void *send, const size_t *scount,
                    const size_t *sdispl, void *recv, const size_t *rcount, const size_t *rdispl, const int *group, int ngroup,
                    int me, hipStream_t stream, int channel)
{
    const int n = comm->nranks, R = ngroup - 1;
    const bool device = relay_on_device(send);
    const char *sb = static_cast<const char *>(send);
    char *rb = static_cast<char *>(recv);
    auto it = cache->meta.find(tag);
    if (it == cache->meta.end()) {
        RelayMeta M;
        if (int r = relay_gather_meta(comm, cache, M, myrank, scount, group, ngroup, me, device, stream, channel)) return r;
        it = cache->meta.emplace(tag, std::move(M)).first;
    }
    const RelayMeta &M = it->second;
    if (M.R != R) { set_error("relay: exchange table changed under its tag"); return 1; }
    // self block: a local copy, as in every transport
    if (rcount[me]) {
        if (device) HIP_TRY(hipMemcpyAsync(rb + rdispl[me], sb + sdispl[me], rcount[me], hipMemcpyDeviceToDevice, stream));
        else memcpy(rb + rdispl[me], sb + sdispl[me], rcount[me]);
    }
    std::vector<int> world(n), src(n);
    for (int y = 0; y < n; y++) world[y] = y;
    std::vector<size_t> sc(n), sd(n), rc(n), rd(n), stoff(n), stlen(n);
    auto helper_index = [](int h, int a, int b) { return h - (a < h) - (b < h); };      // position of h among the ranks outside {a, b}
    for (int k = 1; k <= R; k++) {
        const int ti = (me + k) % ngroup, si = (me - k + ngroup) % ngroup;
        const int t = group[ti], s = group[si];
        const size_t St = scount[ti], Ss = rcount[si];
        if (M.partner[(size_t)myrank * R + k - 1] != t || M.partner[(size_t)s * R + k - 1] != myrank || M.bytes[(size_t)s * R + k - 1] != Ss) {
            set_error("relay: the gathered tables do not match this call");
            return 1;
        }
        for (int y = 0; y < n; y++) src[M.partner[(size_t)y * R + k - 1]] = y;
        // staging: one part from every rank outside {me, s}
        size_t need = 0;
        for (int y = 0; y < n; y++) {
            stoff[y] = stlen[y] = 0;
            if (y == myrank || y == s) continue;
            size_t off, len;
            relay_part(M.bytes[(size_t)y * R + k - 1], n, 2 + helper_index(myrank, y, M.partner[(size_t)y * R + k - 1]), &off, &len);
            stoff[y] = need;
            stlen[y] = len;
            need += (len + 255) & ~(size_t)255;
        }
        RelayBuf &st = cache->staging[std::make_pair((void *)stream, channel)];
        if (int r = relay_reserve(st, need ? need : 256, device, stream)) return r;
        // ---- phase A ----
        const char *sbaseA = sb;
        char *rbaseA = rb < st.p ? rb : st.p;
        for (int x = 0; x < n; x++) {
            sc[x] = sd[x] = rc[x] = rd[x] = 0;
            if (x == myrank) continue;
            size_t off, len;
            relay_part(St, n, x == t ? 0 : 2 + helper_index(x, myrank, t), &off, &len);
            sc[x] = len;
            sd[x] = sdispl[ti] + off;
            if (x == s) {
                relay_part(Ss, n, 0, &off, &len);
                rc[x] = len;
                rd[x] = (size_t)((rb + rdispl[si] + off) - rbaseA);
            } else {
                rc[x] = stlen[x];
                rd[x] = (size_t)((st.p + stoff[x]) - rbaseA);
            }
        }
        if (int r = comm->alltoallv(myrank, sbaseA, sc.data(), sd.data(), rbaseA, rc.data(), rd.data(), world.data(), n, myrank, stream, channel)) return r;
        // ---- phase B ----
        const char *sbaseB = sb < st.p ? sb : st.p;
        for (int x = 0; x < n; x++) {
            sc[x] = sd[x] = rc[x] = rd[x] = 0;
            if (x == myrank) continue;
            size_t off, len;
            if (x == t) {
                relay_part(St, n, 1, &off, &len);
                sc[x] = len;
                sd[x] = (size_t)((sb + sdispl[ti] + off) - sbaseB);
            } else {
                const int y = src[x];      // the rank whose message to x I hold a part of (y != me: only t has me as its source)
                sc[x] = stlen[y];
                sd[x] = (size_t)((st.p + stoff[y]) - sbaseB);
            }
            relay_part(Ss, n, x == s ? 1 : 2 + helper_index(x, s, myrank), &off, &len);
            rc[x] = len;
            rd[x] = rdispl[si] + off;
        }
        if (int r = comm->alltoallv(myrank, sbaseB, sc.data(), sd.data(), rb, rc.data(), rd.data(), world.data(), n, myrank, stream, channel)) return r;
    }
    return 0;
}

dfft_comm *make_local_world(int nranks) { return new LocalWorld(nranks); }
dfft_comm *make_callback_comm(int nranks, int rank, void *fn, void *user)
{
    CallbackComm *c = new CallbackComm;
    c->nranks = nranks;
    c->rank = rank;
    c->fn = (dfft_alltoallv_fn)fn;
    c->user = user;
    return c;
}

}  // namespace dfft
