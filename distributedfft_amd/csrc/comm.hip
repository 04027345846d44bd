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
        const dfft_xfer *sends = nullptr;      // sendrecv_list: the published schedule
        int ns = 0;
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
    // the schedule in one go, "pull" like alltoallv: publish my send pieces, meet, copy every piece addressed to me
    int sendrecv_list(int myrank, const dfft_xfer *sends, int ns, const dfft_xfer *recvs, int nr, int /*nlayers*/, hipStream_t stream,
                      int /*channel*/) override
    {
        int err = 0;
        auto note = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && !err) { err = (int)e; set_error(std::string(what) + ": " + hipGetErrorString(e)); }
        };
        counters.list++;
        Slot &mine = slots[myrank];
        if (!mine.ready) {
            note(hipEventCreateWithFlags(&mine.ready, hipEventDisableTiming), "hipEventCreate");
            if (!err) note(hipEventCreateWithFlags(&mine.done, hipEventDisableTiming), "hipEventCreate");
        }
        mine.sends = sends;
        mine.ns = ns;
        if (!err) note(hipEventRecord(mine.ready, stream), "hipEventRecord");
        barrier(myrank);   // everyone has published
        std::vector<char> waited(nranks, 0);
        for (int i = 0; i < nr && !err; i++) {
            const dfft_xfer &r = recvs[i];
            if (r.peer < 0 || r.peer >= nranks || r.peer == myrank) { if (!err) { err = 1; set_error("sendrecv_list: bad peer"); } break; }
            const Slot &peer = slots[r.peer];
            const dfft_xfer *m = nullptr;
            for (int j = 0; j < peer.ns; j++)
                if (peer.sends[j].peer == myrank && peer.sends[j].layer == r.layer) { m = &peer.sends[j]; break; }
            if (!m || m->bytes != r.bytes) { err = 1; set_error("sendrecv_list: a receive piece has no matching send piece of its size"); break; }
            if (!waited[r.peer] && peer.ready) { note(hipStreamWaitEvent(stream, peer.ready, 0), "hipStreamWaitEvent"); waited[r.peer] = 1; }
            if (!err) note(hipMemcpyAsync(r.ptr, m->ptr, r.bytes, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
        }
        if (!err) note(hipEventRecord(mine.done, stream), "hipEventRecord");
        barrier(myrank);   // everyone has enqueued its pulls
        for (int y = 0; y < nranks && !err; y++)
            if (y != myrank && slots[y].done) note(hipStreamWaitEvent(stream, slots[y].done, 0), "hipStreamWaitEvent");
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
    // duplicate communicators (ncclCommSplit, colour 0): extra[0] for channel 1 (the second exchange of a pencil plan), extra[1] /
    // extra[2] for channels 2 / 3 (the relay's first hop of exchange 1 / 2 while the second hop of the previous chunk is on the wire)
    void *extra[3] = {nullptr, nullptr, nullptr};
    int rank = 0;
    bool self_send = false;    // testing: the self block goes through ncclSend / ncclRecv to the own rank instead of a device copy
                               // (lets a 1-GPU box hand caller buffers -- virtual-memory ranges included -- to RCCL)
    int fixed_rank() const override { return rank; }
    // one ncclComm serialises its operations: only a duplicated communicator allows overlap
    bool concurrent_channels() const override { return extra[0] != nullptr; }
    bool concurrent_hops(int channel) const override { return channel >= 0 && channel < 2 && extra[channel + 1] != nullptr; }
    void *comm_of(int channel) const
    {
        if (channel >= 1 && channel <= 3 && extra[channel - 1]) return extra[channel - 1];
        if (channel == 3 && extra[0]) return extra[0];      // no communicator of its own: at least not the one of the other exchange
        return comm;
    }
    ~RcclComm() override
    {
        RcclApi *R = rccl();
        for (void *c : extra)
            if (R && c && R->CommDestroy) R->CommDestroy(c);
        if (R && comm && R->CommDestroy) R->CommDestroy(comm);
    }
    int transport_nranks() const override
    {
        RcclApi *R = rccl();
        int n = 0;
        if (!R || !R->CommCount || !comm || R->CommCount(comm, &n) != 0) return 0;
        return n;
    }
    // "dup_channel" = 1: second communicator over the same ranks (ncclCommSplit, colour 0) for channel 1; = 3: three of them, for
    // channels 1, 2, 3 (2 / 3 carry the relay's first hop).  Collective: every rank calls it at the same point.  Without it both
    // exchanges of a pencil plan share one communicator and RCCL serialises them.
    int set_option(const char *key, long value) override
    {
        if (std::string(key ? key : "") == "self_send") { self_send = value != 0; return 0; }
        if (std::string(key ? key : "") != "dup_channel") return 1;
        RcclApi *R = rccl();
        if (!R || !comm) { set_error("librccl not available"); return 1; }
        if (value < 0 || value > 3) { set_error("dup_channel: 0 .. 3 extra communicators"); return 2; }
        for (int i = 2; i >= (int)value; i--) {
            if (extra[i] && R->CommDestroy) R->CommDestroy(extra[i]);
            extra[i] = nullptr;
        }
        for (int i = 0; i < (int)value; i++) {
            if (extra[i]) continue;
            if (!R->CommSplit) { set_error("this librccl has no ncclCommSplit"); return 1; }
            NCCL_TRY(R->CommSplit(comm, 0, rank, &extra[i], nullptr));
        }
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
        void *use = comm_of(channel);
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
    // the schedule as ONE group of ncclSend / ncclRecv: RCCL matches the pieces between two ranks in the order they were issued, and
    // both ends list them by layer
    int sendrecv_list(int /*myrank*/, const dfft_xfer *sends, int ns, const dfft_xfer *recvs, int nr, int nlayers, hipStream_t stream,
                      int channel) override
    {
        RcclApi *R = rccl();
        if (!R) { set_error("librccl not available"); return 1; }
        void *use = comm_of(channel);
        counters.list++;
        NCCL_TRY(R->GroupStart());
        int err = 0;
        std::string what;
        for (int layer = 0; layer < nlayers && !err; layer++) {
            for (int i = 0; i < ns && !err; i++)
                if (sends[i].layer == layer && sends[i].bytes && (err = R->Send(sends[i].ptr, sends[i].bytes, /*ncclInt8*/ 0, sends[i].peer, use, stream)) != 0) what = "ncclSend";
            for (int i = 0; i < nr && !err; i++)
                if (recvs[i].layer == layer && recvs[i].bytes && (err = R->Recv(recvs[i].ptr, recvs[i].bytes, 0, recvs[i].peer, use, stream)) != 0) what = "ncclRecv";
        }
        const int end = R->GroupEnd();      // a failing call inside the group must not leave it open
        if (!err && end) { err = end; what = "ncclGroupEnd"; }
        if (err) {
            set_error(what + ": " + (R->GetErrorString ? R->GetErrorString(err) : "rccl error"));
            return 1000 + err;
        }
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
    dfft_sendrecv_list_fn list_fn = nullptr;
    void *list_user = nullptr;
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
    int sendrecv_list(int myrank, const dfft_xfer *sends, int ns, const dfft_xfer *recvs, int nr, int nlayers, hipStream_t stream, int channel) override
    {
        if (!list_fn) return dfft_comm::sendrecv_list(myrank, sends, ns, recvs, nr, nlayers, stream, channel);
        // by layer, so that both ends of a link see its pieces in the same order
        std::vector<int> sp, rp;
        std::vector<void *> sptr, rptr;
        std::vector<size_t> sby, rby;
        for (int layer = 0; layer < nlayers; layer++) {
            for (int i = 0; i < ns; i++)
                if (sends[i].layer == layer && sends[i].bytes) { sp.push_back(sends[i].peer); sptr.push_back(sends[i].ptr); sby.push_back(sends[i].bytes); }
            for (int i = 0; i < nr; i++)
                if (recvs[i].layer == layer && recvs[i].bytes) { rp.push_back(recvs[i].peer); rptr.push_back(recvs[i].ptr); rby.push_back(recvs[i].bytes); }
        }
        counters.list++;
        int r = list_fn(list_user, (int)sp.size(), sp.data(), sptr.data(), sby.data(), (int)rp.size(), rp.data(), rptr.data(), rby.data(), (void *)stream);
        if (r != 0) set_error("send/receive schedule callback failed with code " + std::to_string(r));
        return r;
    }
};
int callback_comm_set_list(dfft_comm *comm, void *fn, void *user)
{
    CallbackComm *c = dynamic_cast<CallbackComm *>(comm);
    if (!c) { set_error("not a callback communicator"); return 1; }
    c->list_fn = (dfft_sendrecv_list_fn)fn;
    c->list_user = user;
    return 0;
}

// ------------------------------------------------------------------------------------------
// Two-hop relay (see comm.hpp), one-shot form.  Round k = 1 .. ngroup - 1 names the partners: member `me` of a group sends to member
// me + k and receives from member me - k; a message of S bytes is cut into nranks parts (relay_part): part 0 and part 1 travel
// directly (hop 1 / hop 2), part 2 + h through helper h, the h-th rank outside the pair.
//   hop 1  to every rank x, for every round k: the part of my round-k message that x carries (part 0 if x is that round's partner);
//          from every rank x, for every round k: part 0 of x's round-k message if it is for me (in place), else the part I carry for
//          that pair (into the staging buffer)
//   hop 2  to every rank x, for every round k: part 1 of my message if x is my round-k partner, else the part I hold of the round-k
//          message for x; from every rank x, for every round k: the part of my round-k source's message that x carried (in place)
// Both ends of a link list its pieces by round, which is the `layer` the transports match them by.
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
    // the schedule of this rank, built once from the gathered tables: where = 0 send buffer, 1 receive buffer, 2 staging
    struct Piece { int peer, layer, where; size_t off, bytes; };
    std::vector<Piece> sendA, recvA, sendB, recvB;
    size_t staging = 0;
    bool built = false;
    // lanes (stream, channel) on which this table's local resources -- staging for both parities, side stream, events -- exist and ALL
    // ranks said so (relay_agree); direct = some rank could not set up: every rank sends this table directly, for good
    std::map<std::pair<void *, int>, bool> agreed;
    bool direct = false;
};
struct RelayBuf { char *p = nullptr; size_t cap = 0; bool device = false; };
// per (stream, channel) of the exchanges that use the relay: staging (two buffers when the hops of neighbouring chunks overlap), the
// side stream of hop 1 and the events that order the two streams
struct RelayLane {
    RelayBuf staging[2];
    hipStream_t side = nullptr;
    hipEvent_t in = nullptr, hop1 = nullptr, hop2[2] = {nullptr, nullptr};
    bool hop2_used[2] = {false, false};
    int parity = 0;
};
struct RelayCache {
    std::map<uint64_t, RelayMeta> meta;
    std::map<std::pair<void *, int>, RelayLane> lanes;
    RelayBuf msend, mrecv;
};
RelayCache *relay_cache_new() { return new RelayCache; }
static void relay_buf_free(RelayBuf &b)
{
    if (!b.p) return;
    if (b.device) (void)dfft_free(b.p); else free(b.p);
    b = RelayBuf();
}
void relay_cache_free(RelayCache *c)
{
    if (!c) return;
    for (auto &kv : c->lanes) {
        RelayLane &L = kv.second;
        if (L.side) { (void)hipStreamSynchronize(L.side); (void)hipStreamDestroy(L.side); }
        for (hipEvent_t e : {L.in, L.hop1, L.hop2[0], L.hop2[1]})
            if (e) (void)hipEventDestroy(e);
        relay_buf_free(L.staging[0]);
        relay_buf_free(L.staging[1]);
    }
    relay_buf_free(c->msend);
    relay_buf_free(c->mrecv);
    delete c;
}
// Device or host buffer?  (The CPU tests drive the exchange with host tensors through the callback transport.)  0 host, 1 device,
// -1 the query failed for another reason than "this is plain host memory" -- the caller returns an error instead of guessing.
static int relay_on_device(const void *ptr)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return 0; }      // no GPU: everything is host memory
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (e == hipErrorInvalidValue) { (void)hipGetLastError(); return 0; }      // a pointer the runtime does not know: malloc'ed host memory
    if (e != hipSuccess) { set_error(std::string("relay: hipPointerGetAttributes: ") + hipGetErrorString(e)); (void)hipGetLastError(); return -1; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeUnified || at.type == hipMemoryTypeManaged ? 1 : 0;
}
static int relay_reserve(RelayBuf &b, size_t bytes, bool device, hipStream_t stream)
{
    if (b.p && b.cap >= bytes && b.device == device) return 0;
    if (b.p && b.device) HIP_TRY(hipDeviceSynchronize());      // earlier exchanges (either hop stream) may still read it
    (void)stream;
    relay_buf_free(b);
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (device) {
        // plain hipMalloc, on purpose.  Staging is written and read front to back by the transport's copies, not scattered into, so the
        // placement of the default backing buys nothing here -- and a buffer that is reallocated while other host threads keep
        // enqueueing work must not be a virtual-memory range: with staging from hipMemCreate / hipMemMap, unmapped and re-created when
        // a larger exchange table came along, the round trips of 8 virtual ranks differed from the direct exchange in 9-12 of 12 runs
        // (never with hipMalloc, never when the ranges were leaked instead of unmapped, poisoned staging made no difference: the
        // relay's ordering is not the cause; tools/exp/r5_relay_stress.py, profiles/r5_relay_stress.txt)
        void *ptr = nullptr;
        static const bool vmm = [] { const char *e = getenv("DFFT_RELAY_STAGING"); return e && std::string(e) == "vmm"; }();      // experiments only
        if (int r = dfft_malloc(bytes, vmm ? 1024 : 0, &ptr)) return r;
        b.p = (char *)ptr;
    } else if (!(b.p = (char *)malloc(bytes))) { set_error("relay: out of host memory"); return 1; }
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
    // once per exchange table and plan: the two copies below are blocking on purpose (the tables are needed on the host)
    if (device) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(cache->msend.p, rep.data(), rec * n, hipMemcpyHostToDevice));
    } else memcpy(cache->msend.p, rep.data(), rec * n);
    comm->counters.relay_meta++;
    comm->counters.alltoallv++;
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

// One word per rank, all to all, blocking (once per exchange table and lane): the largest status any rank reports.  Everything that can
// fail LOCALLY in a relayed exchange (staging, streams, events) is done before this call, so that a rank that could not set up does not
// walk away from a collective the others are already inside (round-5 advice) -- all ranks learn it and send the table directly instead.
static int relay_agree(dfft_comm *comm, RelayCache *cache, int myrank, bool device, hipStream_t stream, int channel, uint64_t mine, uint64_t *worst)
{
    const int n = comm->nranks;
    const size_t rec = sizeof(uint64_t);
    if (cache->msend.cap < rec * n || cache->mrecv.cap < rec * n) { dfft::set_error("relay: table buffers are missing"); return 1; }      // (reserved by the gather)
    std::vector<uint64_t> rep(n, mine), all(n, 0);
    std::vector<size_t> cnt(n, rec), dsp(n);
    std::vector<int> world(n);
    for (int y = 0; y < n; y++) { dsp[y] = (size_t)y * rec; world[y] = y; }
    if (device) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(cache->msend.p, rep.data(), rec * n, hipMemcpyHostToDevice));
    } else memcpy(cache->msend.p, rep.data(), rec * n);
    comm->counters.alltoallv++;
    comm->counters.relay_agree++;
    if (int r = comm->alltoallv(myrank, cache->msend.p, cnt.data(), dsp.data(), cache->mrecv.p, cnt.data(), dsp.data(), world.data(), n,
                                myrank, stream, channel)) return r;
    if (device) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(all.data(), cache->mrecv.p, rec * n, hipMemcpyDeviceToHost));
    } else memcpy(all.data(), cache->mrecv.p, rec * n);
    *worst = 0;
    for (uint64_t v : all) if (v > *worst) *worst = v;
    return 0;
}

// this rank's pieces of both hops, from the gathered tables and its own exchange table (offsets relative to the send / receive
// buffer of the call and to the staging buffer)
static int relay_build(RelayMeta &M, int n, int myrank, const size_t *scount, const size_t *sdispl, const size_t *rcount, const size_t *rdispl,
                       const int *group, int ngroup, int me)
{
    const int R = M.R;
    auto helper_index = [](int h, int a, int b) { return h - (a < h) - (b < h); };      // position of h among the ranks outside {a, b}
    auto partner = [&](int y, int k) { return M.partner[(size_t)y * R + k - 1]; };
    auto bytes = [&](int y, int k) { return M.bytes[(size_t)y * R + k - 1]; };
    std::vector<int> src(n);
    // staged parts: [k][y] for every pair (y, partner(y, k)) I am outside of
    std::vector<size_t> stoff((size_t)R * n, 0), stlen((size_t)R * n, 0);
    size_t need = 0;
    M.sendA.clear(); M.recvA.clear(); M.sendB.clear(); M.recvB.clear();
    for (int k = 1; k <= R; k++) {
        const int ti = (me + k) % ngroup, si = (me - k + ngroup) % ngroup;
        const int t = group[ti], s = group[si];
        const size_t St = scount[ti], Ss = rcount[si];
        if (partner(myrank, k) != t || partner(s, k) != myrank || bytes(s, k) != Ss || bytes(myrank, k) != St) {
            set_error("relay: the gathered tables do not match this call");
            return 1;
        }
        for (int y = 0; y < n; y++) src[partner(y, k)] = y;
        for (int y = 0; y < n; y++) {
            if (y == myrank || y == s) continue;      // (y == s: its partner is me)
            size_t off, len;
            relay_part(bytes(y, k), n, 2 + helper_index(myrank, y, partner(y, k)), &off, &len);
            stoff[(size_t)(k - 1) * n + y] = need;
            stlen[(size_t)(k - 1) * n + y] = len;
            need += (len + 255) & ~(size_t)255;
        }
        for (int x = 0; x < n; x++) {
            if (x == myrank) continue;
            size_t off, len;
            // ---- hop 1 ----
            relay_part(St, n, x == t ? 0 : 2 + helper_index(x, myrank, t), &off, &len);
            if (len) M.sendA.push_back({x, k - 1, 0, sdispl[ti] + off, len});
            if (x == s) {
                relay_part(Ss, n, 0, &off, &len);
                if (len) M.recvA.push_back({x, k - 1, 1, rdispl[si] + off, len});
            } else if (stlen[(size_t)(k - 1) * n + x]) {
                M.recvA.push_back({x, k - 1, 2, stoff[(size_t)(k - 1) * n + x], stlen[(size_t)(k - 1) * n + x]});
            }
            // ---- hop 2 ----
            if (x == t) {
                relay_part(St, n, 1, &off, &len);
                if (len) M.sendB.push_back({x, k - 1, 0, sdispl[ti] + off, len});
            } else {
                const int y = src[x];      // the rank whose round-k message goes to x (y != me: only t has me as its source)
                if (stlen[(size_t)(k - 1) * n + y]) M.sendB.push_back({x, k - 1, 2, stoff[(size_t)(k - 1) * n + y], stlen[(size_t)(k - 1) * n + y]});
            }
            relay_part(Ss, n, x == s ? 1 : 2 + helper_index(x, s, myrank), &off, &len);
            if (len) M.recvB.push_back({x, k - 1, 1, rdispl[si] + off, len});
        }
    }
    M.staging = need ? need : 256;
    M.built = true;
    return 0;
}

int relay_alltoallv(dfft_comm *comm, RelayCache *cache, uint64_t tag, int myrank, const void *send, const size_t *scount,
                    const size_t *sdispl, void *recv, const size_t *rcount, const size_t *rdispl, const int *group, int ngroup,
                    int me, hipStream_t stream, int channel, hipEvent_t ready)
{
    const int n = comm->nranks, R = ngroup - 1;
    const int dev = relay_on_device(send);
    if (dev < 0) return 1;
    const bool device = dev == 1;
    char *sb = const_cast<char *>(static_cast<const char *>(send));
    char *rb = static_cast<char *>(recv);
    auto it = cache->meta.find(tag);
    if (it == cache->meta.end()) {
        RelayMeta M;
        if (int r = relay_gather_meta(comm, cache, M, myrank, scount, group, ngroup, me, device, stream, channel)) return r;
        it = cache->meta.emplace(tag, std::move(M)).first;
    }
    RelayMeta &M = it->second;
    if (M.R != R) { set_error("relay: exchange table changed under its tag"); return 1; }
    if (!M.built)
        if (int r = relay_build(M, n, myrank, scount, sdispl, rcount, rdispl, group, ngroup, me)) return r;
    RelayLane &L = cache->lanes[std::make_pair((void *)stream, channel)];
    // hop 1 on a side stream, so that it overlaps hop 2 of the previous call on this lane (the previous pipeline chunk)?
    const bool overlap = device && comm->relay_overlap && comm->concurrent_hops(channel);
    // First use of this table on this lane: everything that can fail locally, then one word of agreement.  (Receive regions of
    // neighbouring chunks must be disjoint when a `ready` event is passed: hop 1 of chunk c + 1 fills the receive buffer after `ready`
    // only, under hop 2 of chunk c -- true for the chunk-outermost tables of a plan.)
    const auto lane_key = std::make_pair((void *)stream, channel);
    if (!M.direct && !M.agreed.count(lane_key)) {
        int local = 0;
        auto setup = [&]() -> int {
            // (test hook: DFFT_RELAY_TEST_FAIL_RANK makes this step fail on one rank -- tests/test_gpu_relay.py checks that every rank
            // then sends the table directly instead of waiting for it in a collective)
            if (const char *e = getenv("DFFT_RELAY_TEST_FAIL_RANK"))
                if (atoi(e) == myrank) { dfft::set_error("relay: staging setup failed (injected by DFFT_RELAY_TEST_FAIL_RANK)"); return 1; }
            for (int q = 0; q < (overlap ? 2 : 1); q++)
                if (int r = relay_reserve(L.staging[q], M.staging, device, stream)) return r;
            if (overlap && !L.side) {
                HIP_TRY(hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking));
                for (hipEvent_t *e : {&L.in, &L.hop1, &L.hop2[0], &L.hop2[1]}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
            }
            return 0;
        };
        local = setup() != 0;
        std::string local_why;
        if (local) { local_why = dfft_last_error(); (void)hipGetLastError(); }
        uint64_t worst = 0;
        if (int r = relay_agree(comm, cache, myrank, device, stream, channel, (uint64_t)local, &worst)) return r;
        if (worst) {
            M.direct = true;
            fprintf(stderr, "[dfft] relay: rank %d: %s -- this exchange table is sent directly on every rank\n", myrank,
                    local ? ("local setup failed (" + local_why + ")").c_str() : "another rank could not set up its staging");
        } else M.agreed[lane_key] = true;
    }
    if (M.direct) {
        comm->counters.alltoallv++;
        return comm->alltoallv(myrank, send, scount, sdispl, recv, rcount, rdispl, group, ngroup, me, stream, channel);
    }
    const int par = overlap ? L.parity : 0;
    if (overlap) L.parity ^= 1;
    RelayBuf &st = L.staging[par];
    if (st.cap < M.staging) { set_error("relay: staging smaller than the agreed size"); return 1; }      // (cannot happen: reserved above)
    hipStream_t s1 = stream;
    if (overlap) {
        s1 = L.side;
        // hop 1 of this call writes the staging buffer that hop 2 of the call before the previous one read
        if (L.hop2_used[par]) HIP_TRY(hipStreamWaitEvent(s1, L.hop2[par], 0));
        if (ready) HIP_TRY(hipStreamWaitEvent(s1, ready, 0));
        else { HIP_TRY(hipEventRecord(L.in, stream)); HIP_TRY(hipStreamWaitEvent(s1, L.in, 0)); }
    }
    // debugging aid (DFFT_RELAY_POISON=1): the staging buffer is overwritten before hop 1 fills it, on hop 1's stream -- a hop 2 that
    // forwarded a part before hop 1 had written it would then deliver the poison instead of whatever an earlier exchange left there
    static const bool poison = [] { const char *e = getenv("DFFT_RELAY_POISON"); return e && atoi(e) != 0; }();
    if (poison && device) HIP_TRY(hipMemsetAsync(st.p, 0xEE, M.staging, s1));
    // self block: a local copy, as in every transport
    if (rcount[me]) {
        if (device) HIP_TRY(hipMemcpyAsync(rb + rdispl[me], sb + sdispl[me], rcount[me], hipMemcpyDeviceToDevice, s1));
        else memcpy(rb + rdispl[me], sb + sdispl[me], rcount[me]);
    }
    char *const bases[3] = {sb, rb, st.p};
    auto hop = [&](const std::vector<RelayMeta::Piece> &sp, const std::vector<RelayMeta::Piece> &rp, hipStream_t s, int ch) -> int {
        std::vector<dfft_xfer> sx(sp.size()), rx(rp.size());
        for (size_t i = 0; i < sp.size(); i++) sx[i] = dfft_xfer{sp[i].peer, sp[i].layer, bases[sp[i].where] + sp[i].off, sp[i].bytes};
        for (size_t i = 0; i < rp.size(); i++) rx[i] = dfft_xfer{rp[i].peer, rp[i].layer, bases[rp[i].where] + rp[i].off, rp[i].bytes};
        return comm->sendrecv_list(myrank, sx.data(), (int)sx.size(), rx.data(), (int)rx.size(), R, s, ch);
    };
    comm->counters.relayed++;
    if (int r = hop(M.sendA, M.recvA, s1, overlap ? channel + 2 : channel)) return r;
    if (overlap) {
        HIP_TRY(hipEventRecord(L.hop1, s1));
        HIP_TRY(hipStreamWaitEvent(stream, L.hop1, 0));
    }
    if (int r = hop(M.sendB, M.recvB, stream, channel)) return r;
    if (overlap) {
        HIP_TRY(hipEventRecord(L.hop2[par], stream));
        L.hop2_used[par] = true;
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

// The schedule as `nlayers` all-to-all-v calls over the whole communicator (transports without a native form).  The pieces of a layer
// lie in unrelated allocations: the displacements are differences to the lowest address of the layer (flat addressing).
int dfft_comm::sendrecv_list(int myrank, const dfft_xfer *sends, int ns, const dfft_xfer *recvs, int nr, int nlayers, hipStream_t stream,
                             int channel)
{
    const int n = nranks;
    std::vector<int> world(n);
    for (int y = 0; y < n; y++) world[y] = y;
    std::vector<size_t> sc(n), sd(n), rc(n), rd(n);
    for (int layer = 0; layer < nlayers; layer++) {
        const char *smin = nullptr, *rmin = nullptr;
        for (int i = 0; i < ns; i++)
            if (sends[i].layer == layer && (!smin || (const char *)sends[i].ptr < smin)) smin = (const char *)sends[i].ptr;
        for (int i = 0; i < nr; i++)
            if (recvs[i].layer == layer && (!rmin || (const char *)recvs[i].ptr < rmin)) rmin = (const char *)recvs[i].ptr;
        std::fill(sc.begin(), sc.end(), 0); std::fill(sd.begin(), sd.end(), 0);
        std::fill(rc.begin(), rc.end(), 0); std::fill(rd.begin(), rd.end(), 0);
        for (int i = 0; i < ns; i++)
            if (sends[i].layer == layer) {
                if (sends[i].peer < 0 || sends[i].peer >= n || sc[sends[i].peer]) { dfft::set_error("sendrecv_list: two pieces for one peer in one layer"); return 1; }
                sc[sends[i].peer] = sends[i].bytes;
                sd[sends[i].peer] = (size_t)((const char *)sends[i].ptr - smin);
            }
        for (int i = 0; i < nr; i++)
            if (recvs[i].layer == layer) {
                if (recvs[i].peer < 0 || recvs[i].peer >= n || rc[recvs[i].peer]) { dfft::set_error("sendrecv_list: two pieces for one peer in one layer"); return 1; }
                rc[recvs[i].peer] = recvs[i].bytes;
                rd[recvs[i].peer] = (size_t)((const char *)recvs[i].ptr - rmin);
            }
        static const char none = 0;
        counters.alltoallv++;
        counters.layered++;
        const int r = alltoallv(myrank, smin ? smin : &none, sc.data(), sd.data(), rmin ? const_cast<char *>(rmin) : const_cast<char *>(&none), rc.data(),
                                rd.data(), world.data(), n, myrank, stream, channel);
        counters.layered--;
        if (r) return r;
    }
    return 0;
}
