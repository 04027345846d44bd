// comm.hip -- the three exchange transports (see comm.hpp).
#include "comm.hpp"
#include "dfft_internal.hpp"
#include "../../include/dfft_c.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
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
        if (rcount[me])
            HIP_TRY(hipMemcpyAsync(r + rdispl[me], s + sdispl[me], rcount[me], hipMemcpyDeviceToDevice, stream));
        NCCL_TRY(R->GroupStart());
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
