// comm.hpp -- exchange transports behind dfft_comm.
//
// Replaces the reference's MPI layer for the hot path: MPI_Comm_split into row/column
// communicators (src/pencil/mpicufft_pencil_opt1.cpp:103-104) and MPI_Alltoallv on device
// pointers (:784-785, :1297-1298).  A transport only has to provide an all-to-all-v among an
// explicit list of ranks on a HIP stream; sub-communicators are expressed as rank lists of
// the parent, so no split is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>

// one piece of a point-to-point schedule (sendrecv_list): `bytes` at `ptr` to / from global rank `peer`.  Pieces between the same
// ordered pair of ranks are matched by `layer` (both ends give a piece the same layer; at most one piece per peer, direction and layer).
struct dfft_xfer {
    int peer;
    int layer;
    void *ptr;
    size_t bytes;
};

struct dfft_comm {
    int nranks = 1;
    virtual ~dfft_comm() {}
    // rank of the caller if the transport knows it (RCCL, callback), -1 for a local world
    virtual int fixed_rank() const { return -1; }
    // counts/displacements in bytes; group = global ranks, me = my index in group.
    // `channel` (0 or 1) lets a transport keep independent resources per exchange so that the two
    // exchanges of a pencil plan, which use disjoint links, may be in flight at the same time.  It is an
    // argument, not communicator state: several host threads (virtual ranks) share one communicator.
    // true if exchanges issued on channel 0 and channel 1 may run concurrently on two streams
    virtual bool concurrent_channels() const { return true; }
    // true if the relay's first hop (channel + 2, on a side stream) may be in flight together with the second hop on `channel`
    virtual bool concurrent_hops(int /*channel*/) const { return true; }
    virtual int alltoallv(int myrank, const void *send, const size_t *scount, const size_t *sdispl,
                          void *recv, const size_t *rcount, const size_t *rdispl, const int *group,
                          int ngroup, int me, hipStream_t stream, int channel) = 0;
    // A schedule of point-to-point pieces over the WHOLE communicator as ONE grouped transport operation on `stream` (the two-hop relay's
    // hops: several pieces per peer, none of them back to back).  Every rank calls it at the same point with matching lists; `nlayers` is
    // the same on all ranks.  Pointers are absolute.  The default runs the schedule as `nlayers` all-to-all-v calls (layer by layer) and
    // needs a transport that takes base + displacement as a flat address (all of them do: RCCL, the callbacks get raw pointers).
    virtual int sendrecv_list(int myrank, const dfft_xfer *sends, int ns, const dfft_xfer *recvs, int nr, int nlayers, hipStream_t stream,
                              int channel);
    // number of ranks the transport itself reports (ncclCommCount for RCCL); 0 if it has no such notion
    virtual int transport_nranks() const { return 0; }
    // host-side rendezvous of all ranks (used around timing); no-op by default
    virtual void barrier(int /*myrank*/) {}
    // transport knobs (dfft_comm_set_option): 0 = ok, 1 = unknown key / unsupported, else a transport error.
    //   "dup_channel" = 1 (rccl): COLLECTIVE over all ranks of the communicator -- duplicates it (ncclCommSplit) so that
    //   channel 1 (the second exchange of a pencil plan) may be on the wire together with channel 0
    virtual int set_option(const char * /*key*/, long /*value*/) { return 1; }
    // Two-hop relay (dfft_comm_set_option "relay"; every transport, handled above alltoallv: dfft::relay_alltoallv).
    // bit 0: the column-group exchange of pencil plans (exchange 2), bit 1: the row-group exchange (exchange 1).
    int relay = 0;
    // 1 (default): the second hop of a relayed chunk runs on the exchange's stream while the first hop of the NEXT chunk runs on a
    // side stream of the relay (channel + 2: its own communicator where the transport has one), so that the hops of neighbouring
    // pipeline chunks overlap; 0: both hops on the exchange's stream, one after the other
    int relay_overlap = 1;
    int test_channel = 0;      // the channel of the bare transport calls of the C ABI (dfft_comm_alltoallv, dfft_comm_sendrecv_list)
    // what went through the transport since the communicator was made (dfft_comm_get_counter): calls of alltoallv / sendrecv_list
    // made by plans and by the relay, and relayed exchanges
    struct Counters { std::atomic<long> alltoallv{0}, list{0}, relayed{0}, relay_meta{0}, relay_agree{0};
                      // > 0 while sendrecv_list's default runs a schedule as all-to-all-v layers: the pieces of such a call lie in
                      // unrelated allocations ("layered" tells a callback transport not to look for back-to-back blocks)
                      std::atomic<long> layered{0}; } counters;
};

namespace dfft {
dfft_comm *make_local_world(int nranks);
dfft_comm *make_rccl_comm(const void *id128, int nranks, int rank);
dfft_comm *make_callback_comm(int nranks, int rank, void *fn, void *user);
// optional second callback of a callback communicator: the point-to-point schedule in one call (dfft_comm_set_list_callback)
int callback_comm_set_list(dfft_comm *comm, void *fn, void *user);
int rccl_unique_id(void *id128);

// Two-hop relay of a group all-to-all over the WHOLE world (comm.hip).  xGMI is point to point: while the column groups of a
// 2 x 4 pencil grid exchange, each GPU drives ONE of its seven links.  The relay cuts every message into nranks parts: two go
// directly (one per phase), the others to the nranks - 2 ranks outside the pair, which forward them in the second phase, so that
// all links carry the same load in both phases (1 GiB over one link: 7.0 ms at 153 GB/s; as 2 x 1/8 GiB per link: 1.75 ms).
// ONE-SHOT: all partners of a group exchange travel together -- hop 1 is one grouped operation in which every link carries one part of
// EACH of the rank's messages (and, as a helper, receives one part of every other pair's), hop 2 a second one that delivers the direct
// second parts and forwards everything staged: two transport operations (sendrecv_list) per exchange and pipeline chunk, whatever the
// group size (rounds 3-4: 2 (P - 1) world-wide all-to-alls, one pair per partner, strictly one after the other).  Transports without
// a native schedule run each hop as P - 1 all-to-all layers.  With `ready` (an event recorded when the send data was produced) and
// dfft_comm::relay_overlap, hop 1 runs on a side stream ordered after `ready` only, hop 2 on `stream`: hop 1 of chunk c + 1 overlaps
// hop 2 of chunk c (double-buffered staging).  The bytes land exactly where the direct exchange puts them.
// COLLECTIVE over all ranks of the communicator (the groups of one exchange partition the world and every rank is in that exchange at the same point of the plan).  `cache` belongs to the caller (one
// per plan); `tag` names the exchange table: the first call with a tag gathers every rank's partners and message sizes once.
struct RelayCache;
RelayCache *relay_cache_new();
void relay_cache_free(RelayCache *c);
int relay_alltoallv(dfft_comm *comm, RelayCache *cache, uint64_t tag, int myrank, const void *send, const size_t *scount,
                    const size_t *sdispl, void *recv, const size_t *rcount, const size_t *rdispl, const int *group, int ngroup,
                    int me, hipStream_t stream, int channel, hipEvent_t ready = nullptr);
// part p (0 .. nranks - 1) of a message of S bytes: parts 0 and 1 travel directly, part 2 + h through helper h
void relay_part(size_t S, int nranks, int p, size_t *off, size_t *len);
}  // namespace dfft
