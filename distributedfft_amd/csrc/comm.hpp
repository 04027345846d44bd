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
    virtual int alltoallv(int myrank, const void *send, const size_t *scount, const size_t *sdispl,
                          void *recv, const size_t *rcount, const size_t *rdispl, const int *group,
                          int ngroup, int me, hipStream_t stream, int channel) = 0;
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
};

namespace dfft {
dfft_comm *make_local_world(int nranks);
dfft_comm *make_rccl_comm(const void *id128, int nranks, int rank);
dfft_comm *make_callback_comm(int nranks, int rank, void *fn, void *user);
int rccl_unique_id(void *id128);

// Two-hop relay of a group all-to-all over the WHOLE world (comm.hip).  xGMI is point to point: while the column groups of a
// 2 x 4 pencil grid exchange, each GPU drives ONE of its seven links.  The relay cuts every message into nranks parts: two go
// directly (one per phase), the others to the nranks - 2 ranks outside the pair, which forward them in the second phase, so that
// all links carry the same load in both phases (1 GiB over one link: 7.0 ms at 153 GB/s; as 2 x 1/8 GiB per link: 1.75 ms).
// Built from two world-wide alltoallv calls of the underlying transport per partner, so it works on every transport; the bytes
// land exactly where the direct exchange puts them.  COLLECTIVE over all ranks of the communicator (the groups of one exchange
// partition the world and every rank is in that exchange at the same point of the plan).  `cache` belongs to the caller (one
// per plan); `tag` names the exchange table: the first call with a tag gathers every rank's partners and message sizes once.
struct RelayCache;
RelayCache *relay_cache_new();
void relay_cache_free(RelayCache *c);
int relay_alltoallv(dfft_comm *comm, RelayCache *cache, uint64_t tag, int myrank, const void *send, const size_t *scount,
                    const size_t *sdispl, void *recv, const size_t *rcount, const size_t *rdispl, const int *group, int ngroup,
                    int me, hipStream_t stream, int channel);
// part p (0 .. nranks - 1) of a message of S bytes: parts 0 and 1 travel directly, part 2 + h through helper h
void relay_part(size_t S, int nranks, int p, size_t *off, size_t *len);
}  // namespace dfft
