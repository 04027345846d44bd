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
};

namespace dfft {
dfft_comm *make_local_world(int nranks);
dfft_comm *make_rccl_comm(const void *id128, int nranks, int rank);
dfft_comm *make_callback_comm(int nranks, int rank, void *fn, void *user);
int rccl_unique_id(void *id128);
}  // namespace dfft
