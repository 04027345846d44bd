/*
 * dfft_c.h -- C ABI of libdfft_amd.so, the MI355X-native replacement for the hot path of
 * eggersn/DistributedFFT: "plan once, then execR2C / execC2R / (new) execC2C on device buffers".
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  The boundary is the abstract class
 *     template<typename T> class MPIcuFFT            include/mpicufft.hpp:55-105
 * and its concrete decompositions
 *     MPIcuFFT_Slab / _Slab_Opt1                      include/mpicufft_slab.hpp:99-125
 *     MPIcuFFT_Pencil / _Pencil_Opt1                  include/mpicufft_pencil.hpp:88-122
 * Differences by design: plain C, int return codes instead of printf+exit(EXIT_FAILURE)
 * (src/pencil/mpicufft_pencil_opt1.cpp:27-33), the communicator is an explicit dfft_comm
 * (RCCL over xGMI, in-process virtual ranks, or a caller-supplied all-to-all) instead of an
 * MPI_Comm, and complex-to-complex execution is added next to R2C/C2R.
 *
 * Memory layout contract (identical to the reference, README.md:247,
 * include/mpicufft_pencil.hpp:94-122): row-major [x][y][z], z contiguous.
 *   input  block of rank (i,j): [Nx/P1][Ny/P2][Nz]        (real for R2C, complex for C2C)
 *   output block of rank (i,j): [Nx][Ny/P1][Nzc/P2]       (complex; Nzc = Nz/2+1 or Nz)
 * Remainders of uneven divisions go to the lowest ranks.  rank = i*P2 + j.  Slab = P2 == 1.
 * Transforms are unnormalised, forward kernel exp(-2*pi*i*jk/N) (cuFFT convention).
 *
 * All functions return 0 on success; on failure a non-zero code and dfft_last_error() holds
 * a message (thread local).
 */
#ifndef DFFT_C_H
#define DFFT_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dfft_plan dfft_plan;
typedef struct dfft_comm dfft_comm;

/* which reference class the plan stands in for (all four share one engine: the "realigned"
 * opt1 data flow with pack/unpack fused into the FFT kernels; the value is kept for callers
 * that select by name and for the timer/CSV naming) */
enum dfft_kind {
    DFFT_SLAB = 0,         /* MPIcuFFT_Slab       include/mpicufft_slab.hpp          */
    DFFT_SLAB_OPT1 = 1,    /* MPIcuFFT_Slab_Opt1  include/mpicufft_slab_opt1.hpp     */
    DFFT_PENCIL = 2,       /* MPIcuFFT_Pencil     include/mpicufft_pencil.hpp        */
    DFFT_PENCIL_OPT1 = 3,  /* MPIcuFFT_Pencil_Opt1 include/mpicufft_pencil_opt1.hpp  */
    /* alternative slab sequence: 1-D z pass, one all-to-all, 2-D (y,x) pass.  Input split along x,
     * OUTPUT split along z: [Nx][Ny][Nzc/P] (include/mpicufft_slab_z_then_yx.hpp:41-44).  P2 must be 1. */
    DFFT_SLAB_Z_THEN_YX = 4,       /* MPIcuFFT_Slab_Z_Then_YX      include/mpicufft_slab_z_then_yx.hpp      */
    DFFT_SLAB_Z_THEN_YX_OPT1 = 5,  /* MPIcuFFT_Slab_Z_Then_YX_Opt1 include/mpicufft_slab_z_then_yx_opt1.hpp */
    /* forward only, like the reference: R2C along y, one all-to-all, 2-D (z,x) pass.  Output
     * [Nx][(Ny/2+1)/P][Nz] (include/mpicufft_slab_y_then_zx.hpp:40-43).  R2C plans: any Ny dfft_init accepts.  Unlike
     * the reference's single-rank branch (a z-Hermitian cufftPlan3d, mpicufft_slab_y_then_zx.cpp:111-121),
     * one rank produces the same y-Hermitian layout as several. */
    DFFT_SLAB_Y_THEN_ZX = 6        /* MPIcuFFT_Slab_Y_Then_ZX      include/mpicufft_slab_y_then_zx.hpp      */
};
enum dfft_precision { DFFT_F32 = 0, DFFT_F64 = 1 };   /* template parameter T = float|double */
enum dfft_direction { DFFT_FORWARD = -1, DFFT_INVERSE = 1 };

/* struct Configurations, include/params.hpp:83-93.  comm/send method enums are accepted and
 * recorded; the device exchange always runs as one grouped all-to-all per exchange. */
typedef struct dfft_config {
    int cuda_aware;        /* kept for source compatibility; device buffers are always used */
    int warmup_rounds;
    int comm_method;       /* 0 Peer2Peer, 1 All2All   (params.hpp:83) */
    int send_method;       /* 0 Sync, 1 Streams, 2 MPI_Type (params.hpp:84) */
    int comm_method2;
    int send_method2;
} dfft_config;

/* ---------------------------------------------------------------- communicators ---------- */
/* replaces MPI_Comm + MPI_Comm_split (src/mpicufft.cpp:42-51, mpicufft_pencil_opt1.cpp:103-104) */

/* all-to-all-v callback: byte counts/displacements, `group` = global ranks of the members,
 * `me` = caller's index in group.  Buffers are device pointers; `stream` is the hipStream_t
 * the plan enqueues on.  Must return 0 on success. */
typedef int (*dfft_alltoallv_fn)(void *user, const void *sendbuf, const size_t *sendcounts,
                                 const size_t *sdispls, void *recvbuf, const size_t *recvcounts,
                                 const size_t *rdispls, const int *group, int ngroup, int me,
                                 void *stream);

/* Optional second callback of a callback communicator: a schedule of point-to-point pieces over the WHOLE communicator as one
 * grouped operation (the two hops of the relay, see "relay" below: several pieces per peer, not back to back).  Piece i of the send
 * list goes to global rank speer[i], piece j of the receive list comes from rpeer[j]; pointers are absolute; the pieces between two
 * ranks appear in the same order at both ends (match them in that order: grouped ncclSend / ncclRecv, torch.distributed
 * batch_isend_irecv, MPI_Isend / MPI_Irecv with the position as the tag).  Must return 0 on success.  Without it the library runs a
 * schedule as group_size - 1 all-to-all-v callbacks. */
typedef int (*dfft_sendrecv_list_fn)(void *user, int nsend, const int *speer, void *const *sptr, const size_t *sbytes, int nrecv,
                                     const int *rpeer, void *const *rptr, const size_t *rbytes, void *stream);

/* nranks virtual ranks inside this process sharing the current device; each rank's exec must be
 * called from its own host thread (they meet in a barrier like MPI ranks would). */
int dfft_comm_create_local(int nranks, dfft_comm **world);
/* one process per GPU over RCCL/xGMI; id = 128-byte ncclUniqueId made on rank 0 by
 * dfft_rccl_unique_id and broadcast by the caller (MPI_Bcast, torch.distributed, ...) */
int dfft_rccl_unique_id(void *id128);
int dfft_comm_create_rccl(const void *id128, int nranks, int rank, dfft_comm **comm);
/* caller-supplied exchange (e.g. torch.distributed.all_to_all_single, MPI_Alltoallv) */
int dfft_comm_create_callback(int nranks, int rank, dfft_alltoallv_fn fn, void *user, dfft_comm **comm);
int dfft_comm_set_list_callback(dfft_comm *comm, dfft_sendrecv_list_fn fn, void *user);
/* nranks = the size the communicator was created with; transport_nranks = what the transport itself reports
 * (ncclCommCount for the RCCL transport, 0 for the others), so that a caller can verify that RCCL really
 * spans the ranks it claims */
int dfft_comm_info(const dfft_comm *comm, int *nranks, int *transport_nranks);
/* What went through a communicator since it was made: "alltoallv" (all-to-all-v operations of the transport: the direct exchanges, the
 * relay's table gathers, the layers of a schedule on a transport without a native form), "list" (native point-to-point schedules:
 * one per hop of a relayed exchange), "relayed" (relayed exchanges), "relay_meta" (table gathers: one per exchange table).  A relayed
 * exchange of one pipeline chunk is 2 "list" operations on the RCCL, the local-world and the torch transport. */
int dfft_comm_get_counter(const dfft_comm *comm, const char *name, long *value);
/* transport knobs.  "dup_channel" = 1 (RCCL transport only; COLLECTIVE -- every rank of the communicator calls it at
 * the same point): duplicate the communicator (ncclCommSplit) for the second exchange of pencil plans, so that the row-
 * and the column-group exchange -- which use disjoint xGMI links -- may be on the wire at the same time; without it one
 * ncclComm serialises them.  "dup_channel" = 3: two more duplicates for the relay's first hop of either exchange, which then
 * overlaps the second hop of the previous pipeline chunk ("relay_overlap").  Replaces nothing in the reference (its two MPI sub-communicators are independent by
 * construction, src/pencil/mpicufft_pencil_opt1.cpp:103-104).
 * "self_send" = 1 (RCCL transport, testing): a rank's own block goes through ncclSend / ncclRecv to itself instead of a device
 * copy, so that a single-GPU box can hand caller buffers (virtual-memory ranges from dfft_malloc included) to RCCL.
 * "relay" = 0..3 (every transport; a flag, not collective by itself, but every rank must set the same value before the next
 * exec): two-hop relay of the group exchanges.  bit 0 = exchange 2 (column groups), bit 1 = exchange 1 (row groups).  xGMI is
 * point to point, so the column groups of a 2 x 4 pencil grid move half of the volume over ONE of a GPU's seven links; with the
 * relay a message is cut into nranks parts, two travel directly and each of the others through one of the ranks outside the
 * pair.  All partners of an exchange travel together: TWO grouped transport operations per exchange and pipeline chunk (hop 1: every
 * link carries one part of each of the rank's messages; hop 2: the second direct parts and everything staged goes on to its
 * destination), all links equally loaded in both (1 GiB over one link: 7.0 -> 1.75 ms at 153 GB/s per link).  The bytes
 * land exactly where the direct exchange puts them.  "relay_overlap" = 1 (default) / 0: hop 1 of a pipeline chunk runs on a side
 * stream (on the RCCL transport: on its own communicator if "dup_channel" = 3 made one) under hop 2 of the chunk before it.  The reference's answer to its slow exchanges was per-peer overlap
 * (src/pencil/mpicufft_pencil_opt1.cpp:1116-1275); this is the xGMI analogue.  Default off.
 * Returns 0, or nonzero for an unknown key / a failure. */
int dfft_comm_set_option(dfft_comm *comm, const char *key, long value);
/* The transport's all-to-all-v on its own -- what a plan calls for its exchanges (counts / displacements in bytes, `group` =
 * ranks of the communicator taking part, `me` = the caller's index in it; hip_stream may be NULL).  Stream-ordered: returns
 * when the exchange is enqueued.  For callers that move their own blocks with the library's transports, and for tests of a
 * transport by itself (the reference's counterpart is a bare MPI_Alltoallv, src/pencil/mpicufft_pencil_opt1.cpp:784-785). */
int dfft_comm_alltoallv(dfft_comm *comm, int myrank, const void *send, const size_t *scounts, const size_t *sdispls, void *recv,
                        const size_t *rcounts, const size_t *rdispls, const int *group, int ngroup, int me, void *hip_stream);
/* The transport's point-to-point schedule on its own -- what the relay calls for each of its two hops: piece i of the send list goes
 * to rank speer[i], piece j of the receive list comes from rpeer[j]; `slayer` / `rlayer` number the pieces between one ordered pair of
 * ranks (both ends give a piece the same layer, at most one piece per peer, direction and layer; nlayers the same on all ranks).  One
 * grouped operation on the RCCL, local-world and list-callback transports, nlayers all-to-all-v calls elsewhere.  COLLECTIVE over the
 * communicator.  Stream-ordered.  Option "test_channel" = 0..3 picks the channel (communicator) the two bare calls use. */
int dfft_comm_sendrecv_list(dfft_comm *comm, int myrank, int nsend, const int *speer, const int *slayer, void *const *sptr, const size_t *sbytes,
                            int nrecv, const int *rpeer, const int *rlayer, void *const *rptr, const size_t *rbytes, int nlayers, void *hip_stream);
/* destroy the plans that use a communicator before the communicator itself */
int dfft_comm_destroy(dfft_comm *comm);

/* ---------------------------------------------------------------- plan -------------------- */
/* MPIcuFFT<T>::MPIcuFFT(Configurations, MPI_Comm, int max_world_size)   src/mpicufft.cpp:42-66.
 * comm may be NULL for a single rank.  `rank` selects the virtual rank of a local world and is
 * ignored (taken from the comm) otherwise.  max_world_size < 0 = all ranks. */
int dfft_plan_create(dfft_plan **plan, int kind, int precision, const dfft_config *config,
                     dfft_comm *comm, int rank, int max_world_size);
/* ~MPIcuFFT  src/mpicufft.cpp:68-73 (frees the work area if the library allocated it) */
int dfft_plan_destroy(dfft_plan *plan);

/* initFFT(GlobalSize*, Partition*, bool allocate)   include/mpicufft.hpp:60,
 * src/pencil/mpicufft_pencil_opt1.cpp:46-326, src/slab/default/mpicufft_slab.cpp:97-281.
 * P1*P2 must equal the number of ranks.  Axis lengths: powers of two up to 8192 (4096 on the real axis of an R2C
 * plan) and the lengths 2^a 3^b 5^c 7^d <= 2048 listed in csrc/kernels_mixed.inc (native chain), any other length
 * up to 4096 (Bluestein), and beyond that every length N1*N2 <= 2^24 whose factors are each such a length (two-level
 * lines: two launches per pass and a scratch region in the work area, dfft_axis_plan_info shows the split); a length
 * that does not split that way (a prime above 4096, twice such a prime ...) runs Bluestein's algorithm over a two-level padded
 * length (four launches), so every length from 2 to 2^23 has a plan; beyond 2^24 points per line: ERR_UNSUPPORTED.  c2c = 0: R2C/C2R plan (Nz_out = Nz/2+1,
 * include/params.hpp:30); c2c = 1: complex plan (Nz_out = Nz). */
int dfft_init(dfft_plan *plan, size_t Nx, size_t Ny, size_t Nz, int P1, int P2, int c2c, int allocate);
/* setWorkArea(void *device, void *host)   mpicufft_pencil_opt1.cpp:329-387.  NULL device =
 * library allocates dfft_work_size_device() bytes. */
int dfft_set_work_area(dfft_plan *plan, void *device, void *host);
/* Pipeline depth of the exchanges: every pass that feeds an all-to-all is cut into `chunks`
 * pieces so that chunk c travels (communication stream) while chunk c+1 is transformed (compute
 * stream).  The reference's counterpart is the Peer2Peer overlap of
 * src/pencil/mpicufft_pencil_opt1.cpp:601-754.  Takes effect at the next dfft_init; 0 = default (4 when the
 * plan has an exchange, else 1), 1 = no pipelining = the reference's exact message sizes. */
int dfft_set_pipeline_chunks(dfft_plan *plan, int chunks);
int dfft_get_pipeline_chunks(const dfft_plan *plan);
/* Tuning knobs, by name (takes effect at the next dfft_init unless noted).  The reference's counterpart is the
 * Configurations struct (include/params.hpp:85-93); these are the knobs this engine has instead:
 *   "pipeline_chunks"  as dfft_set_pipeline_chunks                      (env default DFFT_CHUNKS)
 *   "mirror_inverse"   1: a single-rank complex inverse runs the multi-rank pass order x, y, z instead of the
 *                      forward order with conjugation (next exec; benchmarking the N > 1 compute path on one GPU;
 *                      env default DFFT_MIRROR)
 *   "point_tables"     per-point address tables: 0 never, 1 segmented sides (default), 2 always (DFFT_TABLES)
 *   "shift"            row-aligned tile windows for odd-pitch output rows: -1 auto, 0 off, 2 always (DFFT_SHIFT)
 *   "single_order"     one rank, complex plan: 1 = pass order z, x, y through a padded private layout, 0 = z, y, x, -1 (default)
 *                      = where it measured faster (env default DFFT_SINGLE_ORDER); "single_layout" (0 | 1) and "single_pad"
 *                      (bytes) shape that layout
 *   "spectral_layout"  0 (default): the spectrum has the reference's layout [Nx][yo][zs] (include/mpicufft_pencil.hpp:94-122).  1: it is kept
 *                      x-contiguous, [yo][zs][Nx] -- entry (kx, ky, kz') at kx + Nx * (kz' + zs * ky), dfft_get_out_strides -- so
 *                      that the forward x pass stores natural lines and the inverse x pass loads them: neither touches the
 *                      point-major layout, whose strided read (16-byte points Nx rows apart) is the slowest pass of every multi-rank
 *                      plan.  For callers that go forward -> pointwise work on the spectrum -> inverse (the reference's testcase 4,
 *                      tests/src/pencil/random_dist_3D.cu:748-793, does exactly that) and can index through the strides.
 *                      getOutSize / getOutStart / the exchange tables are unchanged.  Pencil and default slab plans, any rank count.
 *   "graph"            1: a single-rank plan replays the kernel launches of an exec as one hipGraph from the second call
 *                      with the same (operation, in, out) on; 0 (default): plain launches -- measured 6-8 us faster per
 *                      exec on ROCm 7.2 (profiles/r2_graph_latency.txt)
 *   "native_mixed"     1 (default): lengths that are not powers of two but have a mixed-radix configuration run the
 *                      native chain; 0: they run the Bluestein kernel like every other length (A/B runs, tests)
 *   "debug_skip"       measurement only, a bit set: bit 0 (value 1) = every pass skips its transform and becomes a copy with
 *                      the same access pattern (results are wrong); used to measure the pattern's own roofline.  Bit 1
 *                      (value 2) = the closed-form address sides compute per-point 64-bit vector addresses instead of a scalar
 *                      base + 32-bit lane offset (same results bit for bit; A/B of the address forms)
 *   "variant_<pass>", "order_<pass>", "real_variant"   kernel configuration / workgroup order per pass
 *                      (<pass> = fz fy fx ix iy iz; -1 = the plan's choice), for A/B measurements
 * Returns nonzero for an unknown key.  dfft_get_option returns -1 for an unknown key. */
int dfft_set_option(dfft_plan *plan, const char *key, long value);
long dfft_get_option(dfft_plan *plan, const char *key);
/* HIP stream all kernels/exchanges are enqueued on (default: a non-blocking stream owned by the
 * plan).  exec orders itself only against this stream: work that produces `in` or initialises
 * `out` on another stream must be complete (or that stream handed over here) before exec. */
int dfft_set_stream(dfft_plan *plan, void *hip_stream);

/* execR2C(void *out, const void *in)   include/mpicufft.hpp:63; mpicufft_pencil_opt1.cpp:1422-1519.
 * in: device, real [xs][ys][Nz], not modified.  out: device, >= dfft_domain_size() bytes;
 * holds [Nx][yo][zs] complex on return.  Blocking (returns after the stream is drained), like
 * the reference's trailing cudaDeviceSynchronize. */
int dfft_exec_r2c(dfft_plan *plan, void *out, const void *in);
/* execC2R(void *out, const void *in)   include/mpicufft_pencil.hpp:106-111; :1522-1600.
 * `in` is DESTROYED (used as scratch, as in the reference :1534,1544,1563). */
int dfft_exec_c2r(dfft_plan *plan, void *out, void *in);
/* new: complex-to-complex, same layouts with Nz_out = Nz.  Forward: in [xs][ys][Nz] -> out
 * [Nx][yo][zs].  Inverse: in [Nx][yo][zs] (destroyed) -> out [xs][ys][Nz]. */
int dfft_exec_c2c(dfft_plan *plan, void *out, void *in, int direction);
/* partial transforms: execR2C(out, in, d) / execC2R(out, in, d) of MPIcuFFT_Pencil
 * (include/mpicufft_pencil.hpp:101-111, src/pencil/mpicufft_pencil.cpp:1644-1839).  Works for
 * R2C and C2C plans alike (the plan decides).  d = 3: the full transform.  d = 1: z axis only,
 * stage layout [xs][ys][Nzc] (natural, z contiguous).  d = 2: z then y, stage layout
 * [xs][Ny][zs].  The inverse direction takes those layouts as input and returns [xs][ys][Nz]. */
int dfft_exec_dim(dfft_plan *plan, void *out, void *in, int direction, int d);
/* non-blocking variants: enqueue only (caller synchronises the stream) */
int dfft_enqueue_c2c(dfft_plan *plan, void *out, void *in, int direction);

/* getInSize/getInStart/getOutSize/getOutStart   include/mpicufft_pencil.hpp:112-122
 * (getOutStart returns start_z of the z split -- the reference indexes start_x there, a bug) */
int dfft_get_in_size(const dfft_plan *plan, size_t size[3]);
int dfft_get_in_start(const dfft_plan *plan, size_t start[3]);
int dfft_get_out_size(const dfft_plan *plan, size_t size[3]);
int dfft_get_out_start(const dfft_plan *plan, size_t start[3]);
/* getPartitionDimensions(input_dim, transposed_dim, output_dim)   include/mpicufft_pencil.hpp:112-116, tables of
 * struct Partition_Dimensions (include/params.hpp:58-81) as built in src/pencil/mpicufft_pencil_opt1.cpp:70-93.
 * which = 0 input_dim, 1 transposed_dim, 2 output_dim; axis = 0 x, 1 y, 2 z.  Writes the per-rank extents and
 * offsets of that axis at that stage (one entry = the full extent when the axis is not split there); *count
 * receives the number of entries, at most `capacity` are written (sizes/starts may be NULL to query). */
/* element strides of the spectrum block in `out` along (kx, ky_local, kz_local): entry (kx, ky, kz) of the block dfft_get_out_size
 * describes lives at kx * s[0] + ky * s[1] + kz * s[2].  {yo * zs, zs, 1} for the reference layout, {1, zs * Nx, Nx} with option
 * "spectral_layout" = 1. */
int dfft_get_out_strides(const dfft_plan *plan, size_t s[3]);
int dfft_get_partition_dimensions(const dfft_plan *plan, int which, int axis, size_t *sizes, size_t *starts,
                                  size_t capacity, size_t *count);
/* getDomainSize / getWorkSizeDevice / getWorkSizeHost / getRank / getWorldSize
 * include/mpicufft.hpp:65-78 */
size_t dfft_domain_size(const dfft_plan *plan);        /* bytes `out` must hold */
size_t dfft_work_size_device(const dfft_plan *plan);
size_t dfft_work_size_host(const dfft_plan *plan);
void *dfft_work_area_device(const dfft_plan *plan);
int dfft_rank(const dfft_plan *plan);
int dfft_world_size(const dfft_plan *plan);

/* exchange tables in BYTES exactly as the reference builds them for MPI_Alltoallv
 * (mpicufft_pencil_opt1.cpp:269-273 which=1, :315-319 which=2).  Arrays hold P2 resp. P1 entries. */
int dfft_get_exchange_tables(const dfft_plan *plan, int which, size_t *sendcounts, size_t *sdispls,
                             size_t *recvcounts, size_t *rdispls);
/* run only exchange `which` (1 = row group, 2 = column group) of the plan on caller buffers,
 * forward (send tables) or inverse (tables swapped) direction, blocking.  This is the
 * MPI_Alltoallv step of the reference in isolation (:784-785 / :829-830, :1297-1298 / :1341-1342);
 * used by the transport tests. */
int dfft_exchange(dfft_plan *plan, int which, int direction, const void *sendbuf, void *recvbuf);
/* the tables the pipelined exchanges actually use: chunk `chunk` (0 .. dfft_get_pipeline_chunks-1) of
 * exchange `which`, for the forward or the inverse transform, in send/recv order of that direction
 * (bytes; displacements are absolute offsets in the send resp. receive buffer).  Over all chunks
 * the counts add up to dfft_get_exchange_tables(). */
int dfft_get_pipeline_tables(const dfft_plan *plan, int direction, int which, int chunk, size_t *sendcounts,
                             size_t *sdispls, size_t *recvcounts, size_t *rdispls);
/* lines interleaved per tile in the intermediate layouts */
int dfft_tile_lines(const dfft_plan *plan);

/* per-phase device time of the last exec in milliseconds (Timer sections of the reference,
 * include/mpicufft_pencil.hpp:263-287): up to 8 entries, names via dfft_phase_name */
int dfft_get_phase_times(dfft_plan *plan, float *ms, int max_entries);
const char *dfft_phase_name(int phase, int direction);
int dfft_enable_phase_timing(dfft_plan *plan, int enable);

/* standalone launch of one batched 1-D axis pass on natural lines (in [batch][N] -> out
 * [batch][N]); used by the kernel-level parity tests and the micro-benchmarks */
int dfft_fft1d_batched(int precision, size_t N, size_t batch, void *out, const void *in,
                       int direction, void *hip_stream);

/* same with an explicit kernel configuration (0 = default; unknown values fall back to 0; -1 = the Bluestein kernel
 * even where the length has a native configuration) and the measurement-only debug flags of "debug_skip" */
int dfft_fft1d_batched_ex(int precision, size_t N, size_t batch, void *out, const void *in, int direction,
                          void *hip_stream, int variant, int debug);

/* ---- introspection of the pass descriptors (host only; used by the CPU layout tests) -------- *
 * One axis pass of the plan: which lines it transforms and the address forms of its load and store
 * side (enum values as in distributedfft_amd/csrc/fft_pass.hip.h).  Offsets
 * are in ELEMENTS of the side's type except in_off / out_off (bytes added to the stage buffer). */
typedef struct dfft_pass_desc {
    uint32_t na, LB, nb, LA, T2shift;
    int32_t load_kind;      /* 0 natural lines, 1 tiled (segments), 2 point-major (KS_in, AS_in) */
    int32_t store_kind;     /* 0 lines (row strides if KS_out != 0), 1 point-major, 2 tiled-same, 3 tiled-transpose */
    int32_t swap, shift;
    uint64_t KS_in, KS_out, AS_in, AS_out;      /* load_kind 0 with KS_in != 0: rows a*AS_in + line*KS_in (strided natural lines) */
    uint64_t IA, IB;        /* tiled load, one segment: explicit strides of a and of a tile along b (0 = packed) */
    uint64_t SK, SB;        /* tiled-same store: explicit strides of a point k and of a tile along b (0 = packed) */
    int32_t a_fastest, xcd_swizzle;
    uint64_t in_off, out_off;
    int32_t lnseg, snseg;
    uint32_t lstart[32], llen[32];
    uint64_t lbase[32];
    uint32_t sstart[32], slen[32];
    uint64_t sbase[32];
} dfft_pass_desc;
/* What every pass of the plan runs with right now -- the rules of dfft_init, or what dfft_tune_variants / dfft_tune_placement kept:
 * passes in the order fz fy fx ix iy iz; variant = kernel configuration number (the role numbers of csrc/cfg_f64.hip.h / cfg_f32.hip.h),
 * order = a_fastest + 2 * xcd_swizzle of the workgroup -> tile mapping, addr64 = 1 where the pass keeps per-point 64-bit vector
 * addresses.  Any of the arrays may be NULL.  (A caller can pin the same choices on another plan with the "variant_<pass>" /
 * "order_<pass>" options.) */
int dfft_get_pass_choices(const dfft_plan *plan, int variant[6], int order[6], int addr64[6]);
/* name: "fz" "fy" "ix" "iy" "iz" "py2" "qy2" "zy" "ziy" (index = chunk, or chunk*P + peer for zy/ziy)
 * and "fx" "zix" "yz" "pz1" "qz1" "sz" "sx" "sy" (index 0; the last three: single-rank complex plans, order z, x, y).  Returns nonzero if the plan has no such launch. */
int dfft_debug_get_pass(const dfft_plan *plan, const char *name, int index, dfft_pass_desc *desc);
/* the per-point address table the kernels use for a segmented side of that launch (store = 0: load
 * side, 1: store side); entry i = {base[i], ln[i], aux[i]} as documented for SegEntry.  *count receives
 * the number of points; at most `capacity` entries are written. */
int dfft_debug_get_point_table(const dfft_plan *plan, const char *name, int index, int store, uint64_t *base,
                               uint32_t *ln, uint32_t *aux, size_t capacity, size_t *count);

const char *dfft_last_error(void);
const char *dfft_version(void);
/* kernel configuration for line length N: returns 0 if supported and fills the fields */
int dfft_kernel_info(int precision, size_t N, int *threads, int *lds_bytes, int *points_per_thread,
                     int *lines_per_workgroup);
/* how an axis of N points is transformed (the plan cufftMakePlanMany64 hides, mpicufft_pencil_opt1.cpp:165-197): returns 0 and fills
 * info[0] = 0 native chain / 1 Bluestein / 2 two levels (N = N1*N2 over two launches) / 3 long Bluestein (a length that neither fits
 * one launch nor splits, e.g. a prime above 4096: Bluestein's algorithm on info[1] padded points whose transforms are two-level lines;
 * four launches), info[1] = inner power-of-two length (0 for two levels), and for two levels / long Bluestein info[2..4] = {N1, its
 * inner length, 1 if Bluestein} and info[5..7] likewise for N2 (the levels of the padded length in the long form).
 * two_level = 1: two levels wherever N splits (the plan option of the same name).  ERR_UNSUPPORTED: no plan (beyond 2^23 points). */
int dfft_axis_plan_info(int precision, size_t N, int two_level, size_t info[8]);

/* ---- placement-aware device memory (no counterpart in the reference, whose buffers are plain cudaMalloc,
 * src/pencil/mpicufft_pencil_opt1.cpp:344-365; the analogue of fftw_malloc + FFTW_MEASURE) ----------------
 * The passes that write 128-byte runs (y, x) run 5-10 % faster or slower depending on the PHYSICAL backing of the
 * buffer they scatter to -- per buffer, for the life of the allocation (profiles/r3_placement.txt).
 * dfft_malloc: chunk_mib = 0 is hipMalloc; otherwise one virtual range backed by physical allocations of chunk_mib MiB
 * each (HIP virtual-memory API).  chunk_mib = DFFT_CHUNK_DEFAULT: the library's default backing, which is also what the library
 * uses for a work area it owns (dfft_init(allocate = 1), dfft_set_work_area(plan, NULL, NULL)): 1 GiB chunks (smaller ones and
 * finally hipMalloc if that fails) and, for buffers of 1 GiB and more, PLACEMENT.  Such a buffer is built from chunks that lie far
 * apart (every K-th of K times as many, K = 5 unless DFFT_PLACEMENT_SPREAD says otherwise or memory is short) and a streaming write is
 * timed on it (8 ms per 16 GiB): the good class by construction; the first one of a process gives the device its yardstick, and a
 * built buffer at >= 0.95 x the yardstick is kept.  Where there is no room for the pool, or the built buffer falls short,
 * plain candidates are drawn, up to DFFT_PLACEMENT_TRIES (6), all alive, and the fastest of everything probed is kept.  Nothing is absolute: the
 * yardstick is measured on the device at hand (MI355X: built buffers 6.5 - 7.0 TB/s, bad ones 5.2 - 5.8, a contiguous hipMalloc
 * reference 4.5 - 4.6; the plan's scatter passes follow: 5.5 vs 5.9 - 6.5 ms per pass at 1024^3 fp64, profiles/r4_placement_probe.txt,
 * profiles/r5_allocator.txt); DFFT_PLACEMENT_GOOD_TBPS sets an absolute threshold.  Everything alive during the search stays within
 * half of the free memory divided by DFFT_RANKS_PER_DEVICE (set it when several processes share a GPU), and whatever fails on the way
 * falls back to the plain recipe and then to hipMalloc: the call never fails where hipMalloc would succeed.  1024^3 fp64 forward +
 * inverse on buffers from this call: 33.5 - 33.6 ms, on hipMalloc buffers 37.0 - 38.3.  Local to the device (no plan, no collective):
 * safe on every rank of a multi-rank job.  Costs 2 - 4 s per 16 GiB buffer (the driver clears fresh memory inside hipMemCreate at
 * ~30 ms per GiB: tools/vmm_cycle), 0.3 s per 2 GiB; DFFT_PLACEMENT_SPREAD=1: milliseconds (plain candidates only, no yardstick).  Environment: DFFT_DEFAULT_CHUNK_MIB
 * (0 = hipMalloc), DFFT_PLACEMENT_TRIES (1 = no probe), DFFT_PLACEMENT_SPREAD, DFFT_RANKS_PER_DEVICE, DFFT_PLACEMENT_GOOD_TBPS.
 * Addresses: on this ROCm a virtual address that is mapped a SECOND time (after hipMemAddressFree and a later reservation at the
 * same address, or fresh chunks mapped into a kept reservation) reads and writes wrong bytes -- tools/vmm_reuse_repro.hip reproduces
 * it without this library, with kernels and with runtime copies, in 2 - 90 map / unmap cycles (profiles/r6_vmm_reuse_repro.txt) --
 * and a range that is unmapped but not freed keeps its physical memory (profiles/r6_vmm_cost.txt).  So dfft_free unmaps AND returns
 * the range, and dfft_malloc reserves every range at an address it names itself and never names twice: the library walks once
 * through [DFFT_VMM_BASE_TIB = 4 TiB, 80 TiB) of the process's address space (profiles/r6_vmm_hint.txt: 10^4 cycles, device memory
 * level, not one wrong byte).  When that region is used up the default backing is hipMalloc.
 * dfft_last_placement_info writes what the last placement-aware allocation of the process did (a JSON object: K, candidates drawn,
 * probe / reference / threshold rates, seconds by phase, what was kept).  Free with dfft_free (which also takes pointers it did not
 * allocate: hipFree; it synchronises the owning device first, like hipFree). */
#define DFFT_CHUNK_DEFAULT ((size_t)-1)
int dfft_malloc(size_t bytes, size_t chunk_mib, void **ptr);
int dfft_free(void *ptr);
int dfft_last_placement_info(char *buf, size_t capacity);
/* Tries up to `tries` backings for the plan's own work area (only when the library owns it), for a new output buffer
 * *out (dfft_domain_size bytes) and, if back != NULL, for a new buffer *back of the input block's size (the inverse
 * transform's output), one buffer at a time, and keeps for each the backing on which the plan's own FFT passes
 * (forward in -> out, inverse out -> back; exchanges not counted) run fastest.  `in` must hold a valid input block; it
 * is only read.  Afterwards the y / x passes try their streaming (nontemporal) kernel configuration on the chosen buffers and
 * keep it where it measures faster.  *out / *back are freed with dfft_free.  On a multi-rank plan the call is collective and
 * does NOT search (how many candidates fit is a per-rank matter: ranks running different numbers of trials would strand each
 * other in an exchange): out / back come from the default backing and only the configuration trials run, whose number depends
 * on the global grid alone.  report_ms (optional): the measured pass time of every trial in order, *n_report
 * entries. */
int dfft_tune_placement(dfft_plan *plan, const void *in, int tries, void **out, void **back, float *report_ms,
                        int max_report, int *n_report);
/* The second half of dfft_tune_placement on the caller's own buffers (no allocation).  First every pass tries the four
 * workgroup -> tile orders, then every kernel configuration its line length has (streaming siblings, other lane mappings and
 * tile shapes: the role variants of csrc/cfg_*.hip.h).  A trial sets all passes at once and reads the per-pass times from the
 * phase timers, so each pass picks for itself (1 % threshold) from 4 + (number of configuration numbers) trials; passes the
 * caller pinned with order_* / variant_* keep their setting, Bluestein / two-level axes and the slab sequences are not tuned.
 * A trial executes the plan three times forward in -> out and, if back != NULL, inverse out -> back, which destroys `out` like
 * every inverse.  Collective on a multi-rank plan: which trials run depends on the global grid only, every rank decides for its
 * own kernels.  report_ms: the plan as built, the four order settings, the chosen orders, one entry per configuration number,
 * the final choice. */
int dfft_tune_variants(dfft_plan *plan, const void *in, void *out, void *back, float *report_ms, int max_report, int *n_report);

#ifdef __cplusplus
}
#endif
#endif /* DFFT_C_H */
