// plan.hpp -- the plan object of libdfft_amd.so and what its host translation units share: dfft.hip (axis plans, execution chains,
// tuners, C ABI) and pipeline.hip (the pass descriptors and exchange tables of every decomposition).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/dfft_c.h"
#include "comm.hpp"
#include "dfft_internal.hpp"
#include "fft_pass.hip.h"

namespace dfft {

// remainder to the lowest ranks (mpicufft_pencil_opt1.cpp:71-73)
inline void split(size_t n, int p, std::vector<size_t> &size, std::vector<size_t> &start)
{
    size.assign(p, n / p);
    start.assign(p, 0);
    for (size_t i = 0; i < n % p; i++) size[i]++;
    size_t off = 0;
    for (int i = 0; i < p; i++) { start[i] = off; off += size[i]; }
}

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline size_t next_pow2(size_t n) { size_t m = 1; while (m < n) m <<= 1; return m; }

// One axis of the 3-D transform: native Stockham chain, Bluestein on top of it, or two levels of either (N = N1*N2)
struct Axis {
    size_t N = 0;          // line length
    bool bluestein = false;    // the generic kernel (fft_bluestein_kernel) runs the pass: Bluestein, or the levels of a two-level line
    size_t M = 0;          // inner power-of-two length (== N when native)
    void *tw = nullptr;    // exp(-2 pi i j / M), M entries
    void *chirp = nullptr; // Bluestein: exp(-i pi n^2 / N), N entries
    void *bhat = nullptr;  // Bluestein: FFT_M(conj chirp, wrapped) / M
    // two-level line (fft_pass.hip.h, "Two-level lines"): lv[0] transforms N1 = lv[0].N points, lv[1] N2 = lv[1].N; a level is
    // the plain power-of-two chain (bluestein == false, M == N) or Bluestein on an arbitrary factor
    bool two = false;
    // long Bluestein line (a length that neither fits one launch nor splits into two such factors: a prime above 4096, twice such a
    // prime ...): Bluestein's algorithm whose M-point transforms are two-level lines; lv[0] is that two-level plan of M, chirp / bhat as
    // for the one-launch form.  Four launches of the generic kernel (launch_long_bluestein).
    bool longb = false;
    std::vector<Axis> lv;
    void *twN = nullptr;   // exp(-2 pi i j / N), N entries: twiddles between the levels
};

}  // namespace dfft

struct Launch {
    dfft::PassArgs args{};          // in/out/tw and the device table pointers are filled at enqueue time (zeroed: a plan builds only the launches of its kind)
    dfft::SegTable lseg{}, sseg{};  // host copies of the segment tables (uploaded by upload_tables)
    size_t ltab = 0, stab = 0;   // byte offsets of the tables in the plan's device table buffer
    size_t lent = SIZE_MAX, sent = SIZE_MAX;   // byte offsets of the per-point address tables (SIZE_MAX: none)
    size_t in_off = 0;        // byte offset added to the stage's input buffer
    size_t out_off = 0;       // byte offset added to the stage's output buffer
};
struct A2A {
    std::vector<size_t> sc, sd, rc, rd;   // bytes, absolute displacements in the stage buffers
};

struct Pipeline {
    int C = 1;
    std::vector<Launch> fz, fy, ix, iy, iz;   // per chunk
    Launch fx;                                 // forward x pass (needs complete lines)
    // partial transforms (reference: execR2C/C2R(out, in, d), src/pencil/mpicufft_pencil.cpp:1644-1839)
    Launch pz1, qz1;                           // d = 1: z pass natural -> natural [xs][ys][Nzc] and back
    std::vector<Launch> py2, qy2;              // d = 2: y pass chunk -> [xs][Ny][zs] and back
    // slab sequence Z_Then_YX (src/slab/z_then_yx/): y passes per (chunk, source peer) block and one
    // unchunked inverse x pass; the exchange tables live in f2 / i2
    std::vector<Launch> zy, ziy;
    Launch zix;
    Launch yz;                                 // Y_Then_ZX: final z pass (x pass = fx, y chunks = fy)
    // single-rank complex plans, pass order z, x, y (build_pipeline_single): natural lines -> L1 -> L2 -> natural
    Launch sz, sx, sy;
    bool single = false;
    size_t single_work_elems = 0;              // size of the padded L2 buffer
    std::vector<A2A> f1, f2, i2, i1;          // per chunk exchange tables
    std::vector<hipEvent_t> ev;               // reusable events
    hipStream_t comm_stream = nullptr;
    hipStream_t comm_stream2 = nullptr;       // second exchange of a pencil plan (disjoint links: may overlap the first)
    hipStream_t compute_stream2 = nullptr;    // option compute_streams = 2: the odd pipeline chunks of a pass run here (enqueue_forward)
};

struct TimedSpan { hipEvent_t a = nullptr, b = nullptr; int phase = 0; bool used = false; };

// Tuning knobs of a plan (dfft_set_option; a few have environment defaults read once at plan creation).
// Nothing here is read on the execution path.
struct Options {
    int chunks = 0;          // requested pipeline depth (0 = default)
    int mirror = 0;          // single-rank complex inverse in the mirrored (multi-rank) pass order x, y, z
    int tables = 1;          // per-point address tables: 0 never, 1 sides with more than one segment, 2 always
    int uniform_tables = 1;  // sides whose segments all start at multiples of 16 points: table entries through wave-uniform
                             // (scalar) loads, dfft::PassArgs::luni / suni (0: per-lane vector loads, for A/B runs)
    int shift = -1;          // row-aligned tile windows of odd-pitch point-major stores: -1 auto (fp64), 0 off, 2 always
    int debug = 0;           // dfft::PassArgs::debug of every launch (measurement only; results are wrong when set)
    int real_variant = 0;    // A/B configurations of the real z passes (DFFT_EXPERIMENTS builds)
    int single_order = -1;   // single-rank complex plans: 1 = pass order z, x, y with padded private layouts, 0 = z, y, x,
                             // -1 = by measurement: z, x, y where it won (fp32 with x and y lines of 2048 points or more)
    int single_layout = 1;   // L2 of the z, x, y order: 0 = [kx][kz/TL][y][l], 1 = tile-outer [kz/TL][kx][y][l]
    int single_pad = 128;    // bytes added to every L2 row (a row stride that is an odd multiple of 128 B; 0 = packed)
    int graph = 0;           // 1: single-rank plans replay the launches of an exec as one hipGraph from the second call with the same
                             // buffers on.  Off by default -- measured (profiles/r2_graph_latency.txt): a blocking 64^3 R2C takes 28 us
                             // with three plain launches and 34 us as a graph, 128^3 54 vs 60 us; the launches are already hidden
                             // behind the first kernel (128^3: 50 us of kernels in a 54 us call)
    int native_mixed = 1;    // lengths 2^a 3^b 5^c 7^d with a configuration run the native chain (0: Bluestein, for A/B runs and tests)
    int two_level = 0;       // 1: every axis whose length splits as N1*N2 runs as a two-level line (tests, A/B runs; 0: only lengths
                             // that have no other plan)
    int spectral = 0;        // 1: the spectrum is kept x-contiguous, [yo][zs][Nx] (lines along kx natural), instead of the reference's
                             // [Nx][yo][zs]: the forward x pass stores natural lines and the inverse x pass loads them -- neither
                             // touches the point-major layout whose strided read is the slowest pass of every multi-rank plan
    int compute_streams = -1; // 2: the pipeline chunks of a pass alternate over two compute streams, so that the drain of chunk c
                             // overlaps the ramp of chunk c + 1 (a chunk launch of 0.1-0.2 ms pays ~20 us of launch / drain / ramp when
                             // the chunks queue up behind each other on one stream).  -1 = by measurement
                             // (profiles/r6_compute_streams.txt): two streams from three chunks per pass on (rank 0 of 2x4, 1024^3
                             // fp64: 4 chunks 4.89 -> 4.78 ms, 8 chunks 5.31 -> 4.90; at two chunks there is nothing to gain), 1 = one
    int order[6] = {-1, -1, -1, -1, -1, -1};     // workgroup->tile order per pass: fz fy fx ix iy iz; a_fastest + 2*xcd_swizzle
    int variant[6] = {-1, -1, -1, -1, -1, -1};   // kernel configuration per pass, same order (-1 = the plan's choice)
};

struct dfft_plan {
    int kind = DFFT_PENCIL_OPT1, prec = DFFT_F64;
    dfft_config cfg{};
    dfft_comm *comm = nullptr;
    int rank = 0, nranks = 1;
    bool initialized = false, c2c = false;
    bool spectral_mirror = false;   // one rank with an x-contiguous spectrum (option spectral_layout): the inverse runs the mirrored pass order
    bool zyx = false;            // slab sequence Z_Then_YX: input split along x, output split along z
    bool yzx = false;            // slab sequence Y_Then_ZX: R2C along y, output [Nx][(Ny/2+1)/P][Nz], forward only
    size_t Nyc = 0;              // y extent of the spectrum (Ny/2+1 for a Y_Then_ZX R2C plan, else Ny)
    size_t Nx = 0, Ny = 0, Nz = 0, Nzc = 0;
    int P1 = 1, P2 = 1, pi = 0, pj = 0;
    int TL = 8;
    std::vector<size_t> xs, xstart, ys, ystart, zs, zstart, yo, yostart;
    size_t esz = 16, domain_elems = 0, domainsize = 0, worksize_d = 0;
    void *work_d = nullptr;
    bool work_owned = false;
    dfft::Axis ax[3];                  // [0] = z, [1] = y, [2] = x
    bool zreal_native = false;   // R2C plan whose z axis uses the packed Nz/2-point kernels
    bool yreal_native = false;   // Y_Then_ZX R2C plan whose y axis uses the packed Ny/2-point kernel (strided real lines)
    size_t lv_off = 0, lv_bytes = 0;   // two-level axes: scratch between the levels, a region of the work area (behind the exchange slices)
    void *tw_zr = nullptr;       // split/merge table exp(-2 pi i k / Nz) (or / Ny) of the packed real kernels
    void *tables_d = nullptr;    // segment tables of every launch, device copy
    hipStream_t stream = nullptr;
    bool stream_owned = false;
    bool stream_user = false;    // caller chose the stream (the null stream is a valid choice)
    // exchange tables in bytes (row comm = 1, column comm = 2) and member lists
    std::vector<size_t> sc1, sd1, rc1, rd1, sc2, sd2, rc2, rd2;
    std::vector<int> group1, group2;
    dfft::RelayCache *relay = nullptr;      // two-hop relay of the group exchanges (dfft_comm_set_option "relay"): gathered world tables, staging
    int vfwd[3] = {0, 0, 0}, vinv[3] = {0, 0, 0};   // kernel variant per pass: [0]=z [1]=y [2]=x
    Options opt;
    Pipeline pl;
    // phase timing: (start, stop) event pairs, phases 0..4 = z, exchange 1, y, exchange 2, x
    bool timing = false;
    std::vector<TimedSpan> spans;
    size_t nspans = 0;
    int last_dir = -1;
    // hipGraph replay of single-rank execs (launch-bound small grids): one instantiated graph per (operation, in, out)
    struct GraphEntry { int kind; const void *in; void *out; int uses; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
};

// pass descriptors and exchange tables of a plan (pipeline.hip): the default sequence (pencil and slab ZY_Then_X), the slab sequences
// Z_Then_YX and Y_Then_ZX, and the single-rank order z, x, y.  Return 0 or an error code (message: dfft_last_error).
int build_pipeline(dfft_plan *p, Pipeline &pl);
int build_pipeline_zyx(dfft_plan *p, Pipeline &pl);
int build_pipeline_yzx(dfft_plan *p, Pipeline &pl);
int build_pipeline_single(dfft_plan *p, Pipeline &pl);
// shared by dfft.hip and tune.hip
int check_ready(dfft_plan *p);       // 0 when the plan is initialised and has its device state, else an error code
void graphs_clear(dfft_plan *p);     // drops the captured launch graphs (anything that changes the launches calls it)
