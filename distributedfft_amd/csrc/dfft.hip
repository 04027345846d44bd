// dfft.hip -- plan object, pass descriptors, exec chains and the C ABI of libdfft_amd.so.
//
// Host-side counterpart of the reference's decomposition classes; each block cites what it
// replaces (paths relative to the reference repository):
//   ctor / comm handling     src/mpicufft.cpp:42-66
//   initFFT                  src/pencil/mpicufft_pencil_opt1.cpp:46-326
//   setWorkArea              src/pencil/mpicufft_pencil_opt1.cpp:329-387
//   execR2C / execC2R        src/pencil/mpicufft_pencil_opt1.cpp:1422-1519 / 1522-1600
//   slab (P2 == 1)           src/slab/default/mpicufft_slab_opt1.cpp:38-178, 683-783
//   single rank (fft3d)      src/pencil/mpicufft_pencil_opt1.cpp:132-135, 1434-1436
#include "../../include/dfft_c.h"
#include "comm.hpp"
#include "dfft_internal.hpp"
#include "fft_pass.hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

namespace dfft {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return (int)e_;                                                                    \
        }                                                                                      \
    } while (0)
#define TRY(expr)                                                                              \
    do {                                                                                       \
        int r_ = (expr);                                                                       \
        if (r_ != 0) return r_;                                                                \
    } while (0)

enum { ERR_ARG = 2, ERR_STATE = 3, ERR_UNSUPPORTED = 4 };

static int fail(int code, const std::string &msg)
{
    set_error(msg);
    return code;
}

// remainder to the lowest ranks (mpicufft_pencil_opt1.cpp:71-73)
static void split(size_t n, int p, std::vector<size_t> &size, std::vector<size_t> &start)
{
    size.assign(p, n / p);
    start.assign(p, 0);
    for (size_t i = 0; i < n % p; i++) size[i]++;
    size_t off = 0;
    for (int i = 0; i < p; i++) { start[i] = off; off += size[i]; }
}

static int launch_pass(int prec, int N, int variant, const PassArgs &A, hipStream_t s)
{
    PassInfo pi;
    if (variant && !(prec == DFFT_F64 ? pass_info_f64(N, variant, &pi) : pass_info_f32(N, variant, &pi))) variant = 0;
    int r = prec == DFFT_F64 ? launch_pass_f64(N, variant, A, s) : launch_pass_f32(N, variant, A, s);
    if (r == -1) return fail(ERR_UNSUPPORTED, "unsupported line length " + std::to_string(N));
    if (r != 0) return fail(r, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)r));
    return 0;
}
static bool pass_info(int prec, int N, PassInfo *pi)
{
    return prec == DFFT_F64 ? pass_info_f64(N, 0, pi) : pass_info_f32(N, 0, pi);
}

// twiddle table exp(-2*pi*i*j/N), evaluated in long double, rounded once
static int make_twiddles(int prec, size_t N, void **dev)
{
    const long double PI = 3.141592653589793238462643383279502884L;
    const size_t esz = prec == DFFT_F64 ? 16 : 8;
    std::vector<char> host(esz * N);
    for (size_t j = 0; j < N; j++) {
        long double a = -2.0L * PI * (long double)j / (long double)N;
        if (prec == DFFT_F64) {
            double *d = reinterpret_cast<double *>(host.data()) + 2 * j;
            d[0] = (double)cosl(a); d[1] = (double)sinl(a);
        } else {
            float *d = reinterpret_cast<float *>(host.data()) + 2 * j;
            d[0] = (float)cosl(a); d[1] = (float)sinl(a);
        }
    }
    HIP_TRY(hipMalloc(dev, esz * N));
    HIP_TRY(hipMemcpy(*dev, host.data(), esz * N, hipMemcpyHostToDevice));
    return 0;
}

}  // namespace dfft

using namespace dfft;

struct dfft_plan {
    int kind = DFFT_PENCIL_OPT1, prec = DFFT_F64;
    dfft_config cfg{};
    dfft_comm *comm = nullptr;
    int rank = 0, nranks = 1;
    bool initialized = false, c2c = false;
    size_t Nx = 0, Ny = 0, Nz = 0, Nzc = 0;
    int P1 = 1, P2 = 1, pi = 0, pj = 0;
    int TL = 8;
    std::vector<size_t> xs, xstart, ys, ystart, zs, zstart, yo, yostart;
    size_t esz = 16, domain_elems = 0, domainsize = 0, worksize_d = 0;
    void *work_d = nullptr;
    bool work_owned = false;
    void *tw_x = nullptr, *tw_y = nullptr, *tw_z = nullptr, *tw_zr = nullptr;   // tw_zr: split/merge table (R2C)
    hipStream_t stream = nullptr;
    bool stream_owned = false;
    bool stream_user = false;    // caller chose the stream (the null stream is a valid choice)
    // exchange tables in bytes (row comm = 1, column comm = 2) and member lists
    std::vector<size_t> sc1, sd1, rc1, rd1, sc2, sd2, rc2, rd2;
    std::vector<int> group1, group2;
    // pass descriptors without buffer pointers: [0]=z [1]=y [2]=x
    PassArgs fwd[3], inv[3];
    int vfwd[3] = {0, 0, 0}, vinv[3] = {0, 0, 0};   // kernel variant per pass
    // phase timing
    bool timing = false;
    hipEvent_t ev[8] = {};
    int nev = 0, last_dir = -1;
};

static void seg_from(SegTable &t, const std::vector<size_t> &start, const std::vector<size_t> &len,
                     const std::vector<size_t> &base_elems)
{
    t.nseg = (int)start.size();
    for (int s = 0; s < t.nseg; s++) {
        t.start[s] = (uint32_t)start[s];
        t.len[s] = (uint32_t)len[s];
        t.base[s] = base_elems[s];
    }
}

static int build_passes(dfft_plan *p)
{
    const int TL = p->TL;
    const uint32_t T2shift = ilog2(TL);
    const size_t xs = p->xs[p->pi], ys = p->ys[p->pj], zs = p->zs[p->pj], yo = p->yo[p->pi];
    const size_t e = p->esz;
    auto elems = [&](const std::vector<size_t> &bytes) {
        std::vector<size_t> v(bytes.size());
        for (size_t i = 0; i < bytes.size(); i++) v[i] = bytes[i] / e;
        return v;
    };
    const std::vector<size_t> sd1 = elems(p->sd1), rd1 = elems(p->rd1), sd2 = elems(p->sd2), rd2 = elems(p->rd2);
    auto base = [&](uint32_t na, uint32_t LB, int lk, int sk, int swap) {
        PassArgs A;
        memset(&A, 0, sizeof(A));
        A.na = na; A.LB = LB; A.nb = (LB + TL - 1) / TL; A.ntiles = A.na * A.nb;
        A.load_kind = lk; A.store_kind = sk; A.swap = swap; A.T2shift = T2shift;
        return A;
    };
    // ---------------- forward (mpicufft_pencil_opt1.cpp:1422-1519) ----------------
    {   // z pass: lines along z from the caller's [xs][ys][Nz]; tiles of TL adjacent y.
        // Output = send buffer of exchange 1: block p = [x][kz/TL][y][kz%TL] for kz in zs[p].
        PassArgs A = base((uint32_t)xs, (uint32_t)ys, LOAD_LINES, STORE_TILED_TRANSPOSE, 0);
        seg_from(A.sseg, p->zstart, p->zs, sd1);
        p->fwd[0] = A;
    }
    {   // y pass: lines along y gathered from the P2 received blocks; tiles of TL adjacent kz.
        // Output = send buffer of exchange 2: block p = [ky][kz/TL][x][kz%TL] for ky in yo[p].
        PassArgs A = base((uint32_t)xs, (uint32_t)zs, LOAD_TILED, STORE_TILED_SAME, 0);
        seg_from(A.lseg, p->ystart, p->ys, rd1);
        seg_from(A.sseg, p->yostart, p->yo, sd2);
        A.LA = (uint32_t)xs;
        p->fwd[1] = A;
    }
    {   // x pass: lines along x gathered from the P1 received blocks; writes the API output
        // [kx][y'][z'] (include/mpicufft_pencil.hpp:119-122).
        PassArgs A = base((uint32_t)yo, (uint32_t)zs, LOAD_TILED, STORE_KMAJOR, 0);
        seg_from(A.lseg, p->xstart, p->xs, rd2);
        A.KS_out = (uint64_t)yo * zs;
        p->fwd[2] = A;
    }
    // ---------------- inverse (mpicufft_pencil_opt1.cpp:1522-1600) ----------------
    {   // x^-1: reads the API output layout point-major; block p = [x][kz/TL][ky][kz%TL], x in xs[p]
        PassArgs A = base((uint32_t)yo, (uint32_t)zs, LOAD_KMAJOR, STORE_TILED_SAME, 1);
        A.KS_in = (uint64_t)yo * zs;
        seg_from(A.sseg, p->xstart, p->xs, rd2);
        A.LA = (uint32_t)yo;
        p->inv[2] = A;
    }
    {   // y^-1: lines along ky from the P1 blocks; block p = [x][y/TL][kz][y%TL], y in ys[p]
        PassArgs A = base((uint32_t)xs, (uint32_t)zs, LOAD_TILED, STORE_TILED_TRANSPOSE, 1);
        seg_from(A.lseg, p->yostart, p->yo, sd2);
        seg_from(A.sseg, p->ystart, p->ys, rd1);
        p->inv[1] = A;
    }
    {   // z^-1: lines along kz from the P2 blocks; writes natural [xs][ys][Nz]
        PassArgs A = base((uint32_t)xs, (uint32_t)ys, LOAD_TILED, STORE_LINES, 1);
        seg_from(A.lseg, p->zstart, p->zs, sd1);
        p->inv[0] = A;
    }
    return 0;
}

static int run_pass(dfft_plan *p, const PassArgs &tmpl, int variant, size_t N, const void *tw, const void *in, void *out)
{
    PassArgs A = tmpl;
    A.in = in; A.out = out; A.tw = tw;
    return launch_pass(p->prec, (int)N, variant, A, p->stream);
}

// z pass of an R2C plan: M = Nz/2 point complex FFT + split (mode 1) / merge (mode 2)
static int run_real_pass(dfft_plan *p, const PassArgs &tmpl, int mode, const void *in, void *out)
{
    PassArgs A = tmpl;
    A.in = in; A.out = out; A.tw = p->tw_z; A.tw2 = p->tw_zr;
    const int M = (int)(p->Nz / 2);
    int r = p->prec == DFFT_F64 ? launch_real_f64(M, mode, A, p->stream) : launch_real_f32(M, mode, A, p->stream);
    if (r == -1) return fail(ERR_UNSUPPORTED, "unsupported real line length " + std::to_string(p->Nz));
    if (r != 0) return fail(r, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)r));
    return 0;
}

static int mark(dfft_plan *p)
{
    if (!p->timing) return 0;
    if (p->nev >= 8) return 0;
    if (!p->ev[p->nev]) HIP_TRY(hipEventCreate(&p->ev[p->nev]));
    HIP_TRY(hipEventRecord(p->ev[p->nev], p->stream));
    p->nev++;
    return 0;
}

static int exchange(dfft_plan *p, int which, bool forward, const void *send, void *recv)
{
    const bool first = which == 1;
    const std::vector<int> &grp = first ? p->group1 : p->group2;
    const int me = first ? p->pj : p->pi;
    const std::vector<size_t> &sc = first ? p->sc1 : p->sc2, &sd = first ? p->sd1 : p->sd2;
    const std::vector<size_t> &rc = first ? p->rc1 : p->rc2, &rd = first ? p->rd1 : p->rd2;
    if (!p->comm) return fail(ERR_STATE, "exchange without a communicator");
    // the inverse all-to-all swaps the send and receive tables (mpicufft_pencil_opt1.cpp:829-830)
    if (forward)
        return p->comm->alltoallv(p->rank, send, sc.data(), sd.data(), recv, rc.data(), rd.data(), grp.data(),
                                  (int)grp.size(), me, p->stream);
    return p->comm->alltoallv(p->rank, send, rc.data(), rd.data(), recv, sc.data(), sd.data(), grp.data(),
                              (int)grp.size(), me, p->stream);
}

// forward chain, complex input.  Buffers: A = caller's out, B/C = work area halves.
static int enqueue_forward(dfft_plan *p, void *out, const void *in)
{
    char *A = static_cast<char *>(out), *B = static_cast<char *>(p->work_d), *C = B + p->domainsize;
    p->nev = 0; p->last_dir = DFFT_FORWARD;
    TRY(mark(p));
    if (p->c2c) TRY(run_pass(p, p->fwd[0], p->vfwd[0], p->Nz, p->tw_z, in, A));
    else TRY(run_real_pass(p, p->fwd[0], 1, in, A));
    TRY(mark(p));
    char *cur = A;
    if (p->P2 > 1) { TRY(exchange(p, 1, true, A, B)); cur = B; }
    TRY(mark(p));
    char *ydst = (cur == B) ? C : B;
    TRY(run_pass(p, p->fwd[1], p->vfwd[1], p->Ny, p->tw_y, cur, ydst));
    cur = ydst;
    TRY(mark(p));
    if (p->P1 > 1) { char *dst = (cur == B) ? C : B; TRY(exchange(p, 2, true, cur, dst)); cur = dst; }
    TRY(mark(p));
    TRY(run_pass(p, p->fwd[2], p->vfwd[2], p->Nx, p->tw_x, cur, A));
    TRY(mark(p));
    return 0;
}

// inverse chain, complex output.  `in` is scratch after the first pass, B = work area.
static int enqueue_inverse(dfft_plan *p, void *out, void *in)
{
    char *I = static_cast<char *>(in), *B = static_cast<char *>(p->work_d);
    p->nev = 0; p->last_dir = DFFT_INVERSE;
    TRY(mark(p));
    TRY(run_pass(p, p->inv[2], p->vinv[2], p->Nx, p->tw_x, I, B));
    TRY(mark(p));
    char *cur = B;
    if (p->P1 > 1) { TRY(exchange(p, 2, false, B, I)); cur = I; }
    TRY(mark(p));
    char *ydst = (cur == B) ? I : B;
    TRY(run_pass(p, p->inv[1], p->vinv[1], p->Ny, p->tw_y, cur, ydst));
    cur = ydst;
    TRY(mark(p));
    if (p->P2 > 1) { char *dst = (cur == B) ? I : B; TRY(exchange(p, 1, false, cur, dst)); cur = dst; }
    TRY(mark(p));
    if (p->c2c) TRY(run_pass(p, p->inv[0], p->vinv[0], p->Nz, p->tw_z, cur, out));
    else TRY(run_real_pass(p, p->inv[0], 2, cur, out));
    TRY(mark(p));
    return 0;
}

static int check_ready(dfft_plan *p)
{
    if (!p) return fail(ERR_ARG, "null plan");
    if (!p->initialized) return fail(ERR_STATE, "plan not initialised (call dfft_init first)");
    if (!p->work_d) return fail(ERR_STATE, "no work area (call dfft_set_work_area)");
    return 0;
}

static int ensure_device_state(dfft_plan *p);

extern "C" {

const char *dfft_last_error(void) { return g_error.c_str(); }
const char *dfft_version(void) { return "distributedfft_amd 0.1 (gfx950)"; }

int dfft_comm_create_local(int nranks, dfft_comm **world)
{
    if (nranks < 1 || !world) return fail(ERR_ARG, "bad arguments");
    *world = make_local_world(nranks);
    return 0;
}
int dfft_rccl_unique_id(void *id128) { return rccl_unique_id(id128); }
int dfft_comm_create_rccl(const void *id128, int nranks, int rank, dfft_comm **comm)
{
    if (!id128 || !comm || rank < 0 || rank >= nranks) return fail(ERR_ARG, "bad arguments");
    *comm = make_rccl_comm(id128, nranks, rank);
    return *comm ? 0 : 1;
}
int dfft_comm_create_callback(int nranks, int rank, dfft_alltoallv_fn fn, void *user, dfft_comm **comm)
{
    if (!fn || !comm || rank < 0 || rank >= nranks) return fail(ERR_ARG, "bad arguments");
    *comm = make_callback_comm(nranks, rank, (void *)fn, user);
    return 0;
}
int dfft_comm_destroy(dfft_comm *comm)
{
    delete comm;
    return 0;
}

int dfft_plan_create(dfft_plan **plan, int kind, int precision, const dfft_config *config, dfft_comm *comm,
                     int rank, int max_world_size)
{
    if (!plan) return fail(ERR_ARG, "null plan pointer");
    if (kind < DFFT_SLAB || kind > DFFT_PENCIL_OPT1) return fail(ERR_ARG, "unknown plan kind");
    if (precision != DFFT_F32 && precision != DFFT_F64) return fail(ERR_ARG, "unknown precision");
    dfft_plan *p = new dfft_plan;
    p->kind = kind; p->prec = precision;
    if (config) p->cfg = *config;
    p->comm = comm;
    p->nranks = comm ? comm->nranks : 1;
    p->rank = comm ? (comm->fixed_rank() >= 0 ? comm->fixed_rank() : rank) : 0;
    // max_world_size: the reference truncates the communicator to the first ranks
    // (src/mpicufft.cpp:46-51); here ranks beyond it simply may not create plans.
    if (max_world_size > 0 && max_world_size < p->nranks) p->nranks = max_world_size;
    if (p->rank < 0 || p->rank >= p->nranks) { delete p; return fail(ERR_ARG, "rank outside the world"); }
    p->esz = precision == DFFT_F64 ? 16 : 8;
    p->TL = precision == DFFT_F64 ? TL_F64 : TL_F32;
    *plan = p;
    return 0;
}

int dfft_plan_destroy(dfft_plan *p)
{
    if (!p) return 0;
    if (p->work_owned && p->work_d) (void)hipFree(p->work_d);
    for (void *t : {p->tw_x, p->tw_y, p->tw_z, p->tw_zr}) if (t) (void)hipFree(t);
    for (auto &e : p->ev) if (e) (void)hipEventDestroy(e);
    if (p->stream_owned && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return 0;
}

int dfft_init(dfft_plan *p, size_t Nx, size_t Ny, size_t Nz, int P1, int P2, int c2c, int allocate)
{
    if (!p) return fail(ERR_ARG, "null plan");
    if (!Nx || !Ny || !Nz) return fail(ERR_ARG, "GlobalSize not initialized!");
    if (P1 < 1 || P2 < 1 || P1 * P2 != p->nranks) return fail(ERR_ARG, "Invalid Input Partition!");
    if ((p->kind == DFFT_SLAB || p->kind == DFFT_SLAB_OPT1) && P2 != 1)
        return fail(ERR_ARG, "slab decomposition needs P2 == 1");
    if (P1 > MAXSEG || P2 > MAXSEG) return fail(ERR_UNSUPPORTED, "more than 16 ranks per exchange group");
    if ((size_t)P1 > Nx || (size_t)P1 > Ny || (size_t)P2 > Ny) return fail(ERR_ARG, "partition larger than the grid");
    PassInfo pinfo;
    if (!c2c && (Nz % 2 || Nz < 4 || Nz > 2048))
        return fail(ERR_UNSUPPORTED, "R2C/C2R needs an even Nz in 4..2048");
    for (size_t n : {Nx, Ny, c2c ? Nz : Nz / 2})
        if (!pass_info(p->prec, (int)n, &pinfo))
            return fail(ERR_UNSUPPORTED, "unsupported axis length " + std::to_string(n) +
                                             " (power of two, 2..2048)");
    p->Nx = Nx; p->Ny = Ny; p->Nz = Nz; p->c2c = c2c != 0;
    p->Nzc = c2c ? Nz : Nz / 2 + 1;
    if ((size_t)P2 > p->Nzc) return fail(ERR_ARG, "partition larger than the grid");
    p->P1 = P1; p->P2 = P2;
    p->pi = p->rank / P2; p->pj = p->rank % P2;       // pidx = pidx_i * P2 + pidx_j (:67-68)
    split(Nx, P1, p->xs, p->xstart);
    split(Ny, P2, p->ys, p->ystart);
    split(p->Nzc, P2, p->zs, p->zstart);
    split(Ny, P1, p->yo, p->yostart);
    const size_t xs = p->xs[p->pi], ys = p->ys[p->pj], zs = p->zs[p->pj], yo = p->yo[p->pi];
    // domainsize = largest stage (:203-209)
    p->domain_elems = std::max({xs * ys * p->Nzc, xs * Ny * zs, Nx * yo * zs});
    p->domainsize = p->domain_elems * p->esz;
    p->domainsize = (p->domainsize + 255) & ~(size_t)255;
    const int nexch = (P1 > 1) + (P2 > 1);
    p->worksize_d = p->domainsize * (nexch ? 2 : 1);
    // all-to-all tables in bytes (:269-273, :315-319)
    const size_t e = p->esz;
    p->sc1.assign(P2, 0); p->sd1.assign(P2, 0); p->rc1.assign(P2, 0); p->rd1.assign(P2, 0);
    p->group1.assign(P2, 0);
    for (int q = 0; q < P2; q++) {
        p->sc1[q] = e * p->zs[q] * ys * xs;
        p->sd1[q] = e * p->zstart[q] * ys * xs;
        p->rc1[q] = e * xs * p->ys[q] * zs;
        p->rd1[q] = e * xs * p->ystart[q] * zs;
        p->group1[q] = p->pi * P2 + q;
    }
    p->sc2.assign(P1, 0); p->sd2.assign(P1, 0); p->rc2.assign(P1, 0); p->rd2.assign(P1, 0);
    p->group2.assign(P1, 0);
    for (int q = 0; q < P1; q++) {
        p->sc2[q] = e * xs * zs * p->yo[q];
        p->sd2[q] = e * xs * zs * p->yostart[q];
        p->rc2[q] = e * p->xs[q] * yo * zs;
        p->rd2[q] = e * p->xstart[q] * yo * zs;
        p->group2[q] = q * P2 + p->pj;
    }
    TRY(build_passes(p));
    // point-major (strided) API layouts: neighbouring workgroups step along the outer axis so
    // that concurrent workgroups do not all sit 128 B apart in the same DRAM/L2 channel group
    p->fwd[2].a_fastest = 1;
    p->inv[2].a_fastest = 1;
    if (const char *v = getenv("DFFT_ORDER")) {   // experiment hook: 6 digits like DFFT_VARIANTS
        int k = 0;
        for (const char *c = v; *c && k < 6; c++) {
            if (*c < '0' || *c > '9') continue;
            if (k < 3) p->fwd[k].a_fastest = *c - '0'; else p->inv[5 - k].a_fastest = *c - '0';
            k++;
        }
    }
    // experiment hook: DFFT_VARIANTS="zyx xyz" digits = kernel variant of fwd z,y,x then inv x,y,z
    if (const char *v = getenv("DFFT_VARIANTS")) {
        int k = 0;
        for (const char *c = v; *c && k < 6; c++) {
            if (*c < '0' || *c > '9') continue;
            if (k < 3) p->vfwd[k] = *c - '0'; else p->vinv[5 - k] = *c - '0';
            k++;
        }
    }
    for (void **t : {&p->tw_x, &p->tw_y, &p->tw_z, &p->tw_zr}) if (*t) { (void)hipFree(*t); *t = nullptr; }
    p->initialized = true;
    // device-side state (twiddles, stream, work area) is created by setWorkArea, so that the
    // decomposition tables can be queried on a host without a GPU (allocate = 0).
    if (allocate) return dfft_set_work_area(p, nullptr, nullptr);
    return 0;
}

static int ensure_device_state(dfft_plan *p)
{
    if (!p->tw_x) TRY(make_twiddles(p->prec, p->Nx, &p->tw_x));
    if (!p->tw_y) TRY(make_twiddles(p->prec, p->Ny, &p->tw_y));
    if (!p->tw_z) TRY(make_twiddles(p->prec, p->c2c ? p->Nz : p->Nz / 2, &p->tw_z));
    if (!p->c2c && !p->tw_zr) TRY(make_twiddles(p->prec, p->Nz, &p->tw_zr));
    if (!p->stream && !p->stream_user) {
        HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        p->stream_owned = true;
    }
    return 0;
}

int dfft_set_work_area(dfft_plan *p, void *device, void *host)
{
    (void)host;   // no host staging: device buffers are handed to the transport directly
    if (!p) return fail(ERR_ARG, "null plan");
    if (!p->initialized) return fail(ERR_STATE, "cannot set work area: plan not initialised");
    if (p->work_owned && p->work_d) { (void)hipFree(p->work_d); p->work_d = nullptr; p->work_owned = false; }
    TRY(ensure_device_state(p));
    if (device) {
        p->work_d = device;   // caller keeps ownership (:333-342)
    } else {
        HIP_TRY(hipMalloc(&p->work_d, p->worksize_d));
        p->work_owned = true;
    }
    return 0;
}

int dfft_set_stream(dfft_plan *p, void *hip_stream)
{
    if (!p) return fail(ERR_ARG, "null plan");
    if (p->stream_owned && p->stream) (void)hipStreamDestroy(p->stream);
    p->stream = (hipStream_t)hip_stream;
    p->stream_owned = false;
    p->stream_user = true;
    return 0;
}

int dfft_enqueue_c2c(dfft_plan *p, void *out, void *in, int direction)
{
    TRY(check_ready(p));
    if (!p->c2c) return fail(ERR_STATE, "plan was initialised for R2C/C2R");
    if (!out || !in) return fail(ERR_ARG, "null buffer");
    if (direction == DFFT_FORWARD) return enqueue_forward(p, out, in);
    if (direction == DFFT_INVERSE) return enqueue_inverse(p, out, in);
    return fail(ERR_ARG, "direction must be DFFT_FORWARD or DFFT_INVERSE");
}

int dfft_exec_c2c(dfft_plan *p, void *out, void *in, int direction)
{
    TRY(dfft_enqueue_c2c(p, out, in, direction));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}

int dfft_exchange(dfft_plan *p, int which, int direction, const void *sendbuf, void *recvbuf)
{
    if (!p || !p->initialized) return fail(ERR_STATE, "plan not initialised");
    if (which != 1 && which != 2) return fail(ERR_ARG, "which must be 1 or 2");
    if ((which == 1 ? p->P2 : p->P1) > 1) TRY(exchange(p, which, direction != DFFT_INVERSE, sendbuf, recvbuf));
    if (p->stream || p->stream_user) HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}

int dfft_exec_r2c(dfft_plan *p, void *out, const void *in)
{
    TRY(check_ready(p));
    if (p->c2c) return fail(ERR_STATE, "plan was initialised for C2C");
    if (!out || !in) return fail(ERR_ARG, "null buffer");
    TRY(enqueue_forward(p, out, in));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}
int dfft_exec_c2r(dfft_plan *p, void *out, void *in)
{
    TRY(check_ready(p));
    if (p->c2c) return fail(ERR_STATE, "plan was initialised for C2C");
    if (!out || !in) return fail(ERR_ARG, "null buffer");
    TRY(enqueue_inverse(p, out, in));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}

int dfft_get_in_size(const dfft_plan *p, size_t s[3])
{
    if (!p || !p->initialized) return fail(ERR_STATE, "plan not initialised");
    s[0] = p->xs[p->pi]; s[1] = p->ys[p->pj]; s[2] = p->Nz;
    return 0;
}
int dfft_get_in_start(const dfft_plan *p, size_t s[3])
{
    if (!p || !p->initialized) return fail(ERR_STATE, "plan not initialised");
    s[0] = p->xstart[p->pi]; s[1] = p->ystart[p->pj]; s[2] = 0;
    return 0;
}
int dfft_get_out_size(const dfft_plan *p, size_t s[3])
{
    if (!p || !p->initialized) return fail(ERR_STATE, "plan not initialised");
    s[0] = p->Nx; s[1] = p->yo[p->pi]; s[2] = p->zs[p->pj];
    return 0;
}
int dfft_get_out_start(const dfft_plan *p, size_t s[3])
{
    if (!p || !p->initialized) return fail(ERR_STATE, "plan not initialised");
    s[0] = 0; s[1] = p->yostart[p->pi]; s[2] = p->zstart[p->pj];
    return 0;
}
size_t dfft_domain_size(const dfft_plan *p) { return p ? p->domainsize : 0; }
size_t dfft_work_size_device(const dfft_plan *p) { return p ? p->worksize_d : 0; }
size_t dfft_work_size_host(const dfft_plan *p) { (void)p; return 0; }
void *dfft_work_area_device(const dfft_plan *p) { return p ? p->work_d : nullptr; }
int dfft_rank(const dfft_plan *p) { return p ? p->rank : -1; }
int dfft_world_size(const dfft_plan *p) { return p ? p->nranks : 0; }
int dfft_tile_lines(const dfft_plan *p) { return p ? p->TL : 0; }

int dfft_get_exchange_tables(const dfft_plan *p, int which, size_t *sc, size_t *sd, size_t *rc, size_t *rd)
{
    if (!p || !p->initialized) return fail(ERR_STATE, "plan not initialised");
    if (which != 1 && which != 2) return fail(ERR_ARG, "which must be 1 or 2");
    const auto &a = which == 1 ? p->sc1 : p->sc2, &b = which == 1 ? p->sd1 : p->sd2;
    const auto &c = which == 1 ? p->rc1 : p->rc2, &d = which == 1 ? p->rd1 : p->rd2;
    for (size_t i = 0; i < a.size(); i++) { sc[i] = a[i]; sd[i] = b[i]; rc[i] = c[i]; rd[i] = d[i]; }
    return 0;
}

int dfft_enable_phase_timing(dfft_plan *p, int enable)
{
    if (!p) return fail(ERR_ARG, "null plan");
    p->timing = enable != 0;
    return 0;
}
int dfft_get_phase_times(dfft_plan *p, float *ms, int max_entries)
{
    if (!p) return fail(ERR_ARG, "null plan");
    int n = 0;
    for (int i = 0; i + 1 < p->nev && n < max_entries; i++, n++) {
        if (hipEventElapsedTime(&ms[n], p->ev[i], p->ev[i + 1]) != hipSuccess) ms[n] = -1.f;
    }
    return n;
}
const char *dfft_phase_name(int phase, int direction)
{
    static const char *f[] = {"z-FFT", "exchange 1", "y-FFT", "exchange 2", "x-FFT"};
    static const char *b[] = {"x-FFT^-1", "exchange 2", "y-FFT^-1", "exchange 1", "z-FFT^-1"};
    if (phase < 0 || phase > 4) return "";
    return direction == DFFT_INVERSE ? b[phase] : f[phase];
}

int dfft_fft1d_batched(int precision, size_t N, size_t batch, void *out, const void *in, int direction,
                       void *hip_stream)
{
    PassInfo pi;
    if (!pass_info(precision, (int)N, &pi)) return fail(ERR_UNSUPPORTED, "unsupported line length");
    static thread_local void *tw = nullptr;
    static thread_local size_t twN = 0;
    static thread_local int twP = -1;
    if (twN != N || twP != precision) {
        if (tw) (void)hipFree(tw);
        tw = nullptr;
        TRY(make_twiddles(precision, N, &tw));
        twN = N; twP = precision;
    }
    PassArgs A;
    memset(&A, 0, sizeof(A));
    A.in = in; A.out = out; A.tw = tw;
    A.na = 1; A.LB = (uint32_t)batch; A.nb = ((uint32_t)batch + pi.TL - 1) / pi.TL; A.ntiles = A.nb;
    A.load_kind = LOAD_LINES; A.store_kind = STORE_LINES; A.swap = direction == DFFT_INVERSE;
    return launch_pass(precision, (int)N, getenv("DFFT_VARIANT_1D") ? atoi(getenv("DFFT_VARIANT_1D")) : 0, A, (hipStream_t)hip_stream);
}

int dfft_kernel_info(int precision, size_t N, int *threads, int *lds_bytes, int *points_per_thread,
                     int *lines_per_workgroup)
{
    PassInfo pi;
    if (!pass_info(precision, (int)N, &pi)) return ERR_UNSUPPORTED;
    if (threads) *threads = pi.threads;
    if (lds_bytes) *lds_bytes = pi.lds_bytes;
    if (points_per_thread) *points_per_thread = pi.E;
    if (lines_per_workgroup) *lines_per_workgroup = pi.TL * pi.G;
    return 0;
}

}  // extern "C"
