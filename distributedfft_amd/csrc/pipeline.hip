// pipeline.hip -- the pass descriptors and exchange tables of every decomposition (the layout algebra of DESIGN.md section 2):
// which lines a pass transforms, how its load and store sides address the buffers, which bytes go to which peer, per pipeline chunk.
// Host code only; its own translation unit since round 6 (csrc/dfft.hip keeps axis plans, execution chains, tuners and the C ABI).
#include <string.h>

#include <algorithm>

#include "host_common.hpp"
#include "plan.hpp"

using namespace dfft;

// ------------------------------------------------------------------------------------------
// Pipelined execution plan.
//
// Every axis pass that feeds an exchange is cut into C chunks along its outer line-set axis
// (x for the forward z/y passes and the inverse y/z passes, ky for the inverse x pass).  Send
// and receive buffers are laid out chunk-outermost, [chunk][peer block], so one (chunk, peer)
// message is contiguous: chunk c is exchanged on the communication stream while chunk c+1 is
// still being transformed on the compute stream.  This replaces the reference's only overlap
// mechanism, the Peer2Peer modes with MPI_Waitany / the sender thread
// (src/pencil/mpicufft_pencil_opt1.cpp:601-754, src/pencil/mpicufft_pencil.cpp:513-585), at
// chunk instead of whole-peer granularity.  C = 1 reproduces the reference's message sizes and
// displacements exactly (mpicufft_pencil_opt1.cpp:269-273, 315-319).
// ------------------------------------------------------------------------------------------
// descriptor of one axis pass over `na` outer slices of `LB` lines each (tiles of TL lines)
static PassArgs pass_args(int TL, size_t na, size_t LB, int load_kind, int store_kind, int swap)
{
    PassArgs A;
    memset(&A, 0, sizeof(A));
    A.na = (uint32_t)na; A.LB = (uint32_t)LB; A.nb = (uint32_t)((LB + TL - 1) / TL); A.ntiles = A.na * A.nb;
    A.load_kind = load_kind; A.store_kind = store_kind; A.swap = swap; A.T2shift = ilog2(TL);
    return A;
}

// Point-major store whose row pitch (AS_out) is not a multiple of the tile, e.g. the 513-wide rows
// of an R2C spectrum: let every workgroup's window start at a cache-line boundary of its output row.
// Needs every row start to differ from an aligned address by (a*AS_out) mod TL only, i.e. KS_out a
// multiple of TL (the caller's buffer is assumed 128-byte aligned, like every hipMalloc result).
static void set_shift(const dfft_plan *p, PassArgs &X)
{
    const uint32_t TL = (uint32_t)p->TL;
    if (p->opt.shift == 0) return;
    if (p->ax[2].bluestein) return;               // the Bluestein kernel has no shifted windows
    // fp32 (16-line tiles of 8-byte points) measured 10 % slower with shifted windows, fp64 24 % faster
    if (p->prec != DFFT_F64 && p->opt.shift != 2) return;
    if (X.AS_out % TL == 0 || X.KS_out % TL != 0 || X.LB < TL) return;
    X.shift = 1;
    X.nb += 1;
    X.ntiles = X.na * X.nb;
}

static void seg_push(SegTable &t, size_t start, size_t len, size_t base_elems)
{
    int s = t.nseg++;
    t.start[s] = (uint32_t)start;
    t.len[s] = (uint32_t)len;
    t.base[s] = base_elems;
}

int build_pipeline(dfft_plan *p, Pipeline &pl)
{
    const int TL = p->TL, P1 = p->P1, P2 = p->P2, C = pl.C;
    const size_t xs = p->xs[p->pi], ys = p->ys[p->pj], zs = p->zs[p->pj], yo = p->yo[p->pi];
    const size_t Nx = p->Nx, Ny = p->Ny, Nzc = p->Nzc, e = p->esz;
    // bytes per input/output LINE of the z pass as the caller sees it (real lines in R2C mode)
    const size_t zline_bytes = p->c2c ? p->Nz * e : p->Nz * (e / 2);
    auto base = [&](size_t na, size_t LB, int lk, int sk, int swap) { return pass_args(TL, na, LB, lk, sk, swap); };
    std::vector<size_t> xl, x0, kl, k0;
    split(xs, C, xl, x0);       // my x range in chunks (forward z/y, inverse y/z passes)
    split(yo, C, kl, k0);       // my ky range in chunks (inverse x pass)
    // chunk c of every column peer's x / ky range (they split with the same rule)
    std::vector<std::vector<size_t>> xlq(P1), x0q(P1), klq(P1), k0q(P1);
    for (int q = 0; q < P1; q++) { split(p->xs[q], C, xlq[q], x0q[q]); split(p->yo[q], C, klq[q], k0q[q]); }

    pl.fx = Launch();           // (initFFT may be called again on the same plan)
    pl.fz.assign(C, Launch()); pl.fy.assign(C, Launch()); pl.ix.assign(C, Launch());
    pl.iy.assign(C, Launch()); pl.iz.assign(C, Launch());
    pl.f1.assign(C, A2A()); pl.f2.assign(C, A2A()); pl.i2.assign(C, A2A()); pl.i1.assign(C, A2A());

    // ---------------- forward (mpicufft_pencil_opt1.cpp:1422-1519) ----------------
    size_t R2c = 0;   // running element offset of chunk c in the exchange-2 receive buffer
    {
        PassArgs X = base(yo, zs, LOAD_TILED, STORE_KMAJOR, 0);
        X.KS_out = (uint64_t)yo * zs;
        X.AS_out = zs;
        // neighbouring tiles along z' share cache lines whenever the pitch zs is not a multiple of the
        // tile; keeping consecutive tiles on one XCD lets its L2 merge them (R2C, 513-wide: 6.3 -> 4.5 ms)
        X.a_fastest = 0; X.xcd_swizzle = 1;
        if (p->opt.spectral) {
            // x-contiguous spectrum [ky][kz'][kx]: natural lines out (a = ky, lines of a slice = kz'), no point-major store
            X.store_kind = STORE_LINES;
            X.KS_out = 0; X.AS_out = 0;
        } else
        set_shift(p, X);
        // segments of the x axis, ascending: peer q major, chunk c minor
        std::vector<size_t> r2c_of(C, 0);
        { size_t acc = 0; for (int c = 0; c < C; c++) { r2c_of[c] = acc; for (int q = 0; q < P1; q++) acc += xlq[q][c] * yo * zs; } }
        for (int q = 0; q < P1; q++)
            for (int c = 0; c < C; c++) {
                size_t off = r2c_of[c];
                for (int q2 = 0; q2 < q; q2++) off += xlq[q2][c] * yo * zs;
                if (xlq[q][c]) seg_push(pl.fx.lseg, p->xstart[q] + x0q[q][c], xlq[q][c], off);
            }
        pl.fx.args = X;
    }
    for (int c = 0; c < C; c++) {
        const size_t S1c = x0[c] * Nzc * ys, R1c = x0[c] * zs * Ny, S2c = x0[c] * zs * Ny;
        {   // z pass chunk: natural lines -> send1 block (c,p) = [x][kz/TL][y][kz%TL], kz in zs[p]
            Launch &L = pl.fz[c];
            L.args = base(xl[c], ys, LOAD_LINES, STORE_TILED_TRANSPOSE, 0);
            L.in_off = x0[c] * ys * zline_bytes;
            for (int q = 0; q < P2; q++) seg_push(L.sseg, p->zstart[q], p->zs[q], S1c + xl[c] * p->zstart[q] * ys);
        }
        {   // exchange 1, row group (:269-273 restricted to the chunk)
            A2A &T = pl.f1[c];
            for (int q = 0; q < P2; q++) {
                T.sc.push_back(e * xl[c] * p->zs[q] * ys);
                T.sd.push_back(e * (S1c + xl[c] * p->zstart[q] * ys));
                T.rc.push_back(e * xl[c] * p->ys[q] * zs);
                T.rd.push_back(e * (R1c + xl[c] * p->ystart[q] * zs));
            }
        }
        {   // y pass chunk: lines along y from the P2 blocks -> send2 block (c,p) = [ky][kz/TL][x][kz%TL]
            Launch &L = pl.fy[c];
            L.args = base(xl[c], zs, LOAD_TILED, STORE_TILED_SAME, 0);
            for (int q = 0; q < P2; q++) seg_push(L.lseg, p->ystart[q], p->ys[q], R1c + xl[c] * p->ystart[q] * zs);
            for (int q = 0; q < P1; q++) seg_push(L.sseg, p->yostart[q], p->yo[q], S2c + xl[c] * zs * p->yostart[q]);
            L.args.LA = (uint32_t)xl[c];
        }
        {   // exchange 2, column group (:315-319 restricted to the chunk)
            A2A &T = pl.f2[c];
            size_t roff = R2c;
            for (int q = 0; q < P1; q++) {
                T.sc.push_back(e * xl[c] * zs * p->yo[q]);
                T.sd.push_back(e * (S2c + xl[c] * zs * p->yostart[q]));
                T.rc.push_back(e * xlq[q][c] * yo * zs);
                T.rd.push_back(e * roff);
                roff += xlq[q][c] * yo * zs;
            }
            R2c = roff;
        }
    }
    // ---------------- inverse (mpicufft_pencil_opt1.cpp:1522-1600) ----------------
    std::vector<size_t> r2i_of(C, 0);
    { size_t acc = 0; for (int c = 0; c < C; c++) { r2i_of[c] = acc; for (int q = 0; q < P1; q++) acc += xs * zs * klq[q][c]; } }
    for (int c = 0; c < C; c++) {
        const size_t S2i = k0[c] * zs * Nx;
        {   // x^-1 chunk (ky range): API layout point-major -> block (c,p) = [x][kz/TL][ky][kz%TL], x in xs[p]
            Launch &L = pl.ix[c];
            L.args = base(kl[c], zs, LOAD_KMAJOR, STORE_TILED_SAME, 1);
            L.args.KS_in = (uint64_t)yo * zs;
            L.args.AS_in = zs;
            // aligned pitch: step along ky between neighbouring workgroups (DRAM/TLB spread);
            // odd pitch (R2C): neighbouring z' tiles on one XCD so the shared lines are read once.
            // (Row-aligned LOAD windows -- the mirror image of set_shift -- were measured in round 4 and rejected: the loads become
            // whole cache lines, but every 128-byte run of the private layout is then written in two pieces by two workgroups:
            // 3.63 -> 5.88 ms at 1024^3 on 513-wide rows, 0.80 -> 1.03 ms on rank 0 of 2 x 4, profiles/r4_shift_load_rejected.txt)
            L.args.xcd_swizzle = 1;
            L.args.a_fastest = zs % TL == 0 ? 1 : 0;
            L.in_off = e * k0[c] * zs;
            if (p->opt.spectral) {
                // x-contiguous spectrum: natural lines in, [ky][kz'][kx], this chunk's ky rows first
                L.args.load_kind = LOAD_LINES;
                L.args.KS_in = 0; L.args.AS_in = 0;
                L.args.a_fastest = 0;
                L.in_off = e * k0[c] * zs * Nx;
            }
            for (int q = 0; q < P1; q++) seg_push(L.sseg, p->xstart[q], p->xs[q], S2i + p->xstart[q] * zs * kl[c]);
            L.args.LA = (uint32_t)kl[c];
        }
        {   // exchange 2 backwards
            A2A &T = pl.i2[c];
            size_t roff = r2i_of[c];
            for (int q = 0; q < P1; q++) {
                T.sc.push_back(e * p->xs[q] * zs * kl[c]);
                T.sd.push_back(e * (S2i + p->xstart[q] * zs * kl[c]));
                T.rc.push_back(e * xs * zs * klq[q][c]);
                T.rd.push_back(e * roff);
                roff += xs * zs * klq[q][c];
            }
        }
    }
    for (int c = 0; c < C; c++) {
        const size_t S1i = x0[c] * Ny * zs, R1i = x0[c] * ys * Nzc;
        {   // y^-1 chunk (x range): lines along ky from the (peer, ky-chunk) blocks ->
            // block (c,p) = [x][y/TL][kz][y%TL], y in ys[p]
            Launch &L = pl.iy[c];
            L.args = base(xl[c], zs, LOAD_TILED, STORE_TILED_TRANSPOSE, 1);
            for (int q = 0; q < P1; q++)
                for (int c2 = 0; c2 < C; c2++) {
                    size_t off = r2i_of[c2];
                    for (int q2 = 0; q2 < q; q2++) off += xs * zs * klq[q2][c2];
                    // the block is [x in xs][kz/TL][ky][kz%TL]: skip the x rows before this chunk
                    if (klq[q][c2]) seg_push(L.lseg, p->yostart[q] + k0q[q][c2], klq[q][c2], off + x0[c] * klq[q][c2] * zs);
                }
            for (int q = 0; q < P2; q++) seg_push(L.sseg, p->ystart[q], p->ys[q], S1i + xl[c] * p->ystart[q] * zs);
        }
        {   // exchange 1 backwards
            A2A &T = pl.i1[c];
            for (int q = 0; q < P2; q++) {
                T.sc.push_back(e * xl[c] * p->ys[q] * zs);
                T.sd.push_back(e * (S1i + xl[c] * p->ystart[q] * zs));
                T.rc.push_back(e * xl[c] * ys * p->zs[q]);
                T.rd.push_back(e * (R1i + xl[c] * ys * p->zstart[q]));
            }
        }
        {   // z^-1 chunk: lines along kz from the P2 blocks -> natural [x][y][z]
            Launch &L = pl.iz[c];
            L.args = base(xl[c], ys, LOAD_TILED, STORE_LINES, 1);
            L.args.xcd_swizzle = 1;       // measured +3 % on the natural-line stores
            for (int q = 0; q < P2; q++) seg_push(L.lseg, p->zstart[q], p->zs[q], R1i + xl[c] * ys * p->zstart[q]);
            L.out_off = x0[c] * ys * zline_bytes;
        }
    }
    // ---------------- partial transforms (d = 1, 2) ----------------
    pl.pz1 = Launch(); pl.qz1 = Launch();
    pl.pz1.args = base(xs, ys, LOAD_LINES, STORE_LINES, 0);
    pl.qz1.args = base(xs, ys, LOAD_LINES, STORE_LINES, 1);
    pl.py2.assign(C, Launch()); pl.qy2.assign(C, Launch());
    for (int c = 0; c < C; c++) {
        const size_t R1c = x0[c] * zs * Ny, S1i = x0[c] * Ny * zs;
        {   // forward y pass chunk writing the reference's opt0 stage layout [xs][Ny][zs]
            Launch &L = pl.py2[c];
            L.args = base(xl[c], zs, LOAD_TILED, STORE_KMAJOR, 0);
            for (int q = 0; q < P2; q++) seg_push(L.lseg, p->ystart[q], p->ys[q], R1c + xl[c] * p->ystart[q] * zs);
            L.args.KS_out = zs; L.args.AS_out = Ny * zs; L.args.xcd_swizzle = 1;
            L.out_off = e * x0[c] * Ny * zs;
        }
        {   // inverse y pass chunk reading [xs][Ny][zs]
            Launch &L = pl.qy2[c];
            L.args = base(xl[c], zs, LOAD_KMAJOR, STORE_TILED_TRANSPOSE, 1);
            L.args.KS_in = zs; L.args.AS_in = Ny * zs;
            L.in_off = e * x0[c] * Ny * zs;
            for (int q = 0; q < P2; q++) seg_push(L.sseg, p->ystart[q], p->ys[q], S1i + xl[c] * p->ystart[q] * zs);
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Slab sequence Z_Then_YX (src/slab/z_then_yx/mpicufft_slab_z_then_yx.cpp:74-200): the input is
// split along x, [xs][Ny][Nz]; after the z pass ONE all-to-all over all P ranks sends the slice
// kz in zs[p] to rank p (counts :190-196), and the (y, x) transform runs on [Nx][Ny][zs].  Here:
//   z pass chunk c      natural lines -> block (c,p) = [x][kz/TL][y][kz%TL]           (send)
//   exchange chunk c    receive block (c,q) = [x in chunk c of xs[q]][kz/TL][y][kz%TL]
//   y pass per (c,q)    -> block (c,q) = [ky][kz/TL][x][kz%TL]        (same offsets, other buffer)
//   x pass              lines along x gathered from the P*C blocks -> API layout [kx][ky][kz']
// and the mirror image for the inverse.  Uses p->xs (x split), p->zs (z split over P1 = P ranks).
// ------------------------------------------------------------------------------------------
int build_pipeline_zyx(dfft_plan *p, Pipeline &pl)
{
    const int TL = p->TL, P = p->P1, C = pl.C, r = p->pi;
    const size_t xs = p->xs[r], zs = p->zs[r];
    const size_t Ny = p->Ny, Nzc = p->Nzc, e = p->esz;
    const size_t zline_bytes = p->c2c ? p->Nz * e : p->Nz * (e / 2);
    auto base = [&](size_t na, size_t LB, int lk, int sk, int swap) { return pass_args(TL, na, LB, lk, sk, swap); };
    std::vector<size_t> xl, x0;
    split(xs, C, xl, x0);
    std::vector<std::vector<size_t>> xlq(P), x0q(P);
    for (int q = 0; q < P; q++) split(p->xs[q], C, xlq[q], x0q[q]);
    // element offset of block (c,q) on the z-split side, chunk outermost
    std::vector<std::vector<size_t>> blk(C, std::vector<size_t>(P, 0));
    { size_t acc = 0; for (int c = 0; c < C; c++) for (int q = 0; q < P; q++) { blk[c][q] = acc; acc += xlq[q][c] * Ny * zs; } }

    pl.fx = Launch(); pl.zix = Launch();
    pl.fz.assign(C, Launch()); pl.iz.assign(C, Launch());
    pl.zy.assign((size_t)C * P, Launch()); pl.ziy.assign((size_t)C * P, Launch());
    pl.f2.assign(C, A2A()); pl.i2.assign(C, A2A());
    pl.fy.clear(); pl.ix.clear(); pl.iy.clear(); pl.f1.clear(); pl.i1.clear(); pl.py2.clear(); pl.qy2.clear();
    pl.pz1 = Launch(); pl.qz1 = Launch();

    for (int c = 0; c < C; c++) {
        const size_t S1c = x0[c] * Nzc * Ny;     // send side: my x chunk, every kz
        {   // z pass chunk
            Launch &L = pl.fz[c];
            L.args = base(xl[c], Ny, LOAD_LINES, STORE_TILED_TRANSPOSE, 0);
            L.in_off = x0[c] * Ny * zline_bytes;
            for (int q = 0; q < P; q++) seg_push(L.sseg, p->zstart[q], p->zs[q], S1c + xl[c] * p->zstart[q] * Ny);
        }
        {   // forward exchange (:190-196 restricted to the chunk)
            A2A &T = pl.f2[c];
            for (int q = 0; q < P; q++) {
                T.sc.push_back(e * xl[c] * p->zs[q] * Ny);
                T.sd.push_back(e * (S1c + xl[c] * p->zstart[q] * Ny));
                T.rc.push_back(e * xlq[q][c] * Ny * zs);
                T.rd.push_back(e * blk[c][q]);
            }
        }
        for (int q = 0; q < P; q++) {   // y pass on the block received from q
            Launch &L = pl.zy[(size_t)c * P + q];
            L.args = base(xlq[q][c], zs, LOAD_TILED, STORE_TILED_SAME, 0);
            seg_push(L.lseg, 0, Ny, blk[c][q]);
            seg_push(L.sseg, 0, Ny, blk[c][q]);
            L.args.LA = (uint32_t)xlq[q][c];
        }
    }
    {   // x pass: lines along x from the P*C blocks -> [kx][ky][kz']
        PassArgs X = base(Ny, zs, LOAD_TILED, STORE_KMAJOR, 0);
        X.KS_out = (uint64_t)Ny * zs; X.AS_out = zs; X.xcd_swizzle = 1;
        set_shift(p, X);
        pl.fx.args = X;
        for (int q = 0; q < P; q++)
            for (int c = 0; c < C; c++)
                if (xlq[q][c]) seg_push(pl.fx.lseg, p->xstart[q] + x0q[q][c], xlq[q][c], blk[c][q]);
    }
    // ---------------- inverse ----------------
    {   // x^-1: API layout -> blocks (c,q) = [x][kz/TL][ky][kz%TL]
        Launch &L = pl.zix;
        L.args = base(Ny, zs, LOAD_KMAJOR, STORE_TILED_SAME, 1);
        L.args.KS_in = (uint64_t)Ny * zs; L.args.AS_in = zs;
        L.args.xcd_swizzle = 1; L.args.a_fastest = zs % TL == 0 ? 1 : 0;
        L.args.LA = (uint32_t)Ny;
        for (int q = 0; q < P; q++)
            for (int c = 0; c < C; c++)
                if (xlq[q][c]) seg_push(L.sseg, p->xstart[q] + x0q[q][c], xlq[q][c], blk[c][q]);
    }
    for (int c = 0; c < C; c++) {
        const size_t R1i = x0[c] * Ny * Nzc;
        for (int q = 0; q < P; q++) {   // y^-1 on block (c,q) -> send block [x][y/TL][kz'][y%TL]
            Launch &L = pl.ziy[(size_t)c * P + q];
            L.args = base(xlq[q][c], zs, LOAD_TILED, STORE_TILED_TRANSPOSE, 1);
            seg_push(L.lseg, 0, Ny, blk[c][q]);
            seg_push(L.sseg, 0, Ny, blk[c][q]);
        }
        {   // inverse exchange, already in send/receive order
            A2A &T = pl.i2[c];
            for (int q = 0; q < P; q++) {
                T.sc.push_back(e * xlq[q][c] * Ny * zs);
                T.sd.push_back(e * blk[c][q]);
                T.rc.push_back(e * xl[c] * Ny * p->zs[q]);
                T.rd.push_back(e * (R1i + xl[c] * Ny * p->zstart[q]));
            }
        }
        {   // z^-1 chunk: lines along kz from the P blocks -> natural [x][y][z]
            Launch &L = pl.iz[c];
            L.args = base(xl[c], Ny, LOAD_TILED, STORE_LINES, 1);
            L.args.xcd_swizzle = 1;
            for (int q = 0; q < P; q++) seg_push(L.lseg, p->zstart[q], p->zs[q], R1i + xl[c] * Ny * p->zstart[q]);
            L.out_off = x0[c] * Ny * zline_bytes;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Slab sequence Y_Then_ZX (src/slab/y_then_zx/mpicufft_slab_y_then_zx.cpp:71-175, 268-378; the
// reference provides the forward direction only).  The real-to-complex transform runs along y, the
// output [Nx][(Ny/2+1)/P][Nz] keeps z contiguous:
//   y pass chunk c   real lines along y read in place (lanes along z) -> send block (c,p) =
//                    [ky in yo[p]][z/TL][x][z%TL]
//   exchange chunk c (counts :309-319)  -> recv block (c,q), x in chunk c of xs[q]
//   x pass           lines along x gathered from the P*C blocks -> [ky][kx/TL][z][kx%TL]
//   z pass           -> rows (kx*yo + ky)*Nz of the API layout (strided-lines store)
// ------------------------------------------------------------------------------------------
int build_pipeline_yzx(dfft_plan *p, Pipeline &pl)
{
    const int TL = p->TL, P = p->P1, C = pl.C, r = p->pi;
    const size_t xs = p->xs[r], yo = p->yo[r];
    const size_t Nx = p->Nx, Ny = p->Ny, Nz = p->Nz, e = p->esz;
    auto base = [&](size_t na, size_t LB, int lk, int sk) { return pass_args(TL, na, LB, lk, sk, 0); };
    std::vector<size_t> xl, x0;
    split(xs, C, xl, x0);
    std::vector<std::vector<size_t>> xlq(P), x0q(P);
    for (int q = 0; q < P; q++) split(p->xs[q], C, xlq[q], x0q[q]);
    std::vector<std::vector<size_t>> blk(C, std::vector<size_t>(P, 0));
    { size_t acc = 0; for (int c = 0; c < C; c++) for (int q = 0; q < P; q++) { blk[c][q] = acc; acc += xlq[q][c] * yo * Nz; } }

    pl.fx = Launch(); pl.yz = Launch(); pl.zix = Launch(); pl.pz1 = Launch(); pl.qz1 = Launch();
    pl.fy.assign(C, Launch()); pl.f2.assign(C, A2A());
    pl.fz.clear(); pl.iz.clear(); pl.ix.clear(); pl.iy.clear(); pl.zy.clear(); pl.ziy.clear();
    pl.f1.clear(); pl.i1.clear(); pl.i2.clear(); pl.py2.clear(); pl.qy2.clear();
    // in-place input lines: element (x, y, z) at (x*Ny + y)*Nz + z, in reals (R2C) or complex (C2C)
    const size_t in_elem = p->c2c ? e : e / 2;
    for (int c = 0; c < C; c++) {
        const size_t S1c = x0[c] * p->Nyc * Nz;
        {
            Launch &L = pl.fy[c];
            L.args = base(xl[c], Nz, LOAD_KMAJOR, STORE_TILED_SAME);
            L.args.KS_in = Nz; L.args.AS_in = (uint64_t)Ny * Nz;
            L.args.LA = (uint32_t)xl[c];
            L.in_off = x0[c] * Ny * Nz * in_elem;
            for (int q = 0; q < P; q++) seg_push(L.sseg, p->yostart[q], p->yo[q], S1c + xl[c] * Nz * p->yostart[q]);
        }
        {
            A2A &T = pl.f2[c];
            for (int q = 0; q < P; q++) {
                T.sc.push_back(e * xl[c] * Nz * p->yo[q]);
                T.sd.push_back(e * (S1c + xl[c] * Nz * p->yostart[q]));
                T.rc.push_back(e * xlq[q][c] * Nz * yo);
                T.rd.push_back(e * blk[c][q]);
            }
        }
    }
    {   // x pass: [ky][z/TL][x][z%TL] blocks -> [ky][kx/TL][z][kx%TL]
        PassArgs X = base(yo, Nz, LOAD_TILED, STORE_TILED_TRANSPOSE);
        pl.fx.args = X;
        for (int q = 0; q < P; q++)
            for (int c = 0; c < C; c++)
                if (xlq[q][c]) seg_push(pl.fx.lseg, p->xstart[q] + x0q[q][c], xlq[q][c], blk[c][q]);
        seg_push(pl.fx.sseg, 0, Nx, 0);
    }
    {   // z pass: lines along z, lanes along kx -> out[(kx*yo + ky)*Nz + kz]
        PassArgs Z = base(yo, Nx, LOAD_TILED, STORE_LINES);
        Z.KS_out = (uint64_t)yo * Nz; Z.AS_out = Nz;
        pl.yz.args = Z;
        seg_push(pl.yz.lseg, 0, Nz, 0);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// One rank, complex plan: input and output are both the natural [x][y][z] array, so the pass order is free (the
// reference's fft3d branch is one cuFFT plan, src/pencil/mpicufft_pencil_opt1.cpp:132-135).  Order z, x, y:
//   z pass   lines along z, 8 lines adjacent in x (rows a*AS_in + line*KS_in)  -> L1 = [y][kz/TL][x][kz%TL]   1 KiB runs
//   x pass   chunk (y, kz tile) of L1                                          -> L2 = rows of 128 B, PADDED
//   y pass   chunk (kx, kz tile) of L2                                         -> natural [kx][ky][kz]: a workgroup's
//            1024 rows of 128 B lie 16 KiB apart inside ONE 16 MiB plane (the x-last order puts them 16 MiB apart)
// L2 is private, so its row stride is made an odd multiple of 128 B (profiles/r2_placement_probe.txt: strided 128-byte
// rows whose stride is a multiple of 256 B lose 6 % as a scatter and 15 % as a gather on this part).
// The inverse runs the same launches with conjugation.
// ------------------------------------------------------------------------------------------
int build_pipeline_single(dfft_plan *p, Pipeline &pl)
{
    const int TL = p->TL;
    const size_t Nx = p->Nx, Ny = p->Ny, Nz = p->Nzc;
    pl.single = false;
    // measured (profiles/r2_single_order.txt): the y-last pass gains (2048^3 fp32: 34.7 -> 25.1 ms) but the z pass loses
    // its contiguous 128 KiB read (8 lines from 8 x planes instead): 1024^3 fp64 38.0-38.8 vs 37.5 ms per step, fp32 21.4
    // vs 20.1, 2048^3 fp32 181 vs 192 ms
    const int order = p->opt.single_order >= 0 ? p->opt.single_order : (p->prec == DFFT_F32 && Nx >= 2048 && Ny >= 2048 ? 1 : 0);
    if (p->nranks != 1 || !p->c2c || p->zyx || p->yzx || !order || p->opt.spectral) return 0;
    const size_t nb = (Nz + TL - 1) / TL;
    const size_t pad = (size_t)std::max(0, p->opt.single_pad) / p->esz;      // elements
    pl.sz = Launch(); pl.sx = Launch(); pl.sy = Launch();
    {   // z pass
        PassArgs Z = pass_args(TL, Ny, Nx, LOAD_LINES, STORE_TILED_TRANSPOSE, 0);
        Z.KS_in = (uint64_t)Ny * Nz;      // line stride: the 8 lines of a tile are adjacent in x
        Z.AS_in = Nz;                      // a = y
        Z.a_fastest = 1;                   // neighbouring workgroups read neighbouring (contiguous) lines
        pl.sz.args = Z;
        seg_push(pl.sz.sseg, 0, Nz, 0);
    }
    uint64_t SK, SB;
    if (p->opt.single_layout == 1) { SK = (uint64_t)TL * Ny + pad; SB = (uint64_t)Nx * SK + pad; pl.single_work_elems = nb * SB; }
    else { SK = (uint64_t)Nz * Ny + pad; SB = (uint64_t)TL * Ny; pl.single_work_elems = Nx * SK; }
    {   // x pass: L1 chunk (y, kz tile) -> L2
        PassArgs X = pass_args(TL, Ny, Nz, LOAD_TILED, STORE_TILED_SAME, 0);
        X.LA = (uint32_t)Ny; X.SK = SK; X.SB = SB;
        X.a_fastest = 1; X.xcd_swizzle = 1;      // neighbouring workgroups (y, y+1) write neighbouring 128-byte columns
        pl.sx.args = X;
        seg_push(pl.sx.lseg, 0, Nx, 0);
        seg_push(pl.sx.sseg, 0, Nx, 0);
    }
    {   // y pass: L2 chunk (kx, kz tile) -> natural output
        PassArgs Y = pass_args(TL, Nx, Nz, LOAD_TILED, STORE_KMAJOR, 0);
        Y.IA = SK; Y.IB = SB;
        Y.KS_out = Nz; Y.AS_out = (uint64_t)Ny * Nz;
        Y.a_fastest = 0; Y.xcd_swizzle = 1;      // neighbouring workgroups (kz tiles) write neighbouring 128-byte columns of a row
        if (!p->ax[1].bluestein && p->prec == DFFT_F64 && p->opt.shift != 0 && Y.AS_out % TL != 0 && Y.KS_out % TL == 0 && Y.LB >= (uint32_t)TL) {
            Y.shift = 1; Y.nb += 1; Y.ntiles = Y.na * Y.nb;      // odd row pitch: row-aligned tile windows (see set_shift)
        }
        pl.sy.args = Y;
        seg_push(pl.sy.lseg, 0, Ny, 0);
    }
    pl.single = true;
    return 0;
}

