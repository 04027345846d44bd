/* mpi_pencil.c -- the reference's pencil opt1 path restated for CPU ranks: ONE MPI PROCESS PER RANK.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE (like everything under oracle/): bench.py's cpu_baseline leg times it
 * with `mpiexec -n R` on the GPU box's host cores, tests/test_oracle.py checks it against the
 * in-process restatement (dfft_oracle.c: orc_pencil_forward with virtual ranks).  Never linked into the product.
 *
 * What it follows (reference paths):
 *   grid coordinates, MPI_Comm_split into row / column communicators   src/pencil/mpicufft_pencil_opt1.cpp:67-68, 103-104
 *   partition tables (remainder to the lowest ranks)                    :67-93            (orc_plan_create)
 *   all-to-all counts / displacements                                   :265-274, :311-320 (orc_plan_exchange_tables)
 *   forward chain  z-FFT -> MPI_Alltoallv(row) -> unpack -> y-FFT -> MPI_Alltoallv(column) -> unpack -> x-FFT   :1422-1519
 *   inverse chain, send / receive tables swapped                        :1522-1600, :829-830, :1341-1342
 *   unpack / pack geometry                                              :788-800, :1301-1312, :813-824, :1325-1336
 * The 1-D transforms are the oracle's own (orc_fft1d_many: the reference's arithmetic is closed-source cuFFT).
 *
 * usage: mpiexec -n P1*P2 ./mpi_pencil N P1 P2 iters [Ny Nz]      (C2C, fp64; OMP_NUM_THREADS should be 1)
 * prints one JSON line on rank 0: per-transform wall times (max over ranks), round-trip error, a checksum of the spectrum. */
#include <complex.h>
#include <math.h>
#include <mpi.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;
typedef struct orc_plan orc_plan;
orc_plan *orc_plan_create(size_t Nx, size_t Ny, size_t Nz, int P1, int P2, int c2c);
void orc_plan_destroy(orc_plan *pl);
void orc_plan_in_block(const orc_plan *pl, int rank, size_t size[3], size_t start[3]);
void orc_plan_out_block(const orc_plan *pl, int rank, size_t size[3], size_t start[3]);
size_t orc_plan_domain_elems(const orc_plan *pl, int rank);
void orc_plan_exchange_tables(const orc_plan *pl, int rank, int which, size_t *sc, size_t *sd, size_t *rc, size_t *rd);
void orc_fft1d_many(cplx *x, size_t n, size_t stride, size_t dist, size_t howmany, int sign);
void orc_fill_block(double *dst, size_t Ny, size_t Nz, size_t x0, size_t y0, size_t z0, size_t sx, size_t sy, size_t sz, int ncomp,
                    uint64_t seed);

typedef struct {
    size_t Nx, Ny, Nz;
    int P1, P2, i, j, rank;
    size_t xs, ys, zs, yo;                    /* my extents */
    size_t *xs_of, *xstart, *ys_of, *ystart;  /* peers' extents along the gathered axes */
    MPI_Comm row, col;                        /* (i, *) and (*, j) */
    int *sc1, *sd1, *rc1, *rd1, *sc2, *sd2, *rc2, *rd2;   /* in doubles (2 per element) */
    cplx *a, *b;                              /* two stage buffers of domain size */
} Rank;

static void a2a(Rank *R, int which, int forward, const cplx *send, cplx *recv)
{
    /* the inverse all-to-all swaps the send and receive tables (:829-830, :1341-1342) */
    int *sc = which == 1 ? R->sc1 : R->sc2, *sd = which == 1 ? R->sd1 : R->sd2;
    int *rc = which == 1 ? R->rc1 : R->rc2, *rd = which == 1 ? R->rd1 : R->rd2;
    MPI_Comm c = which == 1 ? R->row : R->col;
    if (forward) MPI_Alltoallv(send, sc, sd, MPI_DOUBLE, recv, rc, rd, MPI_DOUBLE, c);
    else MPI_Alltoallv(send, rc, rd, MPI_DOUBLE, recv, sc, sd, MPI_DOUBLE, c);
}

/* in: [xs][ys][Nz] (kept), out: [Nx][yo][zs]; a / b are scratch */
static void forward(Rank *R, const cplx *in, cplx *out)
{
    const size_t Nx = R->Nx, Ny = R->Ny, Nz = R->Nz, xs = R->xs, ys = R->ys, zs = R->zs, yo = R->yo;
    cplx *a = R->a, *b = R->b;
    /* z-FFT, output transposed [z][x][y] (plan :165-168) */
    size_t B = xs * ys;
    memcpy(b, in, sizeof(cplx) * B * Nz);
    orc_fft1d_many(b, Nz, 1, Nz, B, -1);
    for (size_t l = 0; l < B; l++) for (size_t k = 0; k < Nz; k++) a[k * B + l] = b[l * Nz + k];
    a2a(R, 1, 1, a, b);
    /* unpack [zs][xs][ys_p] -> [zs][xs][Ny] (:788-800), y-FFT -> [y][zs][xs] (:177-180) */
    for (int p = 0; p < R->P2; p++) {
        const cplx *blk = b + xs * R->ystart[p] * zs;
        for (size_t z = 0; z < zs; z++) for (size_t x = 0; x < xs; x++)
            memcpy(a + (z * xs + x) * Ny + R->ystart[p], blk + (z * xs + x) * R->ys_of[p], sizeof(cplx) * R->ys_of[p]);
    }
    B = zs * xs;
    orc_fft1d_many(a, Ny, 1, Ny, B, -1);
    for (size_t l = 0; l < B; l++) for (size_t k = 0; k < Ny; k++) b[k * B + l] = a[l * Ny + k];
    a2a(R, 2, 1, b, a);
    /* unpack [yo][zs][xs_p] -> [yo][zs][Nx] (:1301-1312), x-FFT -> [x][yo][zs] (:189-192) */
    for (int p = 0; p < R->P1; p++) {
        const cplx *blk = a + R->xstart[p] * yo * zs;
        for (size_t y = 0; y < yo; y++) for (size_t z = 0; z < zs; z++)
            memcpy(b + (y * zs + z) * Nx + R->xstart[p], blk + (y * zs + z) * R->xs_of[p], sizeof(cplx) * R->xs_of[p]);
    }
    B = yo * zs;
    orc_fft1d_many(b, Nx, 1, Nx, B, -1);
    for (size_t l = 0; l < B; l++) for (size_t k = 0; k < Nx; k++) out[k * B + l] = b[l * Nx + k];
}

/* in: [Nx][yo][zs] (destroyed, like execC2R's input: :1534,1544,1563), out: [xs][ys][Nz] */
static void inverse(Rank *R, cplx *in, cplx *out)
{
    const size_t Nx = R->Nx, Ny = R->Ny, Nz = R->Nz, xs = R->xs, ys = R->ys, zs = R->zs, yo = R->yo;
    cplx *a = R->a, *b = R->b;
    size_t B = yo * zs;
    for (size_t l = 0; l < B; l++) for (size_t k = 0; k < Nx; k++) a[l * Nx + k] = in[k * B + l];
    orc_fft1d_many(a, Nx, 1, Nx, B, +1);
    for (int p = 0; p < R->P1; p++) {          /* pack (:1325-1336) */
        cplx *blk = b + R->xstart[p] * yo * zs;
        for (size_t y = 0; y < yo; y++) for (size_t z = 0; z < zs; z++)
            memcpy(blk + (y * zs + z) * R->xs_of[p], a + (y * zs + z) * Nx + R->xstart[p], sizeof(cplx) * R->xs_of[p]);
    }
    a2a(R, 2, 0, b, a);                          /* a = [y][zs][xs] */
    B = zs * xs;
    for (size_t l = 0; l < B; l++) for (size_t k = 0; k < Ny; k++) b[l * Ny + k] = a[k * B + l];
    orc_fft1d_many(b, Ny, 1, Ny, B, +1);
    for (int p = 0; p < R->P2; p++) {          /* pack (:813-824) */
        cplx *blk = a + xs * R->ystart[p] * zs;
        for (size_t z = 0; z < zs; z++) for (size_t x = 0; x < xs; x++)
            memcpy(blk + (z * xs + x) * R->ys_of[p], b + (z * xs + x) * Ny + R->ystart[p], sizeof(cplx) * R->ys_of[p]);
    }
    a2a(R, 1, 0, a, b);                          /* b = [z][xs][ys] */
    B = xs * ys;
    for (size_t l = 0; l < B; l++) for (size_t k = 0; k < Nz; k++) out[l * Nz + k] = b[k * B + l];
    orc_fft1d_many(out, Nz, 1, Nz, B, +1);
}

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, size = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    if (argc < 5) { if (!rank) fprintf(stderr, "usage: mpi_pencil N P1 P2 iters [Ny Nz]\n"); MPI_Finalize(); return 2; }
    Rank R;
    memset(&R, 0, sizeof(R));
    R.Nx = (size_t)atoll(argv[1]);
    R.P1 = atoi(argv[2]); R.P2 = atoi(argv[3]);
    const int iters = atoi(argv[4]);
    R.Ny = argc > 5 ? (size_t)atoll(argv[5]) : R.Nx;
    R.Nz = argc > 6 ? (size_t)atoll(argv[6]) : R.Nx;
    if (R.P1 * R.P2 != size) { if (!rank) fprintf(stderr, "P1*P2 must equal the number of ranks\n"); MPI_Finalize(); return 2; }
    R.rank = rank; R.i = rank / R.P2; R.j = rank % R.P2;              /* pidx = pidx_i*P2 + pidx_j (:67-68) */
    MPI_Comm_split(MPI_COMM_WORLD, R.i, R.j, &R.row);                 /* :103 */
    MPI_Comm_split(MPI_COMM_WORLD, R.j, R.i, &R.col);                 /* :104 */
    orc_plan *pl = orc_plan_create(R.Nx, R.Ny, R.Nz, R.P1, R.P2, 1);
    size_t isz[3], ist[3], osz[3], ost[3], s3[3], t3[3];
    orc_plan_in_block(pl, rank, isz, ist);
    orc_plan_out_block(pl, rank, osz, ost);
    R.xs = isz[0]; R.ys = isz[1]; R.yo = osz[1]; R.zs = osz[2];
    R.xs_of = calloc(R.P1, sizeof(size_t)); R.xstart = calloc(R.P1, sizeof(size_t));
    R.ys_of = calloc(R.P2, sizeof(size_t)); R.ystart = calloc(R.P2, sizeof(size_t));
    for (int p = 0; p < R.P1; p++) { orc_plan_in_block(pl, p * R.P2 + R.j, s3, t3); R.xs_of[p] = s3[0]; R.xstart[p] = t3[0]; }
    for (int p = 0; p < R.P2; p++) { orc_plan_in_block(pl, R.i * R.P2 + p, s3, t3); R.ys_of[p] = s3[1]; R.ystart[p] = t3[1]; }
    size_t *t = calloc(4 * (size_t)(R.P1 > R.P2 ? R.P1 : R.P2), sizeof(size_t));
    int **dst1[4] = {&R.sc1, &R.sd1, &R.rc1, &R.rd1}, **dst2[4] = {&R.sc2, &R.sd2, &R.rc2, &R.rd2};
    for (int which = 1; which <= 2; which++) {
        const int n = which == 1 ? R.P2 : R.P1;
        orc_plan_exchange_tables(pl, rank, which, t, t + n, t + 2 * n, t + 3 * n);
        for (int q = 0; q < 4; q++) {
            int *v = calloc(n, sizeof(int));
            for (int p = 0; p < n; p++) {
                if (2 * t[q * n + p] > 0x7fffffffull) { fprintf(stderr, "message too large for MPI_Alltoallv's int counts\n"); MPI_Abort(MPI_COMM_WORLD, 3); }
                v[p] = (int)(2 * t[q * n + p]);                     /* elements -> doubles */
            }
            *(which == 1 ? dst1[q] : dst2[q]) = v;
        }
    }
    const size_t dom = orc_plan_domain_elems(pl, rank), nin = R.xs * R.ys * R.Nz;
    cplx *in = malloc(sizeof(cplx) * nin), *back = malloc(sizeof(cplx) * nin), *out = malloc(sizeof(cplx) * dom);
    R.a = malloc(sizeof(cplx) * dom); R.b = malloc(sizeof(cplx) * dom);
    orc_fill_block((double *)in, R.Ny, R.Nz, ist[0], ist[1], ist[2], isz[0], isz[1], isz[2], 2, 20260921);
    double tf = 0, tb = 0, sum_re = 0, sum_im = 0, sum_abs = 0;
    for (int it = -1; it < iters; it++) {            /* it = -1: warm-up */
        MPI_Barrier(MPI_COMM_WORLD);
        double t0 = MPI_Wtime();
        forward(&R, in, out);
        MPI_Barrier(MPI_COMM_WORLD);
        double t1 = MPI_Wtime();
        if (it == -1) {                              /* checksum of the spectrum block, weighted by the global index */
            for (size_t x = 0; x < osz[0]; x++) for (size_t y = 0; y < osz[1]; y++) for (size_t z = 0; z < osz[2]; z++) {
                const cplx v = out[(x * osz[1] + y) * osz[2] + z];
                const double w = 1.0 + (double)(((x * R.Ny + (ost[1] + y)) * R.Nz + (ost[2] + z)) % 1021) / 1021.0;
                sum_re += w * creal(v); sum_im += w * cimag(v); sum_abs += cabs(v);
            }
        }
        inverse(&R, out, back);
        MPI_Barrier(MPI_COMM_WORLD);
        double t2 = MPI_Wtime();
        if (it >= 0) { tf += t1 - t0; tb += t2 - t1; }
    }
    const double n3 = (double)R.Nx * R.Ny * R.Nz;
    double err = 0;
    for (size_t q = 0; q < nin; q++) { const double d = cabs(back[q] / n3 - in[q]); if (d > err) err = d; }
    double loc[4] = {sum_re, sum_im, sum_abs, 0}, glob[4], gerr = 0, tmax[2] = {tf, tb}, tg[2];
    MPI_Reduce(loc, glob, 4, MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
    MPI_Reduce(&err, &gerr, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    MPI_Reduce(tmax, tg, 2, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    if (!rank)
        printf("{\"ranks\": %d, \"P1\": %d, \"P2\": %d, \"grid\": [%zu, %zu, %zu], \"iters\": %d, \"forward_ms\": %.3f, \"inverse_ms\": %.3f, "
               "\"round_trip_rel_linf\": %.3e, \"checksum\": [%.17g, %.17g, %.17g]}\n",
               size, R.P1, R.P2, R.Nx, R.Ny, R.Nz, iters, iters ? tf / iters * 1e3 : 0.0, iters ? tb / iters * 1e3 : 0.0, gerr / 255.0, glob[0], glob[1],
               glob[2]);
    orc_plan_destroy(pl);
    MPI_Finalize();
    return 0;
}
