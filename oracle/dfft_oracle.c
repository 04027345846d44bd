/*
 * oracle/dfft_oracle.c -- CPU oracle for the distributed 3-D FFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under distributedfft_amd/ may include, link or call
 * this file.  Allowed users: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * What it restates.  The reference (eggersn/DistributedFFT) owns the *decomposition*:
 * partition tables, per-stage layouts, all-to-all counts/displacements, unpack geometry and
 * the exec order.  All butterfly arithmetic in the reference is NVIDIA cuFFT (closed source,
 * not vendored; include/cufft.hpp:23-61), so the 1-D transform below restates cuFFT's
 * *published contract* instead: unnormalised DFT, forward kernel exp(-2*pi*i*j*k/N), inverse
 * exp(+...), R2C keeps k = 0..N/2 of the last axis (include/params.hpp:30).
 *
 * PARITY PINNING STATUS.  PINNED against outputs of the reference itself, as far as the reference holds any (round 6):
 *   (1) KNOWN ANSWERS THE REFERENCE PRODUCED: testcase 4 has a deterministic input (u = sin sin sin,
 *       tests/src/pencil/random_dist_3D.cu:748-762), and the reference ships the logs of its own runs (benchmarks/argon and
 *       benchmarks/pcsgs: `Result (avg)` / `Result (max)` per grid, decomposition and option, double precision, 4 ranks, cuFFT +
 *       its kernels).  tests/golden/ref_testcase4_results.json holds them; this file's decomposed R2C -> derivativeCoefficients
 *       (the reference's arithmetic to the letter, single-precision root included) -> decomposed C2R prints the same average in
 *       all six digits (1.91723e-05 at 128^3) and a maximum inside the logs' own scatter
 *       (tests/test_oracle.py::test_testcase4_reproduces_the_references_own_shipped_results); the HIP path and the C++ drivers are
 *       held to the same numbers at 128^3 and 512^3 (tests/test_gpu_parity.py, tests/test_gpu_cpp_drivers.py);
 *   (2) the reference's own code where it builds here without stand-ins: src/timer.cpp (oracle/_ref: the timer CSV byte for byte)
 *       and include/params.hpp (struct layout of the boundary); its launcher launch.py (the command lines the drivers must accept)
 *       and the 25 000 timer CSV files it ships (section labels, file names);
 *   (3) for inputs the reference holds nothing for -- it seeds its random inputs with clock() (tests/src/pencil/base.cu:49), owns no
 *       arithmetic (closed cuFFT) and cannot be built here (CUDA, cuFFT, OpenMPI's mpi-ext.h; stand-ins are not allowed) --
 *       bit patterns against cuFFT stay UNPINNED, and the oracle is pinned by (a) a long-double O(N^2) DFT for every 1-D length
 *       used in tests, (b) numpy.fft (pocketfft, independent implementation) on 3-D grids, and (c) the properties the reference's
 *       own tests check at this boundary: testcase 3 round trip (:641-666) and testcase 1 distributed == single device (:386-403).
 *   See tests/test_oracle.py, tests/test_ref_timer.py, tests/test_launch_commands.py, tests/test_ref_csv_shapes.py.
 *
 * Citations are paths relative to /root/reference.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

#ifndef M_PIl
#define M_PIl 3.141592653589793238462643383279502884L
#endif

/* ------------------------------------------------------------------------------------------
 * 1-D transforms
 * ---------------------------------------------------------------------------------------- */

/* Ground truth of the ground truth: O(N^2) DFT with long-double twiddles and accumulation.
 * sign = -1 forward, +1 inverse (cuFFT CUFFT_FORWARD / CUFFT_INVERSE), unnormalised. */
void orc_dft_naive(const cplx *in, cplx *out, size_t n, int sign)
{
    for (size_t k = 0; k < n; k++) {
        long double sr = 0.0L, si = 0.0L;
        for (size_t j = 0; j < n; j++) {
            /* reduce j*k mod n first so the angle stays accurate */
            size_t jk = (size_t)(((unsigned __int128)j * k) % n);
            long double a = (long double)sign * 2.0L * M_PIl * (long double)jk / (long double)n;
            long double c = cosl(a), s = sinl(a);
            long double xr = creal(in[j]), xi = cimag(in[j]);
            sr += xr * c - xi * s;
            si += xr * s + xi * c;
        }
        out[k] = (double)sr + (double)si * I;
    }
}

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

/* twiddle table w[j] = exp(sign*2*pi*i*j/n), long-double evaluated, rounded once */
static cplx *make_twiddles(size_t n, int sign)
{
    cplx *w = (cplx *)malloc(sizeof(cplx) * (n ? n : 1));
    for (size_t j = 0; j < n; j++) {
        long double a = (long double)sign * 2.0L * M_PIl * (long double)j / (long double)n;
        w[j] = (double)cosl(a) + (double)sinl(a) * I;
    }
    return w;
}

/* iterative radix-2 decimation-in-time on a contiguous buffer, n a power of two */
static void fft_pow2(cplx *x, size_t n, const cplx *w /* n entries */)
{
    /* bit reversal */
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { cplx t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; k++) {
                cplx u = x[i + k], v = x[i + k + half] * w[k * step];
                x[i + k] = u + v;
                x[i + k + half] = u - v;
            }
    }
}

/* generic length: recursive mixed radix (smallest prime factor), naive DFT on primes */
static void fft_any_rec(const cplx *in, size_t istride, cplx *out, size_t n, size_t wstep,
                        const cplx *w, size_t wn)
{
    if (n == 1) { out[0] = in[0]; return; }
    size_t p = 0;
    for (size_t f = 2; f * f <= n; f++) if (n % f == 0) { p = f; break; }
    if (!p) p = n;
    size_t m = n / p;
    /* p sub-transforms of length m over the decimated inputs */
    for (size_t r = 0; r < p; r++)
        fft_any_rec(in + r * istride, istride * p, out + r * m, m, wstep * p, w, wn);
    /* combine: X[k + m*q] = sum_r w_n^{r(k+mq)} S_r[k] */
    cplx *tmp = (cplx *)malloc(sizeof(cplx) * p);
    for (size_t k = 0; k < m; k++) {
        for (size_t r = 0; r < p; r++) tmp[r] = out[r * m + k];
        for (size_t q = 0; q < p; q++) {
            size_t kk = k + m * q;
            cplx s = 0;
            for (size_t r = 0; r < p; r++) s += tmp[r] * w[((r * kk) % n) * wstep % wn];
            out[kk] = s;
        }
    }
    free(tmp);
}

typedef struct { size_t n; int sign; cplx *w; cplx *buf, *buf2; } line_plan;

static void line_plan_init(line_plan *p, size_t n, int sign)
{
    p->n = n; p->sign = sign; p->w = make_twiddles(n, sign);
    p->buf = (cplx *)malloc(sizeof(cplx) * (n ? n : 1));
    p->buf2 = (cplx *)malloc(sizeof(cplx) * (n ? n : 1));
}
static void line_plan_free(line_plan *p) { free(p->w); free(p->buf); free(p->buf2); }

/* transform one strided line in place */
static void line_exec(line_plan *p, cplx *x, size_t stride)
{
    size_t n = p->n;
    if (n <= 1) return;
    if (is_pow2(n)) {
        if (stride == 1) { fft_pow2(x, n, p->w); return; }
        for (size_t j = 0; j < n; j++) p->buf[j] = x[j * stride];
        fft_pow2(p->buf, n, p->w);
        for (size_t j = 0; j < n; j++) x[j * stride] = p->buf[j];
    } else {
        for (size_t j = 0; j < n; j++) p->buf[j] = x[j * stride];
        fft_any_rec(p->buf, 1, p->buf2, n, 1, p->w, n);
        for (size_t j = 0; j < n; j++) x[j * stride] = p->buf2[j];
    }
}

/* public: in-place 1-D transform of `howmany` lines: line b starts at x + b*dist, elements
 * `stride` apart.  sign -1 forward / +1 inverse.  Unnormalised. */
void orc_fft1d_many(cplx *x, size_t n, size_t stride, size_t dist, size_t howmany, int sign)
{
#pragma omp parallel
    {
        line_plan lp; line_plan_init(&lp, n, sign);
#pragma omp for schedule(static)
        for (long long b = 0; b < (long long)howmany; b++) line_exec(&lp, x + (size_t)b * dist, stride);
        line_plan_free(&lp);
    }
}

/* ------------------------------------------------------------------------------------------
 * single-rank 3-D transforms, C order [x][y][z], z contiguous (README.md:247)
 * ---------------------------------------------------------------------------------------- */

/* lines of length n, element stride `stride`; line (i, j) starts at x + i*dist1 + j*dist2.
 * One parallel region and one twiddle table per thread for the whole batch.  Strided power-of-two
 * lines whose neighbours (j, j+1) are adjacent in memory are gathered four at a time, so that every
 * 64-byte cache line fetched is used in full (the arithmetic per line is unchanged: same fft_pow2). */
#define ORC_BLK 4
static void fft_lines_2d(cplx *x, size_t n, size_t stride, size_t n1, size_t dist1, size_t n2, size_t dist2, int sign)
{
    const int blocked = stride > 1 && dist2 == 1 && is_pow2(n) && n > 1;
#pragma omp parallel
    {
        line_plan lp; line_plan_init(&lp, n, sign);
        if (!blocked) {
#pragma omp for schedule(static) collapse(2)
            for (long long i = 0; i < (long long)n1; i++)
                for (long long j = 0; j < (long long)n2; j++)
                    line_exec(&lp, x + (size_t)i * dist1 + (size_t)j * dist2, stride);
        } else {
            cplx *blk = (cplx *)malloc(sizeof(cplx) * n * ORC_BLK);
            const long long nblk = (long long)((n2 + ORC_BLK - 1) / ORC_BLK);
#pragma omp for schedule(static) collapse(2)
            for (long long i = 0; i < (long long)n1; i++)
                for (long long jb = 0; jb < nblk; jb++) {
                    cplx *base = x + (size_t)i * dist1 + (size_t)jb * ORC_BLK;
                    const size_t w = (size_t)jb * ORC_BLK + ORC_BLK <= n2 ? ORC_BLK : n2 - (size_t)jb * ORC_BLK;
                    for (size_t k = 0; k < n; k++)
                        for (size_t b = 0; b < w; b++) blk[b * n + k] = base[k * stride + b];
                    for (size_t b = 0; b < w; b++) fft_pow2(blk + b * n, n, lp.w);
                    for (size_t k = 0; k < n; k++)
                        for (size_t b = 0; b < w; b++) base[k * stride + b] = blk[b * n + k];
                }
            free(blk);
        }
        line_plan_free(&lp);
    }
}

void orc_fft3d_c2c(cplx *a, size_t Nx, size_t Ny, size_t Nz, int sign)
{
    fft_lines_2d(a, Nz, 1, Nx, Ny * Nz, Ny, Nz, sign);          /* z lines */
    fft_lines_2d(a, Ny, Nz, Nx, Ny * Nz, Nz, 1, sign);          /* y lines */
    fft_lines_2d(a, Nx, Ny * Nz, Ny, Nz, Nz, 1, sign);          /* x lines */
}

/* R2C: in real [x][y][Nz] -> out complex [x][y][Nz/2+1] (cufftMakePlan3d R2C semantics,
 * src/pencil/mpicufft_pencil_opt1.cpp:133) */
void orc_fft3d_r2c(const double *in, cplx *out, size_t Nx, size_t Ny, size_t Nz)
{
    size_t Nzc = Nz / 2 + 1;
#pragma omp parallel
    {
        line_plan lp; line_plan_init(&lp, Nz, -1);
        cplx *line = (cplx *)malloc(sizeof(cplx) * Nz);
#pragma omp for schedule(static)
        for (long long b = 0; b < (long long)(Nx * Ny); b++) {
            for (size_t z = 0; z < Nz; z++) line[z] = in[(size_t)b * Nz + z];
            line_exec(&lp, line, 1);
            memcpy(out + (size_t)b * Nzc, line, sizeof(cplx) * Nzc);
        }
        free(line); line_plan_free(&lp);
    }
    for (size_t x = 0; x < Nx; x++) orc_fft1d_many(out + x * Ny * Nzc, Ny, Nzc, 1, Nzc, -1);
    orc_fft1d_many(out, Nx, Ny * Nzc, 1, Ny * Nzc, -1);
}

/* C2R: in complex [x][y][Nz/2+1] (destroyed, like the reference's execC2R,
 * src/pencil/mpicufft_pencil_opt1.cpp:1534) -> out real [x][y][Nz], unnormalised */
void orc_fft3d_c2r(cplx *in, double *out, size_t Nx, size_t Ny, size_t Nz)
{
    size_t Nzc = Nz / 2 + 1;
    orc_fft1d_many(in, Nx, Ny * Nzc, 1, Ny * Nzc, +1);
    for (size_t x = 0; x < Nx; x++) orc_fft1d_many(in + x * Ny * Nzc, Ny, Nzc, 1, Nzc, +1);
#pragma omp parallel
    {
        line_plan lp; line_plan_init(&lp, Nz, +1);
        cplx *line = (cplx *)malloc(sizeof(cplx) * Nz);
#pragma omp for schedule(static)
        for (long long b = 0; b < (long long)(Nx * Ny); b++) {
            const cplx *src = in + (size_t)b * Nzc;
            /* Hermitian extension; imaginary parts of k=0 (and k=Nz/2 for even Nz) ignored */
            for (size_t k = 0; k < Nzc; k++) line[k] = src[k];
            for (size_t k = Nzc; k < Nz; k++) line[k] = conj(src[Nz - k]);
            line[0] = creal(line[0]);
            if (Nz % 2 == 0) line[Nz / 2] = creal(line[Nz / 2]);
            line_exec(&lp, line, 1);
            for (size_t z = 0; z < Nz; z++) out[(size_t)b * Nz + z] = creal(line[z]);
        }
        free(line); line_plan_free(&lp);
    }
}

/* ------------------------------------------------------------------------------------------
 * Decomposition restatement: pencil opt1 ("Realigned") with P1 x P2 virtual ranks in one
 * process; slab ZY_Then_X is the P2 == 1 case of the same algebra (in [Nx/P][Ny][Nz], out
 * [Nx][Ny/P][Nzc]; src/slab/default/mpicufft_slab_opt1.cpp:95-112).
 *
 * Partition tables:      src/pencil/mpicufft_pencil_opt1.cpp:67-93
 * plan strides/layouts:  :151-197      (z: [x][y][z] -> [z][x][y]; y: -> [y][z'][x]; x: -> [x][y'][z'])
 * all-to-all tables:     :265-274 (first), :311-320 (second)
 * unpack geometry:       :788-800 (first), :1301-1312 (second)
 * exec order:            :1422-1519 forward, :1522-1600 inverse
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    size_t Nx, Ny, Nz, Nzc;   /* Nzc = Nz/2+1 (R2C) or Nz (C2C mode) */
    int P1, P2, c2c;
    /* input_dim.size_x/start_x (P1), input_dim.size_y/start_y (P2),
       transposed_dim.size_z/start_z (P2), output_dim.size_y/start_y (P1) */
    size_t *xs, *xstart, *ys, *ystart, *zs, *zstart, *yo, *yostart;
} orc_plan;

static void split(size_t n, int p, size_t *size, size_t *start)
{
    /* remainder goes to the lowest ranks (mpicufft_pencil_opt1.cpp:71-73) */
    size_t off = 0;
    for (int i = 0; i < p; i++) {
        size[i] = n / p + ((size_t)i < n % p ? 1 : 0);
        start[i] = off; off += size[i];
    }
}

orc_plan *orc_plan_create(size_t Nx, size_t Ny, size_t Nz, int P1, int P2, int c2c)
{
    orc_plan *pl = (orc_plan *)calloc(1, sizeof(orc_plan));
    pl->Nx = Nx; pl->Ny = Ny; pl->Nz = Nz; pl->c2c = c2c;
    pl->Nzc = c2c ? Nz : Nz / 2 + 1;
    pl->P1 = P1; pl->P2 = P2;
    pl->xs = calloc(P1, sizeof(size_t)); pl->xstart = calloc(P1, sizeof(size_t));
    pl->yo = calloc(P1, sizeof(size_t)); pl->yostart = calloc(P1, sizeof(size_t));
    pl->ys = calloc(P2, sizeof(size_t)); pl->ystart = calloc(P2, sizeof(size_t));
    pl->zs = calloc(P2, sizeof(size_t)); pl->zstart = calloc(P2, sizeof(size_t));
    split(Nx, P1, pl->xs, pl->xstart);
    split(Ny, P2, pl->ys, pl->ystart);
    split(pl->Nzc, P2, pl->zs, pl->zstart);
    split(Ny, P1, pl->yo, pl->yostart);
    return pl;
}

void orc_plan_destroy(orc_plan *pl)
{
    free(pl->xs); free(pl->xstart); free(pl->ys); free(pl->ystart);
    free(pl->zs); free(pl->zstart); free(pl->yo); free(pl->yostart); free(pl);
}

/* getInSize/getInStart/getOutSize/getOutStart (include/mpicufft_pencil.hpp:112-122, with
 * the start_x -> start_z fix noted in SURVEY.md section 7).  rank = i*P2 + j. */
void orc_plan_in_block(const orc_plan *pl, int rank, size_t size[3], size_t start[3])
{
    int i = rank / pl->P2, j = rank % pl->P2;
    size[0] = pl->xs[i]; size[1] = pl->ys[j]; size[2] = pl->Nz;
    start[0] = pl->xstart[i]; start[1] = pl->ystart[j]; start[2] = 0;
}
void orc_plan_out_block(const orc_plan *pl, int rank, size_t size[3], size_t start[3])
{
    int i = rank / pl->P2, j = rank % pl->P2;
    size[0] = pl->Nx; size[1] = pl->yo[i]; size[2] = pl->zs[j];
    start[0] = 0; start[1] = pl->yostart[i]; start[2] = pl->zstart[j];
}
/* element count a rank's `out` buffer must hold = max of the three stage sizes
 * (include/mpicufft_pencil.hpp:94-100) */
size_t orc_plan_domain_elems(const orc_plan *pl, int rank)
{
    int i = rank / pl->P2, j = rank % pl->P2;
    size_t a = pl->xs[i] * pl->ys[j] * pl->Nzc;
    size_t b = pl->xs[i] * pl->Ny * pl->zs[j];
    size_t c = pl->Nx * pl->yo[i] * pl->zs[j];
    size_t m = a > b ? a : b; return m > c ? m : c;
}

/* Exchange tables in ELEMENTS for rank (i,j).  which = 1: row comm (P2 peers, :269-273),
 * which = 2: column comm (P1 peers, :315-319).  Arrays must hold P2 resp. P1 entries. */
void orc_plan_exchange_tables(const orc_plan *pl, int rank, int which,
                              size_t *scount, size_t *sdispl, size_t *rcount, size_t *rdispl)
{
    int i = rank / pl->P2, j = rank % pl->P2;
    if (which == 1) {
        for (int p = 0; p < pl->P2; p++) {
            scount[p] = pl->zs[p] * pl->ys[j] * pl->xs[i];
            sdispl[p] = pl->zstart[p] * pl->ys[j] * pl->xs[i];
            rcount[p] = pl->xs[i] * pl->ys[p] * pl->zs[j];
            rdispl[p] = pl->xs[i] * pl->ystart[p] * pl->zs[j];
        }
    } else {
        for (int p = 0; p < pl->P1; p++) {
            scount[p] = pl->xs[i] * pl->zs[j] * pl->yo[p];
            sdispl[p] = pl->xs[i] * pl->zs[j] * pl->yostart[p];
            rcount[p] = pl->xs[p] * pl->yo[i] * pl->zs[j];
            rdispl[p] = pl->xstart[p] * pl->yo[i] * pl->zs[j];
        }
    }
}

/* Forward transform over all P1*P2 virtual ranks.
 *  in[r]  : rank r's input block [xs][ys][Nz]; real (double*) in R2C mode, cplx* in C2C mode
 *  out[r] : rank r's output buffer, >= orc_plan_domain_elems() complex; on return holds
 *           [Nx][yo][zs] (x complete), exactly the reference's opt1 output layout.
 * Every intermediate buffer uses the reference's opt1 layout and the all-to-all is driven by
 * orc_plan_exchange_tables(), i.e. it is the reference's MPI_Alltoallv restated as memcpy
 * between virtual ranks. */
void orc_pencil_forward(const orc_plan *pl, void *const *in, cplx *const *out)
{
    int P1 = pl->P1, P2 = pl->P2, P = P1 * P2;
    size_t Nx = pl->Nx, Ny = pl->Ny, Nz = pl->Nz, Nzc = pl->Nzc;
    cplx **recv = calloc(P, sizeof(cplx *)), **temp = calloc(P, sizeof(cplx *));
    for (int r = 0; r < P; r++) {
        size_t n = orc_plan_domain_elems(pl, r);
        recv[r] = malloc(sizeof(cplx) * n); temp[r] = malloc(sizeof(cplx) * n);
    }
    /* 1. z-FFT, output transposed [z][x][y] (plan :165-168) */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        size_t xs = pl->xs[i], ys = pl->ys[j], B = xs * ys;
        cplx *o = out[r];
#pragma omp parallel
        {
            line_plan lp; line_plan_init(&lp, Nz, -1);
            cplx *line = malloc(sizeof(cplx) * Nz);
#pragma omp for schedule(static)
            for (long long b = 0; b < (long long)B; b++) {
                if (pl->c2c) memcpy(line, (const cplx *)in[r] + (size_t)b * Nz, sizeof(cplx) * Nz);
                else for (size_t z = 0; z < Nz; z++) line[z] = ((const double *)in[r])[(size_t)b * Nz + z];
                line_exec(&lp, line, 1);
                for (size_t k = 0; k < Nzc; k++) o[k * B + (size_t)b] = line[k];
            }
            free(line); line_plan_free(&lp);
        }
    }
    /* 2. first all-to-all, row communicator {(i, p)} (:784-785) */
    size_t sc[64], sd[64], rc[64], rd[64], sc2[64], sd2[64], rc2[64], rd2[64];
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        orc_plan_exchange_tables(pl, r, 1, sc, sd, rc, rd);
        for (int p = 0; p < P2; p++) {
            int src = i * P2 + p;
            orc_plan_exchange_tables(pl, src, 1, sc2, sd2, rc2, rd2);
            /* what src sends to me (its slot j) lands at my rdispl[p] */
            memcpy(recv[r] + rd[p], out[src] + sd2[j], sizeof(cplx) * rc[p]);
        }
    }
    /* 3. unpack [zs][xs][ys_p] -> temp [zs][xs][Ny] (:788-800), 4. y-FFT -> out [y][zs][xs] (:177-180) */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        size_t xs = pl->xs[i], zs = pl->zs[j];
        for (int p = 0; p < P2; p++) {
            const cplx *blk = recv[r] + xs * pl->ystart[p] * zs;
            for (size_t z = 0; z < zs; z++) for (size_t x = 0; x < xs; x++)
                memcpy(temp[r] + (z * xs + x) * Ny + pl->ystart[p],
                       blk + (z * xs + x) * pl->ys[p], sizeof(cplx) * pl->ys[p]);
        }
        size_t B = zs * xs;
        orc_fft1d_many(temp[r], Ny, 1, Ny, B, -1);
        for (size_t b = 0; b < B; b++) for (size_t k = 0; k < Ny; k++) out[r][k * B + b] = temp[r][b * Ny + k];
    }
    /* 5. second all-to-all, column communicator {(p, j)} (:1297-1298) */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        orc_plan_exchange_tables(pl, r, 2, sc, sd, rc, rd);
        for (int p = 0; p < P1; p++) {
            int src = p * P2 + j;
            orc_plan_exchange_tables(pl, src, 2, sc2, sd2, rc2, rd2);
            memcpy(recv[r] + rd[p], out[src] + sd2[i], sizeof(cplx) * rc[p]);
        }
    }
    /* 6. unpack [yo][zs][xs_p] -> [yo][zs][Nx] (:1301-1312), 7. x-FFT -> [x][yo][zs] (:189-192) */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        size_t yo = pl->yo[i], zs = pl->zs[j];
        for (int p = 0; p < P1; p++) {
            const cplx *blk = recv[r] + pl->xstart[p] * yo * zs;
            for (size_t y = 0; y < yo; y++) for (size_t z = 0; z < zs; z++)
                memcpy(temp[r] + (y * zs + z) * Nx + pl->xstart[p],
                       blk + (y * zs + z) * pl->xs[p], sizeof(cplx) * pl->xs[p]);
        }
        size_t B = yo * zs;
        orc_fft1d_many(temp[r], Nx, 1, Nx, B, -1);
        for (size_t b = 0; b < B; b++) for (size_t k = 0; k < Nx; k++) out[r][k * B + b] = temp[r][b * Nx + k];
    }
    for (int r = 0; r < P; r++) { free(recv[r]); free(temp[r]); }
    free(recv); free(temp);
}

/* Inverse: in[r] = [Nx][yo][zs] spectrum (destroyed), out[r] = [xs][ys][Nz] real (R2C mode)
 * or complex (C2C mode).  Mirror of the forward chain (:1522-1600; packs :813-824, :1325-1336). */
void orc_pencil_inverse(const orc_plan *pl, cplx *const *in, void *const *out)
{
    int P1 = pl->P1, P2 = pl->P2, P = P1 * P2;
    size_t Nx = pl->Nx, Ny = pl->Ny, Nz = pl->Nz, Nzc = pl->Nzc;
    cplx **send = calloc(P, sizeof(cplx *)), **temp = calloc(P, sizeof(cplx *));
    for (int r = 0; r < P; r++) {
        size_t n = orc_plan_domain_elems(pl, r);
        send[r] = malloc(sizeof(cplx) * n); temp[r] = malloc(sizeof(cplx) * n);
    }
    size_t sc[64], sd[64], rc[64], rd[64], sc2[64], sd2[64], rc2[64], rd2[64];
    /* x^-1: in [x][yo][zs] (stride B) -> temp [yo][zs][Nx]; pack -> send [p][yo][zs][xs_p] */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        size_t yo = pl->yo[i], zs = pl->zs[j], B = yo * zs;
        for (size_t b = 0; b < B; b++) for (size_t k = 0; k < Nx; k++) temp[r][b * Nx + k] = in[r][k * B + b];
        orc_fft1d_many(temp[r], Nx, 1, Nx, B, +1);
        for (int p = 0; p < P1; p++) {
            cplx *blk = send[r] + pl->xstart[p] * yo * zs;
            for (size_t y = 0; y < yo; y++) for (size_t z = 0; z < zs; z++)
                memcpy(blk + (y * zs + z) * pl->xs[p],
                       temp[r] + (y * zs + z) * Nx + pl->xstart[p], sizeof(cplx) * pl->xs[p]);
        }
    }
    /* all-to-all in the column comm with send/recv tables swapped (:1341-1342) */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        orc_plan_exchange_tables(pl, r, 2, sc, sd, rc, rd);
        for (int p = 0; p < P1; p++) {
            int src = p * P2 + j;
            orc_plan_exchange_tables(pl, src, 2, sc2, sd2, rc2, rd2);
            /* src sends its recv-slot i (rd2[i], rc2[i]); I receive into my send-slot p */
            memcpy(in[r] + sd[p], send[src] + rd2[i], sizeof(cplx) * sc[p]);
        }
    }
    /* y^-1: in[r] = [y][zs][xs] -> temp [zs][xs][Ny]; pack -> send [p][zs][xs][ys_p] */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        size_t xs = pl->xs[i], zs = pl->zs[j], B = zs * xs;
        for (size_t b = 0; b < B; b++) for (size_t k = 0; k < Ny; k++) temp[r][b * Ny + k] = in[r][k * B + b];
        orc_fft1d_many(temp[r], Ny, 1, Ny, B, +1);
        for (int p = 0; p < P2; p++) {
            cplx *blk = send[r] + xs * pl->ystart[p] * zs;
            for (size_t z = 0; z < zs; z++) for (size_t x = 0; x < xs; x++)
                memcpy(blk + (z * xs + x) * pl->ys[p],
                       temp[r] + (z * xs + x) * Ny + pl->ystart[p], sizeof(cplx) * pl->ys[p]);
        }
    }
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        orc_plan_exchange_tables(pl, r, 1, sc, sd, rc, rd);
        for (int p = 0; p < P2; p++) {
            int src = i * P2 + p;
            orc_plan_exchange_tables(pl, src, 1, sc2, sd2, rc2, rd2);
            memcpy(in[r] + sd[p], send[src] + rd2[j], sizeof(cplx) * sc[p]);
        }
    }
    /* z^-1: in[r] = [z][xs][ys] -> out [xs][ys][Nz] */
    for (int r = 0; r < P; r++) {
        int i = r / P2, j = r % P2;
        size_t xs = pl->xs[i], ys = pl->ys[j], B = xs * ys;
#pragma omp parallel
        {
            line_plan lp; line_plan_init(&lp, Nz, +1);
            cplx *line = malloc(sizeof(cplx) * Nz);
#pragma omp for schedule(static)
            for (long long b = 0; b < (long long)B; b++) {
                for (size_t k = 0; k < Nzc; k++) line[k] = in[r][k * B + (size_t)b];
                if (!pl->c2c) {
                    for (size_t k = Nzc; k < Nz; k++) line[k] = conj(in[r][(Nz - k) * B + (size_t)b]);
                    line[0] = creal(line[0]);
                    if (Nz % 2 == 0) line[Nz / 2] = creal(line[Nz / 2]);
                }
                line_exec(&lp, line, 1);
                if (pl->c2c) memcpy((cplx *)out[r] + (size_t)b * Nz, line, sizeof(cplx) * Nz);
                else for (size_t z = 0; z < Nz; z++) ((double *)out[r])[(size_t)b * Nz + z] = creal(line[z]);
            }
            free(line); line_plan_free(&lp);
        }
    }
    for (int r = 0; r < P; r++) { free(send[r]); free(temp[r]); }
    free(send); free(temp);
}

/* ------------------------------------------------------------------------------------------
 * Reference test helpers restated
 * ---------------------------------------------------------------------------------------- */

/* Deterministic input: uniform [0,255) from splitmix64(global index, seed); mirrors the
 * reference's cuRAND-uniform x 255 (tests/src/pencil/base.cu:45-53) but reproducible and
 * decomposition independent (value depends on the GLOBAL linear index only). */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
double orc_uniform255(uint64_t global_index, uint64_t seed)
{
    uint64_t h = splitmix64(global_index ^ splitmix64(seed));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0) * 255.0;
}
/* fill a sub-block [sx][sy][sz] at global offset (x0,y0,z0) of an [Nx][Ny][Nz] grid;
 * ncomp = 1 real, 2 interleaved complex */
void orc_fill_block(double *dst, size_t Ny, size_t Nz, size_t x0, size_t y0, size_t z0,
                    size_t sx, size_t sy, size_t sz, int ncomp, uint64_t seed)
{
#pragma omp parallel for schedule(static) collapse(2)
    for (long long x = 0; x < (long long)sx; x++) for (long long y = 0; y < (long long)sy; y++) for (size_t z = 0; z < sz; z++) {
        uint64_t g = ((uint64_t)(x0 + (size_t)x) * Ny + (y0 + (size_t)y)) * Nz + (z0 + z);
        double *d = dst + (((size_t)x * sy + (size_t)y) * sz + z) * ncomp;
        for (int c = 0; c < ncomp; c++) d[c] = orc_uniform255(g * 2 + c, seed);
    }
}

/* derivativeCoefficients (tests/src/pencil/random_dist_3D.cu:98-121) on a distributed output
 * block [Nx][N2][N1] with offsets; half = 1 for the Hermitian-halved last axis (k3 = z below
 * Nz/2 else 0, :114), half = 0 applies the wrapped wavenumber on z as on x,y (C2C mode). */
void orc_derivative_coefficients(cplx *out, size_t Nx, size_t Ny, size_t Nz, size_t Nz_offset,
                                 size_t Ny_offset, size_t N1, size_t N2, int half)
{
    /* the reference's arithmetic to the letter (tests/src/pencil/random_dist_3D.cu:98-121, the double overload): the divisor is
     * sqrtf -- SINGLE precision -- of the int product Nx*Ny*Nz, the numerator -powf(k1,2)-powf(k2,2)-powf(k3,2) (exact in float
     * for these sizes), and each component is x * scale / root in double.  The float root is why the reference's own runs of
     * testcase 4 print 1.91723e-05 / 7.43e-05 at 128^3 and 1.53465e-04 at 512^3 (3 |N^3/sqrtf(N^3) - sqrt(N^3)| times mean / max |u|;
     * the logs it ships under benchmarks/argon and benchmarks/pcsgs, tests/golden/ref_testcase4_results.json) and ~1e-8 where N^3 is a power of 4 */
    const double root = (double)sqrtf((float)(int)(Nx * Ny * Nz));
    for (size_t x = 0; x < Nx; x++) for (size_t y = 0; y < N2; y++) for (size_t z = 0; z < N1; z++) {
        double k1 = 0, k2 = 0, k3 = 0;
        size_t gy = y + Ny_offset, gz = z + Nz_offset;
        if (x < Nx / 2) k1 = (double)x; else if (x > Nx / 2) k1 = (double)(Nx - x);
        if (gy < Ny / 2) k2 = (double)gy; else if (gy > Ny / 2) k2 = (double)(Ny - gy);
        if (gz < Nz / 2) k3 = (double)gz; else if (!half && gz > Nz / 2) k3 = (double)(Nz - gz);
        const double scale = (double)(-(float)(k1 * k1) - (float)(k2 * k2) - (float)(k3 * k3));
        cplx *v = &out[(x * N2 + y) * N1 + z];
        *v = (creal(*v) * scale / root) + I * (cimag(*v) * scale / root);
    }
}

/* (bench.py's cpu_baseline leg: a job whose cgroup grants fewer CPUs than the host has cores should not run 128 threads on 16) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
