// fft_pass.hip.h -- the one hot kernel of the library: a batched 1-D complex FFT "axis pass"
// for gfx950 (MI355X / CDNA4) with the layout change of the distributed algorithm fused into
// its load and store sides.
//
// Replaces, per axis pass, one cuFFT plan of the reference plus the cudaMemcpy2D/3D pack or
// unpack next to it:
//   cufftMakePlanMany64 z/y/x plans  src/pencil/mpicufft_pencil_opt1.cpp:165-197
//   unpack after exchange 1 / 2      src/pencil/mpicufft_pencil_opt1.cpp:788-800, 1301-1312
//   pack before inverse exchanges    src/pencil/mpicufft_pencil_opt1.cpp:813-824, 1325-1336
//
// Design (see DESIGN.md for the full story):
//  * One workgroup transforms TW = TL*G lines of length N.  A line is owned by N/E threads,
//    each holding E complex points in VGPRs.  The transform is a Stockham autosort chain of
//    up to four radix-R passes (R <= E, all powers of two); butterflies run entirely in
//    registers, data moves between passes through LDS (re/im planes, 8-/4-byte accesses).
//  * Thread <-> data mapping is "line fastest": lane = line + TW * t.  Every global access
//    of a wave then covers TL adjacent lines x 8 adjacent points, which is what makes the
//    line-interleaved intermediate layouts below coalesce.
//  * Intermediate (send/recv) buffers use a tiled layout [a][tile][n][l]: TL lines that are
//    adjacent along the *next* pass's line-set axis are interleaved point by point, so the
//    consumer reads one fully contiguous chunk per workgroup and the producer writes
//    TL x TL (1 KiB) or TL (128 B) contiguous runs.  Per-peer blocks stay contiguous and have
//    exactly the reference's all-to-all byte counts/displacements
//    (mpicufft_pencil_opt1.cpp:269-273, 315-319) -- only the order inside a block differs.
//  * The inverse transform reuses the forward butterflies: conj(forward(conj(x))).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dfft {

constexpr int MAXSEG = 32;   // max segments of a gathered/scattered axis (peers x pipeline chunks)

enum LoadKind : int {
    LOAD_LINES = 0,    // natural lines:        (a*LB + b*TL + l)*N + n, or (KS_in != 0) rows a*AS_in + (b*TL + l)*KS_in + n
    LOAD_TILED = 1,    // tiled, segmented by source peer: base[s] + a*len[s]*LB + b*TL*len[s] + (n-start[s])*tw + l
    LOAD_KMAJOR = 2    // point-major:          n*KS + a*AS + b*TL + l
};
enum StoreKind : int {
    STORE_LINES = 0,           // (a*LB + b*TL + l)*N + k
    STORE_KMAJOR = 1,          // k*KS + a*AS + b*TL + l
    STORE_TILED_SAME = 2,      // base[p] + (k-start[p])*LB*LA + b*TL*LA + a*tw + l
    STORE_TILED_TRANSPOSE = 3  // base[p] + a*len[p]*LB + kt*T2*LB + (b*TL+l)*tw2 + kr
};

struct SegTable {
    int32_t nseg;
    uint32_t start[MAXSEG];   // first point index of the segment
    uint32_t len[MAXSEG];     // points in the segment
    uint64_t base[MAXSEG];    // element offset of the peer block in the buffer
};

// Per-point address table of a segmented (multi-peer) load or store, built by the host once per
// launch descriptor: the segment search, the tile split and the tile widths are done ahead of time,
// the kernel does one cached 16-byte load and two multiply-adds per point.
//   tiled load, point n:          off = base + ln*(a*LB + b*TL) + aux*tw + l        (aux = n - start)
//   transposed-tile store, k:     off = base + ln*(a*LB) + (b*TL + l)*aux           (aux = tw2; base holds bs + kt*T2*LB + kr)
//   same-tile store, k:           off = base + b*TL*LA + a*tw + l                   (base holds bs + (k-start)*LB*LA)
struct SegEntry {
    uint64_t base;
    uint32_t ln;
    uint32_t aux;
};

struct PassArgs {
    const void *in;
    void *out;
    const void *tw;        // twiddle table, N complex entries exp(-2*pi*i*j/N)
    const void *tw2;       // real transforms: exp(-2*pi*i*k/(2N)), k = 0..N (split/merge); Bluestein: chirp exp(-i*pi*n^2/NL)
    const void *tw3;       // Bluestein: FFT_N of the conjugate chirp, already divided by N
    uint32_t NL;           // Bluestein: true line length (<= N/2 + 1/2); 0 otherwise
    int32_t real_mode;     // Bluestein: 0 complex, 1 real input lines (R2C z pass), 2 real output lines (C2R z pass)
    uint32_t NK;           // Bluestein: number of points on the spectral side of a real transform (NL/2+1), else NL
    uint32_t na;           // extent of the outer line-set axis
    uint32_t LB;           // extent of the inner (tiled) line-set axis
    uint32_t nb;           // tiles along LB = ceil(LB / TL)
    uint32_t ntiles;       // na * nb
    int32_t load_kind, store_kind;
    int32_t swap;          // 1 = inverse transform (swap re/im on load and on store)
    int32_t a_fastest;     // workgroup -> tile order: 0: b fastest (w = a*nb + b), 1: a fastest (w = b*na + a)
    int32_t xcd_swizzle;   // 1: consecutive tiles go to the same XCD (block b runs on XCD b % 8), so that
                           //    neighbouring tiles that share a cache line meet in one L2
    int32_t debug;         // measurement only: bit 0 = skip the transform (the pass becomes a copy with the same
                           //    access pattern: its time is the pattern's own roofline), results are then wrong;
                           //    bit 1 = per-point 64-bit vector addresses instead of scalar base + 32-bit lane offset (A/B)
    int32_t addr64;        // 1: this pass keeps the per-point 64-bit vector addresses (dfft_tune_variants measured them faster here)
    int32_t shift;         // 1: STORE_KMAJOR with an odd row pitch: tile windows follow the cache lines of each
                           //    output row (nb counts one extra tile per row); fft_pass_kernel only
    uint32_t LA;           // STORE_TILED_SAME: extent of the a axis
    uint32_t T2shift;      // STORE_TILED_TRANSPOSE: log2 of the consumer's tile size
    uint64_t KS_in;        // LOAD_KMAJOR point stride
    uint64_t KS_out;       // STORE_KMAJOR point stride
    uint64_t AS_in;        // LOAD_KMAJOR stride of the outer axis a (LB for the API output layout)
    uint64_t AS_out;       // STORE_KMAJOR stride of the outer axis a
    // explicit strides of the private tiled layouts (0 = the packed defaults).  They let a plan pad rows so that
    // consecutive rows of a workgroup's scatter differ by an odd multiple of 128 B (profiles/r2_placement_probe.txt)
    uint64_t IA, IB;       // LOAD_TILED, one segment: stride of the outer axis a (default len*LB) and of a tile along b (TL*len)
    uint64_t SK, SB;       // STORE_TILED_SAME: stride of a point k (default LB*LA) and of a tile along b (TL*LA)
    // segment tables live in device memory (plan-owned): dynamically indexed by-value kernel
    // arguments would be copied to scratch
    const SegTable *lseg, *sseg;
    int32_t lnseg, snseg;
    const SegEntry *ltab, *stab;   // per-point tables (null: search the segment table per point)
    int32_t luni, suni;            // the side's table may be read with wave-uniform (scalar) loads: every segment starts at a multiple of 16 points    // two-level lines (fft_bluestein_kernel only): a line of lvN = N1*N2 points over two launches of NL = N1 resp. N2 points
    int32_t lv;            // 0: the launch transforms whole lines; 1 / 2: first / second level
    int32_t plain;         // fft_bluestein_kernel: 1 = NL == the configuration's length (a power of two): plain chain, no chirp
    uint32_t lvQ;          // sub-lines per line: N2 at level 1, N1 at level 2
    uint32_t lvN, lvNK;    // the line's length and its points on the spectral side of a real transform (as NL / NK of a one-launch pass)
    const void *lvtw;      // exp(-2*pi*i*j/lvN), lvN entries: twiddles between the levels
    void *lvw;             // scratch between the levels: ntiles * TL * lvN elements
    int32_t lvqm;          // this level's lanes run over neighbouring sub-lines of one line (its outer side has natural lines)
    int32_t lvlay;         // scratch layout: 0 tiled ((w*N1 + i1)*N2 + i2)*TL + l, 1 per line: line*lvN + i1*lvs1 + i2*lvs2
    uint32_t lvs1, lvs2;
    // long Bluestein lines (a length beyond the one-launch kernel that does not split either, e.g. a prime above 4096): Bluestein's
    // algorithm with M-point transforms that are two-level lines themselves -- four launches, whose outer sides carry the chirp steps.
    // bit 0 (a first-level launch): the points a sub-line fetches are those of a line of lbL points (lbK on the spectral side of a real
    // transform), multiplied by lbtab[point] and zero beyond lbL; bit 1 (a second-level launch): output k is multiplied by lbtab[k];
    // bit 2 (second level): re <-> im are swapped before that product (the inverse M-point transform by the swap trick)
    int32_t lb;
    uint32_t lbL, lbK;
    const void *lbtab;
};

// ------------------------------------------------------------------------------------------
// complex number = native 2-vector.  For fp32 that is the point: a <2 x float> lives in an aligned register pair and its
// arithmetic selects the packed instructions of CDNA3/4 (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32 with op_sel doing the
// re/im broadcasts and swaps): a complex add is one instruction, a complex multiplication two.  (The SLP vectorizer tried
// to find these pairs in scalar code and paid more in register shuffles than it gained; it is switched off, see Makefile.)
typedef float cfloat_t __attribute__((ext_vector_type(2)));
typedef double cdouble_t __attribute__((ext_vector_type(2)));
template <typename R> struct Vec2;
template <> struct Vec2<double> { using type = cdouble_t; };
template <> struct Vec2<float> { using type = cfloat_t; };
template <typename C> struct ScalarOf;
template <> struct ScalarOf<cfloat_t> { using type = float; };
template <> struct ScalarOf<cdouble_t> { using type = double; };
template <typename C> using scalar_t = typename ScalarOf<C>::type;
template <typename C> __host__ __device__ __forceinline__ C cswap(C d) { return __builtin_shufflevector(d, d, 1, 0); }      // (im, re)
template <typename C> __host__ __device__ __forceinline__ C cmake(scalar_t<C> re, scalar_t<C> im) { C r; r.x = re; r.y = im; return r; }
// x * w for a twiddle w given as the pair (w, iw) with iw = i*w = (-w.y, w.x): x.xx * w + x.yy * iw -- two packed
// instructions at fp32.  Keeping iw next to w (instead of building it per multiplication) is what makes it two.
template <typename C> __host__ __device__ __forceinline__ C cmul2(C x, C w, C iw)
{
    const C xx = __builtin_shufflevector(x, x, 0, 0), yy = __builtin_shufflevector(x, x, 1, 1);
    return xx * w + yy * iw;
}
template <typename C> __host__ __device__ __forceinline__ C ci(C w) { C r; r.x = -w.y; r.y = w.x; return r; }            // i * w

template <int I, int N, typename F> __host__ __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// the in-register butterflies split a radix R by its smallest prime factor (2, 3, 5, 7) until nothing is left
constexpr int smallest_factor(int R) { return R % 2 == 0 ? 2 : R % 3 == 0 ? 3 : R % 5 == 0 ? 5 : R % 7 == 0 ? 7 : R; }
// Output index held by register slot m after Dif<R>: digit reversal in the mixed radix system of R's prime factors,
// smallest first (out_R(m) = p * out_{R/p}(m % (R/p)) + m / (R/p)); for a power of two that is the bit reversal.
constexpr int brev(int m, int R)
{
    int r = 0, w = 1;
    while (R > 1) {
        const int p = smallest_factor(R), H = R / p;
        r += w * (m / H);
        m %= H;
        w *= p;
        R = H;
    }
    return r;
}

// slot that holds output index k after Dif<R> (the inverse of brev: the digit reversal is its own inverse only for prime powers)
constexpr int brev_inv(int k, int R)
{
    for (int m = 0; m < R; m++)
        if (brev(m, R) == k) return m;
    return 0;
}

// cos(2*pi*j/64), j = 0..16 (one octant + 1); everything else by symmetry
constexpr double kCos64[17] = {1.0,
                                          0.99518472667219688624,
                                          0.98078528040323044913,
                                          0.95694033573220886494,
                                          0.92387953251128675613,
                                          0.88192126434835502971,
                                          0.83146961230254523708,
                                          0.77301045336273696081,
                                          0.70710678118654752440,
                                          0.63439328416364549822,
                                          0.55557023301960222474,
                                          0.47139673682599764856,
                                          0.38268343236508977173,
                                          0.29028467725446236764,
                                          0.19509032201612826785,
                                          0.09801714032956060199,
                                          0.0};
constexpr double cos64(int j)
{
    j &= 63;
    if (j > 32) j = 64 - j;           // cos even about 32
    if (j > 16) return -kCos64[32 - j];
    return kCos64[j];
}
constexpr double sin64(int j) { return cos64((j + 48) & 63); }   // sin(x) = cos(x - pi/2)

// cos and sin of 2*pi*j/R for any R (mixed-radix butterflies), evaluated at compile time: the angle is reduced to
// [0, pi/4] exactly (integer arithmetic on the fraction j/R), then a Taylor polynomial in Horner form (error ~1 ulp)
struct CosSin { double c, s; };
constexpr double taylor_sin(double x)
{
    const double x2 = x * x;
    double r = 1.0;
    for (int k = 13; k >= 1; k--) r = 1.0 - x2 / (double)((2 * k) * (2 * k + 1)) * r;
    return x * r;
}
constexpr double taylor_cos(double x)
{
    const double x2 = x * x;
    double r = 1.0;
    for (int k = 13; k >= 1; k--) r = 1.0 - x2 / (double)((2 * k - 1) * (2 * k)) * r;
    return r;
}
constexpr CosSin cossin_frac(int j, int R)
{
    long n = (long)(((j % R) + R) % R) * 8;      // angle = 2*pi*n/D with D = 8R, so that D/8 is an integer
    const long D = (long)R * 8;
    bool sneg = false, cneg = false, swp = false;
    if (n > D / 2) { n = D - n; sneg = true; }
    if (n > D / 4) { n = D / 2 - n; cneg = true; }
    if (n > D / 8) { n = D / 4 - n; swp = true; }
    const double x = 0.78539816339744830962 * ((double)n / (double)R);     // (pi/4) * n/R, n <= R
    double c = taylor_cos(x), s = taylor_sin(x);
    if (swp) { const double t = c; c = s; s = t; }
    if (cneg) c = -c;
    if (sneg) s = -s;
    return CosSin{c, s};
}

// multiply by exp(-2*pi*i*J/R) (forward kernel), compile-time J, trivial cases folded.  Vector form: with sw = (d.y, d.x),
// d * (c - i s) = d * c + sw * (s, -s)
template <int R, int J, typename C> __host__ __device__ __forceinline__ C mul_w(C d)
{
    using T = scalar_t<C>;
    constexpr int j = ((J % R) + R) % R;
    if constexpr (j == 0) return d;
    else if constexpr (4 * j == R) { C r; r.x = d.y; r.y = -d.x; return r; }           // * -i
    else if constexpr (2 * j == R) return -d;
    else if constexpr (4 * j == 3 * R) { C r; r.x = -d.y; r.y = d.x; return r; }      // * +i
    else if constexpr (8 * j == R) { constexpr T s = (T)0.70710678118654752440; return (d + cswap(d) * cmake<C>((T)1, (T)-1)) * s; }     // (x+y, y-x) s
    else if constexpr (8 * j == 3 * R) { constexpr T s = (T)0.70710678118654752440; return (cswap(d) * cmake<C>((T)1, (T)-1) - d) * s; }   // (y-x, -x-y) s
    else if constexpr (64 % R == 0) {
        constexpr int j64 = (j * (64 / R)) & 63;
        constexpr T c = (T)cos64(j64), s = (T)sin64(j64);   // w = c - i s
        return d * c + cswap(d) * cmake<C>(s, -s);
    } else {
        constexpr CosSin w = cossin_frac(j, R);
        constexpr T c = (T)w.c, s = (T)w.s;
        return d * c + cswap(d) * cmake<C>(s, -s);
    }
}

// P-point DFT for an odd prime P (3, 5, 7) on x[0..P-1] in place, natural order, forward sign:
//   y_s = a_s - i b_s,  y_{P-s} = a_s + i b_s,   a_s = x_0 + sum_q cos(2 pi q s / P) (x_q + x_{P-q}),
//                                               b_s =       sum_q sin(2 pi q s / P) (x_q - x_{P-q}),   q, s = 1..(P-1)/2
template <int P, typename C> __host__ __device__ __forceinline__ void dft_prime(C *x)
{
    using T = scalar_t<C>;
    constexpr int Hh = (P - 1) / 2;
    C tp[Hh], tm[Hh];
    static_for<0, Hh>([&](auto qc) {
        constexpr int q = decltype(qc)::value + 1;
        tp[q - 1] = x[q] + x[P - q];
        tm[q - 1] = x[q] - x[P - q];
    });
    const C x0 = x[0];
    C y0 = x0;
    static_for<0, Hh>([&](auto qc) { y0 = y0 + tp[decltype(qc)::value]; });
    x[0] = y0;
    static_for<0, Hh>([&](auto sc) {
        constexpr int s = decltype(sc)::value + 1;
        C a = x0, b;
        b.x = 0; b.y = 0;
        static_for<0, Hh>([&](auto qc) {
            constexpr int q = decltype(qc)::value + 1;
            constexpr CosSin w = cossin_frac(q * s, P);
            a = a + tp[q - 1] * (T)w.c;
            b = b + tm[q - 1] * (T)w.s;
        });
        C lo, hi;
        lo.x = a.x + b.y; lo.y = a.y - b.x;      // a - i b
        hi.x = a.x - b.y; hi.y = a.y + b.x;      // a + i b
        x[s] = lo;
        x[P - s] = hi;
    });
}

// radix-R decimation-in-frequency FFT on registers v[OFF + m*STRIDE], m = 0..R-1; R = 2^a 3^b 5^c 7^d.
// Each level splits by the smallest prime factor P of R: P-point DFTs over the legs j + q*R/P, twiddles w_R^(j s)
// on output s, then P sub-transforms of length R/P.  On return slot m holds X[brev(m, R)].
template <int R, int OFF, int STRIDE, typename C> struct Dif {
    static __host__ __device__ __forceinline__ void run(C *v)
    {
        if constexpr (R >= 2) {
            constexpr int P = smallest_factor(R), H = R / P;
            static_for<0, H>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (P == 2) {
                    const C a = v[OFF + j * STRIDE], b = v[OFF + (j + H) * STRIDE];
                    v[OFF + j * STRIDE] = a + b;
                    v[OFF + (j + H) * STRIDE] = mul_w<R, j>(a - b);
                } else {
                    C x[P];
                    static_for<0, P>([&](auto qc) { constexpr int q = decltype(qc)::value; x[q] = v[OFF + (j + q * H) * STRIDE]; });
                    dft_prime<P>(x);
                    static_for<0, P>([&](auto sc) { constexpr int s = decltype(sc)::value; v[OFF + (j + s * H) * STRIDE] = mul_w<R, j * s>(x[s]); });
                }
            });
            static_for<0, P>([&](auto sc) { Dif<H, OFF + decltype(sc)::value * H * STRIDE, STRIDE, C>::run(v); });
        }
    }
};

// ------------------------------------------------------------------------------------------
// TWCHAIN = 1: load only w^k per butterfly and build w^(m k) by successive multiplication
// (2 live registers instead of RP-1 table loads in flight; error grows by ~RP ulp)
// NTMEM: bit 0 = nontemporal global loads, bit 1 = nontemporal global stores (streaming data)
// MAP: thread <-> (line, point) mapping.  0 = line fastest in every pass (lane = line + TW*t).
//      1 = point fastest (16 adjacent points of one line in adjacent lanes) in every pass,
//      2 = line fastest for the first pass (its loads), point fastest afterwards (its stores).
//      3 = point fastest for the first pass (its loads: natural lines), line fastest afterwards (its stores: tiles) -- the mirror
//          image of 2: the fp32 inverse x pass of a plan with an x-contiguous spectrum (option spectral_layout).
//      The point-fastest forms exist for fp32: with 8-byte points and 16-line tiles a line-fastest
//      wave touches natural lines and transposed tiles in 32-byte pieces; point fastest makes
//      those accesses 128-byte runs.  The LDS exchange between passes does the re-mapping for free.
// SUB: sub-tile workgroups.  The layouts interleave TL lines per tile; with SUB = 2 a workgroup transforms
//      only TL/2 of them (half the on-chip footprint: two workgroups per CU at N = 2048) and its sibling
//      (the next logical workgroup, kept on the same XCD so that L2 sees both halves of every 128-byte
//      run) the rest.  Only fft_pass_kernel; needs G == 1.
template <typename R, int N, int E, int TL, int G, int R1, int R2, int R3, int R4, int PLANES, int TWCHAIN = 0, int NTMEM = 0, int MAP = 0, int SUB = 1, int PERSIST = 0>
struct PassCfg {
    using real = R;
    using C = typename Vec2<R>::type;
    static constexpr int kN = N, kE = E, kTL = TL, kG = G;
    static constexpr int r1 = R1, r2 = R2, r3 = R3, r4 = R4, kPLANES = PLANES, kTWCHAIN = TWCHAIN, kNTMEM = NTMEM, kMAP = MAP;
    static constexpr int kSUB = SUB, TLK = TL / SUB;  // lines of a tile one workgroup transforms
    static constexpr int kPERSIST = PERSIST;
    static constexpr int kFIX = 0;                   // address forms fixed at compile time (FixForms below); 0 = all forms, chosen at run time
    static constexpr int RLAST = R4 > 1 ? R4 : (R3 > 1 ? R3 : (R2 > 1 ? R2 : R1));
    static constexpr int NT = N / E;                 // threads per line
    static constexpr int TW = TLK * G;               // lines per workgroup
    static_assert(SUB == 1 || (G == 1 && TL % SUB == 0), "sub-tile workgroups need G == 1");
    static constexpr int THREADS = NT * TW;
    static constexpr int NPASS = (R1 > 1) + (R2 > 1) + (R3 > 1) + (R4 > 1);
    static constexpr int PS = ilog2(R1 * TW);        // pad once per first-pass scatter stride ...
    static constexpr int PWS = ilog2(TW) > 4 ? 4 : ilog2(TW);   // ... by min(TW,16) slots
    static constexpr int SLOTS = N * TW;
    // MAP != 0: line-major LDS plane [line][n + n/PADB] with a pitch that keeps the 16-point groups
    // of the lines sharing a 32-lane LDS group on different banks
    static constexpr int PGRP = 16;
    // one pad slot per PADB points: 32, or 64 when the first pass is a radix-64 one -- its scatter writes points 64 t + m from
    // lanes t = 0..15 of a line, i.e. with a stride of 64 + 64/PADB slots: 66 puts the two lines of a 32-lane group on the
    // same (even) banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 20 %, profiles/r2_pmc_f32_2048.txt), 65 gives each line 16
    // consecutive banks like the linear gathers have
    static constexpr int PADB = R1 == 64 ? 64 : 32, PADSH = ilog2(PADB);
    static constexpr int PITCH_BASE = N + N / PADB;
    static constexpr int PITCH_WANT = MAP == 2 || MAP == 3 ? 17 : 16;
    // which passes run point fastest: the first one (and its loads) / the later ones (and the stores)
    static constexpr bool PF_FIRST = MAP == 1 || MAP == 3;
    static constexpr bool PF_REST = MAP == 1 || (MAP == 2 && (R2 > 1)) || (MAP == 3 && !(R2 > 1));
    static constexpr int PITCH = PITCH_BASE + ((PITCH_WANT - PITCH_BASE % 32 + 32) % 32);
    static constexpr int PLANE_SLOTS = MAP == 0 ? SLOTS + ((SLOTS >> PS) << PWS) : TW * PITCH;
    static_assert(MAP == 0 || ((N / E) % 16 == 0 && G == 1), "point-fastest mapping needs >= 16 threads per line");
    static constexpr size_t LDS_BYTES = NPASS > 1 ? (size_t)PLANES * PLANE_SLOTS * sizeof(R) : 0;
    // consecutive values of t one wave covers on the load side (first-pass mapping) and on the store side (mapping of the
    // later passes): line fastest 64 / TW, point fastest 16.  Wave-uniform table reads (seg_entry_uniform) need them <= 16,
    // the transposed-tile store <= the consumer's tile (TL): sub-tile workgroups with fewer than 4 lines do not qualify.
    static constexpr int SPAN_LF = TW >= 64 ? 1 : 64 / TW;
    static constexpr int SPAN_LOAD = PF_FIRST ? PGRP : SPAN_LF;
    static constexpr int SPAN_STORE = PF_REST ? PGRP : SPAN_LF;
    static constexpr bool UNI_LOAD = is_pow2(N) && SPAN_LOAD <= 16;
    static constexpr bool UNI_STORE_SAME = is_pow2(N) && SPAN_STORE <= 16;
    static constexpr bool UNI_STORE_TRANSPOSE = is_pow2(N) && SPAN_STORE <= TL;
    static_assert((R1 > 1 ? R1 : 1) * (R2 > 1 ? R2 : 1) * (R3 > 1 ? R3 : 1) * (R4 > 1 ? R4 : 1) == N, "radices must multiply to N");
    static_assert(E % R1 == 0 && (R2 <= 1 || E % R2 == 0) && (R3 <= 1 || E % R3 == 0) && (R4 <= 1 || E % R4 == 0), "radix must divide E");
    static_assert(THREADS <= 1024, "workgroup too large");
};

// A configuration compiled for ONE pair of address forms.  The kernel is an "uber-kernel": load_tile / store_tile branch (uniformly,
// at run time) over every address form, which costs nothing in a straight-line kernel but makes the persistent form spill -- every
// branch the transformed registers flow through multiplies their live ranges (PERSIST with all forms: 684 B of scratch at 32 fp64
// points per thread; with one form on each side: 249 VGPRs, no scratch; profiles/r3_persist3_resources.txt).  kFIX = 1: point-major
// load with a scalar base (the strided read of the API layout) and same-tile store into ONE block -- the inverse x pass of a plan
// without a second exchange (P1 = 1).  fix_ok() is the launcher's test that a launch has exactly these forms; every other launch
// goes to Generic.  (The sibling for the table store of P1 > 1 plans was measured in round 4 and dropped: no gain,
// profiles/r4_persist3.txt.)
template <typename Base, typename GenericCfg, int FIX> struct FixForms : Base {
    using Generic = GenericCfg;
    static constexpr int kFIX = FIX;
};
template <typename Cfg> inline bool fix_ok(const PassArgs &A)
{
    if constexpr (Cfg::kFIX == 1) {
        using C = typename Cfg::C;
        return A.load_kind == LOAD_KMAJOR && A.store_kind == STORE_TILED_SAME && !A.stab && A.snseg == 1 && !A.shift && !A.addr64 && !(A.debug & 2) &&
               ((uint64_t)(Cfg::NT - 1) * A.KS_in + (uint64_t)A.na * A.AS_in + A.LB + Cfg::kTL) * sizeof(C) < (1ull << 32);
    } else return true;
}

template <typename Cfg> __host__ __device__ __forceinline__ int lds_pad(int idx)
{
    return idx + ((idx >> Cfg::PS) << Cfg::PWS);
}

// value-level choice between two registers.  Written naively, `cond ? v[a] : v[b]` on a register array becomes a select
// of POINTERS into the array, which keeps the whole array out of registers (scratch); passing both candidates through an
// empty asm makes them values first.
template <typename C> __device__ __forceinline__ C pick(bool cond, C a, C b)
{
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y));
    C r;
    r.x = cond ? a.x : b.x;
    r.y = cond ? a.y : b.y;
    return r;
}

// Conjugate-pair assignment of the N/RP butterflies of a pass (real transforms): register block i < S/2 of thread t is
// butterfly jA = t + NT*i as usual, block i + S/2 is its mirror N/RP - jA, so that points k and N - k -- which the
// Hermitian split / merge of a packed real transform combines -- live in ONE thread and no extra trip through LDS is
// needed.  Butterfly 0 is its own mirror; its slot in thread 0 takes the other self-mirrored butterfly N/(2 RP).
template <typename Cfg, int RP, int I> __host__ __device__ __forceinline__ int pair_j(int t)
{
    constexpr int S = Cfg::kE / RP, H = S / 2, NB = Cfg::kN / RP;
    static_assert(S % 2 == 0, "conjugate-pair assignment needs an even number of butterflies per thread");
    if constexpr (I < H) return t + Cfg::NT * I;
    else {
        const int ja = t + Cfg::NT * (I - H);
        return ja == 0 ? NB / 2 : NB - ja;
    }
}
template <typename Cfg, int RP, int I, int JM> __host__ __device__ __forceinline__ int butterfly_j(int t)
{
    if constexpr (JM) return pair_j<Cfg, RP, I>(t);
    else return t + Cfg::NT * I;
}

// twiddle + butterflies of one Stockham pass: radix RP, previous sub-transform length NS; JM = 1: conjugate-pair
// assignment of the butterflies (pair_j)
template <typename Cfg, int RP, int NS, int JM = 0>
__host__ __device__ __forceinline__ void pass_compute(typename Cfg::C *v, int t, const typename Cfg::C *__restrict__ W)
{
    using C = typename Cfg::C;
    constexpr int S = Cfg::kE / RP;     // butterflies per thread == register stride
    constexpr int N = Cfg::kN;
    static_for<0, S>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (NS > 1) {
            const int j = butterfly_j<Cfg, RP, i, JM>(t);
            const int k = j % NS;      // NS is a compile-time constant (a mask for powers of two)
            constexpr int step = N / (NS * RP);
            if constexpr (Cfg::kTWCHAIN) {
                // w^m by successive multiplication; (cur, icur) advance together (i*cur obeys the same recurrence), so
                // every product below is the two-instruction form
                const C w1 = W[k * step], iw1 = ci(w1);
                C cur = w1, icur = iw1;
                static_for<1, RP>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    v[i + m * S] = cmul2(v[i + m * S], cur, icur);
                    if constexpr (m + 1 < RP) {
                        const C nx = cmul2(cur, w1, iw1);
                        // fp32: advance i*cur by its own packed product (2 instructions); fp64 has no packed arithmetic and
                        // the negation of ci() folds into the next multiplication's source modifiers
                        if constexpr (std::is_same<scalar_t<C>, float>::value) icur = cmul2(icur, w1, iw1);
                        else icur = ci(nx);
                        cur = nx;
                    }
                });
            } else {
                static_for<1, RP>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const C w = W[(m * k) * step];
                    v[i + m * S] = cmul2(v[i + m * S], w, ci(w));
                });
            }
        }
        Dif<RP, i, S, C>::run(v);
    });
}


// thread -> (line within workgroup, thread within line)
template <typename Cfg, bool POINT_FASTEST> __host__ __device__ __forceinline__ void thread_map(int tid, int &lw, int &t)
{
    if constexpr (!POINT_FASTEST) {
        lw = tid % Cfg::TW;
        t = tid / Cfg::TW;
    } else {
        constexpr int PG = Cfg::PGRP;
        lw = (tid / PG) % Cfg::TW;
        t = (tid % PG) + PG * (tid / (PG * Cfg::TW));
    }
}
// LDS slot of point n of line lw
template <typename Cfg> __host__ __device__ __forceinline__ int lds_slot(int lw, int n)
{
    if constexpr (Cfg::kMAP == 0) return lds_pad<Cfg>(n * Cfg::TW + lw);
    else return lw * Cfg::PITCH + n + (n >> Cfg::PADSH);
}

// Slot of point n0 + dn when dn is a compile-time constant whose padding separates from n0's
// (no carry into the padded bits): slot(n0 + dn) = slot(n0) + lds_slot_off(dn).  The per-point LDS
// addresses of a scatter / gather then are one computed base plus immediate offsets.
template <typename Cfg> constexpr int lds_slot_off(int dn)
{
    if constexpr (Cfg::kMAP == 0) return dn * Cfg::TW + (((dn * Cfg::TW) >> Cfg::PS) << Cfg::PWS);
    else return dn + (dn >> Cfg::PADSH);
}
// scatter of pass (RP, NS): point offsets are m*NS.  Line-fastest plane: separable because NS is 1 (then
// RP == R1 and m*TW stays below the padded block) or a multiple of R1.  Line-major plane: 32-point pad blocks.
template <typename Cfg, int RP, int NS> constexpr bool scatter_separable()
{
    if constexpr (!is_pow2(Cfg::kN)) return false;      // mixed radix: the pad blocks are not aligned to the radices
    else if constexpr (Cfg::kMAP == 0) return NS == 1 ? RP == Cfg::r1 : NS % Cfg::r1 == 0;
    else return NS == 1 ? RP % Cfg::PADB == 0 : NS % Cfg::PADB == 0;
}
// gather: point offsets are NT*c
template <typename Cfg> constexpr bool gather_separable()
{
    if constexpr (!is_pow2(Cfg::kN)) return false;
    else if constexpr (Cfg::kMAP == 0) return Cfg::NT % Cfg::r1 == 0;
    else return Cfg::NT % Cfg::PADB == 0 || Cfg::PADB % Cfg::NT == 0;      // t < NT: t + NT*c has the pad count of NT*c either way
}

// scatter the outputs of pass (RP, NS) to LDS plane, Stockham output index
template <typename Cfg, int RP, int NS, int COMP, int JM = 0>
__host__ __device__ __forceinline__ void lds_scatter(const typename Cfg::C *v, typename Cfg::real *plane, int t, int lw)
{
    constexpr int S = Cfg::kE / RP;
    static_for<0, S>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int j = butterfly_j<Cfg, RP, i, JM>(t);
        const int k = j % NS;      // NS is a compile-time constant (a mask for powers of two)
        const int nbase = (j - k) * RP + k;       // (j/NS)*NS*RP + k
        const int base = lds_slot<Cfg>(lw, nbase);
        static_for<0, RP>([&](auto mc) {
            constexpr int mr = decltype(mc)::value;          // register slot
            constexpr int m = brev(mr, RP);                  // output index of that slot
            int idx;
            if constexpr (scatter_separable<Cfg, RP, NS>()) idx = base + lds_slot_off<Cfg>(m * NS);
            else idx = lds_slot<Cfg>(lw, nbase + m * NS);
            plane[idx] = COMP == 0 ? v[i + mr * S].x : v[i + mr * S].y;
        });
    });
}
// gather for a consuming pass of radix RPN whose butterflies are assigned in conjugate pairs: register c = i + m*S holds
// input leg m of butterfly pair_j(i), i.e. point pair_j(i) + m*(N/RPN)
template <typename Cfg, int COMP, int RPN>
__host__ __device__ __forceinline__ void lds_gather_paired(typename Cfg::C *v, const typename Cfg::real *plane, int t, int lw)
{
    constexpr int S = Cfg::kE / RPN, LEG = Cfg::kN / RPN;
    constexpr bool SEP = is_pow2(Cfg::kN) && (Cfg::kMAP == 0 ? LEG % Cfg::r1 == 0 : LEG % Cfg::PADB == 0);      // mixed radix: pad blocks are not aligned to the legs
    static_for<0, S>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int j = pair_j<Cfg, RPN, i>(t);
        const int base = lds_slot<Cfg>(lw, j);
        static_for<0, RPN>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            int idx;
            if constexpr (SEP) idx = base + lds_slot_off<Cfg>(m * LEG);
            else idx = lds_slot<Cfg>(lw, j + m * LEG);
            if (COMP == 0) v[i + m * S].x = plane[idx]; else v[i + m * S].y = plane[idx];
        });
    });
}
template <typename Cfg, int COMP>
__host__ __device__ __forceinline__ void lds_gather(typename Cfg::C *v, const typename Cfg::real *plane, int t, int lw)
{
    const int base = lds_slot<Cfg>(lw, t);
    static_for<0, Cfg::kE>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        int idx;
        if constexpr (gather_separable<Cfg>()) idx = base + lds_slot_off<Cfg>(Cfg::NT * c);
        else idx = lds_slot<Cfg>(lw, t + Cfg::NT * c);
        if (COMP == 0) v[c].x = plane[idx]; else v[c].y = plane[idx];
    });
}

// producer thread coordinates (t, lw), consumer coordinates (t2, lw2): they differ only across the
// first exchange of a MAP == 2 kernel.  JS: the producing pass has the conjugate-pair butterfly assignment; JG: the
// consuming pass (radix RPN) has it
template <typename Cfg, int RP, int NS, int JS = 0, int JG = 0, int RPN = 1>
__device__ __forceinline__ void exchange(typename Cfg::C *v, typename Cfg::real *lds, int t, int lw, int t2, int lw2, bool first)
{
    using R = typename Cfg::real;
    auto gather = [&](auto comp, const R *plane) {
        constexpr int COMP = decltype(comp)::value;
        if constexpr (JG) lds_gather_paired<Cfg, COMP, RPN>(v, plane, t2, lw2);
        else lds_gather<Cfg, COMP>(v, plane, t2, lw2);
    };
    if constexpr (Cfg::kPLANES == 2) {
        R *p0 = lds, *p1 = lds + Cfg::PLANE_SLOTS;
        if (!first) __syncthreads();
        lds_scatter<Cfg, RP, NS, 0, JS>(v, p0, t, lw);
        lds_scatter<Cfg, RP, NS, 1, JS>(v, p1, t, lw);
        __syncthreads();
        gather(std::integral_constant<int, 0>{}, p0);
        gather(std::integral_constant<int, 1>{}, p1);
    } else {
        if (!first) __syncthreads();
        lds_scatter<Cfg, RP, NS, 0, JS>(v, lds, t, lw);     // old re -> LDS
        __syncthreads();
        gather(std::integral_constant<int, 0>{}, lds);      // new re (old im still live in v[].y)
        __syncthreads();
        lds_scatter<Cfg, RP, NS, 1, JS>(v, lds, t, lw);
        __syncthreads();
        gather(std::integral_constant<int, 1>{}, lds);
    }
}

// the whole Stockham chain on the registers of one thread.  (t, lw): coordinates during the first
// pass; (t2, lw2): coordinates from the first exchange on (identical unless MAP == 2)
// PAIR = 1: the LAST pass has the conjugate-pair butterfly assignment (R2C: split in registers afterwards);
// PAIR = 2: the FIRST pass has it (C2R: merge in registers before)
template <typename Cfg, int PAIR = 0>
__device__ __forceinline__ void transform(typename Cfg::C *v, typename Cfg::real *lds,
                                          const typename Cfg::C *__restrict__ W, int t, int lw, int t2, int lw2)
{
    constexpr int R1 = Cfg::r1, R2 = Cfg::r2, R3 = Cfg::r3, R4 = Cfg::r4, NP = Cfg::NPASS;
    static_assert(PAIR == 0 || NP >= 2, "paired butterflies need at least two passes");
    constexpr int JS1 = PAIR == 2;
    pass_compute<Cfg, R1, 1>(v, t, W);
    if constexpr (R2 > 1) {
        constexpr int L = PAIR == 1 && NP == 2;
        exchange<Cfg, R1, 1, JS1, L, R2>(v, lds, t, lw, t2, lw2, true);
        pass_compute<Cfg, R2, R1, L>(v, t2, W);
    }
    if constexpr (R3 > 1) {
        constexpr int L = PAIR == 1 && NP == 3;
        exchange<Cfg, R2, R1, 0, L, R3>(v, lds, t2, lw2, t2, lw2, false);
        pass_compute<Cfg, R3, R1 * R2, L>(v, t2, W);
    }
    if constexpr (R4 > 1) {
        constexpr int L = PAIR == 1 && NP == 4;
        exchange<Cfg, R3, R1 * R2, 0, L, R4>(v, lds, t2, lw2, t2, lw2, false);
        pass_compute<Cfg, R4, R1 * R2 * R3, L>(v, t2, W);
    }
}
// line-fastest kernels (r2c / c2r / Bluestein): one coordinate set
template <typename Cfg>
__device__ __forceinline__ void transform(typename Cfg::C *v, typename Cfg::real *lds,
                                          const typename Cfg::C *__restrict__ W, int t, int lw, int /*tid*/)
{
    transform<Cfg>(v, lds, W, t, lw, t, lw);
}

// one 16-byte load of a table entry (every lane of a wave reads nearby entries: L1 hits)
__device__ __forceinline__ SegEntry seg_entry(const SegEntry *p)
{
    const uint4 r = *reinterpret_cast<const uint4 *>(p);
    SegEntry e;
    e.base = (uint64_t)r.x | ((uint64_t)r.y << 32);
    e.ln = r.z;
    e.aux = r.w;
    return e;
}

// logical workgroup index: identity, or the XCD-aware remap of the guide (T1, bijective form).  id / n = (virtual)
// workgroup index and count: the hardware's for a plain launch, the walked ones of a persistent workgroup.
template <bool ALWAYS = false> __device__ __forceinline__ uint32_t logical_block(const PassArgs &A, uint32_t id, uint32_t n)
{
    if (!ALWAYS && !A.xcd_swizzle) return id;
    const uint32_t cpx = n >> 3;
    return id < (cpx << 3) ? (id & 7) * cpx + (id >> 3) : id;
}
template <bool ALWAYS = false> __device__ __forceinline__ uint32_t logical_block(const PassArgs &A)
{
    return logical_block<ALWAYS>(A, blockIdx.x, gridDim.x);
}

template <typename Cfg, typename C> __device__ __forceinline__ C stream_load(const C *p)
{
    if constexpr (Cfg::kNTMEM & 1) {
        using T = scalar_t<C>;
        typedef T native2 __attribute__((ext_vector_type(2)));
        const native2 r = __builtin_nontemporal_load(reinterpret_cast<const native2 *>(p));
        C v; v.x = r.x; v.y = r.y;
        return v;
    } else return *p;
}
template <typename Cfg, typename C> __device__ __forceinline__ void stream_store(C *p, C v)
{
    if constexpr (Cfg::kNTMEM & 2) {
        using T = scalar_t<C>;
        typedef T native2 __attribute__((ext_vector_type(2)));
        native2 r; r.x = v.x; r.y = v.y;
        __builtin_nontemporal_store(r, reinterpret_cast<native2 *>(p));
    } else *p = v;
}

// tile and line of a lane.  e = index of the lane's line along the tiled axis.  With A.shift (point-major stores whose
// row pitch is not a multiple of the tile) the window of a workgroup is moved back by the row's misalignment, so that
// its TL stores per point fill exactly one aligned 128-byte line; (b, l, tw) then describe where that line lives in
// the (unshifted) tiled input.
template <int TL> struct TilePos {
    uint32_t a, b, tw, e;
    int l;
    bool ok;
};
template <typename Cfg> __device__ __forceinline__ TilePos<Cfg::kTL> tile_pos(const PassArgs &A, uint32_t blk, int lwx)
{
    constexpr int TL = Cfg::kTL;
    TilePos<TL> P;
    uint32_t w;
    if constexpr (Cfg::kSUB > 1) {          // sub-tile workgroup: lines [sub*TLK, sub*TLK + TLK) of tile blk / SUB
        w = blk / Cfg::kSUB;
        P.l = (int)(blk % Cfg::kSUB) * Cfg::TLK + lwx;
    } else if constexpr (Cfg::kG == 1) {    // one tile per workgroup: (a, b, tw) are workgroup-uniform (scalar registers)
        w = blk;
        P.l = lwx;
    } else {
        w = blk * Cfg::kG + lwx / TL;
        P.l = lwx % TL;
    }
    bool ok = w < A.ntiles;
    P.a = !ok ? 0 : (A.a_fastest ? w % A.na : w / A.nb);
    P.b = !ok ? 0 : (A.a_fastest ? w / A.na : w % A.nb);
    if (A.shift) {
        const uint32_t s = (uint32_t)((uint64_t)P.a * A.AS_out) & (uint32_t)(TL - 1);
        const int ei = (int)(P.b * TL + P.l) - (int)s;
        ok = ok && ei >= 0 && (uint32_t)ei < A.LB;
        P.e = ok ? (uint32_t)ei : 0;
        P.b = P.e / TL;
        P.l = (int)(P.e % TL);
        const uint32_t rem = A.LB - P.b * TL;
        P.tw = rem < (uint32_t)TL ? rem : (uint32_t)TL;
        P.ok = ok;
        return P;
    }
    const uint32_t rem = A.LB - P.b * TL;
    P.tw = rem < (uint32_t)TL ? rem : (uint32_t)TL;     // valid lines of this tile
    P.e = P.b * TL + P.l;
    P.ok = ok && (uint32_t)P.l < P.tw;
    return P;
}

// Wave-uniform table entries.  A wave covers SPAN consecutive values of t (line fastest: 64 / TW, point fastest: 16)
// starting at a multiple of SPAN, and every compile-time point offset (NT*c, N/RL) is a multiple of SPAN when N is a
// power of two.  If every segment of a side starts at a multiple of 16 points (the host checks: PassArgs::luni / suni),
// the SPAN points a wave touches per register lie in ONE segment and their table entries differ only by the lane's
// t - t0: the entry of lane 0's point is fetched with a SCALAR load (s_load_dwordx4 through the constant cache, no
// vector-memory issue slot, no VGPRs) and the address becomes a scalar base plus one per-lane offset that does not
// depend on the point -- the segmented side then costs what the single-segment closed form costs (the per-point vector
// load of a 16-byte entry cost as much address-pipeline time as the 8- / 16-byte point it located: +47-73 % on the fp32
// 2048-point passes, profiles/r2_f32_2048_multirank_kernel_stats.csv).
__device__ __forceinline__ SegEntry seg_entry_uniform(const SegEntry *p)      // p must be wave-uniform
{
    const uint4 r = *reinterpret_cast<const uint4 *>(p);
    SegEntry e;
    e.base = (uint64_t)r.x | ((uint64_t)r.y << 32);
    e.ln = r.z;
    e.aux = r.w;
    return e;
}

// loads registers [C0, C1) of the lane (register c holds point t + NT*c); v[c] must be valid for that range
template <typename Cfg, int C0 = 0, int C1 = Cfg::kE>
__device__ __forceinline__ void load_tile(const PassArgs &A, const typename Cfg::C *__restrict__ in, const TilePos<Cfg::kTL> &P,
                                          int t, typename Cfg::C *v)
{
    using C = typename Cfg::C;
    constexpr int N = Cfg::kN, TL = Cfg::kTL, NT = Cfg::NT;
    const uint32_t a = P.a, b = P.b, tw = P.tw;
    const int l = P.l;
    if constexpr (Cfg::kFIX == 1) {          // LOAD_KMAJOR, scalar base (fix_ok has checked the range)
        if (P.ok) {
            const uint32_t lane = (uint32_t)(((uint64_t)t * A.KS_in + (uint64_t)a * A.AS_in + P.e) * sizeof(C));
            const char *ub = reinterpret_cast<const char *>(in);
            static_for<C0, C1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                v[c] = stream_load<Cfg>(reinterpret_cast<const C *>(ub + (uint64_t)(NT * c) * A.KS_in * sizeof(C) + lane));
            });
        } else {
            static_for<C0, C1>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].x = 0; v[c].y = 0; });
        }
        (void)b; (void)tw; (void)l;
        return;
    }
    if (P.ok) {
        if (A.load_kind == LOAD_LINES) {
            const uint64_t row = A.KS_in ? (uint64_t)a * A.AS_in + ((uint64_t)b * TL + l) * A.KS_in
                                         : ((uint64_t)a * A.LB + (uint64_t)b * TL + l) * N;
            const C *p = in + row + t;
            static_for<C0, C1>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = stream_load<Cfg>(p + NT * c); });
        } else if (A.load_kind == LOAD_KMAJOR) {
            if (!(A.debug & 2) && !A.addr64 && ((uint64_t)(NT - 1) * A.KS_in + (uint64_t)A.na * A.AS_in + A.LB + TL) * sizeof(C) < (1ull << 32)) {
                // Address = scalar 64-bit base of the point (SALU: in + NT*c*KS, the same for every lane) + ONE 32-bit byte offset
                // of the lane (its tile, its line and its t rows; below 4 GiB for the API layouts: a lane's points lie N/E rows
                // apart): the loads take the base from a scalar register pair (global_load ... v_off, s[base:base+1]) instead of
                // a 64-bit vector multiply-add, a 32-bit add and a register PAIR per point.
                const uint32_t lane = (uint32_t)(((uint64_t)t * A.KS_in + (uint64_t)a * A.AS_in + P.e) * sizeof(C));
                const char *ub = reinterpret_cast<const char *>(in);
                static_for<C0, C1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    v[c] = stream_load<Cfg>(reinterpret_cast<const C *>(ub + (uint64_t)(NT * c) * A.KS_in * sizeof(C) + lane));
                });
            } else {
                const C *p = in + (uint64_t)a * A.AS_in + (uint64_t)b * TL + l + (uint64_t)t * A.KS_in;
                static_for<C0, C1>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = stream_load<Cfg>(p + (uint64_t)(NT * c) * A.KS_in); });
            }
        } else {
            if (Cfg::UNI_LOAD && A.ltab && A.luni) {
                const int t0 = __builtin_amdgcn_readfirstlane(t);
                const uint32_t Q = a * A.LB + b * TL;
                const uint32_t lane = ((uint32_t)(t - t0) * tw + (uint32_t)l) * (uint32_t)sizeof(C);      // bytes, same for every point
                const SegEntry *tab = A.ltab + t0;
                static_for<C0, C1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const SegEntry e = seg_entry_uniform(tab + NT * c);
                    const char *pu = reinterpret_cast<const char *>(in + (e.base + (uint64_t)e.ln * Q + (uint64_t)e.aux * tw));
                    v[c] = stream_load<Cfg>(reinterpret_cast<const C *>(pu + lane));
                });
            } else if (A.ltab) {
                const uint32_t Q = a * A.LB + b * TL;
                static_for<C0, C1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const SegEntry e = seg_entry(A.ltab + (t + NT * c));
                    v[c] = stream_load<Cfg>(in + e.base + (uint64_t)e.ln * Q + (uint64_t)e.aux * tw + l);
                });
            } else if (A.lnseg == 1) {
                const uint64_t len = A.lseg->len[0];
                const uint64_t ia = A.IA ? A.IA : len * A.LB, ib = A.IB ? A.IB : (uint64_t)TL * len;
                if (Cfg::kG == 1 && Cfg::kSUB == 1 && !A.shift && !(A.debug & 2) && !A.addr64) {
                    // one tile per workgroup: (a, b, tw) are the same in every lane (readfirstlane: the compiler sees them merged
                    // with the per-lane values of the A.shift case), so the tile's chunk and the point's offset in it make a scalar
                    // base; the lane adds its 32-bit (t*tw + l) bytes (see LOAD_KMAJOR)
                    const uint32_t au = (uint32_t)__builtin_amdgcn_readfirstlane((int)a), bu = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    const uint32_t twu = (uint32_t)__builtin_amdgcn_readfirstlane((int)tw);
                    const char *ub = reinterpret_cast<const char *>(in + (A.lseg->base[0] + (uint64_t)au * ia + (uint64_t)bu * ib));
                    const uint32_t lane = ((uint32_t)t * twu + (uint32_t)l) * (uint32_t)sizeof(C);
                    static_for<C0, C1>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        v[c] = stream_load<Cfg>(reinterpret_cast<const C *>(ub + (uint64_t)(NT * c) * twu * sizeof(C) + lane));
                    });
                } else {
                    const C *p = in + A.lseg->base[0] + (uint64_t)a * ia + (uint64_t)b * ib + l + (uint64_t)t * tw;
                    static_for<C0, C1>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = stream_load<Cfg>(p + (uint64_t)(NT * c) * tw); });
                }
            } else {
                static_for<C0, C1>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const uint32_t n = t + NT * c;
                    uint32_t s0 = A.lseg->start[0], ln = A.lseg->len[0];
                    uint64_t bs = A.lseg->base[0];
                    for (int s = 1; s < A.lnseg; s++)
                        if (n >= A.lseg->start[s]) { s0 = A.lseg->start[s]; ln = A.lseg->len[s]; bs = A.lseg->base[s]; }
                    v[c] = stream_load<Cfg>(in + bs + (uint64_t)a * ln * A.LB + (uint64_t)b * TL * ln + (uint64_t)(n - s0) * tw + l);
                });
            }
        }
    } else {
        static_for<C0, C1>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].x = 0; v[c].y = 0; });
    }
}

// register c = i + mr*S holds output k = t2 + NT*i + brev(mr)*(N/RL)   (RL = radix of the last pass, S = E / RL)
// stores registers [C0, C1) of the lane
template <typename Cfg, int C0 = 0, int C1 = Cfg::kE>
__device__ __forceinline__ void store_tile(const PassArgs &A, typename Cfg::C *__restrict__ out, const TilePos<Cfg::kTL> &P,
                                           int t2, const typename Cfg::C *v)
{
    using C = typename Cfg::C;
    constexpr int N = Cfg::kN, E = Cfg::kE, TL = Cfg::kTL, NT = Cfg::NT;
    constexpr int RL = Cfg::RLAST;     // radix of the last pass
    constexpr int S = E / RL;
    if (!P.ok) return;
    const uint32_t a2 = P.a, b2 = P.b, tws = P.tw, e2 = P.e;
    const int l2 = P.l;
    if constexpr (Cfg::kFIX == 1) {          // STORE_TILED_SAME into one block: the lane's part once (tiles may differ from lane to lane:
        // two tiles per workgroup), the point's k0 rows as a scalar offset
        const uint64_t sk = A.SK ? A.SK : (uint64_t)A.LB * A.LA, sb = A.SB ? A.SB : (uint64_t)TL * A.LA;
        C *pl = out + (A.sseg->base[0] - (uint64_t)A.sseg->start[0] * sk + (uint64_t)b2 * sb + (uint64_t)a2 * tws + l2 + (uint64_t)t2 * sk);
        static_for<C0, C1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            stream_store<Cfg>(pl + (uint64_t)k0 * sk, v[c]);
        });
        (void)e2;
        return;
    }
    if (A.store_kind == STORE_LINES) {
        // natural lines, or (KS_out != 0) rows at a*AS_out + line*KS_out: the z pass of the Y_Then_ZX sequence
        const uint64_t row = A.KS_out ? (uint64_t)a2 * A.AS_out + ((uint64_t)b2 * TL + l2) * A.KS_out
                                      : ((uint64_t)a2 * A.LB + (uint64_t)b2 * TL + l2) * N;
        C *p = out + row + t2;
        static_for<C0, C1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            stream_store<Cfg>(p + k0, v[c]);
        });
    } else if (A.store_kind == STORE_KMAJOR) {
        if (!(A.debug & 2) && !A.addr64 && ((uint64_t)(NT - 1) * A.KS_out + (uint64_t)A.na * A.AS_out + A.LB + TL) * sizeof(C) < (1ull << 32)) {
            // scalar base per point + one 32-bit lane offset, as in load_tile
            const uint32_t lane = (uint32_t)(((uint64_t)t2 * A.KS_out + (uint64_t)a2 * A.AS_out + e2) * sizeof(C));
            char *ub = reinterpret_cast<char *>(out);
            static_for<C0, C1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
                stream_store<Cfg>(reinterpret_cast<C *>(ub + (uint64_t)k0 * A.KS_out * sizeof(C) + lane), v[c]);
            });
        } else {
            C *p = out + (uint64_t)a2 * A.AS_out + e2 + (uint64_t)t2 * A.KS_out;
            static_for<C0, C1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
                stream_store<Cfg>(p + (uint64_t)k0 * A.KS_out, v[c]);
            });
        }
    } else if ((A.store_kind == STORE_TILED_SAME ? Cfg::UNI_STORE_SAME : Cfg::UNI_STORE_TRANSPOSE) && A.stab && A.suni) {
        // wave-uniform entries (see seg_entry_uniform): lane 0's entry per register, the lane's own t2 - t0 added in closed form
        const int t0 = __builtin_amdgcn_readfirstlane(t2);
        const uint32_t dt = (uint32_t)(t2 - t0);
        const SegEntry *tab = A.stab + t0;
        if (A.store_kind == STORE_TILED_SAME) {
            // entry.base = block base + (k - start)*SK: the lane part dt*SK + (b*SB + a*tw + l) is the same for every point
            const uint64_t sk = A.SK ? A.SK : (uint64_t)A.LB * A.LA;
            C *pl = out + ((uint64_t)b2 * (A.SB ? A.SB : (uint64_t)TL * A.LA) + (uint64_t)a2 * tws + l2 + (uint64_t)dt * sk);
            static_for<C0, C1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
                const SegEntry e = seg_entry_uniform(tab + k0);
                stream_store<Cfg>(pl + e.base, v[c]);
            });
        } else {
            // entry.base = block base + kt*T2*LB + kr, aux = width of the consumer tile: kr of the lane is kr + dt
            const uint32_t line = b2 * TL + l2;
            const uint32_t aLB = a2 * A.LB;
            static_for<C0, C1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
                const SegEntry e = seg_entry_uniform(tab + k0);
                char *pu = reinterpret_cast<char *>(out + (e.base + (uint64_t)e.ln * aLB));
                const uint32_t lane = (line * e.aux + dt) * (uint32_t)sizeof(C);
                stream_store<Cfg>(reinterpret_cast<C *>(pu + lane), v[c]);
            });
        }
    } else if (A.stab) {
        const bool same = A.store_kind == STORE_TILED_SAME;
        const uint32_t line = b2 * TL + l2;
        const uint64_t fixed = same ? (uint64_t)b2 * (A.SB ? A.SB : (uint64_t)TL * A.LA) + (uint64_t)a2 * tws + l2 : 0;
        const uint32_t aLB = a2 * A.LB;
        static_for<C0, C1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            const SegEntry e = seg_entry(A.stab + (t2 + k0));
            const uint64_t off = same ? e.base + fixed : e.base + (uint64_t)e.ln * aLB + (uint64_t)line * e.aux;
            stream_store<Cfg>(out + off, v[c]);
        });
    } else if (Cfg::kG == 1 && Cfg::kSUB == 1 && A.store_kind == STORE_TILED_SAME && A.snseg == 1 && !A.shift && !(A.debug & 2) && !A.addr64 &&
               ((uint64_t)NT * (A.SK ? A.SK : (uint64_t)A.LB * A.LA) + TL) * sizeof(C) < (1ull << 32)) {
        // one block, one tile per workgroup: scalar base per point (block, tile, (k0 - start) rows of SK) + the lane's 32-bit
        // (t2*SK + l) bytes (see STORE_KMAJOR)
        const uint64_t sk = A.SK ? A.SK : (uint64_t)A.LB * A.LA, sb = A.SB ? A.SB : (uint64_t)TL * A.LA;
        const uint32_t au = (uint32_t)__builtin_amdgcn_readfirstlane((int)a2), bu = (uint32_t)__builtin_amdgcn_readfirstlane((int)b2);
        const uint32_t twu = (uint32_t)__builtin_amdgcn_readfirstlane((int)tws);
        char *ub = reinterpret_cast<char *>(out + (A.sseg->base[0] + (uint64_t)bu * sb + (uint64_t)au * twu - (uint64_t)A.sseg->start[0] * sk));
        const uint32_t lane = (uint32_t)(((uint64_t)t2 * sk + (uint32_t)l2) * sizeof(C));
        static_for<C0, C1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            stream_store<Cfg>(reinterpret_cast<C *>(ub + (uint64_t)k0 * sk * sizeof(C) + lane), v[c]);
        });
    } else if (Cfg::kG == 1 && Cfg::kSUB == 1 && NT % 16 == 0 && (N / RL) % 16 == 0 && A.store_kind == STORE_TILED_TRANSPOSE && A.snseg == 1 &&
               !A.shift && !(A.debug & 2) && !A.addr64 && A.T2shift <= 4 && A.sseg->start[0] == 0 &&
               (A.sseg->len[0] & ((1u << A.T2shift) - 1)) == 0 && ((uint64_t)NT * A.LB + 16 * TL) * sizeof(C) < (1ull << 32)) {
        // one block of whole consumer tiles, one tile per workgroup, every k0 a multiple of the consumer's tile (T2 <= 16): the tile
        // index and the position inside it separate, kt = (t2 >> sh) + k0 / T2 and kr = t2 % T2, so the point's part k0*LB joins the
        // scalar base (block, a*len*LB, b*TL*T2) and the lane adds 32-bit ((t2 / T2)*T2*LB + l*T2 + kr) bytes (see STORE_KMAJOR)
        const uint32_t sh = A.T2shift, T2 = 1u << sh;
        const uint32_t au = (uint32_t)__builtin_amdgcn_readfirstlane((int)a2), bu = (uint32_t)__builtin_amdgcn_readfirstlane((int)b2);
        char *ub = reinterpret_cast<char *>(out + (A.sseg->base[0] + (uint64_t)au * A.sseg->len[0] * A.LB + (uint64_t)bu * TL * T2));
        const uint32_t lane = ((((uint32_t)t2 >> sh) << sh) * A.LB + (uint32_t)l2 * T2 + ((uint32_t)t2 & (T2 - 1))) * (uint32_t)sizeof(C);
        static_for<C0, C1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            stream_store<Cfg>(reinterpret_cast<C *>(ub + (uint64_t)k0 * A.LB * sizeof(C) + lane), v[c]);
        });
    } else {
        static_for<C0, C1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            const uint32_t k = t2 + k0;
            uint32_t s0 = A.sseg->start[0], ln = A.sseg->len[0];
            uint64_t bs = A.sseg->base[0];
            for (int s = 1; s < A.snseg; s++)
                if (k >= A.sseg->start[s]) { s0 = A.sseg->start[s]; ln = A.sseg->len[s]; bs = A.sseg->base[s]; }
            const uint32_t kl = k - s0;
            uint64_t off;
            if (A.store_kind == STORE_TILED_SAME) {
                const uint64_t sk = A.SK ? A.SK : (uint64_t)A.LB * A.LA, sb = A.SB ? A.SB : (uint64_t)TL * A.LA;
                off = bs + (uint64_t)kl * sk + (uint64_t)b2 * sb + (uint64_t)a2 * tws + l2;
            } else {
                const uint32_t T2 = 1u << A.T2shift;
                const uint32_t kt = kl >> A.T2shift, kr = kl & (T2 - 1);
                const uint32_t r2 = ln - kt * T2;
                const uint32_t tw2 = r2 < T2 ? r2 : T2;
                off = bs + (uint64_t)a2 * ln * A.LB + (uint64_t)kt * T2 * A.LB + ((uint64_t)b2 * TL + l2) * tw2 + kr;
            }
            stream_store<Cfg>(out + off, v[c]);
        });
    }
}

template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void fft_pass_kernel(const PassArgs A)
{
    using C = typename Cfg::C;
    using R = typename Cfg::real;
    constexpr int E = Cfg::kE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *lds = reinterpret_cast<R *>(smem);

    const int tid = threadIdx.x;
    // coordinates for the load / first pass, and for the later passes / the store
    constexpr bool PF_FIRST = Cfg::PF_FIRST, PF_REST = Cfg::PF_REST;
    int lw, t, lw2, t2;
    thread_map<Cfg, PF_FIRST>(tid, lw, t);
    thread_map<Cfg, PF_REST>(tid, lw2, t2);

    const C *__restrict__ in = reinterpret_cast<const C *>(A.in);
    C *__restrict__ out = reinterpret_cast<C *>(A.out);
    const C *__restrict__ W = reinterpret_cast<const C *>(A.tw);
    // inverse transform = conj(forward(conj(x))): one multiplication by +-1 per point and side, no
    // branch and no second copy of the data (a conditional re<->im swap costs both: the compiler keeps
    // the swapped and the unswapped registers alive across the branch, +64 VGPRs at 32 points per thread)
    const R sgn = A.swap ? (R)-1 : (R)1;

    if constexpr (!Cfg::kPERSIST) {
        // workgroup -> tile (b fastest or a fastest, optionally XCD-remapped)
        const uint32_t blk = logical_block<(Cfg::kSUB > 1)>(A, blockIdx.x, gridDim.x);
        C v[E];
        load_tile<Cfg>(A, in, tile_pos<Cfg>(A, blk, lw), t, v);
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].y *= sgn; });
        if (!(A.debug & 1)) transform<Cfg>(v, lds, W, t, lw, t2, lw2);
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].y *= sgn; });
        store_tile<Cfg>(A, out, tile_pos<Cfg>(A, blk, lw2), t2, v);
    } else {
        // PERSIST: persistent workgroups with the STORES of a tile fused with the LOADS of the next one, register by register, for
        // configurations that leave room for only ONE workgroup on a CU (a tile of 16 lines x 1024 fp64 points is the LDS plane of a
        // CU): the grid is one workgroup per CU, each walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ...  A register that has
        // been handed to a store is free (gfx9 stores read their data at issue), so the next tile's load goes into the very same
        // register: no second register set -- prefetching the next tile into one was measured in round 3 and spilled (13.3 - 16.2 ms
        // against 7.68, profiles/r3_strided_read_variants.txt) -- and while the loads of tile i + 1 are in flight the stores of tile i
        // drain: max(load, store) + compute per tile instead of load + compute + store.  Measured (profiles/r4_persist3.txt): the
        // strided read of the API layout at 1024^3 fp64 on one rank 8.27 -> 7.64 ms; no gain on the per-GPU plans of 8 ranks (x^-1 is
        // bound by its access pattern there, not by the phases of a tile).  The chunks of CH registers bound the code size.  The
        // virtual workgroup index keeps the XCD of the hardware workgroup (gridDim.x % 8 == 0).
        const uint32_t nwg = (A.ntiles + Cfg::kG - 1) / Cfg::kG * Cfg::kSUB;
        uint32_t vid = blockIdx.x;
        C v[E];
        load_tile<Cfg>(A, in, tile_pos<Cfg>(A, logical_block<(Cfg::kSUB > 1)>(A, vid, nwg), lw), t, v);
        for (;;) {
            static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].y *= sgn; });
            const uint32_t cur = logical_block<(Cfg::kSUB > 1)>(A, vid, nwg);
            vid += gridDim.x;
            const bool more = vid < nwg;
            if (!(A.debug & 1)) transform<Cfg>(v, lds, W, t, lw, t2, lw2);
            static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].y *= sgn; });
            const TilePos<Cfg::kTL> Ps = tile_pos<Cfg>(A, cur, lw2);
            const TilePos<Cfg::kTL> Pn = tile_pos<Cfg>(A, logical_block<(Cfg::kSUB > 1)>(A, more ? vid : cur, nwg), lw);
            constexpr int CH = E >= 16 ? 8 : E;
            static_for<0, E / CH>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                store_tile<Cfg, q * CH, (q + 1) * CH>(A, out, Ps, t2, v);
                if (more) load_tile<Cfg, q * CH, (q + 1) * CH>(A, in, Pn, t, v);
            });
            if (!more) break;
            if constexpr (Cfg::NPASS > 1) __syncthreads();      // the next tile's first scatter reuses the LDS plane
        }
    }
}

// ------------------------------------------------------------------------------------------
// Real transforms on the z axis (the reference's cufftExecD2Z / cufftExecZ2D plans,
// src/pencil/mpicufft_pencil_opt1.cpp:165-173).  A real line of 2M points is transformed as an
// M-point complex FFT of z[j] = x[2j] + i x[2j+1] plus a split (R2C) or merge (C2R) step:
//   R2C:  X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) w^k (Z[k] - conj Z[M-k]),  k = 0..M
//   C2R:  Z[k] = (X[k] + conj X[M-k]) + i w^-k (X[k] - conj X[M-k]),       k = 0..M-1
// with w = exp(-2*pi*i/(2M)).  Cfg::kN is M.
// ------------------------------------------------------------------------------------------
// ONEPLANE: the split step sends re and im through one LDS plane one after the other (two more
// barriers, half the LDS: twice the workgroups or twice the lines per workgroup on a CU)
template <typename Cfg, int ONEPLANE = 0> struct RealCfg {
    static constexpr size_t SPLIT_BYTES = (ONEPLANE == 2 ? 0 : ONEPLANE ? 1 : 2) * (size_t)Cfg::PLANE_SLOTS * sizeof(typename Cfg::real);
    static constexpr size_t LDS_BYTES = SPLIT_BYTES > Cfg::LDS_BYTES ? SPLIT_BYTES : Cfg::LDS_BYTES;
};

template <int TL> struct TileCtx { uint32_t a, b, tw; int l; };

template <int TL>
__device__ __forceinline__ uint64_t tiled_load_offset(const PassArgs &A, const TileCtx<TL> &c, uint32_t n)
{
    if (A.ltab) {
        const SegEntry e = seg_entry(A.ltab + n);
        return e.base + (uint64_t)e.ln * (c.a * A.LB + c.b * TL) + (uint64_t)e.aux * c.tw + c.l;
    }
    uint32_t s0 = A.lseg->start[0], ln = A.lseg->len[0];
    uint64_t bs = A.lseg->base[0];
    for (int s = 1; s < A.lnseg; s++)
        if (n >= A.lseg->start[s]) { s0 = A.lseg->start[s]; ln = A.lseg->len[s]; bs = A.lseg->base[s]; }
    if (A.IA) return bs + (uint64_t)c.a * A.IA + (uint64_t)c.b * A.IB + (uint64_t)(n - s0) * c.tw + c.l;      // one segment, explicit strides
    return bs + (uint64_t)c.a * ln * A.LB + (uint64_t)c.b * TL * ln + (uint64_t)(n - s0) * c.tw + c.l;
}
// The same for the line-fastest waves of the real kernels: a wave covers SPAN consecutive values of t, whose points k run up or
// down from kf (the point of t0 = the wave's first t) to kl (the point of t0 + SPAN - 1); k, kf, kl come from the same formula.
// If both ends lie in one segment -- their entries carry the same block base -- every lane's entry is the first one moved by
// k - kf points: two scalar loads (s_load_dwordx4) instead of a 16-byte vector load per lane and point.  The segments of an R2C
// spectrum start where the reference's split of Nz/2 + 1 puts them (129 + 128 + 128 + 128: not at multiples of 16 points, so the
// host cannot promise it per launch, PassArgs::luni); the few waves that straddle a segment boundary read their entries per lane.
// Rank 0 of 2 x 4 at 1024^3 fp64, C2R z pass: profiles/r5_c2r_uniform_tables.txt.
template <int TL>
__device__ __forceinline__ uint64_t tiled_load_offset_wave(const PassArgs &A, const TileCtx<TL> &c, uint32_t k, uint32_t kf, uint32_t kl, bool linear)
{
    if (A.ltab) {
        const SegEntry e0 = seg_entry_uniform(A.ltab + kf), e1 = seg_entry_uniform(A.ltab + kl);
        if (linear && e0.base == e1.base && (uint32_t)(e1.aux - e0.aux) == (uint32_t)(kl - kf)) {
            const uint32_t aux = e0.aux + (k - kf);      // (k below kf: the unsigned wrap-around is the negative step)
            return e0.base + (uint64_t)e0.ln * (c.a * A.LB + c.b * TL) + (uint64_t)aux * c.tw + c.l;
        }
    }
    return tiled_load_offset<TL>(A, c, k);
}

template <int TL>
__device__ __forceinline__ uint64_t tiled_transpose_store_offset(const PassArgs &A, const TileCtx<TL> &c, uint32_t k)
{
    if (A.stab) {
        const SegEntry e = seg_entry(A.stab + k);
        return e.base + (uint64_t)e.ln * (c.a * A.LB) + (uint64_t)(c.b * TL + c.l) * e.aux;
    }
    uint32_t s0 = A.sseg->start[0], ln = A.sseg->len[0];
    uint64_t bs = A.sseg->base[0];
    for (int s = 1; s < A.snseg; s++)
        if (k >= A.sseg->start[s]) { s0 = A.sseg->start[s]; ln = A.sseg->len[s]; bs = A.sseg->base[s]; }
    const uint32_t kl = k - s0;
    const uint32_t T2 = 1u << A.T2shift;
    const uint32_t kt = kl >> A.T2shift, kr = kl & (T2 - 1);
    const uint32_t r2 = ln - kt * T2;
    const uint32_t tw2 = r2 < T2 ? r2 : T2;
    return bs + (uint64_t)c.a * ln * A.LB + (uint64_t)kt * T2 * A.LB + ((uint64_t)c.b * TL + c.l) * tw2 + kr;
}

// transposed-tile store into the block of a single peer: the per-thread part is computed once
template <int TL> struct TransposeOne {
    uint64_t base;
    uint32_t s0, ln, LB, line, sh;
    __device__ __forceinline__ TransposeOne(const PassArgs &A, const TileCtx<TL> &c)
        : base(A.sseg->base[0] + (uint64_t)c.a * A.sseg->len[0] * A.LB), s0(A.sseg->start[0]), ln(A.sseg->len[0]),
          LB(A.LB), line(c.b * TL + c.l), sh(A.T2shift) {}
    __device__ __forceinline__ uint64_t operator()(uint32_t k) const
    {
        const uint32_t kl = k - s0, T2 = 1u << sh, kt = kl >> sh, kr = kl & (T2 - 1);
        const uint32_t r2 = ln - (kt << sh), tw2 = r2 < T2 ? r2 : T2;
        return base + (uint64_t)(kt << sh) * LB + (uint64_t)line * tw2 + kr;
    }
};

// element offset of point n (k) of the calling lane's line for every load (store) address form
template <int TL>
__device__ __forceinline__ uint64_t generic_load_offset(const PassArgs &A, const TileCtx<TL> &c, uint32_t n, uint32_t NP)
{
    if (A.load_kind == LOAD_LINES)
        return (A.KS_in ? (uint64_t)c.a * A.AS_in + ((uint64_t)c.b * TL + c.l) * A.KS_in : ((uint64_t)c.a * A.LB + (uint64_t)c.b * TL + c.l) * NP) + n;
    if (A.load_kind == LOAD_KMAJOR) return (uint64_t)n * A.KS_in + (uint64_t)c.a * A.AS_in + (uint64_t)c.b * TL + c.l;
    return tiled_load_offset<TL>(A, c, n);
}
template <int TL>
__device__ __forceinline__ uint64_t generic_store_offset(const PassArgs &A, const TileCtx<TL> &c, uint32_t k, uint32_t NP)
{
    if (A.store_kind == STORE_LINES)
        return (A.KS_out ? (uint64_t)c.a * A.AS_out + ((uint64_t)c.b * TL + c.l) * A.KS_out
                         : ((uint64_t)c.a * A.LB + (uint64_t)c.b * TL + c.l) * NP) + k;
    if (A.store_kind == STORE_KMAJOR) return (uint64_t)k * A.KS_out + (uint64_t)c.a * A.AS_out + (uint64_t)c.b * TL + c.l;
    if (A.store_kind == STORE_TILED_TRANSPOSE) return tiled_transpose_store_offset<TL>(A, c, k);
    const uint64_t sk = A.SK ? A.SK : (uint64_t)A.LB * A.LA, sb = A.SB ? A.SB : (uint64_t)TL * A.LA;
    if (A.stab) return seg_entry(A.stab + k).base + (uint64_t)c.b * sb + (uint64_t)c.a * c.tw + c.l;
    uint32_t s0 = A.sseg->start[0], ln = A.sseg->len[0];
    uint64_t bs = A.sseg->base[0];
    for (int s = 1; s < A.snseg; s++)
        if (k >= A.sseg->start[s]) { s0 = A.sseg->start[s]; ln = A.sseg->len[s]; bs = A.sseg->base[s]; }
    (void)ln;
    return bs + (uint64_t)(k - s0) * sk + (uint64_t)c.b * sb + (uint64_t)c.a * c.tw + c.l;
}

// tile of the calling lane for the real-transform kernels; lw = the lane's line within the workgroup in
// the coordinate set in use (first pass or later passes: they differ for the point-fastest mappings)
template <typename Cfg> __device__ __forceinline__ bool real_tile(const PassArgs &A, int lw, TileCtx<Cfg::kTL> &tc)
{
    constexpr int TL = Cfg::kTL;
    static_assert(Cfg::kSUB == 1, "sub-tile workgroups: fft_pass_kernel only");
    const uint32_t w = logical_block<>(A) * Cfg::kG + lw / TL;
    const bool tile_ok = w < A.ntiles;
    tc.a = !tile_ok ? 0 : (A.a_fastest ? w % A.na : w / A.nb);
    tc.b = !tile_ok ? 0 : (A.a_fastest ? w / A.na : w % A.nb);
    tc.l = lw % TL;
    const uint32_t rem = A.LB - tc.b * TL;
    tc.tw = rem < (uint32_t)TL ? rem : (uint32_t)TL;
    return tile_ok && (uint32_t)tc.l < tc.tw;
}

// forward z pass of an R2C plan: real lines [a][LB][2M] -> tiled send buffer with M+1 points.
// Cfg::kMAP as in fft_pass_kernel: the natural-line load of a fp32 plan wants the point-fastest mapping.
// YLINES = 1 is the y pass of the Y_Then_ZX sequence (strided real lines in, point-major / same-tile blocks out), a
// separate instantiation so that its address forms do not cost the z pass registers (fp32: 194 instead of 166 VGPRs,
// i.e. 2 instead of 3 waves per SIMD, when both lived in one kernel).
template <typename Cfg, int ONEPLANE = 0, int YLINES = 0>
__global__ __launch_bounds__(Cfg::THREADS, ONEPLANE == 2 && is_pow2(Cfg::kN) ? 4 : 1) void fft_r2c_kernel(const PassArgs A)
{
    using C = typename Cfg::C;
    using R = typename Cfg::real;
    constexpr int M = Cfg::kN, E = Cfg::kE, TL = Cfg::kTL, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *lds = reinterpret_cast<R *>(smem);
    const int tid = threadIdx.x;
    // (MAP == 3 -- point fastest after the first pass only on the store side, PassCfg::PF_FIRST / PF_REST -- is a form of
    // fft_pass_kernel; the spans and uniform-table paths of the real kernels assume the mappings below)
    static_assert(Cfg::kMAP != 3, "the real z passes support the thread mappings 0, 1 and 2");
    constexpr bool PF_FIRST = Cfg::kMAP == 1, PF_REST = Cfg::kMAP != 0;
    int lw, t, lw2, t2;
    thread_map<Cfg, PF_FIRST>(tid, lw, t);      // coordinates of the load and the first pass
    thread_map<Cfg, PF_REST>(tid, lw2, t2);     // ... of the later passes, the split step and the store
    TileCtx<TL> tc, tc2;
    const bool active = real_tile<Cfg>(A, lw, tc);
    const bool active2 = real_tile<Cfg>(A, lw2, tc2);
    const C *__restrict__ in = reinterpret_cast<const C *>(A.in);
    C *__restrict__ out = reinterpret_cast<C *>(A.out);
    const C *__restrict__ W = reinterpret_cast<const C *>(A.tw);
    const C *__restrict__ W2 = reinterpret_cast<const C *>(A.tw2);

    C v[E];
    if (YLINES && active) {
        // real lines with a stride, lanes along the contiguous axis (the y pass of the Y_Then_ZX sequence reads the
        // input [x][y][z] in place): real point m of the line at a*AS_in + m*KS_in + b*TL + l, in REAL elements;
        // complex point j of the packed transform is (x[2j], x[2j+1])
        const R *rp = reinterpret_cast<const R *>(A.in) + (uint64_t)tc.a * A.AS_in + (uint64_t)tc.b * TL + tc.l;
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const uint64_t m = 2 * (uint64_t)(t + NT * c);
            v[c].x = rp[m * A.KS_in];
            v[c].y = rp[(m + 1) * A.KS_in];
        });
    } else if (!YLINES && active) {
        const C *p = in + ((uint64_t)tc.a * A.LB + (uint64_t)tc.b * TL + tc.l) * M + t;
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = stream_load<Cfg>(p + NT * c); });
    } else {
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].x = 0; v[c].y = 0; });
    }
    if (A.debug & 1) {          // measurement only: copy with this pass's access pattern
        if (!active) return;
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            out[generic_store_offset<TL>(A, tc, (uint32_t)(t + NT * c), M + 1)] = v[c];
        });
        return;
    }
    constexpr int RL = Cfg::RLAST, S = E / RL;
    if constexpr (ONEPLANE == 2) {
        // The last pass owns butterflies in conjugate pairs (pair_j), so Z[k] and Z[M-k] are registers of this thread and
        // the Hermitian split needs no LDS: register c = i + mr*S holds Z[k], k = j(i) + brev(mr)*(M/RL); its partner
        // Z[M-k] is leg RL-1-m of the mirror block i +- S/2 -- except in thread 0's first block pair, which holds the two
        // self-mirrored butterflies 0 (legs m <-> RL-m) and M/(2 RL) (legs m <-> RL-1-m).
        static_assert(S % 2 == 0 && Cfg::NPASS >= 2, "in-register split: even number of last-pass butterflies per thread");
        transform<Cfg, 1>(v, lds, W, t, lw, t2, lw2);
        if (!active2) return;
        constexpr int H = S / 2, LEG = M / RL;
        const bool special = t2 == 0;
        auto emit_paired = [&](auto offset_of) {
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int i = c % S, mr = c / S, m = brev(mr, RL);
                const int k = pair_j<Cfg, RL, i>(t2) + m * LEG;
                constexpr int pn = (i < H ? i + H : i - H) + brev_inv(RL - 1 - m, RL) * S;      // the mirror block's leg RL-1-m
                constexpr int ps = i < H ? i + brev_inv((RL - m) % RL, RL) * S : i + brev_inv(RL - 1 - m, RL) * S;   // self-mirrored butterflies
                C zm = v[pn];
                if constexpr (i % H == 0) zm = pick(special, v[ps], v[pn]);
                const C z = v[c];
                const C wv = W2[k];
                const R Ar = z.x + zm.x, Ai = z.y - zm.y, Br = z.x - zm.x, Bi = z.y + zm.y;
                C x;
                x.x = (R)0.5 * (Ar + wv.x * Bi + wv.y * Br);
                x.y = (R)0.5 * (Ai - wv.x * Br + wv.y * Bi);
                stream_store<Cfg>(out + offset_of((uint32_t)k), x);
                if constexpr (c == 0) {
                    if (special) {          // k = M: X[M] = Re Z[0] - Im Z[0]
                        C xm; xm.x = z.x - z.y; xm.y = 0;
                        out[offset_of((uint32_t)M)] = xm;
                    }
                }
            });
        };
        if constexpr (YLINES) {
            emit_paired([&](uint32_t k) { return generic_store_offset<TL>(A, tc2, k, M + 1); });
        } else if (A.store_kind == STORE_LINES) {
            const uint64_t row = ((uint64_t)tc2.a * A.LB + (uint64_t)tc2.b * TL + tc2.l) * (uint64_t)(M + 1);
            emit_paired([&](uint32_t k) { return row + k; });
        } else if (A.stab || A.snseg != 1) {
            emit_paired([&](uint32_t k) { return tiled_transpose_store_offset<TL>(A, tc2, k); });
        } else {
            const TransposeOne<TL> one(A, tc2);
            emit_paired([&](uint32_t k) { return one(k); });
        }
        return;
    }
    transform<Cfg>(v, lds, W, t, lw, t2, lw2);

    // split step through LDS: scatter Z by natural index, gather the (k, M-k) pairs
    R *p0 = lds, *p1 = ONEPLANE ? lds : lds + Cfg::PLANE_SLOTS;
    R zr_[ONEPLANE ? E : 1], mr_[ONEPLANE ? E : 1];
    if (Cfg::NPASS > 1) __syncthreads();
    if constexpr (!ONEPLANE) {
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (M / RL);
            const int idx = lds_slot<Cfg>(lw2, t2 + k0);
            p0[idx] = v[c].x;
            p1[idx] = v[c].y;
        });
        __syncthreads();
    } else {
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (M / RL);
            lds[lds_slot<Cfg>(lw2, t2 + k0)] = v[c].x;
        });
        __syncthreads();
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int k = t2 + NT * c, km = is_pow2(M) ? (M - k) & (M - 1) : (k == 0 ? 0 : M - k);
            zr_[c] = lds[lds_slot<Cfg>(lw2, k)];
            mr_[c] = lds[lds_slot<Cfg>(lw2, km)];
        });
        __syncthreads();
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (M / RL);
            lds[lds_slot<Cfg>(lw2, t2 + k0)] = v[c].y;
        });
        __syncthreads();
    }
    if (!active2) return;
    // the address form is chosen once per thread, not once per point
    auto emit = [&](auto offset_of) {
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int k = t2 + NT * c;
            const int km = is_pow2(M) ? (M - k) & (M - 1) : (k == 0 ? 0 : M - k);      // (M - k) mod M
            const int i0 = lds_slot<Cfg>(lw2, k), i1 = lds_slot<Cfg>(lw2, km);
            const R zr = ONEPLANE ? zr_[ONEPLANE ? c : 0] : p0[i0], mr = ONEPLANE ? mr_[ONEPLANE ? c : 0] : p0[i1];
            const R zi = p1[i0], mi = p1[i1];
            const C wv = W2[k];
            const R Ar = zr + mr, Ai = zi - mi, Br = zr - mr, Bi = zi + mi;
            C x;
            x.x = (R)0.5 * (Ar + wv.x * Bi + wv.y * Br);
            x.y = (R)0.5 * (Ai - wv.x * Br + wv.y * Bi);
            out[offset_of((uint32_t)k)] = x;
            if (c == 0 && t2 == 0) {          // k = M: X[M] = Re Z[0] - Im Z[0]
                C xm; xm.x = zr - zi; xm.y = 0;
                out[offset_of((uint32_t)M)] = xm;
            }
        });
    };
    if constexpr (YLINES) {
        // point-major or same-tile stores (Y_Then_ZX: the y pass writes [ky][z/TL][x][z%TL] blocks)
        emit([&](uint32_t k) { return generic_store_offset<TL>(A, tc2, k, M + 1); });
    } else if (A.store_kind == STORE_LINES) {
        // natural [line][M+1] rows (partial transform, d = 1)
        const uint64_t row = ((uint64_t)tc2.a * A.LB + (uint64_t)tc2.b * TL + tc2.l) * (uint64_t)(M + 1);
        emit([&](uint32_t k) { return row + k; });
    } else if (A.stab || A.snseg != 1) {
        emit([&](uint32_t k) { return tiled_transpose_store_offset<TL>(A, tc2, k); });
    } else {
        const TransposeOne<TL> one(A, tc2);      // the tiled send buffer of a single peer
        emit([&](uint32_t k) { return one(k); });
    }
}

// inverse z pass of an R2C plan: tiled recv buffer with M+1 points -> real lines [a][LB][2M].
// Cfg::kMAP == 2 (line-fastest tiled load, point-fastest natural-line store) is the fp32 form.
template <typename Cfg, int PAIRED = 0>
__global__ __launch_bounds__(Cfg::THREADS) void fft_c2r_kernel(const PassArgs A)
{
    using C = typename Cfg::C;
    using R = typename Cfg::real;
    constexpr int M = Cfg::kN, E = Cfg::kE, TL = Cfg::kTL, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *lds = reinterpret_cast<R *>(smem);
    const int tid = threadIdx.x;
    static_assert(Cfg::kMAP != 3, "the real z passes support the thread mappings 0, 1 and 2");
    constexpr bool PF_FIRST = Cfg::kMAP == 1, PF_REST = Cfg::kMAP != 0 && Cfg::NPASS > 1 ? true : Cfg::kMAP == 1;
    int lw, t, lw2, t2;
    thread_map<Cfg, PF_FIRST>(tid, lw, t);
    thread_map<Cfg, PF_REST>(tid, lw2, t2);
    TileCtx<TL> tc, tc2;
    const bool active = real_tile<Cfg>(A, lw, tc);
    const bool active2 = real_tile<Cfg>(A, lw2, tc2);
    const C *__restrict__ in = reinterpret_cast<const C *>(A.in);
    C *__restrict__ out = reinterpret_cast<C *>(A.out);
    const C *__restrict__ W = reinterpret_cast<const C *>(A.tw);
    const C *__restrict__ W2 = reinterpret_cast<const C *>(A.tw2);

    C v[E];
    if constexpr (PAIRED) {
        // The FIRST pass owns butterflies in conjugate pairs (pair_j): every X[k] is loaded once, its partner X[M-k] is a
        // register of the same thread (see fft_r2c_kernel), the merge happens in registers before the first butterflies.
        constexpr int R1 = Cfg::r1, S1 = E / R1, H = S1 / 2, LEG = M / R1;
        static_assert(S1 % 2 == 0 && Cfg::NPASS >= 2, "in-register merge: even number of first-pass butterflies per thread");
        C x[E];
        C xM; xM.x = 0; xM.y = 0;
        const bool special = t == 0;
        auto fetch_paired = [&](auto offset_of) {
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int i = c % S1, m = c / S1;                  // input leg m of block i (natural leg order)
                const int k = pair_j<Cfg, R1, i>(t) + m * LEG;
                x[c] = stream_load<Cfg>(in + offset_of((uint32_t)k));
            });
            if (special) xM = in[offset_of((uint32_t)M)];
        };
        // segmented tiled load with wave-uniform table entries where the wave's points lie in one segment (tiled_load_offset_wave).
        // Line-fastest first pass only; thread 0's mirrored butterflies (pair_j: t = 0 is its own mirror) break the run of a wave.
        auto fetch_paired_wave = [&]() {
            constexpr int SPAN = Cfg::SPAN_LF;
            const int t0 = __builtin_amdgcn_readfirstlane(t);
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int i = c % S1, m = c / S1;
                const int k = pair_j<Cfg, R1, i>(t) + m * LEG;
                // (the wave's last t: clamped to the line's last thread -- a workgroup smaller than a wave, round-5 advice)
                const int tl = t0 + SPAN - 1 < NT ? t0 + SPAN - 1 : NT - 1;
                const int kf = pair_j<Cfg, R1, i>(t0) + m * LEG, kl = pair_j<Cfg, R1, i>(tl) + m * LEG;
                const bool linear = i < H || t0 != 0;
                x[c] = stream_load<Cfg>(in + tiled_load_offset_wave<TL>(A, tc, (uint32_t)k, (uint32_t)kf, (uint32_t)kl, linear));
            });
            if (special) xM = in[tiled_load_offset<TL>(A, tc, (uint32_t)M)];
        };
        if (active) {
            if (A.load_kind == LOAD_LINES) {
                const uint64_t row = ((uint64_t)tc.a * A.LB + (uint64_t)tc.b * TL + tc.l) * (uint64_t)(M + 1);
                fetch_paired([&](uint32_t k) { return row + k; });
            } else if (A.ltab && !PF_FIRST && Cfg::TW <= 64) {
                fetch_paired_wave();
            } else if (A.ltab || A.lnseg != 1) {
                fetch_paired([&](uint32_t k) { return tiled_load_offset<TL>(A, tc, k); });
            } else {
                const uint64_t len = A.lseg->len[0];
                const uint64_t base = A.lseg->base[0] + (uint64_t)tc.a * (A.IA ? A.IA : len * A.LB) + (uint64_t)tc.b * (A.IB ? A.IB : (uint64_t)TL * len) + tc.l;
                const uint32_t s0 = A.lseg->start[0], tw = tc.tw;
                fetch_paired([&](uint32_t k) { return base + (uint64_t)(k - s0) * tw; });
            }
        } else {
            static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; x[c].x = 0; x[c].y = 0; });
        }
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int i = c % S1, m = c / S1;
            const int k = pair_j<Cfg, R1, i>(t) + m * LEG;
            constexpr int pn = (i < H ? i + H : i - H) + (R1 - 1 - m) * S1;
            constexpr int ps = i < H ? i + ((R1 - m) % R1) * S1 : i + (R1 - 1 - m) * S1;
            C xk = x[c];
            C xm = x[pn];
            if constexpr (i == 0 && m == 0) xm = pick(special, xM, x[pn]);             // k = 0 pairs with X[M]
            else if constexpr (i % H == 0) xm = pick(special, x[ps], x[pn]);
            if (k == 0) { xk.y = 0; xm.y = 0; }      // imaginary parts of X[0], X[M] are ignored
            const C wv = W2[k];
            const R Ar = xk.x + xm.x, Ai = xk.y - xm.y, Br = xk.x - xm.x, Bi = xk.y + xm.y;
            // swapped on the fly (inverse via re<->im swap): v = (Im Z', Re Z')
            v[c].y = Ar - wv.x * Bi + wv.y * Br;
            v[c].x = Ai + wv.x * Br + wv.y * Bi;
        });
    } else
    if (active) {
        auto fetch = [&](auto offset_of) {
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int k = t + NT * c;
                C x = in[offset_of((uint32_t)k)];
                C m = in[offset_of((uint32_t)(M - k))];
                if (k == 0) { x.y = 0; m.y = 0; }      // imaginary parts of X[0], X[M] are ignored
                const C wv = W2[k];
                const R Ar = x.x + m.x, Ai = x.y - m.y, Br = x.x - m.x, Bi = x.y + m.y;
                // swapped on the fly (inverse via re<->im swap): v = (Im Z', Re Z')
                v[c].y = Ar - wv.x * Bi + wv.y * Br;
                v[c].x = Ai + wv.x * Br + wv.y * Bi;
                // every point needs two loads (X[k], X[M-k]): left alone the scheduler hoists all 2E of them above the arithmetic, which
                // at 24-30 fp64 points per thread is more registers than a lane has (212-516 bytes per lane of scratch in the
                // mixed-radix C2R kernels); a fence every few points keeps a bounded number in flight
                if constexpr (E > 16 && c % 6 == 5) __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (A.load_kind == LOAD_LINES) {
            const uint64_t row = ((uint64_t)tc.a * A.LB + (uint64_t)tc.b * TL + tc.l) * (uint64_t)(M + 1);
            fetch([&](uint32_t k) { return row + k; });
        } else if (A.ltab || A.lnseg != 1) {
            fetch([&](uint32_t k) { return tiled_load_offset<TL>(A, tc, k); });
        } else {
            // one peer: the tile is one contiguous [point][line] block
            const uint64_t len = A.lseg->len[0];
            const uint64_t base = A.lseg->base[0] + (uint64_t)tc.a * (A.IA ? A.IA : len * A.LB) + (uint64_t)tc.b * (A.IB ? A.IB : (uint64_t)TL * len) + tc.l;
            const uint32_t s0 = A.lseg->start[0], tw = tc.tw;
            fetch([&](uint32_t k) { return base + (uint64_t)(k - s0) * tw; });
        }
    } else {
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].x = 0; v[c].y = 0; });
    }
    if (!(A.debug & 1)) transform<Cfg, PAIRED ? 2 : 0>(v, lds, W, t, lw, t2, lw2);
    if (!active2) return;
    constexpr int RL = Cfg::RLAST, S = E / RL;
    C *p = out + ((uint64_t)tc2.a * A.LB + (uint64_t)tc2.b * TL + tc2.l) * M + t2;
    static_for<0, E>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (M / RL);
        C r; r.x = v[c].y; r.y = v[c].x;     // swap back
        stream_store<Cfg>(p + k0, r);
    });
}


#ifdef DFFT_EXPERIMENTS
// ------------------------------------------------------------------------------------------
// A/B only (north_star: "wavefront shuffles for the twiddle/radix stages"): an LDS-free pass.  A line of
// N = 16*E points lives in one 16-lane row (lane t holds x[t + 16 c], c < E): radix-E butterflies in registers,
// twiddles, then a 16-point decimation-in-frequency transform ACROSS the 16 lanes with xor shuffles (distances
// 8, 4, 2, 1).  No LDS, no barriers, four lines per wave.  Its price is structural: lanes end up holding the HIGH
// part of the output index (k = q + E*bitrev(t)), so one side of the pass is accessed with a stride of E points --
// the register<->lane transposition that the LDS exchange of fft_pass_kernel gives for free.  Natural lines only.
// Measured against the LDS configurations in profiles/r2_shuffle_stage.txt.
// ------------------------------------------------------------------------------------------
// xor shuffle inside a 16-lane row: DPP moves (quad_perm, row_mirror, row_half_mirror: gfx9 has no row_xmask) or
// ds_bpermute through __shfl_xor
template <int D, int DPP> __device__ __forceinline__ float row_xor(float x)
{
    if constexpr (!DPP) return __shfl_xor(x, D, 16);
    else {
        int v = __float_as_int(x);
        if constexpr (D == 1) v = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);                 // quad_perm [1,0,3,2]
        if constexpr (D == 2) v = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);                 // quad_perm [2,3,0,1]
        if constexpr (D == 4) { v = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); v = __builtin_amdgcn_mov_dpp(v, 0x1B, 0xF, 0xF, true); }   // half mirror (^7), reverse quads (^3)
        if constexpr (D == 8) { v = __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); v = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); }  // mirror (^15), half mirror (^7)
        return __int_as_float(v);
    }
}

template <typename R, int N, int DPP> __global__ __launch_bounds__(256) void fft_shfl_kernel(const PassArgs A)
{
    using C = typename Vec2<R>::type;
    constexpr int E = N / 16;
    const int tid = threadIdx.x, t = tid & 15;
    const uint32_t line = blockIdx.x * 16 + (tid >> 4);
    if (line >= A.LB * A.na) return;
    const C *__restrict__ in = reinterpret_cast<const C *>(A.in) + (uint64_t)line * N;
    C *__restrict__ out = reinterpret_cast<C *>(A.out) + (uint64_t)line * N;
    const C *__restrict__ W = reinterpret_cast<const C *>(A.tw);
    C v[E];
    static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = in[t + 16 * c]; });
    const R sgn = A.swap ? (R)-1 : (R)1;
    static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c].y *= sgn; });
    Dif<E, 0, 1, C>::run(v);                    // slot m holds Y[t][q], q = brev(m, E)
    static_for<1, E>([&](auto mc) {             // twiddle w_N^(t q)
        constexpr int m = decltype(mc)::value;
        constexpr int q = brev(m, E);
        const C w = W[(t * q) & (N - 1)];
        const C x = v[m];
        v[m].x = x.x * w.x - x.y * w.y;
        v[m].y = x.x * w.y + x.y * w.x;
    });
    static_for<0, 4>([&](auto sc) {             // 16-point DIF across the lanes of a row
        constexpr int d = 8 >> decltype(sc)::value;
        const bool upper = (t & d) != 0;
        const R sg = upper ? (R)-1 : (R)1;
        C tw; tw.x = 1; tw.y = 0;
        if (d > 1 && upper) tw = W[(t & (d - 1)) * (N / (2 * d))];
        static_for<0, E>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            C p;
            p.x = row_xor<d, DPP>(v[m].x);
            p.y = row_xor<d, DPP>(v[m].y);
            C r;
            r.x = p.x + sg * v[m].x;            // lower lane: v + partner, upper lane: partner - v
            r.y = p.y + sg * v[m].y;
            if (d > 1) { v[m].x = r.x * tw.x - r.y * tw.y; v[m].y = r.x * tw.y + r.y * tw.x; }
            else v[m] = r;
        });
    });
    const int k1 = ((t & 1) << 3) | ((t & 2) << 1) | ((t & 4) >> 1) | ((t & 8) >> 3);      // bit reversal of the lane
    static_for<0, E>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int q = brev(m, E);
        C r = v[m];
        r.y *= sgn;
        out[q + E * k1] = r;
    });
}
#endif

// ------------------------------------------------------------------------------------------
// Arbitrary line lengths (the reference accepts any size through cuFFT): Bluestein's chirp-z
// algorithm on top of the power-of-two Stockham chain, all inside one kernel and with the same
// fused load/store address forms as fft_pass_kernel.
//   X[k] = w[k] * sum_n (x[n] w[n]) conj(w)[k-n],   w[n] = exp(-i*pi*n^2/NL)
// = one zero-padded N-point forward transform, a pointwise product with the precomputed
// spectrum of the conjugate chirp (tw3, includes the 1/N of the inverse), one N-point inverse
// transform (re<->im swap), and two chirp multiplications.  N = Cfg::kN >= 2*NL - 1.
// Real plans (real_mode 1/2) run the full complex transform of the real line and keep / rebuild
// the Hermitian half, which also covers odd lengths.
// ------------------------------------------------------------------------------------------
// registers hold the outputs of a transform (slot c <-> index t + k0(c)); bring them back to the
// input order of the next transform (slot c <-> index t + NT*c) through the LDS plane
template <typename Cfg>
__device__ __forceinline__ void reorder_natural(typename Cfg::C *v, typename Cfg::real *lds, int t, int lw, int tid, bool lds_dirty)
{
    using R = typename Cfg::real;
    constexpr int N = Cfg::kN, E = Cfg::kE, NT = Cfg::NT, TW = Cfg::TW, RL = Cfg::RLAST, S = E / RL;
    if constexpr (Cfg::NPASS == 1) {
        // single thread per line: a register permutation
        typename Cfg::C tmp[E];
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL); tmp[k0] = v[c]; });
        static_for<0, E>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = tmp[c]; });
    } else {
        static_for<0, 2>([&](auto pc) {
            constexpr int comp = decltype(pc)::value;
            if (lds_dirty || comp == 1) __syncthreads();
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
                lds[lds_pad<Cfg>((t + k0) * TW + lw)] = comp == 0 ? v[c].x : v[c].y;
            });
            __syncthreads();
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const R val = lds[lds_pad<Cfg>(tid + NT * TW * c)];
                if (comp == 0) v[c].x = val; else v[c].y = val;
            });
        });
    }
}

// Two-level lines (PassArgs::lv; the reference takes any length through cuFFT, mpicufft_pencil_opt1.cpp:165-197): a line of
// N = N1*N2 points that has no kernel of its own is transformed by two launches of this kernel (the four-step algorithm):
//   level 1 (lv = 1, NL = N1, lvQ = N2): sub-line q = n2 of a line holds its points n = n1*N2 + n2, loaded through the pass's
//            own load address form; N1-point transform; times exp(-2 pi i k1 n2 / N) (lvtw); stored to the scratch lvw
//   level 2 (lv = 2, NL = N2, lvQ = N1): sub-line q = k1 from the scratch; N2-point transform; output k2 is the line's point
//            k = k1 + N1*k2, stored through the pass's own store address form
// so that unpack / transpose / pack stay fused exactly as for a one-launch pass, whatever the layouts.  The lanes of a kernel
// tile run over the TL lines of a tile of the pass (outer side tiled or point-major: 128-byte runs) or over TL neighbouring
// sub-lines of one line (outer side natural lines: contiguous points), PassArgs::lvqm, per level.  Scratch layout by the pair:
//   both over lines:                 tiled, element (tile w, k1, n2, line l) at ((w*N1 + k1)*N2 + n2)*TL + l
//   level 1 over sub-lines:          [line][k1][n2] -- level 1 stores runs over n2, level 2 reads runs over n2 (its threads of a line)
//   level 2 only over sub-lines:     [line][n2][k1] -- level 1 stores runs over k1 (its threads of a line), level 2 reads runs over k1
// Each level is either the plain chain (PassArgs::plain: NL == Cfg::kN, a power of two) or Bluestein on an arbitrary factor.
// Real lines: level 1 reads the real line / rebuilds the full spectrum from the Hermitian half, level 2 keeps the half /
// the real parts (the full complex transform of the line, as in the one-launch real modes).
template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void fft_bluestein_kernel(const PassArgs A)
{
    using C = typename Cfg::C;
    using R = typename Cfg::real;
    constexpr int N = Cfg::kN, E = Cfg::kE, TL = Cfg::kTL, NT = Cfg::NT, TW = Cfg::TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *lds = reinterpret_cast<R *>(smem);
    const int tid = threadIdx.x;
    const int lw = tid % TW, t = tid / TW;
    // tile and line of this thread; sub-tile workgroups (inner transforms of 4096 / 8192 points): lines
    // [sub*TLK, sub*TLK + TLK) of tile blk / SUB, as in fft_pass_kernel
    uint32_t w;
    int l;
    if constexpr (Cfg::kSUB > 1) {
        const uint32_t blk = logical_block<true>(A);
        w = blk / Cfg::kSUB;
        l = (int)(blk % Cfg::kSUB) * Cfg::TLK + lw;
    } else {
        w = logical_block<>(A) * Cfg::kG + lw / TL;
        l = lw % TL;
    }
    // two-level lines: the launch's tiles are (tile, sub-line) pairs, sub-line fastest -- or, where the level's outer side has
    // natural lines (lvqm), (line, block of TL sub-lines) pairs: the lanes of a tile then run over neighbouring sub-lines of ONE
    // line, i.e. over contiguous points of the natural line, instead of over the lines of a tile
    const int lv = A.lv;
    const bool qm = lv && A.lvqm;
    const uint32_t Q = lv ? A.lvQ : 1u;
    uint32_t q = 0;
    TileCtx<TL> tc;
    bool active;
    if (!qm) {
        if (lv) { q = w % Q; w /= Q; }
        const bool tile_ok = w < A.ntiles;
        tc.a = !tile_ok ? 0 : (A.a_fastest ? w % A.na : w / A.nb);
        tc.b = !tile_ok ? 0 : (A.a_fastest ? w / A.na : w % A.nb);
        tc.l = l;
        const uint32_t rem = A.LB - tc.b * TL;
        tc.tw = rem < (uint32_t)TL ? rem : (uint32_t)TL;
        active = tile_ok && (uint32_t)l < tc.tw;
    } else {
        const uint32_t QB = (Q + TL - 1) / TL;
        const uint32_t lam = w / QB;                    // line a*LB + (b*TL + l)
        q = (w % QB) * TL + (uint32_t)l;
        active = lam < A.na * A.LB && q < Q;
        const uint32_t r = active ? lam % A.LB : 0u;
        tc.a = active ? lam / A.LB : 0u;
        tc.b = r / TL;
        tc.l = (int)(r % TL);
        const uint32_t rem = A.LB - tc.b * TL;
        tc.tw = rem < (uint32_t)TL ? rem : (uint32_t)TL;
    }
    const uint32_t NL = A.NL;
    // the line as the pass's address forms see it: its length, its points on the spectral side of a real transform
    // (a long-Bluestein launch: the pass's own line has lbL / lbK points, the two-level line around it lvN)
    const uint32_t NS = lv ? A.lvN : NL;                                      // points of the line in the scratch between the levels
    const uint32_t NF = A.lb ? A.lbL : NS, NKF = A.lb ? A.lbK : (lv ? A.lvNK : A.NK);
    const C *__restrict__ LBT = reinterpret_cast<const C *>(A.lbtab);
    const bool plain = A.plain != 0;
    const C *__restrict__ in = reinterpret_cast<const C *>(A.in);
    const R *__restrict__ rin = reinterpret_cast<const R *>(A.in);
    C *__restrict__ out = reinterpret_cast<C *>(A.out);
    R *__restrict__ rout = reinterpret_cast<R *>(A.out);
    const C *__restrict__ W = reinterpret_cast<const C *>(A.tw);
    const C *__restrict__ CH = reinterpret_cast<const C *>(A.tw2);
    const C *__restrict__ BH = reinterpret_cast<const C *>(A.tw3);

    C v[E];
    // fetch(n): point n of this thread's (sub-)line, before the chirp
    auto load_all = [&](auto fetch) {
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const uint32_t n = t + NT * c;
            C x; x.x = 0; x.y = 0;
            if (active && n < NL) {
                x = fetch(n);
                if (!plain) {
                    const C ch = CH[n];
                    C r; r.x = x.x * ch.x - x.y * ch.y; r.y = x.x * ch.y + x.y * ch.x;
                    x = r;
                }
            }
            v[c] = x;
        });
    };
    // the address form (load kind) and the real mode are chosen once per thread, not per point;
    // offset_of(n, NP): element offset of point n of this thread's line, NP = points per natural line
    auto load_ext = [&](auto offset_of) {
        load_all([&](uint32_t n) {
            const uint32_t nf = lv ? n * Q + q : n;
            C x; x.x = 0; x.y = 0;
            if ((A.lb & 1) && nf >= NF) return x;      // zero padding of a long-Bluestein line
            if (A.real_mode == 1) {            // real input line
                x.x = rin[offset_of(nf, NF)];
            } else if (A.real_mode == 2) {     // Hermitian half in, rebuild the full spectrum
                if (nf < NKF) {
                    x = in[offset_of(nf, NKF)];
                    if (nf == 0 || 2 * nf == NF) x.y = 0;
                } else {
                    x = in[offset_of(NF - nf, NKF)];
                    x.y = -x.y;
                }
            } else {
                x = in[offset_of(nf, NF)];
            }
            if (A.swap) { R tmp = x.x; x.x = x.y; x.y = tmp; }
            if (A.lb & 1) {                    // long Bluestein: times the chirp of the LINE's point
                const C ch = LBT[nf];
                C r; r.x = x.x * ch.x - x.y * ch.y; r.y = x.x * ch.y + x.y * ch.x;
                x = r;
            }
            return x;
        });
    };
    const uint64_t rowline = (uint64_t)tc.a * A.LB + (uint64_t)tc.b * TL + tc.l;
    // scratch element (i1, i2) of this thread's line: tiled (lvlay 0, both levels with lanes over the lines of a tile) at
    // ((w*N1 + i1)*N2 + i2)*TL + l, else in the line's own N points at i1*lvs1 + i2*lvs2
    const uint64_t sbase = A.lvlay == 0 ? (uint64_t)w * NS * TL + (uint32_t)l : rowline * NS;
    if (lv == 2) {                                       // second level: sub-line q = i1 of the scratch
        const C *__restrict__ ws = reinterpret_cast<const C *>(A.lvw) + sbase + (uint64_t)q * A.lvs1;
        const uint64_t s2 = A.lvs2;
        load_all([&](uint32_t n) { return ws[(uint64_t)n * s2]; });
    } else if (A.load_kind == LOAD_LINES && A.KS_in) {          // strided natural-line rows
        const uint64_t row = (uint64_t)tc.a * A.AS_in + ((uint64_t)tc.b * TL + tc.l) * A.KS_in;
        load_ext([&](uint32_t n, uint32_t) { return row + n; });
    } else if (A.load_kind == LOAD_LINES) {
        load_ext([&](uint32_t n, uint32_t NP) { return rowline * NP + n; });
    } else if (A.load_kind == LOAD_KMAJOR) {
        const uint64_t base = (uint64_t)tc.a * A.AS_in + (uint64_t)tc.b * TL + tc.l, ks = A.KS_in;
        load_ext([&](uint32_t n, uint32_t) { return base + (uint64_t)n * ks; });
    } else if (A.ltab || A.lnseg != 1) {
        load_ext([&](uint32_t n, uint32_t) { return tiled_load_offset<TL>(A, tc, n); });
    } else {
        const uint64_t len = A.lseg->len[0];
        const uint64_t base = A.lseg->base[0] + (uint64_t)tc.a * (A.IA ? A.IA : len * A.LB) + (uint64_t)tc.b * (A.IB ? A.IB : (uint64_t)TL * len) + tc.l;
        const uint32_t s0 = A.lseg->start[0], tw = tc.tw;
        load_ext([&](uint32_t n, uint32_t) { return base + (uint64_t)(n - s0) * tw; });
    }
    transform<Cfg>(v, lds, W, t, lw, tid);
    if (!plain) {
        {   // pointwise product with the chirp spectrum, then swap for the inverse transform
            constexpr int RL = Cfg::RLAST, S = E / RL;
            static_for<0, E>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
                const C b = BH[t + k0];
                C x = v[c];
                v[c].y = x.x * b.x - x.y * b.y;      // swapped: (im, re)
                v[c].x = x.x * b.y + x.y * b.x;
            });
        }
        reorder_natural<Cfg>(v, lds, t, lw, tid, Cfg::NPASS > 1);
        if (Cfg::NPASS > 1) __syncthreads();
        transform<Cfg>(v, lds, W, t, lw, tid);
    }
    if (!active) return;
    // put(k, r): output k of this thread's (sub-)line, after the chirp
    auto store_all = [&](uint32_t kmax, auto put) {
        constexpr int RL = Cfg::RLAST, S = E / RL;
        static_for<0, E>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int k0 = NT * (c % S) + brev(c / S, RL) * (N / RL);
            const uint32_t k = t + k0;
            if (k < kmax) {
                C r = v[c];
                if (!plain) {
                    C x; x.x = v[c].y; x.y = v[c].x;       // swap back
                    const C ch = CH[k];
                    r.x = x.x * ch.x - x.y * ch.y; r.y = x.x * ch.y + x.y * ch.x;
                }
                put(k, r);
            }
        });
    };
    if (lv == 1) {                                       // first level: twiddle, then the scratch
        C *__restrict__ ws = reinterpret_cast<C *>(A.lvw) + sbase + (uint64_t)q * A.lvs2;      // sub-line q = i2
        const uint64_t s1 = A.lvs1;
        const C *__restrict__ TWN = reinterpret_cast<const C *>(A.lvtw);
        store_all(NL, [&](uint32_t k, C r) {
            const C tw = TWN[k * q];
            C y; y.x = r.x * tw.x - r.y * tw.y; y.y = r.x * tw.y + r.y * tw.x;
            ws[(uint64_t)k * s1] = y;
        });
        return;
    }
    const uint32_t kend = A.real_mode == 1 ? NKF : NF;      // points of the line that are stored
    auto store_ext = [&](auto offset_of) {
        store_all(lv ? NL : kend, [&](uint32_t k, C r) {
            const uint32_t kf = lv ? q + Q * k : k;
            if (lv && kf >= kend) return;
            if (A.lb & 4) { R tmp = r.x; r.x = r.y; r.y = tmp; }
            if (A.lb & 2) {
                const C m = LBT[kf];
                C y; y.x = r.x * m.x - r.y * m.y; y.y = r.x * m.y + r.y * m.x;
                r = y;
            }
            if (A.swap) { R tmp = r.x; r.x = r.y; r.y = tmp; }
            if (A.real_mode == 2) rout[offset_of(kf)] = r.x;
            else out[offset_of(kf)] = r;
        });
    };
    if (A.store_kind == STORE_LINES) {
        const uint64_t row = A.KS_out ? (uint64_t)tc.a * A.AS_out + ((uint64_t)tc.b * TL + tc.l) * A.KS_out
                                      : rowline * (A.real_mode == 2 ? NF : kend);
        store_ext([&](uint32_t k) { return row + k; });
    } else if (A.store_kind == STORE_KMAJOR) {
        const uint64_t base = (uint64_t)tc.a * A.AS_out + (uint64_t)tc.b * TL + tc.l, ks = A.KS_out;
        store_ext([&](uint32_t k) { return base + (uint64_t)k * ks; });
    } else if (A.stab || A.snseg != 1) {
        store_ext([&](uint32_t k) { return generic_store_offset<TL>(A, tc, k, kend); });
    } else if (A.store_kind == STORE_TILED_TRANSPOSE) {
        const TransposeOne<TL> one(A, tc);
        store_ext([&](uint32_t k) { return one(k); });
    } else {
        const uint64_t base = A.sseg->base[0] + (uint64_t)tc.b * (A.SB ? A.SB : (uint64_t)TL * A.LA) + (uint64_t)tc.a * tc.tw + tc.l;
        const uint64_t step = A.SK ? A.SK : (uint64_t)A.LB * A.LA;
        const uint32_t s0 = A.sseg->start[0];
        store_ext([&](uint32_t k) { return base + (uint64_t)(k - s0) * step; });
    }
}

}  // namespace dfft
