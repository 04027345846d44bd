// fp32 instantiations of the axis-pass kernel (TL = 16 lines per tile: 16 x 8 B = 128 B runs)
#include "kernels.hip.inc"

namespace dfft {
using F32_2    = PassCfg<float, 2,    2, 16, 16, 2, 1, 1, 1,   1>;
using F32_4    = PassCfg<float, 4,    4, 16, 16, 4, 1, 1, 1,   1>;
using F32_8    = PassCfg<float, 8,    8, 16, 16, 8, 1, 1, 1,   1>;
using F32_16   = PassCfg<float, 16,  16, 16, 16, 16, 1, 1, 1,  1>;
using F32_32   = PassCfg<float, 32,   8, 16, 4,  8, 4, 1, 1,   2>;
using F32_64   = PassCfg<float, 64,   8, 16, 2,  8, 8, 1, 1,   2>;
using F32_128  = PassCfg<float, 128, 16, 16, 2,  16, 8, 1, 1,  2>;
using F32_256  = PassCfg<float, 256, 16, 16, 1,  16, 16, 1, 1, 2>;
using F32_512  = PassCfg<float, 512, 16, 16, 1,  8, 8, 8, 1,   2>;
// 32 points per thread: fp32 runs out of instruction issue, not bandwidth, at 16
using F32_1024 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1,  1, 1>;
using F32_2048 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1,  1, 1>;
// The variant number of a configuration is its ROLE in a plan (dfft_init picks by role, see PassRole):
//   4 = natural-line load: point-fastest lane mapping in every pass (PassCfg::MAP = 1; forward z pass)
//   5 = natural-line store: line-fastest first pass (tiled load), point-fastest afterwards (MAP = 2; inverse z pass)
//   6 = tiled passes (y, x): two radix passes with a single LDS exchange
using F32_1024_v4 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1, 0, 1>;
using F32_1024_v5 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1, 0, 2>;
using F32_1024_v6 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1>;
using F32_512_v6 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1>;
using F32_512_v4 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 1>;
using F32_512_v5 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 2>;
// 2048: 16 lines x 2048 points are 256 KiB -- one workgroup per CU.  The natural-line passes (4, 5) run 64 points
// per thread (radix 64.32, a single LDS exchange) on sub-tile workgroups of 8 lines (PassCfg::SUB = 2: 256 threads,
// 66 KiB LDS, two per CU): 36.2 -> 30.4 ms per pass at 2048^3.  The tiled passes (6) keep whole tiles (a sub-tile's
// 64-byte runs cost more than its occupancy gives) with the same two-pass chain on 512 threads.
using F32_2048_v4 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 3, 1, 2>;
using F32_2048_v5 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 0, 2, 2>;
using F32_2048_v6 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1>;
// 4096 and 8192 points: sub-tile workgroups of 4 / 2 lines (see kernels_f64.hip), 512 threads, 64 KiB of LDS
using F32_4096 = PassCfg<float, 4096, 32, 16, 1, 32, 16, 8, 1, 1, 1, 0, 0, 4>;
using F32_8192 = PassCfg<float, 8192, 32, 16, 1, 32, 16, 16, 1, 1, 1, 0, 0, 8>;
// A/B-only configurations of earlier measurements (sub-tile workgroups on tiled passes, nontemporal loads-only / stores-only,
// whole-tile 64-point forms, 32-point fp64 2048, ...) were removed after they were measured: results in profiles/r2_*.txt and
// DESIGN.md section 6, definitions in the git history (commit c38cf04).  New ones go here, under -DDFFT_EXPERIMENTS:

// persistent, software-pipelined forms (PassCfg::PERSIST) under test
using F32_2048_v8 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 0, 0, 1, 1>;
using F32_2048_v9 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 3, 0, 1, 1>;
#ifdef DFFT_EXPERIMENTS
#define DFFT_F32_EXP_SMALL(X)
#define DFFT_F32_EXP_1024(X)
#define DFFT_F32_EXP_2048(X) X(2048, 8, F32_2048_v8) X(2048, 9, F32_2048_v9)
#else
#define DFFT_F32_EXP_SMALL(X)
#define DFFT_F32_EXP_1024(X)
#define DFFT_F32_EXP_2048(X)
#endif
#define DFFT_F32_LIST_SMALL(X) X(512, 6, F32_512_v6) X(512, 4, F32_512_v4) X(512, 5, F32_512_v5) X(2, 0, F32_2) X(4, 0, F32_4) X(8, 0, F32_8) X(16, 0, F32_16) X(32, 0, F32_32) X(64, 0, F32_64) X(128, 0, F32_128) X(256, 0, F32_256) X(512, 0, F32_512) DFFT_F32_EXP_SMALL(X)
#define DFFT_F32_LIST_1024(X) X(1024, 4, F32_1024_v4) X(1024, 5, F32_1024_v5) X(1024, 6, F32_1024_v6) X(1024, 0, F32_1024) DFFT_F32_EXP_1024(X)
#define DFFT_F32_LIST_2048(X) X(2048, 4, F32_2048_v4) X(2048, 5, F32_2048_v5) X(2048, 6, F32_2048_v6) X(2048, 0, F32_2048) X(4096, 0, F32_4096) X(8192, 0, F32_8192) DFFT_F32_EXP_2048(X)

// lengths that are not powers of two (mixed radix 2, 3, 5, 7): generated list, slices 5 (N < 512) and 6
#define DFFT_MIXED_F32
#include "kernels_mixed.inc"

DFFT_SLICE_DECLS(f32)
#if DFFT_SLICE == 0
DFFT_SLICE_FUNCS(f32, 0, DFFT_F32_LIST_SMALL)
int launch_pass_f32(int N, int variant, const PassArgs &A, hipStream_t stream)
{
    if (!is_pow2(N)) return N < 512 ? launch_pass_f32_s5(N, variant, A, stream) : launch_pass_f32_s6(N, variant, A, stream);
    return N <= 512 ? launch_pass_f32_s0(N, variant, A, stream) : N == 1024 ? launch_pass_f32_s1(N, variant, A, stream)
                                                                          : launch_pass_f32_s2(N, variant, A, stream);
}
bool pass_info_f32(int N, int variant, PassInfo *pi)
{
    if (!is_pow2(N)) return N < 512 ? pass_info_f32_s5(N, variant, pi) : pass_info_f32_s6(N, variant, pi);
    return N <= 512 ? pass_info_f32_s0(N, variant, pi) : N == 1024 ? pass_info_f32_s1(N, variant, pi) : pass_info_f32_s2(N, variant, pi);
}
#elif DFFT_SLICE == 1
DFFT_SLICE_FUNCS(f32, 1, DFFT_F32_LIST_1024)
#ifdef DFFT_EXPERIMENTS
// the LDS-free shuffle pass (A/B only; natural lines in and out)
int launch_shfl_f32(int N, int dpp, const PassArgs &A, hipStream_t stream)
{
    const uint32_t lines = A.LB * A.na, grid = (lines + 15) / 16;
    if (N == 512 && !dpp) hipLaunchKernelGGL((fft_shfl_kernel<float, 512, 0>), dim3(grid), dim3(256), 0, stream, A);
    else if (N == 512) hipLaunchKernelGGL((fft_shfl_kernel<float, 512, 1>), dim3(grid), dim3(256), 0, stream, A);
    else if (N == 1024 && !dpp) hipLaunchKernelGGL((fft_shfl_kernel<float, 1024, 0>), dim3(grid), dim3(256), 0, stream, A);
    else if (N == 1024) hipLaunchKernelGGL((fft_shfl_kernel<float, 1024, 1>), dim3(grid), dim3(256), 0, stream, A);
    else return -1;
    return (int)hipGetLastError();
}
#endif
#elif DFFT_SLICE == 2
DFFT_SLICE_FUNCS(f32, 2, DFFT_F32_LIST_2048)
#elif DFFT_SLICE == 5
DFFT_SLICE_FUNCS(f32, 5, DFFT_F32_LIST_MIXED0)
#elif DFFT_SLICE == 6
DFFT_SLICE_FUNCS(f32, 6, DFFT_F32_LIST_MIXED1)
#elif DFFT_SLICE == 7
DFFT_REAL_MIXED_FUNCS(f32, 7, DFFT_F32_LIST_RMIXED0)
#elif DFFT_SLICE == 8
DFFT_REAL_MIXED_FUNCS(f32, 8, DFFT_F32_LIST_RMIXED1)
#else
// slices 3 (real z passes) and 4 (Bluestein) share the base list

// real-transform z passes (variant 0 configurations only); M = Nz/2
#define DFFT_F32_BASE(X) X(2, 0, F32_2) X(4, 0, F32_4) X(8, 0, F32_8) X(16, 0, F32_16) X(32, 0, F32_32) X(64, 0, F32_64) \
    X(128, 0, F32_128) X(256, 0, F32_256) X(512, 0, F32_512) X(1024, 0, F32_1024)
#if DFFT_SLICE == 3
// 512 and 1024 (Nz = 1024, 2048): two radix passes (one exchange) + one-plane split, measured +14 % / +5-10 %
// over the three-pass configurations; point-fastest lane mappings on the natural-line side (PassCfg::MAP = 1 for
// the R2C load, 2 for the C2R store): a line-fastest wave touches a real line in 32-byte pieces
using F32_R512_32 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1>;
using F32_R512_pf1 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 1>;
using F32_R512_pf2 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 2>;
using F32_R512_c2r = PassCfg<float, 512, 32, 16, 1, 16, 32, 1, 1, 1, 1, 0, 2>;     // 16 first: two first-pass butterflies per thread (pairs)
// 1024 (Nz = 2048): three passes so that the pass next to the split / merge has two butterflies per thread (pairs)
using F32_R1024_r2c = PassCfg<float, 1024, 32, 16, 1, 8, 8, 16, 1, 1, 1, 0, 1>;
using F32_R1024_c2r = PassCfg<float, 1024, 32, 16, 1, 16, 8, 8, 1, 1, 1, 0, 2>;
using F32_R1024_pf1 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 1>;
using F32_R1024_pf2 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 2>;
// is there a packed real z pass for M = Nz/2 complex points?
bool real_supported_f32(int M)
{
    if (!is_pow2(M)) return M < 320 ? real_mixed_info_f32_s7(M) : real_mixed_info_f32_s8(M);
    switch (M) {
#define X(n, v, cfg) case n: return true;
        DFFT_F32_BASE(X)
        X(2048, 0, F32_2048)
#undef X
    }
    return false;
}
int launch_real_f32(int M, int mode, int variant, const PassArgs &A, hipStream_t stream)
{
    if (!is_pow2(M)) {      // mixed-radix lengths (kernels_mixed.inc); no strided-real-line (Y_Then_ZX) form
        if (A.load_kind == LOAD_KMAJOR && mode == 1) return -1;
        return M < 320 ? launch_real_mixed_f32_s7(M, mode, A, stream) : launch_real_mixed_f32_s8(M, mode, A, stream);
    }
    if (M == 2048 && A.load_kind != LOAD_KMAJOR) return mode == 1 ? launch_real_cfg<F32_2048, 1, 1>(A, stream) : launch_real_cfg<F32_2048, 2>(A, stream);      // Nz = 4096
    if (mode == 1 && A.load_kind == LOAD_KMAJOR) {
        // strided real lines (Y_Then_ZX): the lanes run along the contiguous axis, i.e. the line-fastest mapping
        if (M == 512) return launch_real_cfg<F32_R512_32, 3, 1>(A, stream);
        if (M == 1024) return launch_real_cfg<F32_1024_v6, 3, 1>(A, stream);
        switch (M) {
#define X(n, v, cfg) case n: return launch_real_cfg<cfg, 3>(A, stream);
            DFFT_F32_BASE(X)
#undef X
        }
        return -1;
    }
    if (M == 512 && variant == 0) return mode == 1 ? launch_real_cfg<F32_R512_pf1, 1, 2>(A, stream) : launch_real_cfg<F32_R512_c2r, 2, 2>(A, stream);
    if (M == 1024 && variant == 0) return mode == 1 ? launch_real_cfg<F32_R1024_r2c, 1, 2>(A, stream) : launch_real_cfg<F32_R1024_c2r, 2, 2>(A, stream);
    if (M == 512) return mode == 1 ? launch_real_cfg<F32_R512_pf1, 1, 1>(A, stream) : launch_real_cfg<F32_R512_pf2, 2>(A, stream);
    if (M == 1024) return mode == 1 ? launch_real_cfg<F32_R1024_pf1, 1, 1>(A, stream) : launch_real_cfg<F32_R1024_pf2, 2>(A, stream);
    switch (M) {
#define X(n, v, cfg) case n: return mode == 1 ? launch_real_cfg<cfg, 1>(A, stream) : launch_real_cfg<cfg, 2>(A, stream);
        DFFT_F32_BASE(X)
#undef X
    }
    return -1;
}

// Bluestein passes for arbitrary line lengths: M = power of two >= 2*NL - 1
int launch_bluestein_f32(int M, const PassArgs &A, hipStream_t stream)
{
    switch (M) {
#define X(n, v, cfg) case n: return launch_bluestein_cfg<cfg>(A, stream);
        DFFT_F32_BASE(X)
        X(2048, 0, F32_2048)
        X(4096, 0, F32_4096)      // lines of 1025..2048 / 2049..4096 points: inner transforms on sub-tile workgroups
        X(8192, 0, F32_8192)
#undef X
    }
    return -1;
}
#endif  // DFFT_SLICE == 3
#endif  // DFFT_SLICE
}  // namespace dfft
