// cfg_f32.hip.h -- fp32 instantiations of the axis-pass kernel (TL = 16 lines per tile: 16 x 8 B = 128 B runs): the configurations shared by the translation units of this precision
// (kernels_f32.hip: axis passes, real_f32.hip: packed real z passes, bluestein_f32.hip)
#pragma once
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
// (round 4: the natural-line LOAD runs 32 points per thread -- 512 threads, <= 128 VGPRs, two workgroups = 32 waves per CU, three radix
// passes -- instead of 64: rank 0 of 2 x 4 at 2048^3 3.61 -> 3.09 ms, its copy 3.00; 2048^3 on one GPU 31.7 -> 27.9 ms per z pass;
// the natural-line STORE keeps 64 points: 3.11 vs 3.15 ms; profiles/r4_f32_2048_natural_candidates.txt, r4_f32_2048_single_gpu_candidates.txt)
using F32_2048_v4 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 3, 1, 2>;
using F32_2048_v5 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 0, 2, 2>;
using F32_2048_v6 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1>;
// 4096 and 8192 points: sub-tile workgroups of 4 / 2 lines (see kernels_f64.hip), 512 threads, 64 KiB of LDS
using F32_4096 = PassCfg<float, 4096, 32, 16, 1, 32, 16, 8, 1, 1, 1, 0, 0, 4>;
using F32_8192 = PassCfg<float, 8192, 32, 16, 1, 32, 16, 16, 1, 1, 1, 0, 0, 8>;
// A/B-only configurations of earlier measurements (sub-tile workgroups on tiled passes, nontemporal loads-only / stores-only,
// whole-tile 64-point forms, 32-point fp64 2048, ...) were removed after they were measured: results in profiles/r2_*.txt and
// DESIGN.md section 6, definitions in the git history (commit c38cf04).  New ones go here, under -DDFFT_EXPERIMENTS:

//   9 = tiled passes, streaming: variant 6 with nontemporal loads and stores.  Whether the hints pay depends on the pass, the
//       layout and the buffers (2048 points, rank 0 of the 2 x 4 plan at 2048^3: y 4.86 -> 3.73 ms, x^-1 5.75 -> 4.68, but x
//       3.92 -> 4.27 and y^-1 5.6 -> 9.2; one GPU: the middle pass 9.6 -> 8.5 ms, the last one 6.78 -> 6.85;
//       profiles/r3_f32_2048_tiled_variants.txt), so no plan picks it by rule: dfft_tune_variants tries it per pass and keeps it
//       where that plan's own pass runs faster.  (Radix 32 first instead of 64 first made no difference at 2048 points.)
using F32_512_v9 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 3>;
using F32_1024_v9 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 3>;
using F32_2048_v9 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 3>;
// A/B only, measured and rejected (profiles/r3_f32_inverse_y_point_fastest_store.txt): tiled load + point-fastest store mapping for the
// inverse y pass (transposed-tile stores in whole lines instead of 32-byte pieces): 4.42 vs 4.30 ms at 1024 points, 10.55 vs 10.35 at 2048
using F32_1024_v10 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 2>;
using F32_2048_v10 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 0, 2>;
// Round 4: the tiled 2048-point passes on TWO workgroups per CU.  16 lines x 2048 points are 256 KiB, half of a CU's register file, so
// whole tiles mean one workgroup per CU and load -> compute -> store in sequence.  Measured on rank 0 of the 2 x 4 plan at 2048^3
// (profiles/r4_f32_2048_tiled_candidates.txt), every configuration as a transform and as a copy with the same access pattern:
//   14 / 15 = sub-tile workgroups of 8 lines (PassCfg::SUB = 2) with 32 points per thread: 512 threads, <= 128 VGPRs, 66 KiB of LDS ->
//             two workgroups = 32 waves per CU, three radix passes, without / with nontemporal hints.  SHIPPED as tuner candidates
//             (dfft_tune_variants): the inverse y pass 4.80 (variant 5) -> 4.49 ms with 15, 3 % above its own copy (4.37)
//   12 / 13 = the same sub-tiles with 64 points per thread (two waves per SIMD): 4.84-5.08 ms there, no better than 5; A/B builds only
// What the copies say: the floor of these passes is their access pattern, not the phases of a tile -- y 3.43 ms, x 3.27, x^-1 4.25,
// y^-1 4.16 as pure copies (4.0-5.2 TB/s) against 3.87 / 3.61 / 4.59 / 4.49 as transforms.
using F32_2048_v12 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 0, 0, 2>;
using F32_2048_v13 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 3, 0, 2>;
using F32_2048_v14 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 0, 0, 2>;
using F32_2048_v15 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 3, 0, 2>;
//   7 = transposed-tile store (the inverse y pass of a multi-rank plan): tiled load with the line-fastest mapping, then the point-fastest
//       one, so that a wave stores whole 128-byte lines of the consumer's tiles instead of 32-byte pieces that four waves complete
//       (a line-fastest fp32 wave is 16 lines x 4 points); 32 points per thread on sub-tile workgroups of 8 lines (two workgroups = 32
//       waves per CU), nontemporal hints.  Rank 0 of 2 x 4 at 2048^3: 4.80 ms (variant 5, the tuner's former pick; 5.58 with 6) ->
//       3.63 ms, as a copy 3.50; slab 8: 4.73 -> 3.87.  The same mapping on whole tiles at 64 points per thread (round 3, variant 10)
//       had shown nothing: 5.38.  profiles/r4_f32_2048_inverse_y_whole_lines.txt
using F32_2048_v7 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 3, 2, 2>;
// A/B only: the same without hints (2: 4.19 ms), on whole tiles with 1024 threads (11 / 3: 4.59 / 4.03), and the 1024-point siblings under test
using F32_2048_v2 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 0, 2, 2>;
using F32_2048_v11 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 0, 2>;
using F32_2048_v3 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 3, 2>;
// (the default -- whole tiles, 32 points per thread, 1024 threads -- with hints: y 4.58, x 3.98, x^-1 4.58 ms against 3.93 / 4.03 / 4.55
// for 9: nothing, profiles/r4_f32_2048_whole_tiles_32_hints.txt)
using F32_2048_v8 = PassCfg<float, 2048, 32, 16, 1, 32, 8, 8, 1, 1, 1, 3>;
// 1024 points, the same role (profiles/r4_f32_1024_inverse_y_whole_lines.txt): 7 = 16 points per thread on sub-tiles of 8 lines (512
// threads): rank 0 of 2 x 4 at 1024^3 0.614 (variant 6) -> 0.502 ms, one rank with the mirrored inverse 3.95 -> 3.66; 3 = 32 points per
// thread on whole tiles (512 threads): 0.518 / 3.11 -- a tuner candidate (it wins on the big single-rank grid).  SHIPPED: 7 by rule, 3
// for dfft_tune_variants.  A/B only: 11 (32 points, sub-tiles: 0.524 / 3.62), 2 (16 points, whole tiles, 1024 threads: 0.622 / 4.01)
using F32_1024_v7 = PassCfg<float, 1024, 16, 16, 1, 16, 8, 8, 1, 1, 1, 3, 2, 2>;
using F32_1024_v3 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1, 3, 2>;
using F32_1024_v11 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1, 3, 2, 2>;
using F32_1024_v2 = PassCfg<float, 1024, 16, 16, 1, 16, 8, 8, 1, 1, 1, 3, 2>;
//   1 = natural-line load, tiled store (the inverse x pass of a plan with an x-contiguous spectrum, option spectral_layout): variant 6's
//       two-pass chain with the point-fastest mapping for the FIRST pass only (PassCfg::MAP = 3) -- the loads are 128-byte runs of one
//       line instead of 32-byte pieces of 16 lines, the same-tile stores stay line fastest.  profiles/r5_spectral_layout.txt
using F32_512_v1 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 3>;
using F32_1024_v1 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 3>;
using F32_2048_v1 = PassCfg<float, 2048, 64, 16, 1, 64, 32, 1, 1, 1, 1, 0, 3>;
#ifdef DFFT_EXPERIMENTS
#define DFFT_F32_EXP_SMALL(X)
#define DFFT_F32_EXP_1024(X) X(1024, 10, F32_1024_v10) X(1024, 11, F32_1024_v11) X(1024, 2, F32_1024_v2)
#define DFFT_F32_EXP_2048(X) X(2048, 10, F32_2048_v10) X(2048, 12, F32_2048_v12) X(2048, 13, F32_2048_v13) X(2048, 11, F32_2048_v11) X(2048, 3, F32_2048_v3) X(2048, 2, F32_2048_v2) X(2048, 8, F32_2048_v8)
#else
#define DFFT_F32_EXP_SMALL(X)
#define DFFT_F32_EXP_1024(X)
#define DFFT_F32_EXP_2048(X)
#endif
#define DFFT_F32_LIST_SMALL(X) X(512, 1, F32_512_v1) X(512, 6, F32_512_v6) X(512, 9, F32_512_v9) X(512, 4, F32_512_v4) X(512, 5, F32_512_v5) X(2, 0, F32_2) X(4, 0, F32_4) X(8, 0, F32_8) X(16, 0, F32_16) X(32, 0, F32_32) X(64, 0, F32_64) X(128, 0, F32_128) X(256, 0, F32_256) X(512, 0, F32_512) DFFT_F32_EXP_SMALL(X)
#define DFFT_F32_LIST_1024(X) X(1024, 1, F32_1024_v1) X(1024, 4, F32_1024_v4) X(1024, 5, F32_1024_v5) X(1024, 6, F32_1024_v6) X(1024, 9, F32_1024_v9) X(1024, 7, F32_1024_v7) X(1024, 3, F32_1024_v3) X(1024, 0, F32_1024) DFFT_F32_EXP_1024(X)
#define DFFT_F32_LIST_2048(X) X(2048, 1, F32_2048_v1) X(2048, 4, F32_2048_v4) X(2048, 5, F32_2048_v5) X(2048, 6, F32_2048_v6) X(2048, 9, F32_2048_v9) X(2048, 7, F32_2048_v7) X(2048, 14, F32_2048_v14) X(2048, 15, F32_2048_v15) X(2048, 0, F32_2048) X(4096, 0, F32_4096) X(8192, 0, F32_8192) DFFT_F32_EXP_2048(X)

// lengths with a packed real z pass / a Bluestein inner transform of their own configuration
// real-transform z passes (variant 0 configurations only); M = Nz/2
#define DFFT_F32_BASE(X) X(2, 0, F32_2) X(4, 0, F32_4) X(8, 0, F32_8) X(16, 0, F32_16) X(32, 0, F32_32) X(64, 0, F32_64) \
    X(128, 0, F32_128) X(256, 0, F32_256) X(512, 0, F32_512) X(1024, 0, F32_1024)
}  // namespace dfft
