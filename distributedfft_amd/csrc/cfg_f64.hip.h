// cfg_f64.hip.h -- fp64 instantiations of the axis-pass kernel (TL = 8 lines per tile: 8 x 16 B = 128 B runs): the configurations shared by the translation units of this precision
// (kernels_f64.hip: axis passes, real_f64.hip: packed real z passes, bluestein_f64.hip)
#pragma once
#include "kernels.hip.inc"

namespace dfft {
//                 real    N     E  TL  G   radices      planes  chained twiddles
using F64_2    = PassCfg<double, 2,    2, 8, 32, 2, 1, 1, 1,   1>;
using F64_4    = PassCfg<double, 4,    4, 8, 32, 4, 1, 1, 1,   1>;
using F64_8    = PassCfg<double, 8,    8, 8, 32, 8, 1, 1, 1,   1>;
using F64_16   = PassCfg<double, 16,  16, 8, 32, 16, 1, 1, 1,  1>;
using F64_32   = PassCfg<double, 32,   8, 8, 8,  8, 4, 1, 1,   2>;
using F64_64   = PassCfg<double, 64,   8, 8, 4,  8, 8, 1, 1,   2>;
using F64_128  = PassCfg<double, 128, 16, 8, 4,  16, 8, 1, 1,  2>;
using F64_256  = PassCfg<double, 256, 16, 8, 2,  16, 16, 1, 1, 2>;
using F64_512  = PassCfg<double, 512, 16, 8, 1,  8, 8, 8, 1,   1>;
// 1024: 512 threads, <= 128 VGPRs, 68 KiB LDS -> two workgroups per CU (measured best)
using F64_1024 = PassCfg<double, 1024, 16, 8, 1, 16, 16, 4, 1, 1, 1>;
using F64_2048 = PassCfg<double, 2048, 16, 8, 1, 16, 16, 8, 1, 1, 1>;
// The variant number of a configuration is its ROLE in a plan (dfft_init picks by role, see PassRole):
//   1 = strided read: passes that load the point-major API layout (multi-rank inverse x pass): 16 lines per workgroup
//       (two tiles of 8: 128-byte runs from two neighbouring rows per point), 32 points per thread (twice the loads in
//       flight), nontemporal loads and stores (7.9 -> 7.6 ms at 1024^3; 8.9 ms with the default configuration).  2048 points:
//       the streaming configuration (18.8 -> 17.5 ms on 2048 x 1024 x 1024; 32 points per thread lose there, one or two
//       workgroups per CU alike: profiles/r3_strided_read_variants.txt)
//   3 = streaming: nontemporal loads and stores, for passes whose stores come in long runs (tiled 1 KiB chunks,
//       natural lines): +2-3 %; it costs up to 10 % on 128-byte-run stores, so those keep 0
//   7 = natural lines: a pass with natural lines on one side.  Only 2048 has its own: 8 lines x 2048 points are
//       256 KiB, one workgroup per CU whatever the configuration, so every tile is split between two sibling
//       workgroups of 4 lines (PassCfg::SUB, 68 KiB LDS, two per CU) -- 16.5 -> 12-13 ms on a 1024 x 1024 x 2048 grid.
//       With a tiled side the half-width (64-byte) runs of a sub-tile cost more than the occupancy gives, so the
//       tiled 2048-point passes keep whole tiles.
using F64_1024_v1 = PassCfg<double, 1024, 32, 8, 2, 32, 32, 1, 1, 1, 1, 3>;
using F64_1024_v3 = PassCfg<double, 1024, 16, 8, 1, 16, 16, 4, 1, 1, 1, 3>;
//   2 = strided read of rows whose pitch is not a multiple of the tile (R2C plans: 513-wide rows): as 1, but only the STORES
//       nontemporal -- the 256-byte runs of neighbouring workgroups share a cache line, which a streaming load hint evicts
//       before the neighbour gets to it
using F64_1024_v2 = PassCfg<double, 1024, 32, 8, 2, 32, 32, 1, 1, 1, 1, 2>;
using F64_512_v1 = PassCfg<double, 512, 32, 8, 2, 32, 16, 1, 1, 1, 1>;
using F64_512_v3 = PassCfg<double, 512, 16, 8, 1, 8, 8, 8, 1, 1, 0, 3>;
using F64_2048_v3 = PassCfg<double, 2048, 16, 8, 1, 16, 16, 8, 1, 1, 1, 3>;
using F64_2048_v7 = PassCfg<double, 2048, 16, 8, 1, 16, 16, 8, 1, 1, 1, 3, 0, 2>;
// 4096 and 8192 points: a whole tile of 8 lines does not fit a CU, so a workgroup transforms 2 lines (4096) or 1 line
// (8192) of a tile (PassCfg::SUB = 4 / 8; 64 KiB of LDS, two workgroups per CU) and its siblings -- consecutive logical
// workgroups on one XCD -- the rest.  Natural lines are unaffected; a tiled side is accessed in 32- / 16-byte pieces that
// L2 puts together: these lengths run for completeness (the reference takes any length, mpicufft_pencil_opt1.cpp:165-197),
// not at the speed of the shorter ones.
using F64_4096 = PassCfg<double, 4096, 16, 8, 1, 16, 16, 16, 1, 1, 1, 0, 0, 4>;
using F64_8192 = PassCfg<double, 8192, 32, 8, 1, 32, 16, 16, 1, 1, 1, 0, 0, 8>;
// A/B-only configurations of earlier measurements (sub-tile workgroups on tiled passes, nontemporal loads-only / stores-only,
// whole-tile 64-point forms, 32-point fp64 2048, ...) were removed after they were measured: results in profiles/r2_*.txt and
// DESIGN.md section 6, definitions in the git history (commit c38cf04).  New ones go here, under -DDFFT_EXPERIMENTS:

//   8 = strided read on a plan without a second exchange (P1 = 1: the mirrored inverse of one rank, 1 x P2 grids): variant 1 as
//       persistent workgroups with the stores of a tile fused with the loads of the next (PassCfg::PERSIST), compiled for that one
//       pair of address forms (FixForms: with every form in the loop the registers spill): 8.27 -> 7.64 ms at 1024^3 on plain
//       buffers (profiles/r4_persist3.txt); 249 VGPRs, no scratch.  Any other launch runs variant 1.  Measured and NOT adopted:
//       the same form for the table store of P1 > 1 plans (rank 0 of 2 x 4: 1.02-1.09 -> 1.03-1.10 ms, of 8 x 1: 1.07 -> 1.10-1.17)
//       and, in round 3, prefetching the next tile into a second register set (13.3-16.2 ms: it spills).
using F64_1024_v8 = FixForms<PassCfg<double, 1024, 32, 8, 2, 32, 32, 1, 1, 1, 1, 3, 0, 1, 1>, F64_1024_v1, 1>;
// A/B only (measured, profiles/r3_strided_read_variants.txt): 32 points per thread on ONE tile of 8 lines (256 threads, 68 KiB of
// LDS -> two workgroups per CU) 8.49 ms against 7.68; 16 lines on 1024 threads (16 points per thread) 8.9
using F64_1024_v12 = PassCfg<double, 1024, 32, 8, 1, 32, 32, 1, 1, 1, 1, 3>;
using F64_1024_v14 = PassCfg<double, 1024, 16, 8, 2, 16, 16, 4, 1, 1, 1, 3>;
using F64_2048_v12 = PassCfg<double, 2048, 32, 8, 1, 32, 32, 2, 1, 1, 1, 3, 0, 2>;
#ifdef DFFT_EXPERIMENTS
#define DFFT_F64_EXP_SMALL(X)
#define DFFT_F64_EXP_1024(X) X(1024, 12, F64_1024_v12) X(1024, 14, F64_1024_v14)
#define DFFT_F64_EXP_2048(X) X(2048, 12, F64_2048_v12)
#else
#define DFFT_F64_EXP_SMALL(X)
#define DFFT_F64_EXP_1024(X)
#define DFFT_F64_EXP_2048(X)
#endif
#define DFFT_F64_LIST_SMALL(X) X(512, 1, F64_512_v1) X(512, 3, F64_512_v3) X(2, 0, F64_2) X(4, 0, F64_4) X(8, 0, F64_8) X(16, 0, F64_16) X(32, 0, F64_32) X(64, 0, F64_64) X(128, 0, F64_128) X(256, 0, F64_256) X(512, 0, F64_512) DFFT_F64_EXP_SMALL(X)
#define DFFT_F64_LIST_1024(X) X(1024, 1, F64_1024_v1) X(1024, 2, F64_1024_v2) X(1024, 3, F64_1024_v3) X(1024, 8, F64_1024_v8) X(1024, 0, F64_1024) DFFT_F64_EXP_1024(X)
#define DFFT_F64_LIST_2048(X) X(2048, 1, F64_2048_v3) X(2048, 3, F64_2048_v3) X(2048, 7, F64_2048_v7) X(2048, 0, F64_2048) X(4096, 0, F64_4096) X(8192, 0, F64_8192) DFFT_F64_EXP_2048(X)

// lengths with a packed real z pass / a Bluestein inner transform of their own configuration

// real-transform z passes; M = Nz/2.  512 gets its own 8-points-per-thread configuration: the
// split/merge step needs both LDS planes, and 512 threads x 64 KiB keeps 16 waves on a CU
using F64_R512 = PassCfg<double, 512, 8, 8, 1, 8, 8, 8, 1, 1>;
#define DFFT_F64_BASE(X) X(2, 0, F64_2) X(4, 0, F64_4) X(8, 0, F64_8) X(16, 0, F64_16) X(32, 0, F64_32) X(64, 0, F64_64) \
    X(128, 0, F64_128) X(256, 0, F64_256) X(512, 0, F64_R512) X(1024, 0, F64_1024)
}  // namespace dfft
