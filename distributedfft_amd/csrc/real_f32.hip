// real_f32.hip -- packed real z passes (R2C / C2R) of the power-of-two lengths, f32; M = Nz/2.  Two parts (-DDFFT_PART = 0: the R2C
// kernels and the entry points, 1: the C2R kernels).
#include "cfg_f32.hip.h"

namespace dfft {
int launch_rmixed_f32(int M, int mode, const PassArgs &A, hipStream_t stream);      // rmixed_f32.hip
bool rmixed_info_f32(int M);
// 512 and 1024 (Nz = 1024, 2048): two radix passes (one exchange) + one-plane split, measured +14 % / +5-10 %
// over the three-pass configurations; point-fastest lane mappings on the natural-line side (PassCfg::MAP = 1 for
// the R2C load, 2 for the C2R store): a line-fastest wave touches a real line in 32-byte pieces
using F32_R512_32 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1>;
using F32_R512_pf1 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 1>;
using F32_R512_pf2 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 2>;
using F32_R512_c2r = PassCfg<float, 512, 32, 16, 1, 16, 32, 1, 1, 1, 1, 0, 2>;     // 16 first: two first-pass butterflies per thread (pairs)
// 1024 (Nz = 2048): three passes so that the pass next to the split / merge has an even number of butterflies per thread (pairs).
// R2C: 32.4.8 -- measured against 8.8.16 (which spilled 40 bytes per lane), 16.16.4 and 32.8.4 on 1024 x 1024 x 2048:
// 3.70 / 3.92 / 3.78 / 3.72 ms (profiles/r3_real_pass_variants.txt); 115 VGPRs, no scratch
using F32_R1024_r2c = PassCfg<float, 1024, 32, 16, 1, 32, 4, 8, 1, 1, 1, 0, 1>;
using F32_R1024_c2r = PassCfg<float, 1024, 32, 16, 1, 16, 8, 8, 1, 1, 1, 0, 2>;
// 2048 (Nz = 4096): in-register split / merge as well (radix 8 next to it: two butterflies per thread), point-fastest on the
// natural-line side; 124 / 128 VGPRs, no scratch (the one-plane split and the unpaired merge spilled 292 / 316 bytes per lane)
using F32_R2048_r2c = PassCfg<float, 2048, 32, 16, 1, 16, 16, 8, 1, 1, 1, 0, 1>;
using F32_R2048_c2r = PassCfg<float, 2048, 32, 16, 1, 16, 16, 8, 1, 1, 1, 0, 2>;     // 8.16.16 spills 52 B/lane here
using F32_R512_pf1_nt = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 3, 1>;
using F32_R512_c2r_nt = PassCfg<float, 512, 32, 16, 1, 16, 32, 1, 1, 1, 1, 3, 2>;
using F32_R1024_r2c_nt = PassCfg<float, 1024, 32, 16, 1, 32, 4, 8, 1, 1, 1, 3, 1>;
using F32_R1024_c2r_nt = PassCfg<float, 1024, 32, 16, 1, 16, 8, 8, 1, 1, 1, 3, 2>;
using F32_R1024_pf1 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 1>;
using F32_R1024_pf2 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 2>;
int launch_real_c2r_f32(int M, int variant, const PassArgs &A, hipStream_t stream);      // part 1
#if DFFT_PART == 0
// is there a packed real z pass for M = Nz/2 complex points?
bool real_supported_f32(int M)
{
    if (!is_pow2(M)) return rmixed_info_f32(M);
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
        return launch_rmixed_f32(M, mode, A, stream);
    }
    if (mode != 1) return launch_real_c2r_f32(M, variant, A, stream);
    if (M == 2048 && A.load_kind != LOAD_KMAJOR) return launch_real_cfg<F32_R2048_r2c, 1, 2>(A, stream);      // Nz = 4096
    if (A.load_kind == LOAD_KMAJOR) {
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
    // 512 and 1024: split / merge in registers, nontemporal loads and stores (real z passes stream: 1024^3 R2C 1.57 -> 1.52 ms,
    // C2R 1.73 -> 1.62 ms, profiles/r3_real_pass_nontemporal.txt); real_variant 5 = the same without the hints (A/B), any other
    // value = the round-2 forms that split through one LDS plane
    if (M == 512 && variant == 0) return launch_real_cfg<F32_R512_pf1_nt, 1, 2>(A, stream);
    if (M == 1024 && variant == 0) return launch_real_cfg<F32_R1024_r2c_nt, 1, 2>(A, stream);
    if (M == 512 && variant == 5) return launch_real_cfg<F32_R512_pf1, 1, 2>(A, stream);
    if (M == 1024 && variant == 5) return launch_real_cfg<F32_R1024_r2c, 1, 2>(A, stream);
    if (M == 512) return launch_real_cfg<F32_R512_pf1, 1, 1>(A, stream);
    if (M == 1024) return launch_real_cfg<F32_R1024_pf1, 1, 1>(A, stream);
    switch (M) {
#define X(n, v, cfg) case n: return launch_real_cfg<cfg, 1>(A, stream);
        DFFT_F32_BASE(X)
#undef X
    }
    return -1;
}
#elif DFFT_PART == 1
int launch_real_c2r_f32(int M, int variant, const PassArgs &A, hipStream_t stream)
{
    if (M == 2048 && A.load_kind != LOAD_KMAJOR) return launch_real_cfg<F32_R2048_c2r, 2, 2>(A, stream);      // Nz = 4096
    if (M == 512 && variant == 0) return launch_real_cfg<F32_R512_c2r_nt, 2, 2>(A, stream);
    if (M == 1024 && variant == 0) return launch_real_cfg<F32_R1024_c2r_nt, 2, 2>(A, stream);
    if (M == 512 && variant == 5) return launch_real_cfg<F32_R512_c2r, 2, 2>(A, stream);
    if (M == 1024 && variant == 5) return launch_real_cfg<F32_R1024_c2r, 2, 2>(A, stream);
    if (M == 512) return launch_real_cfg<F32_R512_pf2, 2>(A, stream);
    if (M == 1024) return launch_real_cfg<F32_R1024_pf2, 2>(A, stream);
    switch (M) {
#define X(n, v, cfg) case n: return launch_real_cfg<cfg, 2>(A, stream);
        DFFT_F32_BASE(X)
#undef X
    }
    return -1;
}
#else
#error "DFFT_PART must be 0 or 1"
#endif

}  // namespace dfft
