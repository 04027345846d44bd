// real_f64.hip -- packed real z passes (R2C / C2R) of the power-of-two lengths, f64; M = Nz/2
#include "cfg_f64.hip.h"

namespace dfft {
int launch_rmixed_f64(int M, int mode, const PassArgs &A, hipStream_t stream);      // rmixed_f64.hip
bool rmixed_info_f64(int M);
// 512 and 1024 (Nz = 1024, 2048): the Hermitian split / merge runs in registers -- the pass next to it assigns its
// butterflies in conjugate pairs (pair_j), which needs an even number of butterflies per thread in that pass: 16 points
// per thread with radix 8 (512) or radix 4 (1024: last for R2C, first for C2R)
using F64_R1024_c2r = PassCfg<double, 1024, 16, 8, 1, 4, 16, 16, 1, 1, 1>;
// 2048 (Nz = 4096): the same in-register split / merge -- radix 8 last (R2C: 16.16.8) or first (C2R: 8.16.16) leaves two
// butterflies per thread in that pass.  The forms that went through LDS (one-plane split, unpaired merge) spilled 36 / 324
// bytes per lane at the 128-VGPR budget of a 1024-thread workgroup; these need 98 / 124 VGPRs and no scratch.
using F64_R2048_c2r = PassCfg<double, 2048, 16, 8, 1, 8, 16, 16, 1, 1, 1>;
// is there a packed real z pass for M = Nz/2 complex points?
bool real_supported_f64(int M)
{
    if (!is_pow2(M)) return rmixed_info_f64(M);
    switch (M) {
#define X(n, v, cfg) case n: return true;
        DFFT_F64_BASE(X)
        X(2048, 0, F64_2048)
#undef X
    }
    return false;
}
int launch_real_f64(int M, int mode, int variant, const PassArgs &A, hipStream_t stream)
{
    if (!is_pow2(M)) {      // mixed-radix lengths (kernels_mixed.inc); no strided-real-line (Y_Then_ZX) form
        if (A.load_kind == LOAD_KMAJOR && mode == 1) return -1;
        return launch_rmixed_f64(M, mode, A, stream);
    }
    // Nz = 4096: 8 lines x 2048 points fill the LDS with one plane; split / merge in registers
    if (M == 2048 && A.load_kind != LOAD_KMAJOR) return mode == 1 ? launch_real_cfg<F64_2048, 1, 2>(A, stream) : launch_real_cfg<F64_R2048_c2r, 2, 2>(A, stream);
    // (nontemporal loads / stores were measured on these passes and lose at fp64: the R2C pass goes 3.30 -> 4.16 ms at 1024^3 -- its
    // mirrored butterflies store tiles in 112 + 16 byte pieces that need L2 to merge them --, C2R unchanged; profiles/r3_real_pass_nontemporal.txt)
    if (variant == 0 && A.load_kind != LOAD_KMAJOR) {
        if (M == 512) return mode == 1 ? launch_real_cfg<F64_512, 1, 2>(A, stream) : launch_real_cfg<F64_512, 2, 2>(A, stream);
        if (M == 1024) return mode == 1 ? launch_real_cfg<F64_1024, 1, 2>(A, stream) : launch_real_cfg<F64_R1024_c2r, 2, 2>(A, stream);
    }
    if (mode == 1 && A.load_kind == LOAD_KMAJOR) {      // strided real lines (Y_Then_ZX)
        switch (M) {
#define X(n, v, cfg) case n: return launch_real_cfg<cfg, 3>(A, stream);
            DFFT_F64_BASE(X)
#undef X
        }
        return -1;
    }
    switch (M) {
#define X(n, v, cfg) case n: return mode == 1 ? launch_real_cfg<cfg, 1>(A, stream) : launch_real_cfg<cfg, 2>(A, stream);
        DFFT_F64_BASE(X)
#undef X
    }
    return -1;
}

}  // namespace dfft
