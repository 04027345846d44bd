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
using F32_1024_v1 = PassCfg<float, 1024, 16, 16, 1, 16, 16, 4, 1, 1>;   // round-1 baseline, for A/B runs
// point-fastest lane mappings (see PassCfg::MAP): variant 4 = every pass (natural-line load + transposed-tile
// store: forward z), variant 5 = from the first exchange on (tiled load + natural-line / transposed store)
using F32_1024_v4 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1, 0, 1>;
using F32_1024_v5 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1, 0, 2>;
using F32_512_v1 = PassCfg<float, 512, 32, 16, 1, 32, 4, 4, 1, 1, 1>;

#define DFFT_F32_LIST(X) X(1024, 1, F32_1024_v1) X(512, 1, F32_512_v1) X(1024, 4, F32_1024_v4) X(1024, 5, F32_1024_v5) \
    X(2, 0, F32_2) X(4, 0, F32_4) X(8, 0, F32_8) X(16, 0, F32_16) X(32, 0, F32_32) X(64, 0, F32_64) \
    X(128, 0, F32_128) X(256, 0, F32_256) X(512, 0, F32_512) X(1024, 0, F32_1024) X(2048, 0, F32_2048)

int launch_pass_f32(int N, int variant, const PassArgs &A, hipStream_t stream)
{
    switch (N * 16 + variant) {
#define X(n, v, cfg) case n * 16 + v: return launch_cfg<cfg>(A, stream);
        DFFT_F32_LIST(X)
#undef X
    }
    return -1;
}
bool pass_info_f32(int N, int variant, PassInfo *pi)
{
    switch (N * 16 + variant) {
#define X(n, v, cfg) case n * 16 + v: info_cfg<cfg>(pi); return true;
        DFFT_F32_LIST(X)
#undef X
    }
    return false;
}

// real-transform z passes (variant 0 configurations only); M = Nz/2
#define DFFT_F32_BASE(X) X(2, 0, F32_2) X(4, 0, F32_4) X(8, 0, F32_8) X(16, 0, F32_16) X(32, 0, F32_32) X(64, 0, F32_64) \
    X(128, 0, F32_128) X(256, 0, F32_256) X(512, 0, F32_512) X(1024, 0, F32_1024)
int launch_real_f32(int M, int mode, const PassArgs &A, hipStream_t stream)
{
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
#undef X
    }
    return -1;
}
}  // namespace dfft
