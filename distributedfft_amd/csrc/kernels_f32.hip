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
// two radix-32 passes (one LDS exchange instead of two): variants 6 (line fastest), 7 / 8 (point-fastest forms)
using F32_1024_v6 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1>;
using F32_1024_v7 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 1>;
using F32_1024_v8 = PassCfg<float, 1024, 32, 16, 1, 32, 32, 1, 1, 1, 1, 0, 2>;
using F32_512_v6 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1>;
using F32_512_v4 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 1>;    // point-fastest forms, as for 1024
using F32_512_v5 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1, 0, 2>;
// (2048 with the point-fastest mapping measured slower than line fastest: 0.89 vs 0.81 ms on 256x256x2048)

#define DFFT_F32_LIST(X) X(1024, 1, F32_1024_v1) X(512, 1, F32_512_v1) X(1024, 4, F32_1024_v4) X(1024, 5, F32_1024_v5) \
    X(1024, 6, F32_1024_v6) X(1024, 7, F32_1024_v7) X(1024, 8, F32_1024_v8) X(512, 6, F32_512_v6) X(512, 4, F32_512_v4) X(512, 5, F32_512_v5) \
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
// experiment variants of the real z passes (DFFT_REAL_VARIANT): 1 = one-plane split, same configuration;
// 2 = one-plane split, 32 points per thread
using F32_R512_32 = PassCfg<float, 512, 32, 16, 1, 32, 16, 1, 1, 1, 1>;
static int launch_real_variant_f32(int M, int mode, int variant, const PassArgs &A, hipStream_t stream)
{
    if (M == 512 && variant == 1) return mode == 1 ? launch_real_cfg<F32_512, 1, 1>(A, stream) : launch_real_cfg<F32_512, 2>(A, stream);
    if (M == 512 && variant == 2) return mode == 1 ? launch_real_cfg<F32_R512_32, 1, 1>(A, stream) : launch_real_cfg<F32_R512_32, 2>(A, stream);
    if (M == 1024 && variant == 1) return mode == 1 ? launch_real_cfg<F32_1024, 1, 1>(A, stream) : launch_real_cfg<F32_1024, 2>(A, stream);
    if (M == 1024 && variant == 2) return mode == 1 ? launch_real_cfg<F32_1024_v6, 1, 1>(A, stream) : launch_real_cfg<F32_1024_v6, 2>(A, stream);
    return -2;
}
int launch_real_f32(int M, int mode, const PassArgs &A, hipStream_t stream)
{
    // 512 and 1024: two radix passes (one exchange) + one-plane split, measured +14 % / +5-10 % over the
    // three-pass configurations (DFFT_REAL_VARIANT=0 selects those for A/B runs)
    if (!getenv("DFFT_REAL_VARIANT")) {
        if (M == 512) return mode == 1 ? launch_real_cfg<F32_R512_32, 1, 1>(A, stream) : launch_real_cfg<F32_R512_32, 2>(A, stream);
        if (M == 1024) return mode == 1 ? launch_real_cfg<F32_1024_v6, 1, 1>(A, stream) : launch_real_cfg<F32_1024_v6, 2>(A, stream);
    }
    if (const char *v = getenv("DFFT_REAL_VARIANT")) {
        const int r = launch_real_variant_f32(M, mode, atoi(v), A, stream);
        if (r != -2) return r;
    }
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
