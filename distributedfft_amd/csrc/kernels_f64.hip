// fp64 instantiations of the axis-pass kernel (TL = 8 lines per tile: 8 x 16 B = 128 B runs)
#include "kernels.hip.inc"

namespace dfft {
//                 real    N     E  TL  G   radices      planes
using F64_2    = PassCfg<double, 2,    2, 8, 32, 2, 1, 1, 1,   1>;
using F64_4    = PassCfg<double, 4,    4, 8, 32, 4, 1, 1, 1,   1>;
using F64_8    = PassCfg<double, 8,    8, 8, 32, 8, 1, 1, 1,   1>;
using F64_16   = PassCfg<double, 16,  16, 8, 32, 16, 1, 1, 1,  1>;
using F64_32   = PassCfg<double, 32,   8, 8, 8,  8, 4, 1, 1,   2>;
using F64_64   = PassCfg<double, 64,   8, 8, 4,  8, 8, 1, 1,   2>;
using F64_128  = PassCfg<double, 128, 16, 8, 4,  16, 8, 1, 1,  2>;
using F64_256  = PassCfg<double, 256, 16, 8, 2,  16, 16, 1, 1, 2>;
using F64_512  = PassCfg<double, 512, 16, 8, 1,  8, 8, 8, 1,   1>;
using F64_1024 = PassCfg<double, 1024, 16, 8, 1, 16, 16, 4, 1, 1>;
using F64_2048 = PassCfg<double, 2048, 16, 8, 1, 16, 16, 8, 1, 1>;

#define DFFT_F64_LIST(X) X(2, F64_2) X(4, F64_4) X(8, F64_8) X(16, F64_16) X(32, F64_32) X(64, F64_64) \
    X(128, F64_128) X(256, F64_256) X(512, F64_512) X(1024, F64_1024) X(2048, F64_2048)

int launch_pass_f64(int N, const PassArgs &A, hipStream_t stream)
{
    switch (N) {
#define X(n, cfg) case n: return launch_cfg<cfg>(A, stream);
        DFFT_F64_LIST(X)
#undef X
    }
    return -1;
}
bool pass_info_f64(int N, PassInfo *pi)
{
    switch (N) {
#define X(n, cfg) case n: info_cfg<cfg>(pi); return true;
        DFFT_F64_LIST(X)
#undef X
    }
    return false;
}
}  // namespace dfft
