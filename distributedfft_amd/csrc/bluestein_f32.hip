// bluestein_f32.hip -- Bluestein passes for arbitrary line lengths, f32: M = power of two >= 2*NL - 1.  Two parts
// (-DDFFT_PART = 0: inner transforms up to 1024 points and the entry point, 1: 2048 .. 8192 points).
#include "cfg_f32.hip.h"

namespace dfft {
int launch_bluestein_f32_p1(int M, const PassArgs &A, hipStream_t stream);
#if DFFT_PART == 0
int launch_bluestein_f32(int M, const PassArgs &A, hipStream_t stream)
{
    switch (M) {
#define X(n, v, cfg) case n: return launch_bluestein_cfg<cfg>(A, stream);
        DFFT_F32_BASE(X)
#undef X
    }
    return launch_bluestein_f32_p1(M, A, stream);
}
#elif DFFT_PART == 1
int launch_bluestein_f32_p1(int M, const PassArgs &A, hipStream_t stream)
{
    switch (M) {
#define X(n, v, cfg) case n: return launch_bluestein_cfg<cfg>(A, stream);
        X(2048, 0, F32_2048)
        X(4096, 0, F32_4096)      // lines of 1025..2048 / 2049..4096 points: inner transforms on sub-tile workgroups
        X(8192, 0, F32_8192)
#undef X
    }
    return -1;
}
#else
#error "DFFT_PART must be 0 or 1"
#endif
}  // namespace dfft
