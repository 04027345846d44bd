// kernels_f64.hip -- the axis-pass kernels of the power-of-two lengths, f64.  Compiled in three parts (-DDFFT_PART = 0: up to 512
// points and the entry points, 1: 1024, 2: 2048 and longer) so that a parallel make spreads the instantiations.
#include "cfg_f64.hip.h"

namespace dfft {
int launch_mixed_f64(int N, int variant, const PassArgs &A, hipStream_t stream);      // mixed_f64.hip
bool mixed_info_f64(int N, int variant, PassInfo *pi);
#define DFFT_DECL_PART(k) int launch_pass_f64_p##k(int, int, const PassArgs &, hipStream_t); bool pass_info_f64_p##k(int, int, PassInfo *);
DFFT_DECL_PART(0) DFFT_DECL_PART(1) DFFT_DECL_PART(2)
#undef DFFT_DECL_PART
#if DFFT_PART == 0
DFFT_PASS_FUNCS(launch_pass_f64_p0, pass_info_f64_p0, DFFT_F64_LIST_SMALL)
int launch_pass_f64(int N, int variant, const PassArgs &A, hipStream_t stream)
{
    if (!is_pow2(N)) return launch_mixed_f64(N, variant, A, stream);
    return N <= 512 ? launch_pass_f64_p0(N, variant, A, stream) : N == 1024 ? launch_pass_f64_p1(N, variant, A, stream)
                                                                          : launch_pass_f64_p2(N, variant, A, stream);
}
bool pass_info_f64(int N, int variant, PassInfo *pi)
{
    if (!is_pow2(N)) return mixed_info_f64(N, variant, pi);
    return N <= 512 ? pass_info_f64_p0(N, variant, pi) : N == 1024 ? pass_info_f64_p1(N, variant, pi) : pass_info_f64_p2(N, variant, pi);
}
#elif DFFT_PART == 1
DFFT_PASS_FUNCS(launch_pass_f64_p1, pass_info_f64_p1, DFFT_F64_LIST_1024)
#elif DFFT_PART == 2
DFFT_PASS_FUNCS(launch_pass_f64_p2, pass_info_f64_p2, DFFT_F64_LIST_2048)
#else
#error "DFFT_PART must be 0, 1 or 2"
#endif
}  // namespace dfft
