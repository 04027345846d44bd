// mixed_f32.hip -- axis-pass kernels of the lengths that are not powers of two (mixed radix 2, 3, 5, 7), f32: the generated
// configurations of kernels_mixed.inc, compiled in DFFT_F32_MIXED_PARTS parts (-DDFFT_PART = k; part 0 holds the entry points).
#include "kernels.hip.inc"

namespace dfft {
#define DFFT_MIXED_F32
#include "kernels_mixed.inc"

#define DFFT_DECL_PART(k) int launch_mixed_f32_p##k(int, int, const PassArgs &, hipStream_t); bool mixed_info_f32_p##k(int, int, PassInfo *);
DFFT_F32_MIXED_FOREACH_PART(DFFT_DECL_PART)
#undef DFFT_DECL_PART
DFFT_PASS_FUNCS(DFFT_CAT(launch_mixed_f32_p, DFFT_PART), DFFT_CAT(mixed_info_f32_p, DFFT_PART), DFFT_CAT(DFFT_F32_LIST_MIXED, DFFT_PART))
#if DFFT_PART == 0
int launch_mixed_f32(int N, int variant, const PassArgs &A, hipStream_t stream)
{
    int r = -1;
#define DFFT_TRY_PART(k) if (r == -1) r = launch_mixed_f32_p##k(N, variant, A, stream);
    DFFT_F32_MIXED_FOREACH_PART(DFFT_TRY_PART)
#undef DFFT_TRY_PART
    return r;
}
bool mixed_info_f32(int N, int variant, PassInfo *pi)
{
#define DFFT_TRY_PART(k) if (mixed_info_f32_p##k(N, variant, pi)) return true;
    DFFT_F32_MIXED_FOREACH_PART(DFFT_TRY_PART)
#undef DFFT_TRY_PART
    return false;
}
#endif
}  // namespace dfft
