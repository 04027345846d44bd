// rmixed_f32.hip -- packed real z passes (R2C / C2R) of the mixed-radix lengths, f32: kernels_mixed.inc's RMIXED lists,
// compiled in DFFT_F32_RMIXED_PARTS parts (-DDFFT_PART = k; part 0 holds the entry points).
#include "kernels.hip.inc"

namespace dfft {
#define DFFT_MIXED_F32
#include "kernels_mixed.inc"

#define DFFT_DECL_PART(k) int launch_rmixed_f32_p##k(int, int, const PassArgs &, hipStream_t); bool rmixed_info_f32_p##k(int);
DFFT_F32_RMIXED_FOREACH_PART(DFFT_DECL_PART)
#undef DFFT_DECL_PART
DFFT_REAL_MIXED_FUNCS(DFFT_CAT(launch_rmixed_f32_p, DFFT_PART), DFFT_CAT(rmixed_info_f32_p, DFFT_PART), DFFT_CAT(DFFT_F32_LIST_RMIXED, DFFT_PART))
#if DFFT_PART == 0
int launch_rmixed_f32(int M, int mode, const PassArgs &A, hipStream_t stream)
{
    int r = -1;
#define DFFT_TRY_PART(k) if (r == -1) r = launch_rmixed_f32_p##k(M, mode, A, stream);
    DFFT_F32_RMIXED_FOREACH_PART(DFFT_TRY_PART)
#undef DFFT_TRY_PART
    return r;
}
bool rmixed_info_f32(int M)
{
#define DFFT_TRY_PART(k) if (rmixed_info_f32_p##k(M)) return true;
    DFFT_F32_RMIXED_FOREACH_PART(DFFT_TRY_PART)
#undef DFFT_TRY_PART
    return false;
}
#endif
}  // namespace dfft
