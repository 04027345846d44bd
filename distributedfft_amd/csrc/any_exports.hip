// any_exports.hip -- the C entry points of libdfft_amd_any.so (see any_loader.hip): the kernels of the mixed-radix lengths
// (mixed_*.hip, rmixed_*.hip) and the generic kernel (bluestein_*.hip) behind plain C symbols that the core library looks up with
// dlsym.  The library is linked with -Bsymbolic: the dfft:: functions called here are the ones defined in THIS library's objects,
// not the forwarders of the same name in libdfft_amd.so when that one is in the global scope (the C++ drivers link it directly).
#include "dfft_internal.hpp"

namespace dfft {
int launch_mixed_f64(int N, int variant, const PassArgs &A, hipStream_t stream);
int launch_mixed_f32(int N, int variant, const PassArgs &A, hipStream_t stream);
bool mixed_info_f64(int N, int variant, PassInfo *pi);
bool mixed_info_f32(int N, int variant, PassInfo *pi);
int launch_rmixed_f64(int M, int mode, const PassArgs &A, hipStream_t stream);
int launch_rmixed_f32(int M, int mode, const PassArgs &A, hipStream_t stream);
bool rmixed_info_f64(int M);
bool rmixed_info_f32(int M);
}  // namespace dfft

extern "C" {
__attribute__((visibility("default"))) int dfft_any_launch_mixed_f64(int N, int variant, const dfft::PassArgs *A, hipStream_t s) { return dfft::launch_mixed_f64(N, variant, *A, s); }
__attribute__((visibility("default"))) int dfft_any_launch_mixed_f32(int N, int variant, const dfft::PassArgs *A, hipStream_t s) { return dfft::launch_mixed_f32(N, variant, *A, s); }
__attribute__((visibility("default"))) int dfft_any_mixed_info_f64(int N, int variant, dfft::PassInfo *pi) { return dfft::mixed_info_f64(N, variant, pi) ? 1 : 0; }
__attribute__((visibility("default"))) int dfft_any_mixed_info_f32(int N, int variant, dfft::PassInfo *pi) { return dfft::mixed_info_f32(N, variant, pi) ? 1 : 0; }
__attribute__((visibility("default"))) int dfft_any_launch_rmixed_f64(int M, int mode, const dfft::PassArgs *A, hipStream_t s) { return dfft::launch_rmixed_f64(M, mode, *A, s); }
__attribute__((visibility("default"))) int dfft_any_launch_rmixed_f32(int M, int mode, const dfft::PassArgs *A, hipStream_t s) { return dfft::launch_rmixed_f32(M, mode, *A, s); }
__attribute__((visibility("default"))) int dfft_any_rmixed_info_f64(int M) { return dfft::rmixed_info_f64(M) ? 1 : 0; }
__attribute__((visibility("default"))) int dfft_any_rmixed_info_f32(int M) { return dfft::rmixed_info_f32(M) ? 1 : 0; }
__attribute__((visibility("default"))) int dfft_any_launch_bluestein_f64(int M, const dfft::PassArgs *A, hipStream_t s) { return dfft::launch_bluestein_f64(M, *A, s); }
__attribute__((visibility("default"))) int dfft_any_launch_bluestein_f32(int M, const dfft::PassArgs *A, hipStream_t s) { return dfft::launch_bluestein_f32(M, *A, s); }
}
