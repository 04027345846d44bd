// dfft_internal.hpp -- internal declarations shared by the host code and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

namespace dfft {

struct PassArgs;

struct PassInfo { int N, E, TL, G, threads, lds_bytes, npass, sub; };

// tile size (lines interleaved in the intermediate layouts) per precision: 128 B per run
constexpr int TL_F64 = 8;
constexpr int TL_F32 = 16;

// returns 0 on success, -1 if the length is unsupported, else a hipError_t
int launch_pass_f64(int N, int variant, const PassArgs &A, hipStream_t stream);
int launch_pass_f32(int N, int variant, const PassArgs &A, hipStream_t stream);
bool pass_info_f64(int N, int variant, PassInfo *pi);
// real-transform z pass on M = Nz/2 complex points: mode 1 = R2C (forward), 2 = C2R (inverse)
int launch_real_f64(int M, int mode, int variant, const PassArgs &A, hipStream_t stream);
int launch_real_f32(int M, int mode, int variant, const PassArgs &A, hipStream_t stream);
// Bluestein pass with an M-point inner transform (M = power of two >= 2*A.NL - 1)
int launch_bluestein_f64(int M, const PassArgs &A, hipStream_t stream);
int launch_bluestein_f32(int M, const PassArgs &A, hipStream_t stream);
bool pass_info_f32(int N, int variant, PassInfo *pi);

void set_error(const std::string &msg);

}  // namespace dfft
