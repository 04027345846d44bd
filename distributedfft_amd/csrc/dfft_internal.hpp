// dfft_internal.hpp -- internal declarations shared by the host code and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

namespace dfft {

struct PassArgs;

struct PassInfo { int N, E, TL, G, threads, lds_bytes, npass; };

// tile size (lines interleaved in the intermediate layouts) per precision: 128 B per run
constexpr int TL_F64 = 8;
constexpr int TL_F32 = 16;

// returns 0 on success, -1 if the length is unsupported, else a hipError_t
int launch_pass_f64(int N, const PassArgs &A, hipStream_t stream);
int launch_pass_f32(int N, const PassArgs &A, hipStream_t stream);
bool pass_info_f64(int N, PassInfo *pi);
bool pass_info_f32(int N, PassInfo *pi);

void set_error(const std::string &msg);

}  // namespace dfft
