// dfft_internal.hpp -- internal declarations shared by the host code and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

namespace dfft {

struct PassArgs;

struct PassInfo { int N, E, TL, G, threads, lds_bytes, npass, sub; };

// The variant number of a kernel configuration is its role in a plan; a length that has no configuration for a
// role falls back to ROLE_DEFAULT (launch_pass).  See the comments in kernels_f64.hip / kernels_f32.hip.
enum PassRole {
    ROLE_DEFAULT = 0,
    ROLE_STRIDED_READ = 1,     // fp64: loads the point-major API layout (multi-rank inverse x pass)
    ROLE_NATURAL_LOAD_TILED_STORE = 1,   // fp32: natural lines in (point fastest), same-tile blocks out (line fastest): inverse x pass, x-contiguous spectrum
    ROLE_STREAM = 3,           // fp64: stores in long runs (tiled-transpose chunks, natural lines): nontemporal
    ROLE_NATURAL_LOAD = 4,     // fp32: natural lines in (point-fastest mapping in every pass)
    ROLE_NATURAL_STORE = 5,    // fp32: natural lines out (point-fastest mapping after the first pass)
    ROLE_TILED = 6,            // fp32: tiled on both sides (two radix passes, one exchange)
    ROLE_LINES = 7,            // fp64: natural lines on one side where that differs from ROLE_STREAM (2048: sub-tiles)
    ROLE_TRANSPOSED_STORE = 7, // fp32: tiled load, transposed-tile store in whole 128-byte lines (the inverse y pass of a multi-rank plan)
    ROLE_STRIDED_READ_FUSED = 8,   // fp64: ROLE_STRIDED_READ as persistent workgroups, stores fused with the next tile's loads (plans with P1 = 1)
    ROLE_TILED_STREAM = 9      // fp32: ROLE_TILED with nontemporal loads and stores (chosen by measurement only: dfft_tune_variants)
};

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
bool real_supported_f64(int M);
bool real_supported_f32(int M);
// Bluestein pass with an M-point inner transform (M = power of two >= 2*A.NL - 1)
int launch_bluestein_f64(int M, const PassArgs &A, hipStream_t stream);
int launch_bluestein_f32(int M, const PassArgs &A, hipStream_t stream);
bool pass_info_f32(int N, int variant, PassInfo *pi);
#ifdef DFFT_EXPERIMENTS
int launch_shfl_f32(int N, int dpp, const PassArgs &A, hipStream_t stream);      // LDS-free shuffle pass (A/B only)
#endif

// libdfft_amd_any.so (any_loader.hip): the kernels of every length that is not a power of two up to 8192 -- mixed radix, Bluestein,
// two-level lines -- loaded at the first plan that needs them.  The launchers of those kernels answer -1 / false while it is missing.
bool any_available(std::string *why);

void set_error(const std::string &msg);

}  // namespace dfft
