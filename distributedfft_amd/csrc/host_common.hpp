// host_common.hpp -- error plumbing shared by the host translation units of libdfft_amd.so (dfft.hip, alloc.hip): every failure
// sets the thread's message (dfft_last_error) and returns a code, nothing throws.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "dfft_internal.hpp"

namespace dfft {
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return (int)e_;                                                                    \
        }                                                                                      \
    } while (0)
#define TRY(expr)                                                                              \
    do {                                                                                       \
        int r_ = (expr);                                                                       \
        if (r_ != 0) return r_;                                                                \
    } while (0)

enum { ERR_ARG = 2, ERR_STATE = 3, ERR_UNSUPPORTED = 4 };

static inline int fail(int code, const std::string &msg)
{
    set_error(msg);
    return code;
}
}  // namespace dfft
