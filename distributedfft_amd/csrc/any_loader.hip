// any_loader.hip -- the core library's side of the second, lazily loaded library libdfft_amd_any.so.
//
// libdfft_amd.so holds the kernels of the power-of-two lengths 2 .. 8192 (complex axis passes and the packed real z passes): every
// BASELINE configuration, ~13 core-minutes to compile.  The kernels of every OTHER length -- the 49 mixed-radix lengths 2^a 3^b 5^c 7^d
// <= 2048, their packed real forms, and the generic kernel (Bluestein passes, two-level lines, long Bluestein) -- are ~45 core-minutes
// of template instantiations that no BASELINE configuration touches (profiles/r6_cold_build_per_object.txt); they live in
// libdfft_amd_any.so next to this library (any_exports.hip), which is opened at the first plan that needs one of them.  The
// functions below are the entry points the rest of the core calls (kernels_*.hip, real_*.hip, dfft.hip: launch_generic); they forward
// through the C symbols of the second library.  There is no fallback: when it is missing, dfft_init of such a plan fails and says
// which file to build.
#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>
#include <string>

#include "dfft_internal.hpp"

namespace dfft {
namespace {
using launch3_fn = int (*)(int, int, const PassArgs *, hipStream_t);
using launch2_fn = int (*)(int, const PassArgs *, hipStream_t);
using info3_fn = int (*)(int, int, PassInfo *);
using info1_fn = int (*)(int);

struct AnyLib {
    void *handle = nullptr;
    std::string why;                       // why it is not available
    launch3_fn launch_mixed[2] = {nullptr, nullptr}, launch_rmixed[2] = {nullptr, nullptr};      // [0] f64, [1] f32
    info3_fn mixed_info[2] = {nullptr, nullptr};
    info1_fn rmixed_info[2] = {nullptr, nullptr};
    launch2_fn launch_bluestein[2] = {nullptr, nullptr};
};
AnyLib g_any;
std::once_flag g_any_once;

void load_once()
{
    AnyLib &lib = g_any;
    std::string path;
    if (const char *e = getenv("DFFT_ANY_LIBRARY")) path = e;      // tests: another build, or a name that does not exist
    else {
        Dl_info di;
        if (!dladdr(reinterpret_cast<const void *>(&load_once), &di) || !di.dli_fname) { lib.why = "dladdr failed"; return; }
        path = di.dli_fname;
        const size_t slash = path.rfind('/');
        path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "libdfft_amd_any.so";
    }
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        const char *de = dlerror();
        lib.why = "cannot load " + path + (de ? std::string(": ") + de : std::string()) + " -- build it with `make -C distributedfft_amd/csrc any` (or __graft_entry__.build())";
        return;
    }
    bool ok = true;
    auto sym = [&](const char *name) { void *s = dlsym(h, name); if (!s) { ok = false; lib.why = path + " lacks the symbol " + name; } return s; };
    lib.launch_mixed[0] = (launch3_fn)sym("dfft_any_launch_mixed_f64");   lib.launch_mixed[1] = (launch3_fn)sym("dfft_any_launch_mixed_f32");
    lib.mixed_info[0] = (info3_fn)sym("dfft_any_mixed_info_f64");         lib.mixed_info[1] = (info3_fn)sym("dfft_any_mixed_info_f32");
    lib.launch_rmixed[0] = (launch3_fn)sym("dfft_any_launch_rmixed_f64"); lib.launch_rmixed[1] = (launch3_fn)sym("dfft_any_launch_rmixed_f32");
    lib.rmixed_info[0] = (info1_fn)sym("dfft_any_rmixed_info_f64");       lib.rmixed_info[1] = (info1_fn)sym("dfft_any_rmixed_info_f32");
    lib.launch_bluestein[0] = (launch2_fn)sym("dfft_any_launch_bluestein_f64");
    lib.launch_bluestein[1] = (launch2_fn)sym("dfft_any_launch_bluestein_f32");
    if (ok) lib.handle = h; else dlclose(h);
}
AnyLib *any()
{
    std::call_once(g_any_once, load_once);
    return g_any.handle ? &g_any : nullptr;
}
}  // namespace

// is libdfft_amd_any.so there?  (*why: the reason when it is not)
bool any_available(std::string *why)
{
    if (any()) return true;
    if (why) *why = g_any.why;
    return false;
}

// "no kernel for this length" (-1 / false) when the library is missing, exactly like a length without a configuration: the plan
// then looks for another form of the axis, ends at the generic kernel, and dfft_init reports why that is not available either
int launch_mixed_f64(int N, int variant, const PassArgs &A, hipStream_t s) { AnyLib *l = any(); return l ? l->launch_mixed[0](N, variant, &A, s) : -1; }
int launch_mixed_f32(int N, int variant, const PassArgs &A, hipStream_t s) { AnyLib *l = any(); return l ? l->launch_mixed[1](N, variant, &A, s) : -1; }
bool mixed_info_f64(int N, int variant, PassInfo *pi) { AnyLib *l = any(); return l && l->mixed_info[0](N, variant, pi) != 0; }
bool mixed_info_f32(int N, int variant, PassInfo *pi) { AnyLib *l = any(); return l && l->mixed_info[1](N, variant, pi) != 0; }
int launch_rmixed_f64(int M, int mode, const PassArgs &A, hipStream_t s) { AnyLib *l = any(); return l ? l->launch_rmixed[0](M, mode, &A, s) : -1; }
int launch_rmixed_f32(int M, int mode, const PassArgs &A, hipStream_t s) { AnyLib *l = any(); return l ? l->launch_rmixed[1](M, mode, &A, s) : -1; }
bool rmixed_info_f64(int M) { AnyLib *l = any(); return l && l->rmixed_info[0](M) != 0; }
bool rmixed_info_f32(int M) { AnyLib *l = any(); return l && l->rmixed_info[1](M) != 0; }
int launch_bluestein_f64(int M, const PassArgs &A, hipStream_t s) { AnyLib *l = any(); return l ? l->launch_bluestein[0](M, &A, s) : -1; }
int launch_bluestein_f32(int M, const PassArgs &A, hipStream_t s) { AnyLib *l = any(); return l ? l->launch_bluestein[1](M, &A, s) : -1; }
}  // namespace dfft
