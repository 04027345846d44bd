// tests/cpp/chain_check.hip -- host-side emulation of the whole Stockham chain of fft_pass.hip.h (no GPU needed).
// The kernel's own building blocks -- pass_compute (twiddles + butterflies), lds_scatter / lds_gather (the exchange through
// the padded LDS plane) and the slot -> output index map used by the store -- are compiled for the host and driven for
// every (thread, line) of a workgroup in program order, with an array standing in for LDS and the loop boundaries standing
// in for the barriers.  Every line of the workgroup gets its own random input; the result is compared with a long-double
// DFT.  Covers a selection of power-of-two configurations and every generated mixed-radix configuration of
// kernels_mixed.inc with a second template argument (-DCHAIN_PREC=...).  Built and run by tests/test_cpu_host.py.
#include "../../distributedfft_amd/csrc/fft_pass.hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <complex>
#include <vector>

namespace dfft {
#ifdef CHAIN_F32
#define DFFT_MIXED_F32
#else
#define DFFT_MIXED_F64
#endif
#include "../../distributedfft_amd/csrc/kernels_mixed.inc"
}  // namespace dfft
using namespace dfft;

static int failures = 0, checked = 0;

template <typename Cfg, int RP, int NS> static void run_pass(std::vector<typename Cfg::C> &regs, const typename Cfg::C *W)
{
    for (int tid = 0; tid < Cfg::THREADS; tid++) {
        int lw, t;
        thread_map<Cfg, false>(tid, lw, t);
        pass_compute<Cfg, RP, NS>(&regs[(size_t)tid * Cfg::kE], t, W);
    }
}
template <typename Cfg, int RP, int NS> static void run_exchange(std::vector<typename Cfg::C> &regs, std::vector<typename Cfg::real> &plane)
{
    static_for<0, 2>([&](auto pc) {
        constexpr int comp = decltype(pc)::value;
        for (int tid = 0; tid < Cfg::THREADS; tid++) {
            int lw, t;
            thread_map<Cfg, false>(tid, lw, t);
            lds_scatter<Cfg, RP, NS, comp>(&regs[(size_t)tid * Cfg::kE], plane.data(), t, lw);
        }
        for (int tid = 0; tid < Cfg::THREADS; tid++) {
            int lw, t;
            thread_map<Cfg, false>(tid, lw, t);
            lds_gather<Cfg, comp>(&regs[(size_t)tid * Cfg::kE], plane.data(), t, lw);
        }
    });
}

template <typename Cfg> static void check_cfg(const char *name)
{
    using C = typename Cfg::C;
    using R = typename Cfg::real;
    static_assert(Cfg::kMAP == 0, "line-fastest configurations only");
    constexpr int N = Cfg::kN, E = Cfg::kE, NT = Cfg::NT, TW = Cfg::TW;
    constexpr int R1 = Cfg::r1, R2 = Cfg::r2, R3 = Cfg::r3, R4 = Cfg::r4;
    const long double PI = 3.141592653589793238462643383279502884L;
    std::vector<C> W(N);
    for (int j = 0; j < N; j++) { W[j].x = (R)cosl(-2 * PI * j / N); W[j].y = (R)sinl(-2 * PI * j / N); }
    // input: line lw, point n
    std::vector<std::complex<long double>> x((size_t)TW * N);
    srand(N * 31 + TW);
    for (auto &v : x) v = std::complex<long double>((R)(rand() / (double)RAND_MAX - 0.5), (R)(rand() / (double)RAND_MAX - 0.5));
    std::vector<C> regs((size_t)Cfg::THREADS * E);
    for (int tid = 0; tid < Cfg::THREADS; tid++) {
        int lw, t;
        thread_map<Cfg, false>(tid, lw, t);
        for (int c = 0; c < E; c++) {          // the kernel's load: register c holds point t + NT*c of the lane's line
            const auto v = x[(size_t)lw * N + t + NT * c];
            regs[(size_t)tid * E + c].x = (R)v.real();
            regs[(size_t)tid * E + c].y = (R)v.imag();
        }
    }
    std::vector<R> plane(Cfg::PLANE_SLOTS + 1, (R)0);
    run_pass<Cfg, R1, 1>(regs, W.data());
    if constexpr (R2 > 1) { run_exchange<Cfg, R1, 1>(regs, plane); run_pass<Cfg, R2, R1>(regs, W.data()); }
    if constexpr (R3 > 1) { run_exchange<Cfg, R2, R1>(regs, plane); run_pass<Cfg, R3, R1 * R2>(regs, W.data()); }
    if constexpr (R4 > 1) { run_exchange<Cfg, R3, R1 * R2>(regs, plane); run_pass<Cfg, R4, R1 * R2 * R3>(regs, W.data()); }
    // the kernel's store: register c = i + mr*S of thread t holds output k = t + NT*i + brev(mr, RL)*(N/RL)
    constexpr int RL = Cfg::RLAST, S = E / RL;
    double worst = 0;
    std::vector<std::complex<long double>> X(N);
    for (int lw = 0; lw < TW; lw += (TW > 2 ? TW - 1 : 1)) {            // first and last line of the workgroup
        for (int k = 0; k < N; k++) {
            std::complex<long double> s(0, 0);
            for (int n = 0; n < N; n++) {
                const long double a = -2 * PI * (long double)(((long)k * n) % N) / N;
                s += x[(size_t)lw * N + n] * std::complex<long double>(cosl(a), sinl(a));
            }
            X[k] = s;
        }
        for (int tid = 0; tid < Cfg::THREADS; tid++) {
            int l2, t;
            thread_map<Cfg, false>(tid, l2, t);
            if (l2 != lw) continue;
            for (int c = 0; c < E; c++) {
                const int k = t + NT * (c % S) + brev(c / S, RL) * (N / RL);
                const C g = regs[(size_t)tid * E + c];
                const double err = (double)std::abs(X[k] - std::complex<long double>(g.x, g.y));
                if (err > worst) worst = err;
            }
        }
    }
    const double tol = (sizeof(R) == 8 ? 2e-15 : 1e-6) * sqrt((double)N) * log2((double)N);
    checked++;
    if (!(worst <= tol)) { failures++; printf("%-14s N = %4d  max abs error %.2e  > %.2e  FAIL\n", name, N, worst, tol); }
}

int main()
{
#ifdef CHAIN_F32
    using P64 = PassCfg<float, 64, 8, 16, 2, 8, 8, 1, 1, 2>;
    using P1024 = PassCfg<float, 1024, 32, 16, 1, 32, 8, 4, 1, 1, 1>;
    check_cfg<P64>("pow2"); check_cfg<P1024>("pow2");
#define X(n, v, cfg) check_cfg<cfg>(#cfg);
    DFFT_F32_LIST_MIXED_ALL(X)
#undef X
#else
    using P64 = PassCfg<double, 64, 8, 8, 4, 8, 8, 1, 1, 2>;
    using P512 = PassCfg<double, 512, 16, 8, 1, 8, 8, 8, 1, 1>;
    using P1024 = PassCfg<double, 1024, 16, 8, 1, 16, 16, 4, 1, 1, 1>;
    check_cfg<P64>("pow2"); check_cfg<P512>("pow2"); check_cfg<P1024>("pow2");
#define X(n, v, cfg) check_cfg<cfg>(#cfg);
    DFFT_F64_LIST_MIXED_ALL(X)
#undef X
#endif
    printf("%d configurations checked, %d failed\n%s\n", checked, failures, failures ? "FAILED" : "ALL OK");
    return failures ? 1 : 0;
}
