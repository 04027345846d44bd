// tests/cpp/butterfly_check.hip -- host-side check of the in-register butterflies of fft_pass.hip.h (no GPU needed):
// Dif<R> (mixed radix 2, 3, 5, 7) against a long-double DFT, the slot -> output index map brev(), and the compile-time
// trigonometry cossin_frac() against cosl / sinl.  Built and run by tests/test_cpu_host.py.
#include "../../distributedfft_amd/csrc/fft_pass.hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <complex>

using namespace dfft;

static int failures = 0;

template <int R, typename C> static double check_dif()
{
    using T = scalar_t<C>;
    C v[R];
    std::complex<long double> x[R];
    srand(1234 + R);
    for (int m = 0; m < R; m++) {
        const T re = (T)(rand() / (double)RAND_MAX - 0.5), im = (T)(rand() / (double)RAND_MAX - 0.5);
        v[m].x = re; v[m].y = im;
        x[m] = std::complex<long double>(re, im);
    }
    Dif<R, 0, 1, C>::run(v);
    const long double PI = 3.141592653589793238462643383279502884L;
    double worst = 0;
    bool seen[R] = {};
    for (int m = 0; m < R; m++) {
        const int k = brev(m, R);
        if (k < 0 || k >= R || seen[k]) { printf("brev(%d, %d) = %d is not a permutation\n", m, R, k); failures++; return 1; }
        seen[k] = true;
        std::complex<long double> X(0, 0);
        for (int n = 0; n < R; n++) X += x[n] * std::complex<long double>(cosl(2 * PI * k * n / R), -sinl(2 * PI * k * n / R));
        const double err = (double)std::abs(X - std::complex<long double>(v[m].x, v[m].y));
        if (err > worst) worst = err;
    }
    return worst;
}

template <int R> static void check_radix()
{
    const double e64 = check_dif<R, cdouble_t>(), e32 = check_dif<R, cfloat_t>();
    const double tol64 = 4e-16 * R, tol32 = 3e-7 * R;      // inputs are O(1); the sum grows like R
    const bool ok = e64 <= tol64 && e32 <= tol32;
    printf("radix %2d  max abs error fp64 %.2e  fp32 %.2e  %s\n", R, e64, e32, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

int main()
{
    check_radix<2>(); check_radix<3>(); check_radix<4>(); check_radix<5>(); check_radix<6>(); check_radix<7>(); check_radix<8>();
    check_radix<9>(); check_radix<10>(); check_radix<12>(); check_radix<14>(); check_radix<15>(); check_radix<16>();
    check_radix<18>(); check_radix<20>(); check_radix<21>(); check_radix<24>(); check_radix<25>(); check_radix<27>();
    check_radix<28>(); check_radix<30>(); check_radix<32>(); check_radix<64>();
    // compile-time trigonometry: relative to 1 ulp of the larger component
    const long double PI = 3.141592653589793238462643383279502884L;
    double worst = 0;
    for (int R = 3; R <= 2048; R++) {
        for (int j = 0; j < R; j += (R > 64 ? 7 : 1)) {
            const CosSin w = cossin_frac(j, R);
            const double ec = fabs((double)(w.c - cosl(2 * PI * j / R))), es = fabs((double)(w.s - sinl(2 * PI * j / R)));
            if (ec > worst) worst = ec;
            if (es > worst) worst = es;
        }
    }
    printf("cossin_frac max abs error %.2e %s\n", worst, worst < 3e-16 ? "ok" : "FAIL");
    if (!(worst < 3e-16)) failures++;
    printf(failures ? "FAILED\n" : "ALL OK\n");
    return failures ? 1 : 0;
}
