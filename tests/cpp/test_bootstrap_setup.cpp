// Host-only checks of the bootstrapping SETUP code (no device, no seal:: objects): the minimax polynomials of the
// modular reduction (bootstrapping/moai_remez.h) and the transform diagonals (bootstrapping/moai_fft_diagonals.h).
// The reference derives both with NTL-backed code that cannot be built here, so they are pinned to their
// mathematical definitions: equioscillation (which characterises the unique polynomial the reference's Remez
// converges to) and the canonical embedding's matrix.  `--print-cos K loge deg sf` prints coefficients for the
// fixture comparison in tests/test_bootstrap_setup.py.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "moai_fft_diagonals.h"
#include "moai_remez.h"

using namespace std;
using namespace moai_boot;

static int g_checks = 0, g_fail = 0;
#define CHECK(cond)                                                         \
    do                                                                      \
    {                                                                       \
        g_checks++;                                                         \
        if (!(cond))                                                        \
        {                                                                   \
            g_fail++;                                                       \
            printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
        }                                                                   \
    } while (0)

static void remez_checks()
{
    // MOAI's parameters (include/test/test_full_scheme.hpp:345-348): K = 25, degree 59, loge = 10, two double-angle steps
    const long K = 25, deg = 59;
    RemezResult r = remez_cos(K, 10.0, deg, 4);
    CHECK(r.chebcoeff.size() == (size_t)deg + 1);
    CHECK(r.reference.size() == (size_t)deg + 2);
    CHECK((double)r.level_spread < 1e-15);
    CHECK(fabs((double)r.error - 1.9362149866592e-10) < 1e-20);
    // equioscillation: deg + 2 points of the set, increasing, error alternating in sign with equal magnitude, and no
    // point of the set with a larger error (dense scan)
    const f128 w = (f128)pow(2.0, -10.0);
    MultiIntervalRemez chk([](f128 x) { return q::cos(2 * q::pi() * (x - (f128)0.25) / 4); }, K, w, deg);
    int sign = 0;
    bool alternates = true, inside = true, increasing = true;
    for (size_t i = 0; i < r.reference.size(); i++)
    {
        const f128 x = r.reference[i];
        const f128 e = chk.error_at(r.chebcoeff, x);
        const int s = e > 0 ? 1 : -1;
        alternates = alternates && (i == 0 || s == -sign);
        sign = s;
        const f128 frac = x - q::floor(x + (f128)0.5);
        inside = inside && q::fabs(frac) <= w * (1 + (f128)1e-20) && q::fabs(x) < (f128)K - (f128)0.5;
        increasing = increasing && (i == 0 || x > r.reference[i - 1]);
        CHECK(fabs((double)(q::fabs(e) / r.error) - 1.0) < 1e-12);
    }
    CHECK(alternates);
    CHECK(inside);
    CHECK(increasing);
    double worst = 0;
    for (long k = -(K - 1); k <= K - 1; k++)
    {
        for (int i = 0; i <= 400; i++)
        {
            const f128 x = (f128)k - w + 2 * w * (f128)i / 400;
            worst = max(worst, fabs((double)chk.error_at(r.chebcoeff, x)));
        }
    }
    CHECK(worst <= (double)r.error * (1 + 1e-9));
    printf("modular-reduction cosine: degree %ld, %d exchange steps, minimax error %.6e, level spread %.1e\n", deg, r.iterations,
           (double)r.error, (double)r.level_spread);
    // the polynomial, evaluated in double from the double-rounded coefficients, keeps that accuracy
    {
        vector<double> c = r.chebcoeff_double();
        double werr = 0;
        for (long k = -(K - 1); k <= K - 1; k++)
        {
            for (int i = 0; i <= 16; i++)
            {
                double x = k + (i - 8) / 8.0 * pow(2.0, -10.0), t = x / K, b1 = 0, b2 = 0;
                for (size_t j = c.size(); j-- > 1;)
                {
                    double tmp = 2 * t * b1 - b2 + c[j];
                    b2 = b1;
                    b1 = tmp;
                }
                werr = max(werr, fabs(t * b1 - b2 + c[0] - cos(2 * M_PI * (x - 0.25) / 4)));
            }
        }
        CHECK(werr < 2.5e-10);
    }
    // the inverse-sine scaling MOAI configures (inverse_deg = 1, ModularReducer.cpp:11): degree-1 minimax of
    // arcsin(x) / 2 pi on |x| <= sin(2 pi 2^-10).  For an odd convex function the answer is a x with
    // a w + E = f(w) and a x0 - E = f(x0), f'(x0) = a: checked in closed form.
    {
        const double lw = -log2(sin(2 * M_PI * pow(2.0, -10.0)));
        RemezResult a = remez_arcsin(lw, 1);
        const double slope = (double)a.chebcoeff[1];
        CHECK(fabs((double)a.chebcoeff[0]) < 1e-25);
        const f128 wa = (f128)pow(2.0, -lw);
        const f128 A = a.chebcoeff[1];
        const f128 two_pi = 2 * q::pi();
        const f128 x0 = q::sqrt(1 - 1 / (two_pi * A * two_pi * A)); // f'(x0) = A
        const f128 fw = q::asin(wa) / two_pi, f0 = q::asin(x0) / two_pi;
        const f128 balance = (fw - A * wa) + (f0 - A * x0);
        CHECK(fabs((double)(balance / a.error)) < 1e-10);
        CHECK(fabs(slope - 0.15915569210818443) < 1e-16);
        printf("inverse sine: slope %.17g (1 / 2 pi = %.17g), minimax error %.3e\n", slope, 1 / (2 * M_PI), (double)a.error);
    }
    // other shapes the class must handle: odd scale factor (sine), a small set, low degree
    {
        RemezResult s = remez_cos(3, 6.0, 9, 1);
        CHECK((double)s.level_spread < 1e-12 && (double)s.error < 0.1);
        RemezResult t = remez_cos(12, 10.0, 31, 2);
        CHECK((double)t.level_spread < 1e-12 && (double)t.error < 1e-2);
        bool threw = false;
        try
        {
            MultiIntervalRemez bad([](f128 x) { return x; }, 0, (f128)0.1, 3);
        }
        catch (const invalid_argument &)
        {
            threw = true;
        }
        CHECK(threw);
    }
}

static int bitrev(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++)
    {
        r |= ((v >> i) & 1) << (bits - 1 - i);
    }
    return r;
}

static void diagonal_checks()
{
    mt19937_64 rng(7);
    uniform_real_distribution<double> ud(-1, 1);
    for (int logn : { 3, 4, 5, 6, 7, 9 })
    {
        const int n = 1 << logn;
        const long K = 25;
        LevelThreeDiagonals d = level_three_diagonals(logn, K);
        const LevelThreeSplit f = forward_split(logn), v = inverse_split(logn);
        CHECK(f.part[0] + f.part[1] + f.part[2] == logn && v.part[0] + v.part[1] + v.part[2] == logn);
        CHECK(d.fftcoeff1.size() == (size_t)(2 * f.totlen[0] + 1) && d.fftcoeff2.size() == (size_t)(2 * f.totlen[1] + 1) &&
              d.fftcoeff3.size() == (size_t)(f.totlen[2] + 1));
        CHECK(d.invfftcoeff1.size() == (size_t)(v.totlen[0] + 1) && d.invfftcoeff2.size() == (size_t)(2 * v.totlen[1] + 1) &&
              d.invfftcoeff3.size() == (size_t)(2 * v.totlen[2] + 1));
        // apply a stored set the way the homomorphic transform does: sum_i diag_i (*) rot(x, offset_i)
        auto apply_centred = [&](const DiagonalSet &s, int totlen, int step, const vector<cplx> &x) {
            vector<cplx> y(n, 0);
            for (int i = 0; i <= 2 * totlen; i++)
                for (int k = 0; k < n; k++) y[k] += s[i][k] * x[(((k + (i - totlen) * step) % n) + n) % n];
            return y;
        };
        auto apply_rotated = [&](const DiagonalSet &s, int totlen, int step, const vector<cplx> &x) {
            vector<cplx> y(n, 0);
            for (int i = 0; i <= totlen; i++)
                for (int k = 0; k < n; k++) y[k] += s[i][k] * x[(k + i * step) % n];
            return y;
        };
        vector<cplx> x(n);
        for (auto &z : x) z = { ud(rng), ud(rng) };
        // slot-to-coefficient direction: F (P x) = U x with U[j][k] = exp(2 pi i 5^j k / 4n)
        vector<cplx> px(n);
        for (int k = 0; k < n; k++) px[k] = x[bitrev(k, logn)];
        vector<cplx> y = apply_centred(d.fftcoeff1, f.totlen[0], f.basicstep[0], px);
        y = apply_centred(d.fftcoeff2, f.totlen[1], f.basicstep[1], y);
        y = apply_rotated(d.fftcoeff3, f.totlen[2], f.basicstep[2], y);
        double err = 0;
        long long p5 = 1;
        for (int j = 0; j < n; j++)
        {
            cplx want = 0;
            for (int k = 0; k < n; k++)
            {
                long long e = (p5 * k) % (4LL * n);
                want += x[k] * polar(1.0, 2 * M_PI * (double)e / (4.0 * n));
            }
            err = max(err, abs(want - y[j]));
            p5 = (p5 * 5) % (4LL * n);
        }
        CHECK(err < 1e-9 * n);
        // coefficient-to-slot direction undoes it up to the two scalings: G F = identity / (2 K)
        vector<cplx> z = apply_centred(d.fftcoeff1, f.totlen[0], f.basicstep[0], x);
        z = apply_centred(d.fftcoeff2, f.totlen[1], f.basicstep[1], z);
        z = apply_rotated(d.fftcoeff3, f.totlen[2], f.basicstep[2], z);
        z = apply_rotated(d.invfftcoeff1, v.totlen[0], v.basicstep[0], z);
        z = apply_centred(d.invfftcoeff2, v.totlen[1], v.basicstep[1], z);
        z = apply_centred(d.invfftcoeff3, v.totlen[2], v.basicstep[2], z);
        double err2 = 0;
        for (int k = 0; k < n; k++) err2 = max(err2, abs(z[k] * (2.0 * K) - x[k]));
        CHECK(err2 < 1e-11 * n);
        printf("logn %d: |F P x - U x| %.2e, |2K G F x - x| %.2e, diagonals %zu+%zu+%zu / %zu+%zu+%zu\n", logn, err, err2,
               d.fftcoeff1.size(), d.fftcoeff2.size(), d.fftcoeff3.size(), d.invfftcoeff1.size(), d.invfftcoeff2.size(),
               d.invfftcoeff3.size());
    }
    // a merged stage pair against the enumerated products (the reference's 3^p sums, on the smallest case): every
    // entry is one product of roots, so the two orders of evaluation agree bit for bit
    {
        const int logn = 4, n = 16;
        DiagonalMatrix m = merge_stages(logn, 0, 2, special_fft_stage);
        DiagonalMatrix s0 = special_fft_stage(logn, 0), s1 = special_fft_stage(logn, 1);
        bool same = true;
        for (int pos = -3; pos <= 3; pos++)
        {
            vector<cplx> sum(n, 0);
            for (int a = -1; a <= 1; a++)
                for (int b = -1; b <= 1; b++)
                {
                    if (a + 2 * b != pos) continue;
                    const auto *da = s0.find(a), *db = s1.find(2 * b);
                    for (int k = 0; k < n; k++)
                    {
                        cplx t = 1.0;
                        t = t * (*da)[(k + 2 * b + 16) % n]; // stage 0 sees the index shifted by the later stage's offset
                        t = t * (*db)[k];
                        sum[k] += t;
                    }
                }
            const auto *got = m.find(pos);
            for (int k = 0; k < n; k++)
            {
                cplx g = got ? (*got)[k] : cplx(0, 0);
                same = same && g.real() == sum[k].real() && g.imag() == sum[k].imag();
            }
        }
        CHECK(same);
    }
}

int main(int argc, char **argv)
{
    if (argc == 6 && !strcmp(argv[1], "--print-cos"))
    {
        RemezResult r = remez_cos(atol(argv[2]), atof(argv[3]), atol(argv[4]), atol(argv[5]));
        printf("%.17g\n", (double)r.error);
        for (double c : r.chebcoeff_double()) printf("%.17g\n", c);
        return 0;
    }
    remez_checks();
    diagonal_checks();
    printf("%d checks, %d failed\n", g_checks, g_fail);
    if (!g_fail)
    {
        printf("ALL PASS\n");
    }
    return g_fail ? 1 : 0;
}
