// bootstrapping/moai_remez.h -- the minimax polynomials of MOAI's modular reduction, computed without NTL.
//
// What the reference computes (include/source/bootstrapping/common/Remez.cpp:577-586, RemezCos.h:11-16,
// RemezArcsin.h:10-13): the best uniform approximation of degree `deg` of
//     cos(2 pi (x - 1/4) / sf)   (sin(2 pi x / sf) for odd sf)      on  U = union of [k - w, k + w], |k| <= K - 1,
// and of arcsin(x) / (2 pi) on [-w', w'], in the basis T_j(x / K), by a multi-interval Remez iteration in
// 1000-bit NTL floating point that stops when the error levels at the deg + 2 reference points agree to 2^-120.
// That limit is THE minimax polynomial of the function on the set (unique by the equioscillation theorem), so any
// convergent exchange iteration reproduces it; the coefficients are used as doubles (Polynomial.cpp `to_double`).
// This file runs its own exchange iteration in 113-bit binary128 arithmetic (__float128: +, -, *, / come from
// libgcc; cos, sin and arcsin are evaluated here from their series, so no libquadmath and no NTL), which leaves the
// double-rounded coefficients with a relative error around 1e-15 (checked against a 400-bit computation of the
// same polynomial, tests/golden/remez_cos_K25_deg59_loge10_r2.json, and against the equioscillation property).
//
// Own design, not a transcription: least-squares start, all local extrema of the error per interval by scanning and
// golden-section refinement, the classical "drop the weakest, keep alternation" selection, Gaussian elimination with
// partial pivoting.  Host-only setup code (runs once per Bootstrapper).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <stdexcept>
#include <vector>

namespace moai_boot
{
    using f128 = __float128;

    namespace q
    {
        inline f128 from_parts(double a, double b, double c)
        {
            return static_cast<f128>(a) + static_cast<f128>(b) + static_cast<f128>(c);
        }
        // pi as a triple-double: 3.14159265358979311600e+00 + 1.22464679914735320717e-16 - 2.99476980971833966728e-33
        inline f128 pi()
        {
            return from_parts(3.141592653589793116, 1.2246467991473532072e-16, -2.9947698097183396659e-33);
        }
        inline f128 fabs(f128 x)
        {
            return x < 0 ? -x : x;
        }
        inline f128 floor(f128 x)
        {
            // |x| below 2^62 is all this file needs
            long long i = static_cast<long long>(x);
            f128 f = static_cast<f128>(i);
            return f > x ? f - 1 : f;
        }
        // sin and cos of r in [-pi/4, pi/4] by Taylor series (34 digits need 30-odd terms at most)
        inline void sincos_small(f128 r, f128 &s, f128 &c)
        {
            const f128 r2 = r * r;
            f128 ts = r, tc = 1;
            s = r;
            c = 1;
            for (int k = 1; k < 40; k++)
            {
                tc = -tc * r2 / static_cast<f128>((2 * k - 1) * (2 * k));
                ts = -ts * r2 / static_cast<f128>((2 * k) * (2 * k + 1));
                c += tc;
                s += ts;
                if (fabs(tc) < static_cast<f128>(1e-40) && fabs(ts) < static_cast<f128>(1e-40))
                {
                    break;
                }
            }
        }
        inline void sincos(f128 x, f128 &s, f128 &c)
        {
            // x = n * pi/2 + r
            const f128 half_pi = pi() / 2;
            f128 n = floor(x / half_pi + static_cast<f128>(0.5));
            f128 r = x - n * half_pi;
            f128 sr, cr;
            sincos_small(r, sr, cr);
            long long m = static_cast<long long>(n);
            switch (((m % 4) + 4) % 4)
            {
            case 0: s = sr; c = cr; break;
            case 1: s = cr; c = -sr; break;
            case 2: s = -sr; c = -cr; break;
            default: s = -cr; c = sr; break;
            }
        }
        inline f128 cos(f128 x)
        {
            f128 s, c;
            sincos(x, s, c);
            return c;
        }
        inline f128 sin(f128 x)
        {
            f128 s, c;
            sincos(x, s, c);
            return s;
        }
        inline f128 sqrt(f128 x)
        {
            if (x <= 0)
            {
                return 0;
            }
            f128 y = static_cast<f128>(std::sqrt(static_cast<double>(x)));
            for (int i = 0; i < 4; i++)
            {
                y = (y + x / y) / 2;
            }
            return y;
        }
        // arcsin for |x| <= 0.5: Newton's iteration on sin from the double-precision value
        inline f128 asin(f128 x)
        {
            if (fabs(x) > static_cast<f128>(0.5))
            {
                throw std::invalid_argument("asin: argument out of the range this file needs");
            }
            f128 y = static_cast<f128>(std::asin(static_cast<double>(x)));
            for (int i = 0; i < 4; i++)
            {
                f128 s, c;
                sincos(y, s, c);
                y -= (s - x) / c;
            }
            return y;
        }
    } // namespace q

    // sum_j c[j] T_j(t), Clenshaw
    inline f128 cheb_eval(const std::vector<f128> &c, f128 t)
    {
        f128 b1 = 0, b2 = 0;
        for (std::size_t j = c.size(); j-- > 1;)
        {
            f128 tmp = 2 * t * b1 - b2 + c[j];
            b2 = b1;
            b1 = tmp;
        }
        return t * b1 - b2 + c[0];
    }

    struct RemezResult
    {
        std::vector<f128> chebcoeff;     // basis T_j(x / K)
        f128 error = 0;                  // the levelled (minimax) error
        f128 level_spread = 0;           // (max - min) / min of |error| over the final reference
        std::vector<f128> reference;     // deg + 2 alternation points
        int iterations = 0;
        std::vector<double> chebcoeff_double() const
        {
            std::vector<double> d(chebcoeff.size());
            for (std::size_t i = 0; i < d.size(); i++)
            {
                d[i] = static_cast<double>(chebcoeff[i]);
            }
            return d;
        }
    };

    class MultiIntervalRemez
    {
    public:
        // the set: [k - width, k + width] for k = -(K-1) .. K-1 (Remez.cpp:304-306: 2K-1 intervals); basis T_j(x / K)
        MultiIntervalRemez(std::function<f128(f128)> f, long boundary_K, f128 width, long deg)
            : f_(std::move(f)), K_(boundary_K), w_(width), deg_(deg)
        {
            if (K_ < 1 || deg_ < 1 || !(w_ > 0))
            {
                throw std::invalid_argument("invalid Remez parameters");
            }
        }

        // tolerance on (max - min) / min of the error levels: binary128 bottoms out near 1e-22 for errors around 1e-10
        // (the error is a difference of values near 1), far below what survives the rounding of the coefficients to double
        RemezResult run(f128 tolerance = static_cast<f128>(1e-18), int max_iterations = 100) const
        {
            RemezResult r;
            r.chebcoeff = least_squares_start();
            std::vector<Extremum> ref;
            for (int it = 0; it < max_iterations; it++)
            {
                r.iterations = it + 1;
                ref = select(extrema(r.chebcoeff));
                f128 hi = 0, lo = q::fabs(ref[0].e);
                for (auto &p : ref)
                {
                    hi = std::max(hi, q::fabs(p.e));
                    lo = std::min(lo, q::fabs(p.e));
                }
                r.level_spread = (hi - lo) / lo;
                r.error = hi;
                if (it > 0 && r.level_spread < tolerance)
                {
                    break;
                }
                if (it + 1 == max_iterations)
                {
                    throw std::runtime_error("Remez iteration did not converge");
                }
                solve_on_reference(ref, r.chebcoeff);
            }
            for (auto &p : ref)
            {
                r.reference.push_back(p.x);
            }
            return r;
        }

        f128 error_at(const std::vector<f128> &c, f128 x) const
        {
            return cheb_eval(c, x / static_cast<f128>(K_)) - f_(x);
        }

    private:
        struct Extremum
        {
            f128 x, e;
        };

        void basis_row(f128 x, std::vector<f128> &row) const
        {
            const f128 t = x / static_cast<f128>(K_);
            row[0] = 1;
            if (deg_ >= 1)
            {
                row[1] = t;
            }
            for (long j = 2; j <= deg_; j++)
            {
                row[static_cast<std::size_t>(j)] = 2 * t * row[static_cast<std::size_t>(j - 1)] - row[static_cast<std::size_t>(j - 2)];
            }
        }

        // least squares on 8 Chebyshev points per interval by Householder QR: its error already has about the right
        // number of sign changes, which makes the first exchange well defined for any (K, deg) the set supports
        std::vector<f128> least_squares_start() const
        {
            const long per = std::max<long>(8, (deg_ + 1) / (2 * K_ - 1) + 4);
            const std::size_t cols = static_cast<std::size_t>(deg_ + 1);
            std::vector<f128> pts;
            for (long k = -(K_ - 1); k <= K_ - 1; k++)
            {
                for (long j = 0; j < per; j++)
                {
                    pts.push_back(static_cast<f128>(k) + w_ * q::cos(q::pi() * static_cast<f128>(2 * j + 1) / static_cast<f128>(2 * per)));
                }
            }
            const std::size_t rows = pts.size();
            if (rows < cols)
            {
                throw std::invalid_argument("degree too large for the set");
            }
            std::vector<std::vector<f128>> A(rows, std::vector<f128>(cols));
            std::vector<f128> b(rows);
            for (std::size_t i = 0; i < rows; i++)
            {
                basis_row(pts[i], A[i]);
                b[i] = f_(pts[i]);
            }
            for (std::size_t c = 0; c < cols; c++)
            {
                f128 norm = 0;
                for (std::size_t i = c; i < rows; i++)
                {
                    norm += A[i][c] * A[i][c];
                }
                norm = q::sqrt(norm);
                if (norm == 0)
                {
                    throw std::runtime_error("rank-deficient least-squares start");
                }
                const f128 alpha = A[c][c] > 0 ? -norm : norm;
                std::vector<f128> v(rows - c);
                for (std::size_t i = c; i < rows; i++)
                {
                    v[i - c] = A[i][c];
                }
                v[0] -= alpha;
                f128 vv = 0;
                for (auto &x : v)
                {
                    vv += x * x;
                }
                if (vv == 0)
                {
                    continue;
                }
                auto reflect = [&](auto &&get, auto &&set) {
                    f128 dot = 0;
                    for (std::size_t i = c; i < rows; i++)
                    {
                        dot += v[i - c] * get(i);
                    }
                    const f128 s = 2 * dot / vv;
                    for (std::size_t i = c; i < rows; i++)
                    {
                        set(i, get(i) - s * v[i - c]);
                    }
                };
                for (std::size_t cc = c; cc < cols; cc++)
                {
                    reflect([&](std::size_t i) { return A[i][cc]; }, [&](std::size_t i, f128 val) { A[i][cc] = val; });
                }
                reflect([&](std::size_t i) { return b[i]; }, [&](std::size_t i, f128 val) { b[i] = val; });
            }
            std::vector<f128> x(cols);
            for (std::size_t c = cols; c-- > 0;)
            {
                f128 s = b[c];
                for (std::size_t j = c + 1; j < cols; j++)
                {
                    s -= A[c][j] * x[j];
                }
                x[c] = s / A[c][c];
            }
            return x;
        }

        // every local maximum of |error| on every interval, endpoints included, in increasing x
        std::vector<Extremum> extrema(const std::vector<f128> &c) const
        {
            const int grid = 48;
            const f128 g1 = static_cast<f128>(0.381966011250105151795413165634362L);
            const f128 g2 = static_cast<f128>(0.618033988749894848204586834365638L);
            std::vector<Extremum> out;
            std::vector<f128> xs(grid + 1), es(grid + 1);
            for (long k = -(K_ - 1); k <= K_ - 1; k++)
            {
                for (int i = 0; i <= grid; i++)
                {
                    xs[static_cast<std::size_t>(i)] = static_cast<f128>(k) - w_ + 2 * w_ * static_cast<f128>(i) / static_cast<f128>(grid);
                    es[static_cast<std::size_t>(i)] = error_at(c, xs[static_cast<std::size_t>(i)]);
                }
                for (int i = 0; i <= grid; i++)
                {
                    const f128 a = q::fabs(es[static_cast<std::size_t>(i)]);
                    const bool left_ok = i == 0 || a >= q::fabs(es[static_cast<std::size_t>(i - 1)]);
                    const bool right_ok = i == grid || a > q::fabs(es[static_cast<std::size_t>(i + 1)]);
                    if (!left_ok || !right_ok)
                    {
                        continue;
                    }
                    f128 x = xs[static_cast<std::size_t>(i)];
                    if (i > 0 && i < grid)
                    {
                        f128 lo = xs[static_cast<std::size_t>(i - 1)], hi = xs[static_cast<std::size_t>(i + 1)];
                        const f128 sgn = es[static_cast<std::size_t>(i)] > 0 ? 1 : -1;
                        f128 m1 = lo + (hi - lo) * g1, m2 = lo + (hi - lo) * g2;
                        f128 e1 = sgn * error_at(c, m1), e2 = sgn * error_at(c, m2);
                        for (int s = 0; s < 90; s++)
                        {
                            if (e1 < e2)
                            {
                                lo = m1;
                                m1 = m2;
                                e1 = e2;
                                m2 = lo + (hi - lo) * g2;
                                e2 = sgn * error_at(c, m2);
                            }
                            else
                            {
                                hi = m2;
                                m2 = m1;
                                e2 = e1;
                                m1 = lo + (hi - lo) * g1;
                                e1 = sgn * error_at(c, m1);
                            }
                        }
                        x = (lo + hi) / 2;
                    }
                    out.push_back({ x, error_at(c, x) });
                }
            }
            return out;
        }

        // deg + 2 of the extrema with alternating signs: the largest of every run of equal sign, then the weakest
        // dropped (an end point alone, an interior point together with its weaker neighbour, which keeps alternation)
        std::vector<Extremum> select(const std::vector<Extremum> &cand) const
        {
            std::vector<Extremum> m;
            for (auto &p : cand)
            {
                if (!m.empty() && ((m.back().e > 0) == (p.e > 0)))
                {
                    if (q::fabs(p.e) > q::fabs(m.back().e))
                    {
                        m.back() = p;
                    }
                }
                else
                {
                    m.push_back(p);
                }
            }
            const std::size_t want = static_cast<std::size_t>(deg_ + 2);
            if (m.size() < want)
            {
                throw std::runtime_error("the error has too few alternations for this degree");
            }
            while (m.size() > want)
            {
                if ((m.size() - want) % 2 == 1)
                {
                    if (q::fabs(m.front().e) < q::fabs(m.back().e))
                    {
                        m.erase(m.begin());
                    }
                    else
                    {
                        m.pop_back();
                    }
                }
                else
                {
                    std::size_t best = 0;
                    f128 best_val = -1;
                    for (std::size_t i = 0; i + 1 < m.size(); i++)
                    {
                        const f128 v = std::max(q::fabs(m[i].e), q::fabs(m[i + 1].e));
                        if (best_val < 0 || v < best_val)
                        {
                            best_val = v;
                            best = i;
                        }
                    }
                    m.erase(m.begin() + static_cast<std::ptrdiff_t>(best), m.begin() + static_cast<std::ptrdiff_t>(best) + 2);
                }
            }
            return m;
        }

        // p(x_i) + (-1)^i E = f(x_i), i = 0 .. deg + 1
        void solve_on_reference(const std::vector<Extremum> &ref, std::vector<f128> &coeff) const
        {
            const std::size_t n = static_cast<std::size_t>(deg_ + 2);
            std::vector<std::vector<f128>> M(n, std::vector<f128>(n + 1));
            std::vector<f128> row(static_cast<std::size_t>(deg_ + 1));
            for (std::size_t i = 0; i < n; i++)
            {
                basis_row(ref[i].x, row);
                std::copy(row.begin(), row.end(), M[i].begin());
                M[i][n - 1] = (i % 2) ? -1 : 1;
                M[i][n] = f_(ref[i].x);
            }
            for (std::size_t c = 0; c < n; c++)
            {
                std::size_t piv = c;
                for (std::size_t i = c + 1; i < n; i++)
                {
                    if (q::fabs(M[i][c]) > q::fabs(M[piv][c]))
                    {
                        piv = i;
                    }
                }
                if (M[piv][c] == 0)
                {
                    throw std::runtime_error("singular exchange system");
                }
                std::swap(M[c], M[piv]);
                for (std::size_t i = c + 1; i < n; i++)
                {
                    const f128 f = M[i][c] / M[c][c];
                    if (f == 0)
                    {
                        continue;
                    }
                    for (std::size_t j = c; j <= n; j++)
                    {
                        M[i][j] -= f * M[c][j];
                    }
                }
            }
            std::vector<f128> x(n);
            for (std::size_t c = n; c-- > 0;)
            {
                f128 s = M[c][n];
                for (std::size_t j = c + 1; j < n; j++)
                {
                    s -= M[c][j] * x[j];
                }
                x[c] = s / M[c][c];
            }
            coeff.assign(x.begin(), x.begin() + static_cast<std::ptrdiff_t>(deg_ + 1));
        }

        std::function<f128(f128)> f_;
        long K_;
        f128 w_;
        long deg_;
    };

    // RemezCos (RemezCos.h:7-17): scale_factor even -> cos(2 pi (x - 1/4) / scale_factor), odd -> sin(2 pi x / scale_factor)
    inline RemezResult remez_cos(long boundary_K, double log_width, long deg, long scale_factor)
    {
        const f128 width = static_cast<f128>(std::pow(2.0, -log_width)); // Remez.cpp:8
        const f128 sf = static_cast<f128>(scale_factor);
        std::function<f128(f128)> f;
        if (scale_factor % 2 == 0)
        {
            f = [sf](f128 x) { return q::cos(2 * q::pi() * (x - static_cast<f128>(0.25)) / sf); };
        }
        else
        {
            f = [sf](f128 x) { return q::sin(2 * q::pi() * x / sf); };
        }
        return MultiIntervalRemez(f, boundary_K, width, deg).run();
    }

    // RemezArcsin (RemezArcsin.h:5-14): arcsin(x) / (2 pi) on [-2^-log_width, 2^-log_width], boundary_K = 1
    inline RemezResult remez_arcsin(double log_width, long deg)
    {
        const f128 width = static_cast<f128>(std::pow(2.0, -log_width));
        auto f = [](f128 x) { return q::asin(x) / (2 * q::pi()); };
        return MultiIntervalRemez(f, 1, width, deg).run();
    }
} // namespace moai_boot
