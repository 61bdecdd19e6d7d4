// bootstrapping/moai_fft_diagonals.h -- the plaintext diagonals of the coefficient-to-slot and slot-to-coefficient
// transforms of MOAI's bootstrapping (the "level-3" split used by Bootstrapper::bootstrap_3), from the definition.
//
// Mathematics.  CKKS decoding evaluates the plaintext polynomial at zeta^(5^j), zeta = exp(2 pi i / 4n) for n slots:
// slots = U (t_lo + i t_hi) with U[j][k] = zeta^(5^j k).  The "special FFT" (Cheon-Kim-Kim-Song, HEAAN fftSpecial)
// factors U P (P = bit reversal) into log2 n butterfly stages
//     stage s (span h = 2^s, block 2h, position j < h inside a block, w = exp(i pi (5^j mod 8h) / 4h)):
//         out[k]     = in[k] + w in[k + h]
//         out[k + h] = in[k] - w in[k + h]
// and its inverse runs the stages backwards (span n/2 first) with the conjugate roots and a factor 1/2 per stage:
//         out[k]     = (in[k] + in[k + h]) / 2
//         out[k + h] = (in[k] - in[k + h]) conj(w) / 2
// Every stage is a matrix with three generalised diagonals (offsets -h, 0, +h).  Homomorphically a diagonal costs one
// plaintext product and an offset one rotation, so the reference merges the stages into three groups per direction
// and evaluates each group as one baby-step / giant-step transform (Bootstrapper.cpp:1997-2129); the bit reversal is
// never applied (slot-wise operations between the two transforms commute with it).
//
// What the reference stores (Bootstrapper.h:43-46; built by genorigcoeff Bootstrapper.cpp:522-603, genfftcoeff_3
// :1144-1417 and geninvfftcoeff_3 :1552-1820 for logn == logNh by enumerating the 3^p products of stage diagonals):
//   fftcoeff1/2   [2 totlen + 1][n]: diagonal of offset (i - totlen) * basicstep at index i
//   fftcoeff3     [totlen + 1][n]  : offsets taken mod (totlen + 1) (basicstep * (totlen + 1) == n, the "rotated" form)
//   invfftcoeff1  [totlen + 1][n]  : rotated form, scaled by 1 / boundary_K
//   invfftcoeff2/3[2 totlen + 1][n], the third scaled by 1/2
// This file produces the same arrays by multiplying the stage matrices in their diagonal representation
// ((A B)_{a+b}[k] += A_a[k] B_b[k + a]) -- a dynamic-programming form of the same sums: every entry of a merged
// matrix has exactly one non-zero path through the butterflies, so each value is the same product of the same roots,
// multiplied in the same (first stage first) order.  Host-only setup code, double precision like the reference's.
// The reference's own constants cannot be produced here (its Bootstrapper needs NTL), so nothing compares the arrays
// with its output ("parity unpinned"); they are pinned to the definition instead: tests/cpp/test_bootstrap_setup.cpp
// checks F P = U against the O(n^2) sum, F^-1 F = identity / (2 K), and the GPU tests that a bootstrapped ciphertext
// decrypts to its message.
#pragma once
#include <cmath>
#include <complex>
#include <cstddef>
#include <map>
#include <stdexcept>
#include <vector>

namespace moai_boot
{
    using cplx = std::complex<double>;

    // (M x)[k] = sum over stored offsets d of diag[d][k] * x[(k + d) mod n]; offsets are kept in [0, n)
    class DiagonalMatrix
    {
    public:
        explicit DiagonalMatrix(int n) : n_(n)
        {
        }
        int n() const
        {
            return n_;
        }
        std::vector<cplx> &diagonal(int offset)
        {
            auto &v = d_[norm(offset)];
            if (v.empty())
            {
                v.assign(static_cast<std::size_t>(n_), cplx(0.0, 0.0));
            }
            return v;
        }
        const std::vector<cplx> *find(int offset) const
        {
            auto it = d_.find(norm(offset));
            return it == d_.end() ? nullptr : &it->second;
        }
        std::size_t diagonal_count() const
        {
            return d_.size();
        }
        // this <- S * this (S acts after this)
        void apply_left(const DiagonalMatrix &S)
        {
            if (S.n_ != n_)
            {
                throw std::invalid_argument("dimension mismatch");
            }
            std::map<int, std::vector<cplx>> out;
            for (const auto &sa : S.d_)
            {
                const int a = sa.first;
                for (const auto &mb : d_)
                {
                    auto &dst = out[norm(a + mb.first)];
                    if (dst.empty())
                    {
                        dst.assign(static_cast<std::size_t>(n_), cplx(0.0, 0.0));
                    }
                    for (int k = 0; k < n_; k++)
                    {
                        const cplx &m = mb.second[static_cast<std::size_t>((k + a) % n_)];
                        const cplx &s = sa.second[static_cast<std::size_t>(k)];
                        if ((m.real() == 0.0 && m.imag() == 0.0) || (s.real() == 0.0 && s.imag() == 0.0))
                        {
                            continue;
                        }
                        dst[static_cast<std::size_t>(k)] += m * s;
                    }
                }
            }
            d_.swap(out);
        }
        void scale(double f)
        {
            for (auto &kv : d_)
            {
                for (auto &z : kv.second)
                {
                    z *= f;
                }
            }
        }
        std::vector<cplx> apply(const std::vector<cplx> &x) const
        {
            std::vector<cplx> y(static_cast<std::size_t>(n_), cplx(0.0, 0.0));
            for (const auto &kv : d_)
            {
                for (int k = 0; k < n_; k++)
                {
                    y[static_cast<std::size_t>(k)] += kv.second[static_cast<std::size_t>(k)] * x[static_cast<std::size_t>((k + kv.first) % n_)];
                }
            }
            return y;
        }

    private:
        int norm(int offset) const
        {
            return ((offset % n_) + n_) % n_;
        }
        int n_;
        std::map<int, std::vector<cplx>> d_;
    };

    // stage s of the forward special FFT on n = 2^logn slots
    inline DiagonalMatrix special_fft_stage(int logn, int s)
    {
        const int n = 1 << logn, h = 1 << s, block = 2 * h;
        DiagonalMatrix M(n);
        auto &lower = M.diagonal(-h), &mid = M.diagonal(0), &upper = M.diagonal(h);
        const double theta = (M_PI / (2 * n)) * (1 << (logn - 1 - s)); // pi / 4h
        int power = 1;
        for (int j = 0; j < h; j++)
        {
            const cplx w = std::polar(1.0, theta * power);
            for (int k = j; k < n; k += block)
            {
                mid[static_cast<std::size_t>(k)] = 1;
                upper[static_cast<std::size_t>(k)] = w;
                mid[static_cast<std::size_t>(k + h)] = -w;
                lower[static_cast<std::size_t>(k + h)] = 1;
            }
            power = (5 * power) % (4 * block);
        }
        return M;
    }

    // stage s of the inverse (s = 0 has span n/2)
    inline DiagonalMatrix special_ifft_stage(int logn, int s)
    {
        const int n = 1 << logn, block = n >> s, h = block / 2;
        DiagonalMatrix M(n);
        auto &lower = M.diagonal(-h), &mid = M.diagonal(0), &upper = M.diagonal(h);
        const double theta = (-M_PI / (2 * n)) * (1 << s); // -pi / 4h
        int power = 1;
        for (int j = 0; j < h; j++)
        {
            const cplx w = std::polar(1.0, theta * power);
            for (int k = j; k < n; k += block)
            {
                mid[static_cast<std::size_t>(k)] = 0.5;
                upper[static_cast<std::size_t>(k)] = 0.5;
                mid[static_cast<std::size_t>(k + h)] = -0.5 * w;
                lower[static_cast<std::size_t>(k + h)] = 0.5 * w;
            }
            power = (5 * power) % (4 * block);
        }
        return M;
    }

    // product of stages first .. first + count - 1 (the first acts first)
    template <typename Stage>
    DiagonalMatrix merge_stages(int logn, int first, int count, Stage stage)
    {
        DiagonalMatrix M = stage(logn, first);
        for (int s = first + 1; s < first + count; s++)
        {
            M.apply_left(stage(logn, s));
        }
        return M;
    }

    using DiagonalSet = std::vector<std::vector<cplx>>;

    // index i holds the diagonal of offset (i - totlen) * basicstep, i = 0 .. 2 totlen
    inline DiagonalSet centred_layout(const DiagonalMatrix &M, int totlen, int basicstep)
    {
        DiagonalSet out(static_cast<std::size_t>(2 * totlen + 1), std::vector<cplx>(static_cast<std::size_t>(M.n()), cplx(0.0, 0.0)));
        for (int i = 0; i <= 2 * totlen; i++)
        {
            if (const auto *d = M.find((i - totlen) * basicstep))
            {
                out[static_cast<std::size_t>(i)] = *d;
            }
        }
        if (M.diagonal_count() > static_cast<std::size_t>(2 * totlen + 1))
        {
            throw std::logic_error("merged transform has diagonals outside its layout");
        }
        return out;
    }
    // index i holds the diagonal of offset i * basicstep, i = 0 .. totlen, with basicstep * (totlen + 1) == n
    inline DiagonalSet rotated_layout(const DiagonalMatrix &M, int totlen, int basicstep)
    {
        if (static_cast<long>(basicstep) * (totlen + 1) != M.n())
        {
            throw std::logic_error("rotated layout needs basicstep * (totlen + 1) == n");
        }
        DiagonalSet out(static_cast<std::size_t>(totlen + 1), std::vector<cplx>(static_cast<std::size_t>(M.n()), cplx(0.0, 0.0)));
        for (int i = 0; i <= totlen; i++)
        {
            if (const auto *d = M.find(i * basicstep))
            {
                out[static_cast<std::size_t>(i)] = *d;
            }
        }
        if (M.diagonal_count() > static_cast<std::size_t>(totlen + 1))
        {
            throw std::logic_error("merged transform has diagonals outside its layout");
        }
        return out;
    }

    // the six sets of the full-slot (logn == logNh) level-3 bootstrapping
    struct LevelThreeDiagonals
    {
        DiagonalSet fftcoeff1, fftcoeff2, fftcoeff3;          // slot-to-coefficient, applied 1, 2, 3
        DiagonalSet invfftcoeff1, invfftcoeff2, invfftcoeff3; // coefficient-to-slot, applied 1, 2, 3
    };

    struct LevelThreeSplit
    {
        int part[3];      // stages per group, in order of application
        int totlen[3];
        int basicstep[3];
    };
    // genfftcoeff_3's split (Bootstrapper.cpp:1159-1170): the LAST group gets floor(logn / 3) stages
    inline LevelThreeSplit forward_split(int logn)
    {
        LevelThreeSplit s;
        s.part[2] = static_cast<int>(std::floor(logn / 3.0));
        s.part[1] = static_cast<int>(std::floor((logn - s.part[2]) / 2.0));
        s.part[0] = logn - s.part[2] - s.part[1];
        s.basicstep[0] = 1;
        s.basicstep[1] = 1 << s.part[0];
        s.basicstep[2] = 1 << (s.part[0] + s.part[1]);
        for (int i = 0; i < 3; i++)
        {
            s.totlen[i] = (1 << s.part[i]) - 1;
        }
        return s;
    }
    // geninvfftcoeff_3's split (:1567-1578): the FIRST group gets floor(logn / 3) stages
    inline LevelThreeSplit inverse_split(int logn)
    {
        LevelThreeSplit s;
        s.part[0] = static_cast<int>(std::floor(logn / 3.0));
        s.part[1] = static_cast<int>(std::floor((logn - s.part[0]) / 2.0));
        s.part[2] = logn - s.part[0] - s.part[1];
        s.basicstep[0] = 1 << (logn - s.part[0]);
        s.basicstep[1] = 1 << (logn - s.part[0] - s.part[1]);
        s.basicstep[2] = 1;
        for (int i = 0; i < 3; i++)
        {
            s.totlen[i] = (1 << s.part[i]) - 1;
        }
        return s;
    }

    inline LevelThreeDiagonals level_three_diagonals(int logn, long boundary_K)
    {
        if (logn < 3)
        {
            throw std::invalid_argument("the level-3 split needs at least three stages");
        }
        LevelThreeDiagonals out;
        {
            const LevelThreeSplit f = forward_split(logn);
            DiagonalMatrix g1 = merge_stages(logn, 0, f.part[0], special_fft_stage);
            DiagonalMatrix g2 = merge_stages(logn, f.part[0], f.part[1], special_fft_stage);
            DiagonalMatrix g3 = merge_stages(logn, f.part[0] + f.part[1], f.part[2], special_fft_stage);
            out.fftcoeff1 = centred_layout(g1, f.totlen[0], f.basicstep[0]);
            out.fftcoeff2 = centred_layout(g2, f.totlen[1], f.basicstep[1]);
            out.fftcoeff3 = rotated_layout(g3, f.totlen[2], f.basicstep[2]);
        }
        {
            const LevelThreeSplit v = inverse_split(logn);
            DiagonalMatrix g1 = merge_stages(logn, 0, v.part[0], special_ifft_stage);
            DiagonalMatrix g2 = merge_stages(logn, v.part[0], v.part[1], special_ifft_stage);
            DiagonalMatrix g3 = merge_stages(logn, v.part[0] + v.part[1], v.part[2], special_ifft_stage);
            g1.scale(1.0 / boundary_K); // :1684-1687
            g3.scale(0.5);              // :1689-1692
            out.invfftcoeff1 = rotated_layout(g1, v.totlen[0], v.basicstep[0]);
            out.invfftcoeff2 = centred_layout(g2, v.totlen[1], v.basicstep[1]);
            out.invfftcoeff3 = centred_layout(g3, v.totlen[2], v.basicstep[2]);
        }
        return out;
    }
} // namespace moai_boot
