// seal/moai_bootstrap_eval.h -- the evaluation half of MOAI's bootstrapping (Bootstrapper::bootstrap_full_3)
// on PACKED ciphertexts: modulus raise, coefficient-to-slot, the modular reduction's
// Chebyshev polynomial in baby-step / giant-step form with its double-angle steps, slot-to-coefficient.
//
// What it follows, call for call:
//   Bootstrapper::bootstrap_full_3        include/source/bootstrapping/Bootstrapper.cpp:3231-3251
//   ::modraise_inplace                    :2938-2992   (moai_modraise)
//   ::coefftoslot_full_3 / sflinv_full_3  :2742-2759, :2602-2623
//   ::slottocoeff_full_3 / sfl_full_3     :2760-2777, :2460-2497
//   ModularReducer::modular_reduction     ModularReducer.cpp:58-78 (the inverse_deg == 1 branch MOAI configures)
//   Polynomial::generate_poly_heap        common/Polynomial.cpp:168-214, babycount common/func.cpp:120-142
//   Polynomial::homomorphic_poly_evaluation   common/Polynomial.cpp:255-520
// Every evaluator call of those routines is issued here in the same order with the same arguments (a multiply_const
// that is followed by rescale_to_next_inplace goes out as Evaluator::multiply_const_rescale, one pass, same result), on a
// ciphertext that carries a whole batch (seal::Ciphertext::batch()), so each call is ONE batched device call;
// the three transforms of each linear part go through BsgsLinearTransform (cached diagonals).
//
// What it does NOT reproduce: the constants.  The reference derives the polynomial with a multi-precision
// Remez iteration and the transform diagonals with NTL-backed setup code; NTL is not in this image, so the
// reference's Bootstrapper cannot be compiled here and this file is "parity unpinned" against it.  The
// constants are inputs: Chebyshev coefficients (chebyshev_interpolant() gives a double-precision stand-in that
// approximates the same function) and the six diagonal sets.  What the tests pin instead
// (tests/cpp/test_bootstrap_lt.cpp): the heap decomposition evaluates to the polynomial it was built from;
// the homomorphic evaluation decrypts to that polynomial's values; a packed run is bit-identical to the same
// calls made per ciphertext.
#pragma once
#include <cmath>
#include <complex>
#include <functional>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "seal/moai_bootstrap_lt.h"
#include "seal/seal.h"

namespace moai_fused
{
    // common/func.cpp:120-142
    inline void babycount(long &mink, long &minm, long deg)
    {
        int curr_mul, min_mul;
        mink = 2;
        double d_over_k = static_cast<double>(deg) / mink;
        int log2_d_over_k = static_cast<int>(std::ceil(std::log2(d_over_k)));
        int ceil_d_over_k = static_cast<int>(std::ceil(d_over_k));
        min_mul = log2_d_over_k + static_cast<int>(mink) + ceil_d_over_k - 3;
        minm = log2_d_over_k;
        for (int i = 3; i < 2 * std::sqrt(deg); i++)
        {
            d_over_k = static_cast<double>(deg) / i;
            log2_d_over_k = static_cast<int>(std::ceil(std::log2(d_over_k)));
            ceil_d_over_k = static_cast<int>(std::ceil(d_over_k));
            curr_mul = log2_d_over_k + i + ceil_d_over_k - 3;
            if (min_mul > curr_mul)
            {
                mink = i;
                min_mul = curr_mul;
                minm = log2_d_over_k;
            }
        }
    }

    // Chebyshev coefficients c[0..deg] of the interpolant of f on [-1, 1] at `nodes` Chebyshev points
    // (c[0] is the plain constant term, not the halved one).  Stand-in for Remez::generate_optimal_poly.
    inline std::vector<double> chebyshev_interpolant(const std::function<double(double)> &f, long deg, long nodes = 0)
    {
        const long M = nodes > deg ? nodes : deg + 1;
        const long double pi = 3.141592653589793238462643383279502884L;
        std::vector<long double> fx(static_cast<std::size_t>(M));
        for (long k = 0; k < M; k++)
        {
            fx[static_cast<std::size_t>(k)] = f(static_cast<double>(std::cos(pi * (k + 0.5L) / M)));
        }
        std::vector<double> c(static_cast<std::size_t>(deg + 1));
        for (long j = 0; j <= deg; j++)
        {
            long double s = 0;
            for (long k = 0; k < M; k++)
            {
                s += fx[static_cast<std::size_t>(k)] * std::cos(pi * j * (k + 0.5L) / M);
            }
            c[static_cast<std::size_t>(j)] = static_cast<double>((j == 0 ? 1.0L : 2.0L) * s / M);
        }
        return c;
    }

    // The heap of quotients and remainders of common/Polynomial.cpp:168-205: node j splits into
    // quotient 2(j+1)-1 and remainder 2(j+1) by T_{k 2^(m-1-depth)}; a node of smaller degree passes itself on
    // as the remainder.  The division runs in the Chebyshev basis (T_i T_d = (T_{i+d} + T_{|i-d|}) / 2), which
    // gives the same unique quotient and remainder as the reference's power-basis long division without its
    // need for multi-precision arithmetic.
    class ChebyshevHeap
    {
    public:
        struct Node
        {
            bool present = false;
            std::vector<double> cheb; // chebcoeff[0..deg]
            long deg() const
            {
                return static_cast<long>(cheb.size()) - 1;
            }
        };

        ChebyshevHeap() = default;
        explicit ChebyshevHeap(const std::vector<double> &chebcoeff) : root_(chebcoeff)
        {
            if (chebcoeff.size() < 2)
            {
                throw std::invalid_argument("polynomial of degree 0");
            }
            rebuild();
        }
        // Polynomial::constmul followed by generate_poly_heap (ModularReducer.cpp:45-46)
        void constmul(double constant)
        {
            for (auto &c : root_)
            {
                c *= constant;
            }
            rebuild();
        }
        long deg() const
        {
            return static_cast<long>(root_.size()) - 1;
        }
        long heap_k() const
        {
            return heap_k_;
        }
        long heap_m() const
        {
            return heap_m_;
        }
        const std::vector<Node> &nodes() const
        {
            return nodes_;
        }
        const std::vector<double> &chebcoeff() const
        {
            return root_;
        }
        // power-basis coefficient i of the root (the deg <= 3 branches of the evaluation use coeff[], not chebcoeff[])
        double power_coeff(long i) const
        {
            const long d = deg();
            const double c0 = root_[0], c1 = root_[1], c2 = d >= 2 ? root_[2] : 0, c3 = d >= 3 ? root_[3] : 0;
            switch (i) // T2 = 2x^2 - 1, T3 = 4x^3 - 3x
            {
            case 0:
                return c0 - c2;
            case 1:
                return c1 - 3 * c3;
            case 2:
                return 2 * c2;
            case 3:
                return 4 * c3;
            default:
                throw std::logic_error("power_coeff is for degree <= 3");
            }
        }
        static double cheb_value(const std::vector<double> &c, double x)
        {
            // Clenshaw
            double b1 = 0, b2 = 0;
            for (std::size_t j = c.size(); j-- > 1;)
            {
                double t = 2 * x * b1 - b2 + c[j];
                b2 = b1;
                b1 = t;
            }
            return x * b1 - b2 + c[0];
        }
        // value of the root polynomial
        double value(double x) const
        {
            return cheb_value(root_, x);
        }
        // value recombined from the leaves the way the homomorphic evaluation recombines them: checks the heap
        double heap_value(double x) const
        {
            if (deg() <= 3)
            {
                return value(x);
            }
            std::vector<double> v(nodes_.size(), 0.0);
            std::vector<bool> have(nodes_.size(), false);
            long first = (1L << heap_m_) - 1, last = (1L << (heap_m_ + 1)) - 1;
            for (long i = first; i < last; i++)
            {
                if (nodes_[static_cast<std::size_t>(i)].present)
                {
                    v[static_cast<std::size_t>(i)] = cheb_value(nodes_[static_cast<std::size_t>(i)].cheb, x);
                    have[static_cast<std::size_t>(i)] = true;
                }
            }
            long depth = heap_m_, g = heap_k_;
            while (depth != 0)
            {
                depth--;
                first = (1L << depth) - 1;
                last = (1L << (depth + 1)) - 1;
                const double tg = std::cos(static_cast<double>(g) * std::acos(x));
                for (long i = first; i < last; i++)
                {
                    if (nodes_[static_cast<std::size_t>(i)].present)
                    {
                        const std::size_t q = static_cast<std::size_t>(2 * (i + 1) - 1), r = q + 1;
                        v[static_cast<std::size_t>(i)] = have[q] ? v[q] * tg + v[r] : v[r];
                        have[static_cast<std::size_t>(i)] = true;
                    }
                }
                g *= 2;
            }
            return v[0];
        }

        // Polynomial::homomorphic_poly_evaluation, common/Polynomial.cpp:255-520
        void evaluate(const seal::Evaluator &evaluator, const seal::RelinKeys &relin_keys, seal::Ciphertext &rtn,
                      const seal::Ciphertext &cipher) const
        {
            using seal::Ciphertext;
            double zero = 1. / cipher.scale();
            const long d = deg();
            if (d == 1)
            {
                evaluator.multiply_const_rescale(cipher, power_coeff(1), rtn);
                evaluator.add_const(rtn, power_coeff(0), rtn);
                return;
            }
            else if (d == 2)
            {
                Ciphertext squared;
                evaluator.square(cipher, squared);
                evaluator.relinearize_inplace(squared, relin_keys);
                evaluator.rescale_to_next_inplace(squared);
                evaluator.multiply_const_inplace(squared, power_coeff(2));
                evaluator.rescale_to_next_inplace(squared);
                if (std::abs(power_coeff(1)) >= zero)
                {
                    evaluator.multiply_const_rescale(cipher, power_coeff(1), rtn);
                    evaluator.add_reduced_error(rtn, squared, rtn);
                }
                else
                {
                    rtn = squared;
                }
                evaluator.add_const_inplace(rtn, power_coeff(0));
                return;
            }
            else if (d == 3)
            {
                Ciphertext squared, cubic;
                evaluator.square(cipher, squared);
                evaluator.relinearize_inplace(squared, relin_keys);
                evaluator.rescale_to_next_inplace(squared);
                evaluator.multiply_const_rescale(cipher, power_coeff(3), cubic);
                evaluator.multiply_inplace_reduced_error(cubic, squared, relin_keys);
                evaluator.rescale_to_next_inplace(cubic);
                if (std::abs(power_coeff(1)) >= zero)
                {
                    evaluator.multiply_const_rescale(cipher, power_coeff(1), rtn);
                    evaluator.add_reduced_error(rtn, cubic, rtn);
                }
                else
                {
                    rtn = cubic;
                }
                if (std::abs(power_coeff(2)) >= zero)
                {
                    evaluator.multiply_const_inplace(squared, power_coeff(2));
                    evaluator.rescale_to_next_inplace(squared);
                    evaluator.add_reduced_error(rtn, squared, rtn);
                }
                evaluator.add_const_inplace(rtn, power_coeff(0));
                return;
            }

            const long heap_k = heap_k_, heap_m = heap_m_;
            std::vector<Ciphertext> baby(static_cast<std::size_t>(heap_k));
            std::vector<bool> babybool(static_cast<std::size_t>(heap_k), false);
            baby[1] = cipher;
            babybool[1] = true;
            for (long i = 2; i < heap_k; i *= 2) // :355-364
            {
                evaluator.square(baby[static_cast<std::size_t>(i / 2)], baby[static_cast<std::size_t>(i)]);
                evaluator.relinearize_inplace(baby[static_cast<std::size_t>(i)], relin_keys);
                evaluator.rescale_to_next_inplace(baby[static_cast<std::size_t>(i)]);
                evaluator.double_inplace(baby[static_cast<std::size_t>(i)]);
                evaluator.add_const(baby[static_cast<std::size_t>(i)], -1.0, baby[static_cast<std::size_t>(i)]);
                babybool[static_cast<std::size_t>(i)] = true;
            }
            long lpow2, res, diff;
            Ciphertext tmp;
            for (long i = 1; i < heap_k; i++) // :369-395
            {
                if (!babybool[static_cast<std::size_t>(i)])
                {
                    lpow2 = (1 << static_cast<int>(std::floor(std::log(i) / std::log(2))));
                    res = i - lpow2;
                    diff = std::abs(lpow2 - res);
                    auto &b = baby[static_cast<std::size_t>(i)];
                    evaluator.multiply_reduced_error(baby[static_cast<std::size_t>(lpow2)], baby[static_cast<std::size_t>(res)], relin_keys, b);
                    evaluator.rescale_to_next_inplace(b);
                    evaluator.double_inplace(b);
                    evaluator.sub_reduced_error(b, baby[static_cast<std::size_t>(diff)], b);
                    babybool[static_cast<std::size_t>(i)] = true;
                }
            }
            std::vector<Ciphertext> giant(static_cast<std::size_t>(heap_m));
            lpow2 = (1 << (static_cast<int>(std::ceil(std::log(heap_k) / std::log(2))) - 1)); // :401-403
            res = heap_k - lpow2;
            diff = std::abs(lpow2 - res);
            if (res == 0)
            {
                giant[0] = baby[static_cast<std::size_t>(lpow2)];
            }
            else if (diff == 0)
            {
                evaluator.square(baby[static_cast<std::size_t>(lpow2)], giant[0]);
                evaluator.relinearize_inplace(giant[0], relin_keys);
                evaluator.rescale_to_next_inplace(giant[0]);
                evaluator.double_inplace(giant[0]);
                evaluator.add_const(giant[0], -1.0, giant[0]);
            }
            else
            {
                evaluator.multiply_reduced_error(baby[static_cast<std::size_t>(lpow2)], baby[static_cast<std::size_t>(res)], relin_keys, giant[0]);
                evaluator.rescale_to_next_inplace(giant[0]);
                evaluator.double_inplace(giant[0]);
                evaluator.sub_reduced_error(giant[0], baby[static_cast<std::size_t>(diff)], giant[0]);
            }
            for (long i = 1; i < heap_m; i++) // :436-446
            {
                auto &g = giant[static_cast<std::size_t>(i)];
                evaluator.square(giant[static_cast<std::size_t>(i - 1)], g);
                evaluator.relinearize_inplace(g, relin_keys);
                evaluator.rescale_to_next_inplace(g);
                evaluator.double_inplace(g);
                evaluator.add_const_inplace(g, -1.0);
            }
            const std::size_t heaplen = (static_cast<std::size_t>(1) << (heap_m + 1)) - 1;
            std::vector<Ciphertext> cipherheap(heaplen);
            std::vector<bool> cipherheapbool(heaplen, false);
            long heapfirst = (1L << heap_m) - 1;
            long heaplast = (1L << (heap_m + 1)) - 1;
            zero = 1. / cipher.scale();
            for (long i = heapfirst; i < heaplast; i++) // :458-487
            {
                const Node &node = nodes_[static_cast<std::size_t>(i)];
                if (node.present)
                {
                    auto &h = cipherheap[static_cast<std::size_t>(i)];
                    cipherheapbool[static_cast<std::size_t>(i)] = true;
                    evaluator.multiply_const_rescale(baby[1], node.cheb[1], h);
                    if (!(std::abs(node.cheb[1]) <= zero))
                    {
                        evaluator.add_const_inplace(h, node.cheb[0]);
                    }
                    for (long j = 2; j <= node.deg(); j++)
                    {
                        if (std::abs(node.cheb[static_cast<std::size_t>(j)]) <= zero)
                        {
                            continue;
                        }
                        const Ciphertext &term = j < heap_k ? baby[static_cast<std::size_t>(j)] : giant[0];
                        if (evaluator.rides_on_rescale_of(h, term))
                        {
                            // the product lands on h's level: add_reduced_error would be a plain addition (same level: the scale
                            // of the new term is taken, evaluator.cpp:447-452), and it rides on the rescale
                            evaluator.multiply_const_rescale(term, node.cheb[static_cast<std::size_t>(j)], h, 0, &h);
                            continue;
                        }
                        evaluator.multiply_const_rescale(term, node.cheb[static_cast<std::size_t>(j)], tmp);
                        evaluator.add_reduced_error(h, tmp, h);
                    }
                }
            }
            long depth = heap_m;
            std::size_t gindex = 0;
            while (depth != 0) // :491-510
            {
                depth--;
                heapfirst = (1L << depth) - 1;
                heaplast = (1L << (depth + 1)) - 1;
                for (long i = heapfirst; i < heaplast; i++)
                {
                    if (nodes_[static_cast<std::size_t>(i)].present)
                    {
                        const std::size_t q = static_cast<std::size_t>(2 * (i + 1) - 1), r = q + 1;
                        auto &h = cipherheap[static_cast<std::size_t>(i)];
                        cipherheapbool[static_cast<std::size_t>(i)] = true;
                        if (!cipherheapbool[q])
                        {
                            h = cipherheap[r];
                        }
                        else
                        {
                            evaluator.multiply_reduced_error(cipherheap[q], giant[gindex], relin_keys, h);
                            if (evaluator.rides_on_rescale_of(cipherheap[r], h))
                            {
                                // rescale, then an addition at equal levels (which takes the second operand's scale): one pass,
                                // accumulated into the remainder's buffer, which nothing reads after this
                                const double keep = cipherheap[r].scale();
                                evaluator.rescale_to_next_add_inplace(h, cipherheap[r]);
                                cipherheap[r].scale() = keep;
                                h = std::move(cipherheap[r]);
                            }
                            else
                            {
                                evaluator.rescale_to_next_inplace(h);
                                evaluator.add_reduced_error(h, cipherheap[r], h);
                            }
                        }
                    }
                }
                gindex++;
            }
            rtn = cipherheap[0];
        }

    private:
        // p = q T_d + r with deg r = d - 1 (the reference's remainder always carries d coefficients)
        static void divide(const std::vector<double> &p, long d, std::vector<double> &q, std::vector<double> &r)
        {
            std::vector<long double> c(p.begin(), p.end());
            const long deg = static_cast<long>(p.size()) - 1;
            q.assign(static_cast<std::size_t>(deg - d + 1), 0.0);
            for (long j = deg - d; j >= 1; j--)
            {
                const long double top = c[static_cast<std::size_t>(j + d)];
                q[static_cast<std::size_t>(j)] = static_cast<double>(2 * top);
                c[static_cast<std::size_t>(std::labs(j - d))] -= top;
                c[static_cast<std::size_t>(j + d)] = 0;
            }
            q[0] = static_cast<double>(c[static_cast<std::size_t>(d)]);
            c[static_cast<std::size_t>(d)] = 0;
            r.assign(static_cast<std::size_t>(d), 0.0);
            for (long i = 0; i < d; i++)
            {
                r[static_cast<std::size_t>(i)] = static_cast<double>(c[static_cast<std::size_t>(i)]);
            }
        }
        void rebuild()
        {
            babycount(heap_k_, heap_m_, deg()); // :211-214
            const std::size_t heaplen = (static_cast<std::size_t>(1) << (heap_m_ + 1)) - 1;
            nodes_.assign(heaplen, Node());
            nodes_[0].present = true;
            nodes_[0].cheb = root_;
            long chebdeg = heap_k_ << heap_m_;
            for (long i = 0; i < heap_m_; i++) // :183-204
            {
                chebdeg >>= 1;
                const long first = (1L << i) - 1, last = (1L << (i + 1)) - 1;
                for (long j = first; j < last; j++)
                {
                    const Node &cur = nodes_[static_cast<std::size_t>(j)];
                    if (!cur.present)
                    {
                        continue;
                    }
                    Node &quo = nodes_[static_cast<std::size_t>(2 * (j + 1) - 1)];
                    Node &rem = nodes_[static_cast<std::size_t>(2 * (j + 1))];
                    if (cur.deg() < chebdeg)
                    {
                        rem.present = true;
                        rem.cheb = cur.cheb;
                    }
                    else
                    {
                        quo.present = rem.present = true;
                        divide(cur.cheb, chebdeg, quo.cheb, rem.cheb);
                    }
                }
            }
        }

        std::vector<double> root_;
        long heap_k_ = 0, heap_m_ = 0;
        std::vector<Node> nodes_;
    };

    // ModularReducer with inverse_deg == 1 (ModularReducer.cpp:26-32, 40-47, 58-71): the cosine polynomial is
    // pre-multiplied by scale_inverse_coeff^(1/2^r) and every double-angle step subtracts the squared-up constant.
    class ModularReducer3
    {
    public:
        ModularReducer3(const std::vector<double> &sin_cos_chebcoeff, double inverse_coeff1, long num_double_formula)
            : sin_cos_(sin_cos_chebcoeff), num_double_formula_(num_double_formula)
        {
            scale_inverse_coeff_ = inverse_coeff1;
            for (long i = 0; i < num_double_formula_; i++)
            {
                scale_inverse_coeff_ = std::sqrt(scale_inverse_coeff_);
            }
            sin_cos_.constmul(scale_inverse_coeff_);
        }
        const ChebyshevHeap &polynomial() const
        {
            return sin_cos_;
        }
        double scale_inverse_coeff() const
        {
            return scale_inverse_coeff_;
        }
        long num_double_formula() const
        {
            return num_double_formula_;
        }
        // what modular_reduction computes on a plain value (for tests)
        double value(double x) const
        {
            double v = sin_cos_.value(x), curr = scale_inverse_coeff_;
            for (long i = 0; i < num_double_formula_; i++)
            {
                curr = curr * curr;
                v = 2 * v * v - curr;
            }
            return v;
        }
        void double_angle_formula_scaled(const seal::Evaluator &evaluator, const seal::RelinKeys &relin_keys, seal::Ciphertext &cipher,
                                         double scale_coeff) const
        {
            evaluator.square_inplace(cipher);
            evaluator.relinearize_inplace(cipher, relin_keys);
            evaluator.rescale_to_next_inplace(cipher);
            evaluator.double_inplace(cipher);
            evaluator.add_const(cipher, -scale_coeff, cipher);
        }
        void modular_reduction(const seal::Evaluator &evaluator, const seal::RelinKeys &relin_keys, seal::Ciphertext &rtn,
                               const seal::Ciphertext &cipher) const
        {
            seal::Ciphertext tmp1 = cipher, tmp2;
            sin_cos_.evaluate(evaluator, relin_keys, tmp2, tmp1);
            double curr_scale = scale_inverse_coeff_;
            for (long i = 0; i < num_double_formula_; i++)
            {
                curr_scale = curr_scale * curr_scale;
                double_angle_formula_scaled(evaluator, relin_keys, tmp2, curr_scale);
            }
            rtn = tmp2;
        }

    private:
        ChebyshevHeap sin_cos_;
        double scale_inverse_coeff_ = 1;
        long num_double_formula_ = 0;
    };

    // Bootstrapper::addLeftRotKeys_Linear_to_vector_3, Bootstrapper.cpp:89-184 (appends the missing steps)
    inline void boot_rotation_steps_3(int logn, int logNh, std::vector<int> &gal_steps_vector)
    {
        const int Nh = 1 << logNh;
        int div_part1 = static_cast<int>(std::floor(logn / 3.0));
        int div_part2 = static_cast<int>(std::floor((logn - div_part1) / 2.0));
        int div_part3 = logn - div_part1 - div_part2;
        int totlen[3] = { (1 << div_part1) - 1, (1 << div_part2) - 1, (1 << div_part3) - 1 };
        int basicstep[3] = { 1 << (logn - div_part1), 1 << (logn - div_part1 - div_part2), 1 };
        int gs1_e = 0;
        int gs[3] = { giantstep(totlen[0] + 1), giantstep(2 * totlen[1] + 1), giantstep(2 * totlen[2] + 1) };
        if (logn != logNh)
        {
            gs1_e = giantstep(2 * totlen[0] + 1);
        }
        int basicstart[3], giantfirst[3], giantlast[3];
        for (int s = 0; s < 3; s++)
        {
            basicstart[s] = -totlen[s] + gs[s] * static_cast<int>(std::floor((totlen[s] + 0.0) / (gs[s] + 0.0)));
            giantfirst[s] = -static_cast<int>(std::floor((totlen[s] + 0.0) / (gs[s] + 0.0)));
            giantlast[s] = static_cast<int>(std::floor((2 * totlen[s] + 0.0) / (gs[s] + 0.0))) + giantfirst[s];
        }
        int giantlast1_e = logn != logNh ? static_cast<int>(std::floor((totlen[0] + 0.0) / (gs[0] + 0.0))) : 0;
        auto add = [&](int step) {
            if (std::find(gal_steps_vector.begin(), gal_steps_vector.end(), step) == gal_steps_vector.end())
            {
                gal_steps_vector.push_back(step);
            }
        };
        auto babies = [&](int s) {
            for (int i = basicstart[s]; i < basicstart[s] + gs[s]; i++)
            {
                if (i != 0)
                {
                    add((Nh + i * basicstep[s]) % Nh);
                }
            }
        };
        auto giants = [&](int s) {
            for (int i = giantfirst[s]; i <= giantlast[s]; i++)
            {
                if (i != 0)
                {
                    add((Nh + i * gs[s] * basicstep[s]) % Nh);
                }
            }
        };
        babies(0);
        for (int i = 1; i < gs1_e; i++)
        {
            add(i * basicstep[0]);
        }
        babies(1);
        babies(2);
        giants(0);
        for (int i = 1; i <= giantlast1_e; i++)
        {
            add(i * gs1_e * basicstep[0]);
        }
        giants(1);
        giants(2);
    }

    // The diagonals of the six transforms, in the reference's layout (Bootstrapper.h: fftcoeff1..3 and
    // invfftcoeff1..3 of one slot index): 2 totlen + 1 diagonals for a plain transform, the first
    // totlen + 1 are used by a rotated one.
    struct BootDiagonals3
    {
        std::vector<std::vector<std::complex<double>>> invfftcoeff1, invfftcoeff2, invfftcoeff3; // coefficient-to-slot
        std::vector<std::vector<std::complex<double>>> fftcoeff1, fftcoeff2, fftcoeff3;          // slot-to-coefficient
    };

    // Bootstrapper::bootstrap_3 for logn == logNh (bootstrap_full_3) on packed ciphertexts.  Like the reference's
    // Bootstrapper it keeps references to the encoder, evaluator and keys: they must outlive it.
    class PackedBootstrapper3
    {
    public:
        PackedBootstrapper3(const seal::SEALContext &context, const seal::CKKSEncoder &encoder, const seal::Evaluator &evaluator,
                            const seal::RelinKeys &relin_keys, const seal::GaloisKeys &gal_keys, int logn, int logNh,
                            double final_scale, const BootDiagonals3 &diagonals, const ModularReducer3 &mod_reducer)
            : context_(context), encoder_(encoder), evaluator_(evaluator), relin_keys_(relin_keys), gal_keys_(gal_keys), logn_(logn),
              Nh_(1 << logNh), n_(1 << logn), final_scale_(final_scale), fftcoeff3_(diagonals.fftcoeff3), mod_reducer_(mod_reducer)
        {
            if (logn != logNh)
            {
                throw std::invalid_argument("bootstrap_full_3 is the logn == logNh case");
            }
            // sflinv_full_3's split, Bootstrapper.cpp:2603-2613
            {
                int p1 = static_cast<int>(std::floor(logn / 3.0)), p2 = static_cast<int>(std::floor((logn - p1) / 2.0));
                int p3 = logn - p1 - p2;
                inv_[0].reset(new BsgsLinearTransform(context, Nh_, (1 << p1) - 1, 1 << (logn - p1), logn, diagonals.invfftcoeff1, true));
                inv_[1].reset(new BsgsLinearTransform(context, Nh_, (1 << p2) - 1, 1 << (logn - p1 - p2), logn, diagonals.invfftcoeff2, false));
                inv_[2].reset(new BsgsLinearTransform(context, Nh_, (1 << p3) - 1, 1, logn, diagonals.invfftcoeff3, false));
            }
            // sfl_full_3's split, :2461-2471
            {
                int p3 = static_cast<int>(std::floor(logn / 3.0)), p2 = static_cast<int>(std::floor((logn - p3) / 2.0));
                int p1 = logn - p3 - p2;
                fwd_totlen2_ = (1 << p2) - 1;
                fwd_totlen3_ = (1 << p3) - 1;
                fwd_basicstep3_ = 1 << (p1 + p2);
                fwd_[0].reset(new BsgsLinearTransform(context, Nh_, (1 << p1) - 1, 1, logn, diagonals.fftcoeff1, false));
                fwd_[1].reset(new BsgsLinearTransform(context, Nh_, fwd_totlen2_, 1 << p1, logn, diagonals.fftcoeff2, false));
            }
        }

        // :2938-2992; the ciphertext must sit at the lowest level
        void modraise_inplace(seal::Ciphertext &cipher) const
        {
            using namespace seal;
            if (cipher.size() != 2)
            {
                throw std::invalid_argument("Ciphertexts of size 2 are supported only!");
            }
            if (cipher.coeff_modulus_size() != 1)
            {
                throw std::invalid_argument("Ciphertexts in the lowest level are supported only!");
            }
            if (!cipher.is_ntt_form())
            {
                evaluator_.transform_to_ntt_inplace(cipher); // moai_modraise reads NTT form; the round trip is exact
            }
            Ciphertext raised;
            raised.resize_batch(context_, context_.first_parms_id(), 2, cipher.batch());
            util::hip_check(moai_modraise(context_.device(), cipher.device_data(), raised.device_data(), raised.coeff_modulus_size(),
                                          cipher.batch(), context_.stream()));
            raised.is_ntt_form() = true;
            raised.scale() = cipher.scale();
            cipher = std::move(raised);
        }

        // :2602-2623
        void sflinv_full_3(seal::Ciphertext &rtncipher, const seal::Ciphertext &cipher)
        {
            seal::Ciphertext tmpct, tmpct2;
            inv_[0]->apply(cipher, tmpct, gal_keys_);
            evaluator_.rescale_to_next_inplace(tmpct);
            inv_[1]->apply(tmpct, tmpct2, gal_keys_);
            evaluator_.rescale_to_next_inplace(tmpct2);
            inv_[2]->apply(tmpct2, rtncipher, gal_keys_);
            evaluator_.rescale_to_next_inplace(rtncipher);
        }

        // :2460-2497
        void sfl_full_3(seal::Ciphertext &rtncipher, const seal::Ciphertext &cipher)
        {
            using namespace seal;
            Ciphertext tmpct, tmpct2;
            fwd_[0]->apply(cipher, tmpct, gal_keys_);
            evaluator_.rescale_to_next_inplace(tmpct);
            fwd_[1]->apply(tmpct, tmpct2, gal_keys_);
            evaluator_.rescale_to_next_inplace(tmpct2);

            const auto &modulus = context_.first_context_data()->parms().coeff_modulus();
            auto curr_level = context_.get_context_data(tmpct2.parms_id())->chain_index();
            double mod_zero = static_cast<double>(modulus[0].value());
            double curr_mod = static_cast<double>(modulus[curr_level].value());
            // the third set is rescaled by a factor that depends on the running scale; it is the same for every
            // ciphertext that went through the same pipeline, so the scaled transform is kept per factor
            auto key = std::make_tuple(curr_mod, tmpct2.scale(), initial_scale_);
            auto it = fwd3_.find(key);
            if (it == fwd3_.end())
            {
                // the reference fills totlen2 + 1 entries of an array of 2 totlen3 + 1 and the rotated transform reads
                // totlen3 + 1 of them: defined only when the last two parts of the split are equal (logn = 15: 5 + 5 + 5)
                if (fwd_totlen2_ != fwd_totlen3_)
                {
                    throw std::invalid_argument("sfl_full_3 is undefined in the reference for this logn");
                }
                std::vector<std::vector<std::complex<double>>> scaled(static_cast<std::size_t>(fwd_totlen3_ + 1));
                for (int i = 0; i < fwd_totlen2_ + 1; i++)
                {
                    scaled[static_cast<std::size_t>(i)].resize(static_cast<std::size_t>(n_));
                    for (int j = 0; j < n_; j++)
                    {
                        scaled[static_cast<std::size_t>(i)][static_cast<std::size_t>(j)] =
                            fftcoeff3_.at(static_cast<std::size_t>(i)).at(static_cast<std::size_t>(j)) * curr_mod * mod_zero * final_scale_ /
                            (tmpct2.scale() * tmpct2.scale() * initial_scale_);
                    }
                }
                it = fwd3_.emplace(key, std::unique_ptr<BsgsLinearTransform>(new BsgsLinearTransform(context_, Nh_, fwd_totlen3_, fwd_basicstep3_,
                                                                                                    logn_, scaled, true)))
                         .first;
            }
            it->second->apply(tmpct2, rtncipher, gal_keys_);
            evaluator_.rescale_to_next_inplace(rtncipher);
        }

        // :2742-2759
        void coefftoslot_full_3(seal::Ciphertext &rtncipher1, seal::Ciphertext &rtncipher2, const seal::Ciphertext &cipher)
        {
            using namespace seal;
            Ciphertext tmpct1, tmpct2, tmpct3, tmpct4;
            sflinv_full_3(tmpct1, cipher);
            std::complex<double> iunit(0.0, 1.0);
            std::vector<std::complex<double>> tmpvec(static_cast<std::size_t>(Nh_), 0);
            for (auto &z : tmpvec)
            {
                z -= iunit;
            }
            Plaintext tmpplain;
            encoder_.encode(tmpvec, 1.0, tmpplain);
            evaluator_.mod_switch_to_inplace(tmpplain, tmpct1.parms_id());
            evaluator_.multiply_plain(tmpct1, tmpplain, tmpct2);
            evaluator_.complex_conjugate(tmpct2, gal_keys_, tmpct3);
            evaluator_.complex_conjugate(tmpct1, gal_keys_, tmpct4);
            evaluator_.add_reduced_error(tmpct1, tmpct4, rtncipher1);
            evaluator_.add_reduced_error(tmpct2, tmpct3, rtncipher2);
        }

        // :2760-2777
        void slottocoeff_full_3(seal::Ciphertext &rtncipher, const seal::Ciphertext &cipher1, const seal::Ciphertext &cipher2)
        {
            using namespace seal;
            Ciphertext tmpct1, tmpct3;
            std::complex<double> iunit(0.0, 1.0);
            std::vector<std::complex<double>> tmpvec(static_cast<std::size_t>(Nh_), 0);
            for (auto &z : tmpvec)
            {
                z += iunit;
            }
            Plaintext tmpplain;
            encoder_.encode(tmpvec, 1.0, tmpplain);
            evaluator_.mod_switch_to_inplace(tmpplain, cipher2.parms_id());
            evaluator_.multiply_plain(cipher2, tmpplain, tmpct1);
            evaluator_.add_reduced_error(cipher1, tmpct1, tmpct3);
            sfl_full_3(rtncipher, tmpct3);
        }

        // bootstrap_3 (:3496-3502) -> bootstrap_full_3 (:3231-3251); `cipher` is consumed like the reference's
        void bootstrap_3(seal::Ciphertext &rtncipher, seal::Ciphertext &cipher)
        {
            using namespace seal;
            initial_scale_ = cipher.scale();
            modraise_inplace(cipher);
            const auto &modulus = context_.first_context_data()->parms().coeff_modulus();
            cipher.scale() = static_cast<double>(modulus[0].value());
            Ciphertext rtn1, rtn2;
            coefftoslot_full_3(rtn1, rtn2, cipher);
            Ciphertext modrtn1, modrtn2;
            mod_reducer_.modular_reduction(evaluator_, relin_keys_, modrtn1, rtn1);
            mod_reducer_.modular_reduction(evaluator_, relin_keys_, modrtn2, rtn2);
            slottocoeff_full_3(rtncipher, modrtn1, modrtn2);
            rtncipher.scale() = final_scale_;
        }

        double &initial_scale()
        {
            return initial_scale_;
        }

    private:
        seal::SEALContext context_;
        const seal::CKKSEncoder &encoder_;
        const seal::Evaluator &evaluator_;
        const seal::RelinKeys &relin_keys_;
        const seal::GaloisKeys &gal_keys_;
        int logn_, Nh_, n_;
        double final_scale_, initial_scale_ = 1;
        std::vector<std::vector<std::complex<double>>> fftcoeff3_; // rescaled per running scale in sfl_full_3; the other sets live in their transforms
        ModularReducer3 mod_reducer_;
        std::unique_ptr<BsgsLinearTransform> inv_[3], fwd_[2];
        int fwd_totlen2_ = 0, fwd_totlen3_ = 0, fwd_basicstep3_ = 1;
        std::map<std::tuple<double, double, double>, std::unique_ptr<BsgsLinearTransform>> fwd3_;
    };
} // namespace moai_fused
