// bootstrapping/ModularReducer.h -- drop-in for MOAI's include/source/bootstrapping/ModularReducer.h (and the
// boot::Polynomial of common/Polynomial.h as far as the reducer uses it), without NTL.
//
// Same class name, constructor and methods as the reference (ModularReducer.h:15-56, ModularReducer.cpp:3-78):
//   ModularReducer(boundary_K, log_width, deg, num_double_formula, inverse_deg, context, encoder, encryptor,
//                  evaluator, relin_keys, decryptor)
//   generate_sin_cos_polynomial(), generate_inverse_sine_polynomial(), modular_reduction(rtn, cipher),
//   double_angle_formula(cipher), double_angle_formula_scaled(cipher, scale_coeff), write_polynomials()
// What differs inside: the minimax polynomials come from bootstrapping/moai_remez.h (binary128 exchange iteration
// instead of 1000-bit NTL; same unique polynomial, see that file) and the polynomial evaluation is
// moai_fused::ChebyshevHeap::evaluate (seal/moai_bootstrap_eval.h), which issues the evaluator calls of
// Polynomial::homomorphic_poly_evaluation (common/Polynomial.cpp:255-520) in its order; all of them run on the
// device and accept packed ciphertexts.  Supported: inverse_deg 1 (MOAI's configuration: the inverse sine folded
// into the cosine's scale, ModularReducer.cpp:40-47, 62-69) and 2..3 (direct evaluation); above that the reference
// switches to an odd-polynomial heap (generate_poly_heap_odd) that is not provided.
#pragma once
#include <cmath>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <vector>

#include "moai_remez.h"
#include "seal/moai_bootstrap_eval.h"
#include "seal/seal.h"

namespace boot
{
    // Chebyshev-basis polynomial with the quotient / remainder heap of the baby-step / giant-step evaluation
    // (common/Polynomial.h:20-55).  Coefficients are doubles: that is what the reference's evaluation reads
    // (`to_double(chebcoeff[j])`, common/Polynomial.cpp:271-480).
    class Polynomial
    {
    public:
        long deg = -1, heap_k = 0, heap_m = 0, heaplen = 0;
        std::vector<double> chebcoeff;

        Polynomial() = default;
        void set_polynomial(long _deg, const std::vector<double> &cheb)
        {
            if (_deg < 1 || cheb.size() != static_cast<std::size_t>(_deg + 1))
            {
                throw std::invalid_argument("invalid polynomial");
            }
            deg = _deg;
            chebcoeff = cheb;
            heap_.reset();
        }
        // coeff[i] of the power basis for degree <= 3 (the only degrees whose power coefficients the reducer reads)
        double power_coeff(long i) const
        {
            if (deg < 1 || deg > 3 || i < 0 || i > deg)
            {
                throw std::logic_error("power_coeff is for degree <= 3");
            }
            const double c0 = chebcoeff[0], c1 = chebcoeff[1], c2 = deg >= 2 ? chebcoeff[2] : 0, c3 = deg >= 3 ? chebcoeff[3] : 0;
            switch (i) // T2 = 2x^2 - 1, T3 = 4x^3 - 3x
            {
            case 0:
                return c0 - c2;
            case 1:
                return c1 - 3 * c3;
            case 2:
                return 2 * c2;
            default:
                return 4 * c3;
            }
        }
        void constmul(double constant)
        {
            for (auto &c : chebcoeff)
            {
                c *= constant;
            }
            heap_.reset();
        }
        void generate_poly_heap()
        {
            heap_.reset(new moai_fused::ChebyshevHeap(chebcoeff));
            heap_k = heap_->heap_k();
            heap_m = heap_->heap_m();
            heaplen = (1L << (heap_m + 1)) - 1;
        }
        double evaluate(double value) const
        {
            return moai_fused::ChebyshevHeap::cheb_value(chebcoeff, value);
        }
        const moai_fused::ChebyshevHeap &heap()
        {
            if (!heap_)
            {
                generate_poly_heap();
            }
            return *heap_;
        }
        void homomorphic_poly_evaluation(seal::SEALContext &, seal::CKKSEncoder &, seal::Encryptor &, seal::Evaluator &evaluator,
                                         seal::RelinKeys &relin_keys, seal::Ciphertext &rtn, seal::Ciphertext &cipher, seal::Decryptor &)
        {
            heap().evaluate(evaluator, relin_keys, rtn, cipher);
        }
        void write_heap_to_file(std::ofstream &out)
        {
            out.precision(17);
            out << deg << "\n";
            for (double c : chebcoeff)
            {
                out << c << "\n";
            }
        }

    private:
        std::shared_ptr<moai_fused::ChebyshevHeap> heap_;
    };
} // namespace boot

class ModularReducer
{
public:
    long boundary_K;
    double log_width;
    long deg;
    long num_double_formula;

    double inverse_log_width;
    long inverse_deg;

    double scale_inverse_coeff = 1.0;

    seal::SEALContext &context;
    seal::CKKSEncoder &encoder;
    seal::Encryptor &encryptor;
    seal::Evaluator &evaluator;
    seal::RelinKeys &relin_keys;
    seal::Decryptor &decryptor;

    boot::Polynomial sin_cos_polynomial;
    boot::Polynomial inverse_sin_polynomial;
    // the levelled errors of the two fits (diagnostics; the reference prints nothing comparable)
    double sin_cos_minimax_error = 0, inverse_sin_minimax_error = 0;

    ModularReducer(long _boundary_K, double _log_width, long _deg, long _num_double_formula, long _inverse_deg,
                   seal::SEALContext &_context, seal::CKKSEncoder &_encoder, seal::Encryptor &_encryptor, seal::Evaluator &_evaluator,
                   seal::RelinKeys &_relin_keys, seal::Decryptor &_decryptor)
        : boundary_K(_boundary_K), log_width(_log_width), deg(_deg), num_double_formula(_num_double_formula), inverse_deg(_inverse_deg),
          context(_context), encoder(_encoder), encryptor(_encryptor), evaluator(_evaluator), relin_keys(_relin_keys),
          decryptor(_decryptor)
    {
        inverse_log_width = -std::log2(std::sin(2 * M_PI * std::pow(2.0, -log_width))); // ModularReducer.cpp:10
    }

    // ModularReducer.cpp:18-24
    void double_angle_formula(seal::Ciphertext &cipher)
    {
        evaluator.square_inplace(cipher);
        evaluator.relinearize_inplace(cipher, relin_keys);
        evaluator.rescale_to_next_inplace(cipher);
        evaluator.double_inplace(cipher);
        evaluator.add_const(cipher, -1.0, cipher);
    }
    // :26-32
    void double_angle_formula_scaled(seal::Ciphertext &cipher, double scale_coeff)
    {
        evaluator.square_inplace(cipher);
        evaluator.relinearize_inplace(cipher, relin_keys);
        evaluator.rescale_to_next_inplace(cipher);
        evaluator.double_inplace(cipher);
        evaluator.add_const(cipher, -scale_coeff, cipher);
    }
    // :34-37: RemezCos(rmparm, boundary_K, log_width, deg, 1 << num_double_formula) (ModularReducer.cpp:11)
    void generate_sin_cos_polynomial()
    {
        moai_boot::RemezResult r = moai_boot::remez_cos(boundary_K, log_width, deg, 1L << num_double_formula);
        sin_cos_minimax_error = static_cast<double>(r.error);
        sin_cos_polynomial.set_polynomial(deg, r.chebcoeff_double());
        sin_cos_polynomial.generate_poly_heap();
        reducer3_.reset();
    }
    // :39-48
    void generate_inverse_sine_polynomial()
    {
        if (inverse_deg > 3)
        {
            throw std::invalid_argument("inverse_deg > 3 (odd-polynomial heap) is not provided");
        }
        moai_boot::RemezResult r = moai_boot::remez_arcsin(inverse_log_width, inverse_deg);
        inverse_sin_minimax_error = static_cast<double>(r.error);
        inverse_sin_polynomial.set_polynomial(inverse_deg, r.chebcoeff_double());
        if (inverse_deg == 1)
        {
            if (sin_cos_polynomial.deg < 1)
            {
                throw std::logic_error("generate_sin_cos_polynomial() comes first");
            }
            unscaled_sin_cos_ = sin_cos_polynomial.chebcoeff;
            scale_inverse_coeff = inverse_sin_polynomial.power_coeff(1);
            for (int i = 0; i < num_double_formula; i++)
            {
                scale_inverse_coeff = std::sqrt(scale_inverse_coeff);
            }
            sin_cos_polynomial.constmul(scale_inverse_coeff);
            sin_cos_polynomial.generate_poly_heap();
        }
        reducer3_.reset();
    }
    // :50-56
    void write_polynomials()
    {
        std::ofstream sin_cos_out("cosine.txt"), inverse_out("inverse_sine.txt");
        sin_cos_polynomial.write_heap_to_file(sin_cos_out);
        inverse_sin_polynomial.write_heap_to_file(inverse_out);
    }
    // :58-78
    void modular_reduction(seal::Ciphertext &rtn, seal::Ciphertext &cipher)
    {
        seal::Ciphertext tmp1 = cipher, tmp2;
        sin_cos_polynomial.heap().evaluate(evaluator, relin_keys, tmp2, tmp1);
        if (inverse_deg == 1)
        {
            double curr_scale = scale_inverse_coeff;
            for (int i = 0; i < num_double_formula; i++)
            {
                curr_scale = curr_scale * curr_scale;
                double_angle_formula_scaled(tmp2, curr_scale);
            }
            rtn = tmp2;
        }
        else
        {
            for (int i = 0; i < num_double_formula; i++)
            {
                double_angle_formula(tmp2);
            }
            inverse_sin_polynomial.heap().evaluate(evaluator, relin_keys, rtn, tmp2);
        }
    }

    // the same reducer in the form the packed pipeline takes (seal/moai_bootstrap_eval.h); inverse_deg == 1 only
    const moai_fused::ModularReducer3 &packed_reducer()
    {
        if (inverse_deg != 1)
        {
            throw std::invalid_argument("the packed bootstrapping pipeline is the inverse_deg == 1 configuration");
        }
        if (unscaled_sin_cos_.empty())
        {
            throw std::logic_error("prepare_mod_polynomial() has not run");
        }
        if (!reducer3_)
        {
            reducer3_.reset(new moai_fused::ModularReducer3(unscaled_sin_cos_, inverse_sin_polynomial.power_coeff(1), num_double_formula));
        }
        return *reducer3_;
    }
    // what modular_reduction computes on a plain value (tests)
    double plain_value(double x) const
    {
        double v = sin_cos_polynomial.evaluate(x);
        if (inverse_deg == 1)
        {
            double curr = scale_inverse_coeff;
            for (int i = 0; i < num_double_formula; i++)
            {
                curr = curr * curr;
                v = 2 * v * v - curr;
            }
            return v;
        }
        for (int i = 0; i < num_double_formula; i++)
        {
            v = 2 * v * v - 1;
        }
        return inverse_sin_polynomial.evaluate(v);
    }

private:
    std::vector<double> unscaled_sin_cos_;
    std::unique_ptr<moai_fused::ModularReducer3> reducer3_;
};
