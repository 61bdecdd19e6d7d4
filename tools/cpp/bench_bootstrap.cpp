// bench_bootstrap.cpp -- one whole Bootstrapper::bootstrap_3 (include/source/bootstrapping/Bootstrapper.cpp:3496,
// bootstrap_full_3 :3231-3251) at MOAI's parameters: N = 2^16, the 36-prime chain, logn = 15, degree-59 cosine with
// two double-angle steps and boundary K = 25 (include/test/test_full_scheme.hpp:345-368), the key list of
// addLeftRotKeys_Linear_to_vector_3 -- on a PACK of ciphertexts through moai_fused::PackedBootstrapper3, and, for
// comparison, the same sequence of evaluator calls made per ciphertext from OpenMP threads (the reference's way of
// calling it, test_full_scheme.hpp:654-660), with the two results compared bit for bit.
// The constants are stand-ins (random diagonals, a double-precision Chebyshev interpolant of the cosine): the
// reference's setup needs NTL; the timing depends on the shape of the computation only.
#include <omp.h>

#include <chrono>
#include <complex>
#include <cstdio>
#include <random>

#include "seal/moai_bootstrap_eval.h"
#include "seal/seal.h"

using namespace seal;
using namespace std;

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

#include "../../tests/cpp/ref_bootstrap_calls.h"

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 16;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    const int Bref = argc > 3 ? atoi(argv[3]) : min(B, threads);
    omp_set_num_threads(threads);
    EncryptionParameters parms(scheme_type::ckks);
    const size_t n = 65536;
    const int logn = 15;
    parms.set_poly_modulus_degree(n);
    vector<int> bits{ 51 };
    for (int i = 0; i < 20; i++) bits.push_back(46);
    for (int i = 0; i < 14; i++) bits.push_back(51);
    bits.push_back(58);
    parms.set_coeff_modulus(CoeffModulus::Create(n, bits));
    parms.set_secret_key_hamming_weight(192);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys relin_keys;
    keygen.create_relin_keys(relin_keys);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    const int Nh = (int)encoder.slot_count();
    vector<int> steps{ 0 };
    for (int i = 0; i < 15; i++) steps.push_back(1 << i);
    moai_fused::boot_rotation_steps_3(logn, logn, steps);
    double t0 = now_s();
    GaloisKeys gal_keys;
    keygen.create_galois_keys(steps, gal_keys);
    context.sync();
    printf("%zu Galois keys (test_full_scheme.hpp:436-443): %.1f s\n", steps.size(), now_s() - t0);

    mt19937_64 rng(3);
    uniform_real_distribution<double> ud(-1.0, 1.0);
    const int p = logn / 3, totlen = (1 << p) - 1, slotlen = 1 << logn;
    auto random_set = [&](int count) {
        vector<vector<complex<double>>> c(count, vector<complex<double>>(slotlen));
        for (auto &d : c)
            for (auto &z : d) z = { ud(rng) * 0.1, ud(rng) * 0.1 };
        return c;
    };
    moai_fused::BootDiagonals3 dg;
    dg.invfftcoeff1 = random_set(2 * totlen + 1);
    dg.invfftcoeff2 = random_set(2 * totlen + 1);
    dg.invfftcoeff3 = random_set(2 * totlen + 1);
    dg.fftcoeff1 = random_set(2 * totlen + 1);
    dg.fftcoeff2 = random_set(2 * totlen + 1);
    dg.fftcoeff3 = random_set(2 * totlen + 1);
    const long K = 25, deg = 59, r = 2;
    const double two_pi = 2 * M_PI;
    moai_fused::ModularReducer3 reducer(
        moai_fused::chebyshev_interpolant([=](double t) { return cos(two_pi * (K * t - 0.25) / 4.0); }, deg, 4 * deg), 1 / two_pi, r);
    const double scale = pow(2.0, 46);
    moai_fused::PackedBootstrapper3 boot(context, encoder, evaluator, relin_keys, gal_keys, logn, logn, scale, dg, reducer);

    vector<Ciphertext> in(B);
    {
        vector<complex<double>> v(Nh);
        for (auto &z : v) z = { ud(rng) * 0.01, ud(rng) * 0.01 };
        Plaintext pl;
        encoder.encode(v, scale, pl);
        Ciphertext c;
        encryptor.encrypt(pl, c);
        evaluator.mod_switch_to_inplace(c, context.last_parms_id());
        for (int b = 0; b < B; b++) in[b] = c;
    }
    context.sync();

    // ---- packed: the first call encodes and caches the diagonals, the second is the steady state --------------------
    Ciphertext packed_out;
    double t_first = 0, t_steady = 0;
    for (int rep = 0; rep < 2; rep++)
    {
        Ciphertext packed = moai_fused::pack(in, context);
        context.sync();
        t0 = now_s();
        boot.bootstrap_3(packed_out, packed);
        context.sync();
        (rep == 0 ? t_first : t_steady) = now_s() - t0;
    }
    printf("packed, %d ciphertexts: first call %.3f s (encodes the diagonals), then %.3f s = %.2f ms per bootstrap; chain index %zu -> %zu\n", B,
           t_first, t_steady, t_steady * 1e3 / B, context.first_context_data()->chain_index(),
           context.get_context_data(packed_out.parms_id())->chain_index());
    // stage split of the steady state
    {
        Ciphertext packed = moai_fused::pack(in, context), rtn1, rtn2, m1, m2, out;
        boot.initial_scale() = packed.scale();
        context.sync();
        t0 = now_s();
        boot.modraise_inplace(packed);
        packed.scale() = (double)context.first_context_data()->parms().coeff_modulus()[0].value();
        context.sync();
        double t_raise = now_s() - t0;
        t0 = now_s();
        boot.coefftoslot_full_3(rtn1, rtn2, packed);
        context.sync();
        double t_cts = now_s() - t0;
        t0 = now_s();
        reducer.modular_reduction(evaluator, relin_keys, m1, rtn1);
        reducer.modular_reduction(evaluator, relin_keys, m2, rtn2);
        context.sync();
        double t_mod = now_s() - t0;
        t0 = now_s();
        boot.slottocoeff_full_3(out, m1, m2);
        context.sync();
        double t_stc = now_s() - t0;
        printf("  per bootstrap: modraise %.2f ms, coefficient-to-slot %.2f ms, 2 modular reductions %.2f ms, slot-to-coefficient %.2f ms\n",
               t_raise * 1e3 / B, t_cts * 1e3 / B, t_mod * 1e3 / B, t_stc * 1e3 / B);
    }

    // ---- per ciphertext, as the reference calls it: one bootstrap per OpenMP thread ---------------------------------
    if (Bref > 0)
    {
        const auto &modulus = context.first_context_data()->parms().coeff_modulus();
        const int bs_inv[3] = { 1 << (logn - p), 1 << (logn - 2 * p), 1 }, bs_fwd[3] = { 1, 1 << p, 1 << (2 * p) };
        vector<Ciphertext> outr(Bref);
        t0 = now_s();
#pragma omp parallel for
        for (int b = 0; b < Bref; b++)
        {
            Ciphertext cipher = in[b];
            const double initial_scale = cipher.scale();
            boot.modraise_inplace(cipher);
            cipher.scale() = (double)modulus[0].value();
            Ciphertext t1, t2, t3, t4, rtn1, rtn2, a, c;
            ref_rotated_bsgs(evaluator, gal_keys, Nh, a, cipher, totlen, bs_inv[0], logn, dg.invfftcoeff1);
            evaluator.rescale_to_next_inplace(a);
            ref_bsgs(evaluator, gal_keys, Nh, c, a, totlen, bs_inv[1], logn, dg.invfftcoeff2);
            evaluator.rescale_to_next_inplace(c);
            ref_bsgs(evaluator, gal_keys, Nh, t1, c, totlen, bs_inv[2], logn, dg.invfftcoeff3);
            evaluator.rescale_to_next_inplace(t1);
            {
                vector<complex<double>> tmpvec(Nh, 0);
                for (auto &z : tmpvec) z -= complex<double>(0.0, 1.0);
                Plaintext tmpplain;
                encoder.encode(tmpvec, 1.0, tmpplain);
                evaluator.mod_switch_to_inplace(tmpplain, t1.parms_id());
                evaluator.multiply_plain(t1, tmpplain, t2);
            }
            evaluator.complex_conjugate(t2, gal_keys, t3);
            evaluator.complex_conjugate(t1, gal_keys, t4);
            evaluator.add_reduced_error(t1, t4, rtn1);
            evaluator.add_reduced_error(t2, t3, rtn2);
            Ciphertext m1, m2, s1, s3, f1, f2, want;
            reducer.modular_reduction(evaluator, relin_keys, m1, rtn1);
            reducer.modular_reduction(evaluator, relin_keys, m2, rtn2);
            {
                vector<complex<double>> tmpvec(Nh, 0);
                for (auto &z : tmpvec) z += complex<double>(0.0, 1.0);
                Plaintext tmpplain;
                encoder.encode(tmpvec, 1.0, tmpplain);
                evaluator.mod_switch_to_inplace(tmpplain, m2.parms_id());
                evaluator.multiply_plain(m2, tmpplain, s1);
            }
            evaluator.add_reduced_error(m1, s1, s3);
            ref_bsgs(evaluator, gal_keys, Nh, f1, s3, totlen, bs_fwd[0], logn, dg.fftcoeff1);
            evaluator.rescale_to_next_inplace(f1);
            ref_bsgs(evaluator, gal_keys, Nh, f2, f1, totlen, bs_fwd[1], logn, dg.fftcoeff2);
            evaluator.rescale_to_next_inplace(f2);
            {
                auto curr_level = context.get_context_data(f2.parms_id())->chain_index();
                double mod_zero = (double)modulus[0].value();
                double curr_mod = (double)modulus[curr_level].value();
                vector<vector<complex<double>>> fftcoeff3_scale(2 * totlen + 1);
                for (int i = 0; i < totlen + 1; i++) fftcoeff3_scale[i].resize(slotlen);
                for (int i = 0; i < totlen + 1; i++)
                    for (int j = 0; j < slotlen; j++)
                        fftcoeff3_scale[i][j] = dg.fftcoeff3[i][j] * curr_mod * mod_zero * scale / (f2.scale() * f2.scale() * initial_scale);
                ref_rotated_bsgs(evaluator, gal_keys, Nh, want, f2, totlen, bs_fwd[2], logn, fftcoeff3_scale);
            }
            evaluator.rescale_to_next_inplace(want);
            want.scale() = scale;
            outr[b] = want;
        }
        context.sync();
        double t_ref = now_s() - t0;
        printf("per ciphertext (the Bootstrapper's call sequence, %d OpenMP threads), %d bootstraps: %.3f s = %.2f ms per bootstrap\n", threads, Bref,
               t_ref, t_ref * 1e3 / Bref);
        vector<Ciphertext> got;
        moai_fused::unpack(packed_out, context, got);
        bool same = true;
        for (int b = 0; b < Bref; b++) same = same && outr[b].download() == got[b].download() && outr[b].scale() == got[b].scale();
        printf("results: %s\n", same ? "bit-identical" : "DIFFERENT");
        return same ? 0 : 1;
    }
    return 0;
}
