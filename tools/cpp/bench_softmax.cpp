// bench_softmax.cpp -- the softmax of one attention head as MOAI's 12-layer run calls it
// (softmax_boot, include/source/non_linear_func/softmax.hpp:307-580, from single_att_block.hpp:163 with iter = 16,
// input_num = 5, 128 ciphertexts at the chain index Q K^T leaves them at) at N = 2^16 on the 36-prime chain.
// softmax.hpp includes Bootstrapper.h and so cannot be compiled without NTL; this file issues the same evaluator
// calls in the same order -- exp (:9-27), the masks (:337-424), the sum and its bootstrap (:470-537), inverse (:29-52),
// the final products (:560-570) -- leaving out only the decrypt-and-print blocks, with the bootstrap from
// seal/moai_bootstrap_eval.h (stand-in constants, see there).  Timed twice:
//   per ciphertext   the calls as written, OpenMP loops over the 128 ciphertexts as in the reference
//   packed           the same calls on moai_fused::pack'ed ciphertexts where all 128 take the same operation
// and the two results are compared bit for bit.
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

// softmax.hpp:9-27
static Ciphertext exp_ct(const Ciphertext &x, CKKSEncoder &encoder, Evaluator &evaluator, const RelinKeys &relin_keys)
{
    Plaintext inverse_128;
    encoder.encode(0.0078125, x.parms_id(), x.scale(), inverse_128);
    Ciphertext output;
    evaluator.multiply_plain(x, inverse_128, output);
    evaluator.rescale_to_next_inplace(output);
    Plaintext one;
    encoder.encode(1.0, output.parms_id(), output.scale(), one);
    evaluator.add_plain_inplace(output, one);
    for (int i = 0; i < log2(128); ++i)
    {
        evaluator.square_inplace(output);
        evaluator.relinearize_inplace(output, relin_keys);
        evaluator.rescale_to_next_inplace(output);
    }
    return output;
}

// softmax.hpp:29-52
static Ciphertext inverse_ct(const Ciphertext &x, CKKSEncoder &encoder, Evaluator &evaluator, const RelinKeys &relin_keys, int iter)
{
    Plaintext one;
    encoder.encode(1.0, x.parms_id(), x.scale(), one);
    Ciphertext y;
    evaluator.sub_plain(x, one, y);
    evaluator.negate_inplace(y);
    Ciphertext tmp;
    evaluator.add_plain(y, one, tmp);
    Ciphertext res = tmp;
    for (int i = 0; i < iter; ++i)
    {
        evaluator.square_inplace(y);
        evaluator.relinearize_inplace(y, relin_keys);
        evaluator.rescale_to_next_inplace(y);
        encoder.encode(1.0, y.parms_id(), y.scale(), one);
        evaluator.add_plain(y, one, tmp);
        evaluator.mod_switch_to_inplace(res, tmp.parms_id());
        evaluator.multiply_inplace(res, tmp);
        evaluator.relinearize_inplace(res, relin_keys);
        evaluator.rescale_to_next_inplace(res);
    }
    return res;
}

// the plaintext vector ciphertext i is masked with (softmax.hpp:337-372 with value = minus_index, :388-424 with
// value = 1); empty when the ciphertext takes no vector (the branch `i > input_num && i <= num - input_num`)
static vector<double> mask_vector(int i, int num, int input_num, int num_batch, const vector<int> &bias_vec, double value)
{
    const int slot_count = (int)bias_vec.size();
    vector<double> v;
    if (i == 0)
    {
        v.assign(slot_count, 0);
        for (int s = 0; s < slot_count; ++s)
            if (bias_vec[s] == 1) v[s] = value;
    }
    else if (i > input_num && i <= (num - input_num))
    {
    }
    else if (i <= input_num)
    {
        v.assign(slot_count, 0);
        int index = num_batch * (input_num - i);
        for (int s = 0; s < slot_count; ++s)
            if (bias_vec[s] == 1 && s < index) v[s] = value;
    }
    else if (i > num - input_num)
    {
        v.assign(slot_count, 0);
        int index = (num - i) * num_batch;
        for (int s = 0; s < slot_count; ++s)
            if (bias_vec[s] == 1 && s >= index) v[s] = value;
    }
    return v;
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 16;
    const int start_index = argc > 2 ? atoi(argv[2]) : 13; // chain index of Q K^T's result (bench_attention)
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
    const int slot_count = (int)encoder.slot_count();
    vector<int> steps{ 0 };
    for (int i = 0; i < 15; i++) steps.push_back(1 << i);
    moai_fused::boot_rotation_steps_3(logn, logn, steps);
    double t0 = now_s();
    GaloisKeys gal_keys;
    keygen.create_galois_keys(steps, gal_keys);
    context.sync();
    printf("keys: %.1f s\n", now_s() - t0);

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
    const double two_pi = 2 * M_PI;
    moai_fused::ModularReducer3 reducer(
        moai_fused::chebyshev_interpolant([=](double t) { return cos(two_pi * (25 * t - 0.25) / 4.0); }, 59, 4 * 59), 1 / two_pi, 2);
    const double scale = pow(2.0, 46);
    moai_fused::PackedBootstrapper3 boot(context, encoder, evaluator, relin_keys, gal_keys, logn, logn, scale, dg, reducer);

    const int num = 128, input_num = 5, iter = 16, num_batch = slot_count / 128;
    const double minus_index = 7.5; // minus_index_vec[0], softmax.hpp:321
    vector<int> bias_vec(slot_count, 0);
    for (int s = 0; s < num_batch * input_num; ++s) bias_vec[s] = 1;
    vector<Ciphertext> enc_X(num);
    {
        vector<double> v(slot_count);
        for (auto &z : v) z = ud(rng);
        Plaintext pl;
        encoder.encode(v, scale, pl);
        Ciphertext c;
        encryptor.encrypt(pl, c);
        while (context.get_context_data(c.parms_id())->chain_index() != (size_t)start_index) evaluator.mod_switch_to_next_inplace(c);
        for (int i = 0; i < num; i++) enc_X[i] = c;
    }
    context.sync();

    // ---- per ciphertext, as written --------------------------------------------------------------------------------
    vector<Ciphertext> out_ref(num);
    double t_ref;
    {
        t0 = now_s();
        vector<Ciphertext> enc_x_minus(num), exp_x(num);
#pragma omp parallel for
        for (int i = 0; i < num; ++i)
        {
            enc_x_minus[i] = enc_X[i];
            vector<double> m = mask_vector(i, num, input_num, num_batch, bias_vec, minus_index);
            if (!m.empty())
            {
                Plaintext one;
                encoder.encode(m, enc_x_minus[i].scale(), one);
                evaluator.mod_switch_to_inplace(one, enc_x_minus[i].parms_id());
                evaluator.sub_plain_inplace(enc_x_minus[i], one);
            }
        }
#pragma omp parallel for
        for (int i = 0; i < num; ++i)
        {
            exp_x[i] = exp_ct(enc_x_minus[i], encoder, evaluator, relin_keys);
            vector<double> m = mask_vector(i, num, input_num, num_batch, bias_vec, 1.0);
            Plaintext one;
            if (m.empty())
                encoder.encode(0, exp_x[i].scale(), one);
            else
                encoder.encode(m, exp_x[i].scale(), one);
            evaluator.mod_switch_to_inplace(one, exp_x[i].parms_id());
            evaluator.multiply_plain_inplace(exp_x[i], one);
            evaluator.rescale_to_next_inplace(exp_x[i]);
            exp_x[i].scale() = scale;
        }
        Ciphertext sum_exp_x = exp_x[0];
        for (int i = 1; i < num; ++i) evaluator.add_inplace(sum_exp_x, exp_x[i]);
        Plaintext eps;
        encoder.encode(0.00001, sum_exp_x.parms_id(), sum_exp_x.scale(), eps);
        evaluator.add_plain_inplace(sum_exp_x, eps);
        sum_exp_x.scale() = scale;
        while (context.get_context_data(sum_exp_x.parms_id())->chain_index() != 0) evaluator.mod_switch_to_next_inplace(sum_exp_x);
        Ciphertext rtn;
        boot.bootstrap_3(rtn, sum_exp_x);
        while (context.get_context_data(rtn.parms_id())->chain_index() > (size_t)(iter + 1 + 3)) evaluator.mod_switch_to_next_inplace(rtn);
        Ciphertext inv_sum = inverse_ct(rtn, encoder, evaluator, relin_keys, iter);
        inv_sum.scale() = scale;
        if (context.get_context_data(exp_x[0].parms_id())->chain_index() < context.get_context_data(inv_sum.parms_id())->chain_index())
            evaluator.mod_switch_to_inplace(inv_sum, exp_x[0].parms_id());
#pragma omp parallel for
        for (int i = 0; i < num; ++i)
        {
            if (context.get_context_data(exp_x[i].parms_id())->chain_index() > context.get_context_data(inv_sum.parms_id())->chain_index())
                evaluator.mod_switch_to_inplace(exp_x[i], inv_sum.parms_id());
            evaluator.multiply(exp_x[i], inv_sum, out_ref[i]);
            evaluator.relinearize_inplace(out_ref[i], relin_keys);
            evaluator.rescale_to_next_inplace(out_ref[i]);
            out_ref[i].scale() = scale;
        }
        context.sync();
        t_ref = now_s() - t0;
    }
    printf("softmax_boot as written (%d OpenMP threads), 128 ciphertexts from chain index %d: %.3f s; result at chain index %zu\n", threads,
           start_index, t_ref, context.get_context_data(out_ref[0].parms_id())->chain_index());

    // ---- packed ----------------------------------------------------------------------------------------------------
    vector<Ciphertext> out_packed;
    double t_packed = 0;
    for (int rep = 0; rep < 2; rep++)
    {
        t0 = now_s();
        vector<Ciphertext> enc_x_minus(num);
        for (int i = 0; i < num; ++i)
        {
            enc_x_minus[i] = enc_X[i];
            vector<double> m = mask_vector(i, num, input_num, num_batch, bias_vec, minus_index);
            if (!m.empty())
            {
                Plaintext one;
                encoder.encode(m, enc_x_minus[i].scale(), one);
                evaluator.mod_switch_to_inplace(one, enc_x_minus[i].parms_id());
                evaluator.sub_plain_inplace(enc_x_minus[i], one);
            }
        }
        Ciphertext pack_exp = exp_ct(moai_fused::pack(enc_x_minus, context), encoder, evaluator, relin_keys);
        vector<Ciphertext> exp_x;
        moai_fused::unpack(pack_exp, context, exp_x);
        for (int i = 0; i < num; ++i)
        {
            vector<double> m = mask_vector(i, num, input_num, num_batch, bias_vec, 1.0);
            Plaintext one;
            if (m.empty())
                encoder.encode(0, exp_x[i].scale(), one);
            else
                encoder.encode(m, exp_x[i].scale(), one);
            evaluator.mod_switch_to_inplace(one, exp_x[i].parms_id());
            evaluator.multiply_plain_inplace(exp_x[i], one);
        }
        Ciphertext pack_masked = moai_fused::pack(exp_x, context);
        evaluator.rescale_to_next_inplace(pack_masked);
        pack_masked.scale() = scale;
        moai_fused::unpack(pack_masked, context, exp_x);
        Ciphertext sum_exp_x = exp_x[0];
        for (int i = 1; i < num; ++i) evaluator.add_inplace(sum_exp_x, exp_x[i]);
        Plaintext eps;
        encoder.encode(0.00001, sum_exp_x.parms_id(), sum_exp_x.scale(), eps);
        evaluator.add_plain_inplace(sum_exp_x, eps);
        sum_exp_x.scale() = scale;
        while (context.get_context_data(sum_exp_x.parms_id())->chain_index() != 0) evaluator.mod_switch_to_next_inplace(sum_exp_x);
        Ciphertext rtn;
        boot.bootstrap_3(rtn, sum_exp_x);
        while (context.get_context_data(rtn.parms_id())->chain_index() > (size_t)(iter + 1 + 3)) evaluator.mod_switch_to_next_inplace(rtn);
        Ciphertext inv_sum = inverse_ct(rtn, encoder, evaluator, relin_keys, iter);
        inv_sum.scale() = scale;
        if (context.get_context_data(pack_masked.parms_id())->chain_index() < context.get_context_data(inv_sum.parms_id())->chain_index())
            evaluator.mod_switch_to_inplace(inv_sum, pack_masked.parms_id());
        if (context.get_context_data(pack_masked.parms_id())->chain_index() > context.get_context_data(inv_sum.parms_id())->chain_index())
            evaluator.mod_switch_to_inplace(pack_masked, inv_sum.parms_id());
        Ciphertext pack_inv = moai_fused::pack(vector<Ciphertext>(num, inv_sum), context), pack_out;
        evaluator.multiply(pack_masked, pack_inv, pack_out);
        evaluator.relinearize_inplace(pack_out, relin_keys);
        evaluator.rescale_to_next_inplace(pack_out);
        pack_out.scale() = scale;
        moai_fused::unpack(pack_out, context, out_packed);
        context.sync();
        t_packed = now_s() - t0;
    }
    printf("the same calls on packed ciphertexts: %.3f s\n", t_packed);
    bool same = true;
    for (int i = 0; i < num; ++i)
        same = same && out_packed[i].parms_id() == out_ref[i].parms_id() && out_packed[i].scale() == out_ref[i].scale() &&
               out_packed[i].download() == out_ref[i].download();
    printf("results: %s\n", same ? "bit-identical" : "DIFFERENT");
    return same ? 0 : 1;
}
