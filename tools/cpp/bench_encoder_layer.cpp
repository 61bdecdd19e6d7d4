// bench_encoder_layer.cpp -- ONE encoder layer of MOAI's 12-layer run (include/test/test_full_scheme.hpp:524-1095)
// with its data flow and levels, at the real parameters: N = 2^16, the 36-prime chain, 768 ciphertexts carrying 256
// packed inputs of 128 tokens, 12 heads, 3072 intermediate ciphertexts, four bootstrapping rounds of 768.
// Every stage runs through the batched replacements of this repository (each of which is tested bit-identical to
// MOAI's own loop or call sequence) or through MOAI's unchanged headers (layernorm.hpp, gelu_others.hpp on packs):
//   attention head x 12   Q, K, V products + bias (single_att_block.hpp:30-98), Q K^T (:119-125), softmax_boot
//                         (softmax.hpp:307-580, as its call sequence: the header needs NTL), softmax . V (:186-197)
//   self-output product   ct_pt_matrix_mul_wo_pre_w_mask 768 x 768 + bias (test_full_scheme.hpp:601-617)
//   bootstrap round 1     768 x bootstrap_3 (:654-660), residual add (:663-667)
//   LayerNorm 1           layernorm() (:668)
//   bootstrap round 2     (:758-765), 11 levels dropped (:768-773)
//   intermediate product  ct_pt_matrix_mul_wo_pre_large 768 x 3072 + bias (:776-793)
//   GELU                  gelu_v2 on 3072 ciphertexts (:797-803)
//   final product         ct_pt_matrix_mul_wo_pre_w_mask 3072 x 768 + bias (:806-824)
//   bootstrap round 3     (:990-995), residual add (:998-1002)
//   LayerNorm 2           layernorm2() (:1004)
//   bootstrap round 4     (:1080-1087)
// Weights are synthetic N(0, 0.02) (the reference's dense weights are not in the checkout); the bootstrap uses
// stand-in constants (seal/moai_bootstrap_eval.h), so values lose their meaning after the first bootstrap: this
// binary measures time, the per-stage correctness checks live in the tests and the per-stage drivers.
// usage: bench_encoder_layer [heads = 12] [bootstrap packs per round = all] [gelu packs = 48] [bootstrap pack size = 48]
//   smaller numbers make a quick plumbing run: the remaining work is skipped and its results are copies.
#include "seal/seal.h"

#include <omp.h>
#include <sys/time.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>
#include <vector>

#include "Batch_encode_encrypt.hpp"
#include "Ct_pt_matrix_mul.hpp"
#include "gelu_others.hpp"
#include "layernorm.hpp"

#include "seal/moai_bootstrap_eval.h"
#include "seal/moai_fused.h"

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

// softmax_boot (softmax.hpp:307-580) on packed ciphertexts; tools/cpp/bench_softmax.cpp checks this sequence bit for
// bit against the per-ciphertext one
static vector<Ciphertext> softmax_boot_packed(const vector<Ciphertext> &enc_X, const vector<int> &bias_vec, int input_num, const SEALContext &context,
                                              CKKSEncoder &encoder, Evaluator &evaluator, const RelinKeys &relin_keys, int iter,
                                              moai_fused::PackedBootstrapper3 &boot, double minus_index)
{
    const int num = (int)enc_X.size(), num_batch = (int)encoder.slot_count() / 128;
    const double scale = enc_X[0].scale();
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
    vector<Ciphertext> out;
    moai_fused::unpack(pack_out, context, out);
    return out;
}

int main(int argc, char **argv)
{
    const int heads = argc > 1 ? atoi(argv[1]) : 12;
    const int boot_B = argc > 4 ? atoi(argv[4]) : 48; // must divide 768; 63.8 ms per bootstrap at 48, 65.6 ms at 16
    const int boot_packs = argc > 2 ? atoi(argv[2]) : 768 / boot_B;
    const int gelu_packs = argc > 3 ? atoi(argv[3]) : 48;
    omp_set_num_threads(16);
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
    double t0 = now_s();
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    SecretKey sk = keygen.secret_key();
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys relin_keys;
    keygen.create_relin_keys(relin_keys);
    GaloisKeys gal_keys, gal_keys_boot;
    keygen.create_galois_keys(gal_keys); // test_full_scheme.hpp: the rotation keys of the matrix products
    vector<int> steps{ 0 };
    for (int i = 0; i < 15; i++) steps.push_back(1 << i);
    moai_fused::boot_rotation_steps_3(logn, logn, steps);
    keygen.create_galois_keys(steps, gal_keys_boot); // :436-443
    context.sync();
    fprintf(stderr, "keys (relin, 31 + %zu Galois): %.1f s\n", steps.size(), now_s() - t0);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 46);
    const size_t slots = encoder.slot_count();
    const int num_col = 768, num_inter = 3072, col_W = 64, tokens = 128, num_batch = 256, num_input = 5, iter = 16;
    const size_t after_boot = 20, att_level = 15;

    mt19937_64 rng(1);
    normal_distribution<double> nd(0.0, 0.5), wd(0.0, 0.02);
    uniform_real_distribution<double> ud(-1.0, 1.0);
    vector<int> b_vec(slots, 0);
    for (size_t s = 0; s < (size_t)num_batch * num_input; s++) b_vec[s] = 1;

    // bootstrapper with stand-in constants
    const int p3 = logn / 3, totlen = (1 << p3) - 1, slotlen = 1 << logn;
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
    moai_fused::PackedBootstrapper3 boot(context, encoder, evaluator, relin_keys, gal_keys_boot, logn, logn, scale, dg, reducer);

    // one bootstrapping round: every ciphertext to the lowest level (:642-646), then bootstrap_3 in packs of boot_B
    auto bootstrap_round = [&](vector<Ciphertext> &cts, const char *name) {
        const int B = boot_B, packs = (int)cts.size() / B;
        double t = now_s();
        vector<Ciphertext> out(cts.size());
        for (int pk_i = 0; pk_i < packs; pk_i++)
        {
            if (pk_i >= boot_packs)
            {
                for (int b = 0; b < B; b++) out[pk_i * B + b] = out[b]; // quick run: copies
                continue;
            }
            vector<Ciphertext> part(cts.begin() + pk_i * B, cts.begin() + (pk_i + 1) * B);
            for (auto &c : part)
            {
                while (context.get_context_data(c.parms_id())->chain_index() != 0) evaluator.mod_switch_to_next_inplace(c);
            }
            Ciphertext packed = moai_fused::pack(part, context), res;
            boot.bootstrap_3(res, packed);
            vector<Ciphertext> un;
            moai_fused::unpack(res, context, un);
            for (int b = 0; b < B; b++)
            {
                out[pk_i * B + b] = std::move(un[b]);
                cts[pk_i * B + b].release();
            }
        }
        context.sync();
        t = now_s() - t;
        fprintf(stderr, "%-28s %8.2f s   (%d of %d packs of %d; chain index -> %zu)\n", name, t, min(packs, boot_packs), packs, B,
                context.get_context_data(out[0].parms_id())->chain_index());
        cts = std::move(out);
        return t;
    };
    auto add_bias = [&](vector<Ciphertext> &cts, double bias) {
        // :604-617: the bias as a masked vector, one encode per ciphertext
        vector<double> bias_vec(slots, 0);
        for (size_t j = 0; j < slots; ++j)
            if (b_vec[j] == 1) bias_vec[j] = bias;
        for (auto &c : cts)
        {
            Plaintext ecd;
            encoder.encode(bias_vec, c.parms_id(), c.scale(), ecd);
            evaluator.mod_switch_to_inplace(ecd, c.parms_id());
            c.scale() = scale;
            ecd.scale() = scale;
            evaluator.add_plain_inplace(c, ecd);
        }
    };

    // the layer's input: 768 ciphertexts as a bootstrapping round leaves them
    vector<Ciphertext> enc_ecd_x(num_col), enc_ecd_x_copy;
    {
        vector<Ciphertext> base(8);
        for (int i = 0; i < 8; i++)
        {
            vector<double> vals(slots);
            for (size_t s = 0; s < slots; s++) vals[s] = b_vec[s] ? nd(rng) : 0.0;
            Plaintext p;
            encoder.encode(vals, scale, p);
            encryptor.encrypt(p, base[i]);
            evaluator.mod_switch_to_inplace(base[i], context.data_level(after_boot + 1)->parms_id());
        }
        for (int i = 0; i < num_col; i++) enc_ecd_x[i] = base[i % 8];
    }
    enc_ecd_x_copy = enc_ecd_x;
    context.sync();
    fprintf(stderr, "layer input: %d ciphertexts at chain index %zu\n", num_col, context.get_context_data(enc_ecd_x[0].parms_id())->chain_index());

    const double t_layer = now_s();
    double t_att = 0, t_boot = 0;
    // ---- attention: 12 heads -------------------------------------------------------------------------------------------
    vector<Ciphertext> att_output(num_col);
    {
        t0 = now_s();
        vector<Ciphertext> X(num_col), Xv(num_col);
        for (int i = 0; i < num_col; i++)
        {
            X[i] = enc_ecd_x[i];
            evaluator.mod_switch_to_inplace(X[i], context.data_level(att_level + 1)->parms_id());
            Xv[i] = X[i];
            evaluator.mod_switch_to_inplace(Xv[i], context.data_level(3 + 1)->parms_id()); // single_att_block.hpp:76-84
        }
        double t_qkv = 0, t_qk = 0, t_sm = 0, t_sv = 0;
        for (int h = 0; h < 12; h++)
        {
            if (h >= heads)
            {
                for (int j = 0; j < col_W; j++) att_output[h * col_W + j] = att_output[j]; // quick run: copies
                continue;
            }
            vector<vector<double>> WQ(num_col, vector<double>(col_W)), WK = WQ, WV = WQ;
            for (auto *W : { &WQ, &WK, &WV })
                for (auto &r : *W)
                    for (auto &x : r) x = wd(rng);
            double t1 = now_s();
            vector<Ciphertext> Q = moai_fused::ct_pt_matrix_mul_wo_pre(X, WQ, num_col, col_W, num_col, context);
            vector<Ciphertext> K = moai_fused::ct_pt_matrix_mul_wo_pre(X, WK, num_col, col_W, num_col, context);
            vector<Ciphertext> V = moai_fused::ct_pt_matrix_mul_wo_pre(Xv, WV, num_col, col_W, num_col, context);
            for (auto *M : { &Q, &K, &V })
            {
                for (auto &c : *M)
                {
                    // bias as a full vector, single_att_block.hpp:44-55
                    Plaintext ecd;
                    vector<double> bvec(slots, 0.01);
                    encoder.encode(bvec, c.parms_id(), c.scale(), ecd);
                    evaluator.mod_switch_to_inplace(ecd, c.parms_id());
                    c.scale() = scale;
                    ecd.scale() = scale;
                    evaluator.add_plain_inplace(c, ecd);
                }
            }
            context.sync();
            t_qkv += now_s() - t1;
            t1 = now_s();
            vector<Ciphertext> QK = moai_fused::ct_ct_matrix_mul_colpacking(Q, K, gal_keys, relin_keys, context, col_W, tokens, col_W, tokens, num_batch);
            for (auto &c : QK) c.scale() = scale;
            context.sync();
            t_qk += now_s() - t1;
            t1 = now_s();
            vector<Ciphertext> sm = softmax_boot_packed(QK, b_vec, num_input, context, encoder, evaluator, relin_keys, iter, boot, 7.5);
            context.sync();
            t_sm += now_s() - t1;
            t1 = now_s();
            for (auto &c : V)
            {
                if (context.get_context_data(c.parms_id())->chain_index() > context.get_context_data(sm[0].parms_id())->chain_index())
                    evaluator.mod_switch_to_inplace(c, sm[0].parms_id());
            }
            vector<Ciphertext> out = moai_fused::ct_ct_matrix_mul_diagpacking(sm, V, gal_keys, relin_keys, context, tokens, tokens, col_W, tokens, num_batch);
            context.sync();
            t_sv += now_s() - t1;
            if (h == 0)
            {
                fprintf(stderr, "  head 0 chain indices: Q %zu, QK^T %zu, softmax %zu, output %zu\n", context.get_context_data(Q[0].parms_id())->chain_index(),
                        context.get_context_data(QK[0].parms_id())->chain_index(), context.get_context_data(sm[0].parms_id())->chain_index(),
                        context.get_context_data(out[0].parms_id())->chain_index());
            }
            for (int j = 0; j < col_W; j++)
            {
                att_output[h * col_W + j] = std::move(out[j]);
                att_output[h * col_W + j].scale() = scale;
            }
        }
        vector<Ciphertext>().swap(enc_ecd_x); // the next layer's input is written by the last bootstrapping round (:1084-1086)
        t_att = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (%d heads: Q,K,V %.2f, Q K^T %.2f, softmax %.2f, . V %.2f)\n", "attention", t_att, min(heads, 12), t_qkv, t_qk, t_sm, t_sv);
    }
    // ---- self-output product + bias -------------------------------------------------------------------------------------
    vector<Ciphertext> work;
    double t_so;
    {
        vector<vector<double>> W(num_col, vector<double>(num_col));
        for (auto &r : W)
            for (auto &x : r) x = wd(rng);
        t0 = now_s();
        work = moai_fused::ct_pt_matrix_mul_wo_pre_w_mask(att_output, W, b_vec, num_col, num_col, num_col, context);
        add_bias(work, 0.01);
        context.sync();
        t_so = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (chain index %zu -> %zu)\n", "self-output product", t_so, context.get_context_data(att_output[0].parms_id())->chain_index(),
                context.get_context_data(work[0].parms_id())->chain_index());
        vector<Ciphertext>().swap(att_output);
    }
    t_boot += bootstrap_round(work, "bootstrap round 1");
    // ---- residual + LayerNorm 1 -------------------------------------------------------------------------------------------
    vector<double> gamma(num_col, 1.0), beta(num_col, 0.1);
    double t_ln1;
    {
        t0 = now_s();
#pragma omp parallel for
        for (int i = 0; i < num_col; ++i)
        {
            evaluator.mod_switch_to_inplace(enc_ecd_x_copy[i], work[i].parms_id());
            evaluator.add_inplace(work[i], enc_ecd_x_copy[i]);
        }
        vector<Ciphertext>().swap(enc_ecd_x_copy);
        vector<Ciphertext> y = layernorm(work, gamma, beta, b_vec, context, relin_keys, sk);
        context.sync();
        t_ln1 = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (chain index %zu -> %zu)\n", "residual + LayerNorm 1", t_ln1, context.get_context_data(work[0].parms_id())->chain_index(),
                context.get_context_data(y[0].parms_id())->chain_index());
        work = std::move(y);
    }
    t_boot += bootstrap_round(work, "bootstrap round 2");
    vector<Ciphertext> boot_layer = work;
    // ---- intermediate product + bias ------------------------------------------------------------------------------------
    vector<Ciphertext> inter;
    double t_inter;
    {
        vector<vector<double>> W(num_col, vector<double>(num_inter));
        for (auto &r : W)
            for (auto &x : r) x = wd(rng);
        t0 = now_s();
#pragma omp parallel for
        for (int i = 0; i < num_col; ++i)
        {
            for (int j = 0; j < 11; ++j) evaluator.mod_switch_to_next_inplace(work[i]); // :768-773
        }
        inter = moai_fused::ct_pt_matrix_mul_wo_pre_large(work, W, num_col, num_inter, num_col, context);
        add_bias(inter, 0.01);
        context.sync();
        t_inter = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (chain index %zu -> %zu)\n", "intermediate product", t_inter, context.get_context_data(work[0].parms_id())->chain_index(),
                context.get_context_data(inter[0].parms_id())->chain_index());
        vector<Ciphertext>().swap(work);
    }
    // ---- GELU ---------------------------------------------------------------------------------------------------------------
    vector<Ciphertext> gelu_out(num_inter);
    double t_gelu;
    {
        t0 = now_s();
        const int chunk = 64;
        for (int c0 = 0, pk_i = 0; c0 < num_inter; c0 += chunk, pk_i++)
        {
            if (pk_i >= gelu_packs)
            {
                for (int i = 0; i < chunk; i++) gelu_out[c0 + i] = gelu_out[i]; // quick run: copies
                continue;
            }
            vector<Ciphertext> part(inter.begin() + c0, inter.begin() + c0 + chunk), res;
            Ciphertext packed = moai_fused::pack(part, context);
            Ciphertext g = gelu_v2(packed, context, relin_keys, sk);
            moai_fused::unpack(g, context, res);
            for (int i = 0; i < chunk; i++)
            {
                gelu_out[c0 + i] = std::move(res[i]);
                inter[c0 + i].release();
            }
        }
        context.sync();
        t_gelu = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (gelu_v2 on packs of 64; chain index -> %zu)\n", "GELU", t_gelu, context.get_context_data(gelu_out[0].parms_id())->chain_index());
        vector<Ciphertext>().swap(inter);
    }
    // ---- final product + bias -------------------------------------------------------------------------------------------------
    double t_final;
    {
        vector<vector<double>> W(num_inter, vector<double>(num_col));
        for (auto &r : W)
            for (auto &x : r) x = wd(rng);
        t0 = now_s();
        work = moai_fused::ct_pt_matrix_mul_wo_pre_w_mask(gelu_out, W, b_vec, num_inter, num_col, num_inter, context);
        add_bias(work, 0.01);
        context.sync();
        t_final = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (chain index %zu -> %zu)\n", "final product", t_final, context.get_context_data(gelu_out[0].parms_id())->chain_index(),
                context.get_context_data(work[0].parms_id())->chain_index());
        vector<Ciphertext>().swap(gelu_out);
    }
    t_boot += bootstrap_round(work, "bootstrap round 3");
    // ---- residual + LayerNorm 2 -----------------------------------------------------------------------------------------------
    double t_ln2;
    {
        t0 = now_s();
#pragma omp parallel for
        for (int i = 0; i < num_col; ++i)
        {
            evaluator.mod_switch_to_inplace(boot_layer[i], work[i].parms_id());
            evaluator.add_inplace(work[i], boot_layer[i]);
        }
        vector<Ciphertext>().swap(boot_layer);
        vector<Ciphertext> y = layernorm2(work, gamma, beta, b_vec, context, relin_keys, sk);
        context.sync();
        t_ln2 = now_s() - t0;
        fprintf(stderr, "%-28s %8.2f s   (chain index -> %zu)\n", "residual + LayerNorm 2", t_ln2, context.get_context_data(y[0].parms_id())->chain_index());
        work = std::move(y);
    }
    t_boot += bootstrap_round(work, "bootstrap round 4");
    context.sync();
    const double total = now_s() - t_layer;
    const bool full = heads >= 12 && boot_packs >= 768 / boot_B && gelu_packs >= 48 && 768 % boot_B == 0;
    fprintf(stderr, "one encoder layer, 256 packed inputs, %s: %.1f s wall (bootstrapping %.1f s = %.0f %%)\n", full ? "complete" : "QUICK RUN (work skipped)", total,
            t_boot, 100 * t_boot / total);
    if (full)
    {
        fprintf(stderr, "x 12 layers = %.0f s per batch of 256 inputs = %.2f s per encrypted input (paper: 574.6 s on 56 cores)\n", total * 12, total * 12 / 256);
    }
    return 0;
}
