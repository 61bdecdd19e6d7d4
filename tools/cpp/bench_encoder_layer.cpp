// bench_encoder_layer.cpp -- ONE encoder layer of MOAI's 12-layer run (include/test/test_full_scheme.hpp:524-1095)
// with its data flow and levels, at the real parameters: N = 2^16, the 36-prime chain, 768 ciphertexts carrying 256
// packed inputs of 128 tokens, 12 heads, 3072 intermediate ciphertexts, four bootstrapping rounds of 768.
// Every stage runs through the batched replacements of this repository (each of which is tested bit-identical to
// MOAI's own loop or call sequence) or through MOAI's unchanged headers (layernorm.hpp, gelu_others.hpp on packs):
//   attention head x 12   Q, K, V products + bias (single_att_block.hpp:30-98), Q K^T (:119-125), MOAI's own softmax_boot
//                         (softmax.hpp:307-580, header unchanged), softmax . V (:186-197)
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
// Weights are synthetic N(0, 0.02) (the reference's dense weights are not in the checkout); the bootstrap is the drop-in
// Bootstrapper with its real constants.  This binary measures time; the per-stage correctness checks live in the tests
// (tests/cpp/test_moai_headers.cpp, test_moai_attention.cpp, test_bootstrap_real.cpp).
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
#include "softmax.hpp" // MOAI's own softmax_boot, unchanged; brings in the drop-in Bootstrapper (seal_shim/bootstrapping)

#include "seal/moai_fused.h"

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

static int run(int argc, char **argv)
{
    const int heads = argc > 1 ? atoi(argv[1]) : 12;
    const int boot_B = argc > 4 ? atoi(argv[4]) : 48; // must divide 768; 63.8 ms per bootstrap at 48, 65.6 ms at 16
    const int boot_packs = argc > 2 ? atoi(argv[2]) : 768 / boot_B;
    const int gelu_packs = argc > 3 ? atoi(argv[3]) : 48;
    const int layers = argc > 5 ? atoi(argv[5]) : 1; // consecutive layers: the output of one is the input of the next
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
    // The default rotation keys serve Q K^T and softmax . V only, at chain index <= 14 (Ct_ct_matrix_mul.hpp:29,95,112,147 called from
    // single_att_block.hpp:119,197): keep on the device what those levels read -- 19 percent of 41 GB -- and park the rest in host
    // memory (KSwitchKeys::limit_to_chain_index; a switch at a higher level would bring a key back whole, same bits either way).
    // MOAI_KEEP_FULL_KEYS=1 leaves them whole.
    if (!getenv("MOAI_KEEP_FULL_KEYS"))
    {
        gal_keys.limit_to_chain_index(context, 14);
    }
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

    // the Bootstrapper exactly as MOAI's driver sets it up (test_full_scheme.hpp:413-448), with the real constants
    Decryptor decryptor(context, sk);
    Bootstrapper bootstrapper(10, logn, 15, 34, scale, 25, 59, 2, 1, context, keygen, encoder, encryptor, decryptor, evaluator, relin_keys,
                              gal_keys_boot);
    bootstrapper.prepare_mod_polynomial();
    bootstrapper.slot_vec.push_back(logn);
    bootstrapper.generate_LT_coefficient_3();

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
            const double t_pack = now_s();
            const auto fresh0 = seal::util::DevicePool::instance().fresh_allocations();
            Ciphertext packed = moai_fused::pack(part, context), res;
            bootstrapper.bootstrap_full_3(res, packed);
            if (getenv("MOAI_POOL_DEBUG"))
            {
                context.sync();
                const auto fresh1 = seal::util::DevicePool::instance().fresh_allocations();
                fprintf(stderr, "  [pack %d] %.2f s; %llu device allocations, %.0f ms in them\n", pk_i, now_s() - t_pack,
                        (unsigned long long)(fresh1.first - fresh0.first), fresh1.second - fresh0.second);
            }
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

    for (int layer = 0; layer < layers; layer++)
    {
    if (layers > 1)
    {
        fprintf(stderr, "---- layer %d of %d (fresh synthetic weights; input = the previous layer's output) ----\n", layer + 1, layers);
    }
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
            vector<Ciphertext> sm = softmax_boot(QK, b_vec, num_input, context, relin_keys, iter, sk, bootstrapper, 0); // softmax.hpp:308, unchanged
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
    vector<double> gamma(num_col, 0.5), beta(num_col, 0.05); // LayerNorm hands on values with the spread of the synthetic layer input (standard deviation 0.5), so that consecutive layers see what the first one sees
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
    // one machine-readable line per layer on stdout (bench.py's end-to-end slice reads it)
    printf("LAYER_JSON {\"complete\": %s, \"layer_s\": %.3f, \"attention_s\": %.3f, \"selfout_s\": %.3f, \"layernorm1_s\": %.3f, "
           "\"intermediate_s\": %.3f, \"gelu_s\": %.3f, \"final_s\": %.3f, \"layernorm2_s\": %.3f, \"bootstrap_s\": %.3f, "
           "\"bootstrap_pack\": %d}\n",
           full ? "true" : "false", total, t_att, t_so, t_ln1, t_inter, t_gelu, t_final, t_ln2, t_boot, boot_B);
    fflush(stdout);
    if (full)
    {
        fprintf(stderr, "x 12 layers = %.0f s per batch of 256 inputs = %.2f s per encrypted input (paper: 574.6 s on 56 cores)\n", total * 12, total * 12 / 256);
    }
    {
        // what the layer hands on: the first output ciphertext decrypted, over the slots that carry tokens (LayerNorm 2's
        // output through a bootstrap: values of order one if everything before it stayed inside its approximation ranges)
        Plaintext pt;
        decryptor.decrypt(work[0], pt);
        vector<double> vals;
        encoder.decode(pt, vals);
        double mx = 0, sum = 0;
        size_t cnt = 0;
        for (size_t sidx = 0; sidx < slots; sidx++)
        {
            if (b_vec[sidx])
            {
                mx = max(mx, fabs(vals[sidx]));
                sum += fabs(vals[sidx]);
                cnt++;
            }
        }
        fprintf(stderr, "layer output, ciphertext 0 decrypted: max |value| %.3f, mean |value| %.3f over %zu token slots\n", mx, sum / max<size_t>(cnt, 1), cnt);
    }
    // test_full_scheme.hpp:1084-1086: the last bootstrapping round writes the next layer's input
    enc_ecd_x = std::move(work);
    enc_ecd_x_copy = enc_ecd_x;
    }
    return 0;
}

// a failure on the device (out of memory above all) is reported and ends the process with a non-zero status instead of
// std::terminate
int main(int argc, char **argv)
{
    try
    {
        return run(argc, argv);
    }
    catch (const std::exception &e)
    {
        size_t f = 0, t = 0;
        moai_mem_info(&f, &t);
        fprintf(stderr, "FAILED: %s  [device memory: %.1f GiB free of %.1f]\n", e.what(), f / 1073741824.0, t / 1073741824.0);
        return 3;
    }
}
