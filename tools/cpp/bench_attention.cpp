// bench_attention.cpp -- BASELINE.json config 4 as far as it can be built here: the matrix products of
// one encrypted self-attention block (include/source/att_block/single_att_block.hpp:30-197) run through
// MOAI's OWN headers (include/source/matrix_mul/*.hpp, included unchanged from the reference checkout at
// build time) against the seal:: shim, at the real parameters: N = 2^16, the 36-prime chain
// (include/test/test_full_scheme.hpp:356-378), 768 input ciphertexts at chain index 15, 64-column heads,
// 128 tokens, 256 packed inputs.  softmax_boot (softmax.hpp) is NOT run: it needs the Bootstrapper, whose
// setup needs NTL, which this image lacks; its inputs/outputs are replaced by ciphertexts at the levels
// the reference has there.  Every stage is timed wall-clock with the stream drained.
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
#include "Ct_ct_matrix_mul.hpp"
#include "gelu_others.hpp"

#include "seal/moai_fused.h"

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    int threads = argc > 1 ? atoi(argv[1]) : 16;
    int num_col = argc > 2 ? atoi(argv[2]) : 768; // hidden size
    omp_set_num_threads(threads);
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 65536;
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
    GaloisKeys gal_keys;
    keygen.create_galois_keys(gal_keys); // the reference's key set: powers of two (galois.cpp:106-131)
    context.sync();
    printf("context + relin key + 31 Galois keys: %.1f s (%d OpenMP threads for the evaluator calls)\n", now_s() - t0, threads);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, sk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 46);
    const size_t slots = encoder.slot_count();
    const int col_W = 64, tokens = 128, num_batch = 256;

    // 768 input ciphertexts at chain index 15 (single_att_block's input level, test_full_scheme.hpp:363-368)
    t0 = now_s();
    mt19937_64 rng(1);
    normal_distribution<double> nd(0.0, 0.5);
    vector<Ciphertext> base(8);
    vector<vector<double>> base_vals(8, vector<double>(slots));
    for (int i = 0; i < 8; i++)
    {
        for (auto &x : base_vals[i]) x = nd(rng);
        Plaintext p;
        encoder.encode(base_vals[i], scale, p);
        encryptor.encrypt(p, base[i]);
        while (context.get_context_data(base[i].parms_id())->chain_index() > 15)
        {
            evaluator.mod_switch_to_next_inplace(base[i]);
        }
    }
    vector<Ciphertext> enc_X(num_col);
    for (int i = 0; i < num_col; i++) enc_X[i] = base[i % 8];
    context.sync();
    printf("%d input ciphertexts at chain index %zu: %.2f s\n", num_col, context.get_context_data(enc_X[0].parms_id())->chain_index(), now_s() - t0);

    normal_distribution<double> wd(0.0, 0.02);
    vector<vector<double>> WQ(num_col, vector<double>(col_W)), WK = WQ, WV = WQ;
    for (auto *W : { &WQ, &WK, &WV })
        for (auto &r : *W)
            for (auto &x : r) x = wd(rng);
    vector<double> bQ(col_W, 0.01), bias_vec(slots, 1.0);

    // ---- Q, K, V = X W + b (single_att_block.hpp:30-98) ------------------------------------------------
    t0 = now_s();
    vector<Ciphertext> Q = ct_pt_matrix_mul_wo_pre(enc_X, WQ, num_col, col_W, num_col, context);
    context.sync();
    double t_q = now_s() - t0;
    t0 = now_s();
    for (int i = 0; i < col_W; ++i)
    {
        Plaintext ecd_b_q;
        vector<double> bq_vec(slots, 0);
        for (size_t j = 0; j < slots; ++j) bq_vec[j] = bQ[i];
        encoder.encode(bq_vec, Q[i].parms_id(), Q[i].scale(), ecd_b_q);
        evaluator.mod_switch_to_inplace(ecd_b_q, Q[i].parms_id());
        Q[i].scale() = scale;
        ecd_b_q.scale() = scale;
        evaluator.add_plain_inplace(Q[i], ecd_b_q);
    }
    context.sync();
    double t_bias = now_s() - t0;
    t0 = now_s();
    vector<Ciphertext> K = ct_pt_matrix_mul_wo_pre(enc_X, WK, num_col, col_W, num_col, context);
    for (auto &c : K) c.scale() = scale;
    context.sync();
    double t_k = now_s() - t0;
    vector<Ciphertext> enc_X_v(num_col);
    t0 = now_s();
#pragma omp parallel for
    for (int i = 0; i < num_col; ++i)
    {
        enc_X_v[i] = enc_X[i];
        while (context.get_context_data(enc_X_v[i].parms_id())->chain_index() > 3)
        {
            evaluator.mod_switch_to_next_inplace(enc_X_v[i]);
        }
    }
    vector<Ciphertext> V = ct_pt_matrix_mul_wo_pre(enc_X_v, WV, num_col, col_W, num_col, context);
    for (auto &c : V) c.scale() = scale;
    context.sync();
    double t_v = now_s() - t0;
    printf("Q = X WQ (MOAI's loop, %d x %d):   %8.3f s\n", num_col, col_W, t_q);
    printf("Q += bias (64 vector encodes):       %8.3f s\n", t_bias);
    printf("K = X WK:                            %8.3f s\n", t_k);
    printf("V = X WV at chain index 3 (+ drops): %8.3f s\n", t_v);
    // the fused replacements are timed on their second call: the first one also pays for the device blocks the
    // shim's pool does not hold yet (hipMalloc of multi-GB blocks), which a 12-layer run pays once
    for (int rep = 0; rep < 2; rep++)
    {
        t0 = now_s();
        vector<Ciphertext> Qf = moai_fused::ct_pt_matrix_mul_wo_pre(enc_X, WQ, num_col, col_W, num_col, context);
        context.sync();
        printf("Q again with the fused product (%s):  %8.3f s\n", rep ? "warm pool" : "cold pool", now_s() - t0);
    }

    // spot check of Q[0] against the plaintext product
    {
        Plaintext p;
        vector<double> out;
        decryptor.decrypt(Q[0], p);
        encoder.decode(p, out);
        double err = 0;
        for (size_t s = 0; s < slots; s += 997)
        {
            double e = bQ[0];
            for (int r = 0; r < num_col; r++) e += base_vals[r % 8][s] * WQ[r][0];
            err = max(err, fabs(out[s] - e));
        }
        printf("   Q[0] max |error| vs plaintext: %.2e\n", err);
    }

    // ---- Q K^T (single_att_block.hpp:119-125) --------------------------------------------------------
    t0 = now_s();
    vector<Ciphertext> QK = ct_ct_matrix_mul_colpacking(Q, K, gal_keys, relin_keys, context, col_W, tokens, col_W, tokens, num_batch);
    context.sync();
    double t_qk = now_s() - t0;
    printf("Q K^T (colpacking, 128 x 64, NAF rotations): %8.3f s, result at chain index %zu\n", t_qk,
           context.get_context_data(QK[0].parms_id())->chain_index());
    {
        Plaintext p;
        vector<double> q0, k0, out;
        decryptor.decrypt(QK[1], p);
        encoder.decode(p, out);
        // row 1: sum_j Q[j][s] * K[j][s + 256]
        vector<vector<double>> qd(col_W), kd(col_W);
        for (int j = 0; j < col_W; j++)
        {
            decryptor.decrypt(Q[j], p);
            encoder.decode(p, qd[j]);
            decryptor.decrypt(K[j], p);
            encoder.decode(p, kd[j]);
        }
        double err = 0;
        for (size_t s = 0; s < slots; s += 1013)
        {
            double e = 0;
            for (int j = 0; j < col_W; j++) e += qd[j][s] * kd[j][(s + num_batch) % slots];
            err = max(err, fabs(out[s] - e));
        }
        printf("   QK[1] max |error| vs decrypted Q, K: %.2e\n", err);
    }

    vector<Ciphertext> QKf;
    double t_qkf = 0;
    for (int rep = 0; rep < 2; rep++)
    {
        t0 = now_s();
        QKf = moai_fused::ct_ct_matrix_mul_colpacking(Q, K, gal_keys, relin_keys, context, col_W, tokens, col_W, tokens, num_batch);
        context.sync();
        t_qkf = now_s() - t0;
        if (!rep) printf("Q K^T again, batched + shared prefixes (cold pool): %8.3f s\n", t_qkf);
    }
    {
        bool same = true;
        for (int i : { 0, 1, 3, 77, 127 }) same = same && (QKf[i].download() == QK[i].download());
        printf("Q K^T again, batched + shared prefixes (warm pool): %8.3f s (%s MOAI's loop)\n", t_qkf, same ? "bit-identical to" : "DIFFERS from");
    }

    // ---- softmax(QK^T) V (single_att_block.hpp:186-197); softmax output stands at V's level ---------
    vector<Ciphertext> enc_softmax(tokens);
    for (int i = 0; i < tokens; i++)
    {
        enc_softmax[i] = base[i % 8];
        evaluator.mod_switch_to_inplace(enc_softmax[i], V[0].parms_id());
    }
    context.sync();
    t0 = now_s();
    vector<Ciphertext> out = ct_ct_matrix_mul_diagpacking(enc_softmax, V, gal_keys, relin_keys, context, tokens, tokens, col_W, tokens, num_batch);
    context.sync();
    double t_sv = now_s() - t0;
    printf("softmax(QK^T) V (diagpacking, chain index %zu): %8.3f s\n", context.get_context_data(V[0].parms_id())->chain_index(), t_sv);
    {
        vector<Ciphertext> outf;
        double t_f = 0;
        for (int rep = 0; rep < 2; rep++)
        {
            t0 = now_s();
            outf = moai_fused::ct_ct_matrix_mul_diagpacking(enc_softmax, V, gal_keys, relin_keys, context, tokens, tokens, col_W, tokens, num_batch);
            context.sync();
            t_f = now_s() - t0;
        }
        bool same = true;
        for (int i : { 0, 31, 63 }) same = same && (outf[i].download() == out[i].download());
        printf("softmax(QK^T) V again, batched:                 %8.3f s (%s MOAI's loop)\n", t_f, same ? "bit-identical to" : "DIFFERS from");
    }

    // ---- self-output product (test_full_scheme.hpp:601): 768 x 768 with vector-encoded masked weights, on the
    // concatenated head outputs (chain index of the .V result) ---------------------------------------------------
    {
        vector<Ciphertext> att_output(num_col);
        for (int i = 0; i < num_col; i++) att_output[i] = out[i % col_W];
        vector<int> b_vec(slots, 0);
        for (size_t s = 0; s < slots; s++) b_vec[s] = (s % 128) < 100 ? 1 : 0;
        vector<vector<double>> Wso(num_col, vector<double>(num_col));
        for (auto &r : Wso)
            for (auto &x : r) x = wd(rng);
        // MOAI's loop on the first 128 of the 768 columns (a sixth of the work), then the fused product on all
        vector<vector<double>> W128(num_col, vector<double>(128));
        for (int r = 0; r < num_col; r++)
            for (int c = 0; c < 128; c++) W128[r][c] = Wso[r][c];
        context.sync();
        t0 = now_s();
        vector<Ciphertext> so_ref = ct_pt_matrix_mul_wo_pre_w_mask(att_output, W128, b_vec, num_col, 128, num_col, context);
        context.sync();
        double t_ref = now_s() - t0;
        vector<Ciphertext> so;
        double t_f = 0;
        for (int rep = 0; rep < 2; rep++)
        {
            t0 = now_s();
            so = moai_fused::ct_pt_matrix_mul_wo_pre_w_mask(att_output, Wso, b_vec, num_col, num_col, num_col, context);
            context.sync();
            t_f = now_s() - t0;
        }
        bool same = true;
        for (int c : { 0, 17, 127 }) same = same && (so[c].download() == so_ref[c].download());
        printf("self-output X W (768 x 768, masked vector weights, chain index %zu): MOAI's loop %.2f s for 128 columns (= %.1f s for 768), fused %.3f s for all 768 (%s)\n",
               context.get_context_data(att_output[0].parms_id())->chain_index(), t_ref, t_ref * 6, t_f,
               same ? "bit-identical" : "DIFFERENT");
    }

    // ---- GELU on one intermediate ciphertext (gelu_others.hpp gelu_v2; 3072 of these per layer) --------
    t0 = now_s();
    const int gelu_n = 16;
#pragma omp parallel for
    for (int i = 0; i < gelu_n; i++)
    {
        Ciphertext gi = gelu_v2(Q[i], context, relin_keys, sk);
    }
    context.sync();
    printf("gelu_v2 on %d ciphertexts at chain index 14: %8.3f s (%.1f ms each)\n", gelu_n, now_s() - t0, (now_s() - t0) * 1e3 / gelu_n);

    {
        // the same routine on a pack of 64 ciphertexts (moai_fused::pack): MOAI's source, batched kernels
        vector<Ciphertext> some(Q.begin(), Q.begin() + 64);
        vector<Ciphertext> gs;
        double t_pack = 0;
        for (int rep = 0; rep < 2; rep++)
        {
            t0 = now_s();
            Ciphertext packed = moai_fused::pack(some, context);
            Ciphertext gp = gelu_v2(packed, context, relin_keys, sk);
            moai_fused::unpack(gp, context, gs);
            context.sync();
            t_pack = now_s() - t0;
            if (!rep) printf("gelu_v2 on a pack of 64 ciphertexts (cold pool): %8.3f s\n", t_pack);
        }
        Ciphertext g0 = gelu_v2(Q[0], context, relin_keys, sk);
        printf("gelu_v2 on a pack of 64 ciphertexts (warm pool): %8.3f s (%.1f ms each), member 0 %s the single call\n", t_pack,
               t_pack * 1e3 / 64, g0.download() == gs[0].download() ? "bit-identical to" : "DIFFERS from");
    }

    double total = t_q + t_bias + t_k + t_v + t_qk + t_sv;
    printf("matrix products of one head (Q,K,V, QK^T, .V), wall: %.2f s for 256 packed inputs = %.1f ms per input\n", total,
           total * 1e3 / 256);
    return 0;
}
