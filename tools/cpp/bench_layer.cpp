// bench_layer.cpp -- the parts of one BERT encoder layer of MOAI's 12-layer run
// (include/test/test_full_scheme.hpp:596-1095) that can be built without the Bootstrapper, at the real
// parameters (N = 2^16, 36-prime chain, 768 / 3072 ciphertexts, 256 packed inputs), after the attention heads
// (tools/cpp/bench_attention.cpp):
//   LayerNorm 1        layernorm()  -- MOAI's own header, unchanged, on 768 ciphertexts at the post-bootstrap level
//   intermediate       ct_pt_matrix_mul_wo_pre_large 768 x 3072 (moai_fused:: replacement; MOAI's loop on one
//                      128-column block for comparison)
//   GELU               gelu_v2 on 3072 ciphertexts (MOAI's header on moai_fused::pack'ed batches of 64; per
//                      ciphertext on a sample for comparison)
//   final              ct_pt_matrix_mul_wo_pre_w_mask 3072 x 768 (moai_fused:: replacement; MOAI's loop on one block)
//   LayerNorm 2        layernorm2()
// The four bootstrapping rounds between them (3072 bootstraps) and the softmax are NOT here: their setup needs NTL.
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

#include "seal/moai_fused.h"

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 16;
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
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    SecretKey sk = keygen.secret_key();
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys relin_keys;
    keygen.create_relin_keys(relin_keys);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, sk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 46);
    const size_t slots = encoder.slot_count();
    const int num_col = 768, num_inter = 3072;
    const size_t after_boot = 20; // remaining_level of test_full_scheme.hpp:363-368

    mt19937_64 rng(2);
    normal_distribution<double> nd(0.0, 0.5), wd(0.0, 0.02);
    vector<int> b_vec(slots, 0);
    for (size_t s = 0; s < slots; s++) b_vec[s] = (s % 128) < 100 ? 1 : 0;
    auto fresh = [&](int count, size_t index, double mean) {
        vector<Ciphertext> base(8), v(count);
        for (int i = 0; i < 8; i++)
        {
            vector<double> vals(slots);
            for (size_t s = 0; s < slots; s++) vals[s] = b_vec[s] ? mean + nd(rng) : 0.0;
            Plaintext p;
            encoder.encode(vals, scale, p);
            encryptor.encrypt(p, base[i]);
            evaluator.mod_switch_to_inplace(base[i], context.data_level(index + 1)->parms_id());
        }
        for (int i = 0; i < count; i++) v[i] = base[i % 8];
        context.sync();
        return v;
    };
    double t0, total = 0;

    // ---- LayerNorm 1 (MOAI's header, per-ciphertext calls from its own OpenMP loops) ------------------------
    vector<double> gamma(num_col, 1.0), beta(num_col, 0.1);
    {
        vector<Ciphertext> x = fresh(num_col, after_boot, 0.0);
        t0 = now_s();
        vector<Ciphertext> y = layernorm(x, gamma, beta, b_vec, context, relin_keys, sk);
        context.sync();
        double t = now_s() - t0;
        total += t;
        printf("LayerNorm 1 (layernorm.hpp unchanged, 768 ciphertexts, chain index %zu -> %zu): %8.2f s\n", after_boot,
               context.get_context_data(y[0].parms_id())->chain_index(), t);
    }
    // ---- intermediate product 768 x 3072, scalar weights ---------------------------------------------------------
    vector<Ciphertext> inter;
    {
        vector<Ciphertext> x = fresh(num_col, after_boot, 0.0);
        vector<vector<double>> W(num_col, vector<double>(num_inter));
        for (auto &r : W)
            for (auto &v : r) v = wd(rng);
        vector<vector<double>> W128(num_col, vector<double>(128));
        for (int r = 0; r < num_col; r++)
            for (int c = 0; c < 128; c++) W128[r][c] = W[r][c];
        t0 = now_s();
        vector<Ciphertext> ref = ct_pt_matrix_mul_wo_pre_large(x, W128, num_col, 128, num_col, context);
        context.sync();
        double t_ref = now_s() - t0;
        double t = 0;
        for (int rep = 0; rep < 2; rep++)
        {
            t0 = now_s();
            inter = moai_fused::ct_pt_matrix_mul_wo_pre_large(x, W, num_col, num_inter, num_col, context);
            context.sync();
            t = now_s() - t0;
        }
        total += t;
        printf("intermediate X W (768 x 3072): MOAI's loop %.2f s for 128 columns (= %.1f s for 3072), fused %.2f s for all (%s), chain index %zu\n",
               t_ref, t_ref * 24, t, ref[7].download() == inter[7].download() ? "bit-identical" : "DIFFERENT",
               context.get_context_data(inter[0].parms_id())->chain_index());
    }
    // ---- GELU on the 3072 intermediate ciphertexts ------------------------------------------------------------------
    vector<Ciphertext> gelu_out(num_inter);
    {
        t0 = now_s();
        const int sample = 32;
#pragma omp parallel for
        for (int i = 0; i < sample; i++)
        {
            Ciphertext g = gelu_v2(inter[num_inter - 1 - i], context, relin_keys, sk);
        }
        context.sync();
        double t_ref = now_s() - t0;
        t0 = now_s();
        const int chunk = 64;
        for (int c0 = 0; c0 < num_inter; c0 += chunk)
        {
            vector<Ciphertext> part(inter.begin() + c0, inter.begin() + c0 + chunk), res;
            Ciphertext packed = moai_fused::pack(part, context);
            Ciphertext g = gelu_v2(packed, context, relin_keys, sk);
            moai_fused::unpack(g, context, res);
            for (int i = 0; i < chunk; i++)
            {
                gelu_out[c0 + i] = std::move(res[i]);
                inter[c0 + i].release(); // consumed
            }
        }
        context.sync();
        double t = now_s() - t0;
        total += t;
        printf("GELU (gelu_v2 unchanged): per ciphertext %.1f ms (%d OpenMP threads) = %.1f s for 3072; on packs of 64: %.2f s for 3072 (%.2f ms each), chain index %zu\n",
               t_ref * 1e3 / sample, threads, t_ref / sample * num_inter, t, t * 1e3 / num_inter,
               context.get_context_data(gelu_out[0].parms_id())->chain_index());
        vector<Ciphertext>().swap(inter);
    }
    // ---- final product 3072 x 768, masked vector weights ------------------------------------------------------------
    {
        vector<vector<double>> W(num_inter, vector<double>(num_col));
        for (auto &r : W)
            for (auto &v : r) v = wd(rng);
        vector<vector<double>> W128(num_inter, vector<double>(128));
        for (int r = 0; r < num_inter; r++)
            for (int c = 0; c < 128; c++) W128[r][c] = W[r][c];
        // MOAI's loop on 8 columns' worth of work: 1/16 of a 128-column block is not expressible, so time one block of a
        // thinner matrix (384 rows) and scale by rows and columns
        vector<Ciphertext> xs(gelu_out.begin(), gelu_out.begin() + 384);
        vector<vector<double>> Ws(W128.begin(), W128.begin() + 384);
        t0 = now_s();
        vector<Ciphertext> ref = ct_pt_matrix_mul_wo_pre_w_mask(xs, Ws, b_vec, 384, 128, 384, context);
        context.sync();
        double t_ref = now_s() - t0;
        t0 = now_s();
        vector<Ciphertext> fin = moai_fused::ct_pt_matrix_mul_wo_pre_w_mask(gelu_out, W, b_vec, num_inter, num_col, num_inter, context);
        context.sync();
        double t = now_s() - t0;
        total += t;
        vector<Ciphertext> chk = moai_fused::ct_pt_matrix_mul_wo_pre_w_mask(xs, Ws, b_vec, 384, 128, 384, context);
        printf("final X W (3072 x 768, masked vector weights): MOAI's loop %.2f s for 384 x 128 (= %.0f s for 3072 x 768), fused %.2f s for all (%s on the 384 x 128 block)\n",
               t_ref, t_ref * 8 * 6, t, ref[3].download() == chk[3].download() ? "bit-identical" : "DIFFERENT");
    }
    // ---- LayerNorm 2 ----------------------------------------------------------------------------------------------------
    {
        vector<Ciphertext> x = fresh(num_col, after_boot, 0.0);
        t0 = now_s();
        vector<Ciphertext> y = layernorm2(x, gamma, beta, b_vec, context, relin_keys, sk);
        context.sync();
        double t = now_s() - t0;
        total += t;
        printf("LayerNorm 2 (layernorm.hpp unchanged): %8.2f s\n", t);
    }
    printf("feed-forward half of one layer without its bootstraps, 256 packed inputs: %.1f s = %.1f ms per input\n", total, total * 1e3 / 256);
    return 0;
}
