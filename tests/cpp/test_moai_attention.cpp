// test_moai_attention.cpp -- MOAI's attention head, UNCHANGED: include/source/att_block/single_att_block.hpp with
// include/source/non_linear_func/softmax.hpp (softmax_boot: exp, masks, one bootstrap_3, Goldschmidt inverse) and the
// matrix products of include/source/matrix_mul, included from the reference checkout at build time and compiled against
// the seal:: shim plus the drop-in Bootstrapper (seal_shim/bootstrapping).  BASELINE configs[3] ("one encrypted
// self-attention block: HE QKV matmul + poly-approx softmax, 128 tokens") through the reference's own code path, at
// N = 2^13 with MOAI's 36-prime chain, levels and constants (chain index 15 in, V at index 3, iter = 16;
// include/test/test_full_scheme.hpp:494-532), 32 packed inputs x 128 token slots, 8 input columns, head width 4.
//
// Unlike the reference's driver, the result is checked: decrypt(single_att_block(...)) against the same attention
// computed in the clear with the approximations MOAI uses -- (1 + x/128)^128 for exp (softmax.hpp:9-47), the masks of
// softmax_boot (:330-466), the 16-step Goldschmidt inverse (:49-82) -- and the bootstrap's own transfer function
// a sin(2 pi r m) / r (r = scale / q0; a = the inverse-sine slope).
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
#include "softmax.hpp"
#include "single_att_block.hpp"

static int g_fail = 0;
#define CHECK(cond)                                                        \
    do                                                                     \
    {                                                                      \
        if (!(cond))                                                       \
        {                                                                  \
            g_fail++;                                                      \
            printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                  \
    } while (0)

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    omp_set_num_threads(8);
    // include/test/test_full_scheme.hpp:345-389 at a smaller ring
    long boundary_K = 25, deg = 59, scale_factor = 2, inverse_deg = 1;
    long logN = 13, loge = 10, logn = logN - 1;
    int logp = 46, logq = 51, log_special_prime = 58;
    int remaining_level = 20, boot_level = 14, total_level = remaining_level + boot_level;
    vector<int> coeff_bit_vec;
    coeff_bit_vec.push_back(logq);
    for (int i = 0; i < remaining_level; i++) coeff_bit_vec.push_back(logp);
    for (int i = 0; i < boot_level; i++) coeff_bit_vec.push_back(logq);
    coeff_bit_vec.push_back(log_special_prime);
    EncryptionParameters parms(scheme_type::ckks);
    size_t poly_modulus_degree = (size_t)(1 << logN);
    parms.set_poly_modulus_degree(poly_modulus_degree);
    parms.set_coeff_modulus(CoeffModulus::Create(poly_modulus_degree, coeff_bit_vec));
    parms.set_secret_key_hamming_weight(192);
    double scale = pow(2.0, logp);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    SecretKey secret_key = keygen.secret_key();
    PublicKey public_key;
    keygen.create_public_key(public_key);
    RelinKeys relin_keys;
    keygen.create_relin_keys(relin_keys);
    GaloisKeys gal_keys;
    keygen.create_galois_keys(gal_keys);
    GaloisKeys gal_keys_boot;
    Encryptor encryptor(context, public_key);
    Decryptor decryptor(context, secret_key);
    CKKSEncoder encoder(context);
    Evaluator evaluator(context, encoder);
    size_t slot_count = encoder.slot_count();

    Bootstrapper bootstrapper(loge, logn, logN - 1, total_level, scale, boundary_K, deg, scale_factor, inverse_deg, context, keygen, encoder,
                              encryptor, decryptor, evaluator, relin_keys, gal_keys_boot);
    bootstrapper.prepare_mod_polynomial();
    vector<int> gal_steps_vector;
    gal_steps_vector.push_back(0);
    for (int i = 0; i < logN - 1; i++) gal_steps_vector.push_back((1 << i));
    bootstrapper.addLeftRotKeys_Linear_to_vector_3(gal_steps_vector);
    keygen.create_galois_keys(gal_steps_vector, gal_keys_boot);
    bootstrapper.slot_vec.push_back(logn);
    bootstrapper.generate_LT_coefficient_3();

    // ---- inputs: num_X inputs x 128 token rows x num_col columns; the first three inputs hold 5 tokens, the rest none
    const int num_row = 128, num_X = (int)slot_count / num_row, num_col = 8, col_W = 4, input_num = 5, layer_id = 0, iter = 16;
    const double minus_index = 7.5; // softmax.hpp:324, layer 0
    mt19937_64 rng(42);
    uniform_real_distribution<double> ud(-1.0, 1.0);
    vector<vector<vector<double>>> X(num_X, vector<vector<double>>(num_row, vector<double>(num_col, 0.0)));
    vector<int> input_len(num_X, 0);
    for (int j = 0; j < 3; j++)
    {
        input_len[j] = input_num;
        for (int k = 0; k < input_num; k++)
            for (int i = 0; i < num_col; i++) X[j][k][i] = 0.5 * ud(rng);
    }
    vector<int> b_vec = bias_vec(input_len, num_X, num_row);
    vector<vector<double>> WQ(num_col, vector<double>(col_W)), WK(num_col, vector<double>(col_W)), WV(num_col, vector<double>(col_W));
    vector<double> bQ(col_W), bK(col_W), bV(col_W);
    for (int r = 0; r < num_col; r++)
        for (int c = 0; c < col_W; c++)
        {
            WQ[r][c] = 0.12 * ud(rng);
            WK[r][c] = 0.12 * ud(rng);
            WV[r][c] = 0.4 * ud(rng);
        }
    for (int c = 0; c < col_W; c++)
    {
        bQ[c] = 1.1;
        bK[c] = 1.1 + 0.05 * c;
        bV[c] = 0.3 * ud(rng);
    }
    vector<Ciphertext> enc_X = batch_input(X, num_X, num_row, num_col, scale, context, public_key);
    // test_full_scheme.hpp:468-507: fresh ciphertexts are switched down to chain index 15 before the attention block
    for (auto &c : enc_X)
        while (context.get_context_data(c.parms_id())->chain_index() > 15) evaluator.mod_switch_to_next_inplace(c);

    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    vector<Ciphertext> out = single_att_block(enc_X, WQ, WK, WV, bQ, bK, bV, b_vec, input_num, context, relin_keys, gal_keys, bootstrapper,
                                              num_X, secret_key, iter, layer_id);
    context.sync();
    gettimeofday(&t1, NULL);
    printf("\nsingle_att_block (MOAI's header, unchanged): %.2f s\n", t1.tv_sec - t0.tv_sec + (t1.tv_usec - t0.tv_usec) / 1e6);
    CHECK(out.size() == (size_t)col_W);
    // QK^T at index 13, exp and its mask 13 -> 4, the 16-step inverse 20 -> 3, the normalisation -> 2 = V's level, the product
    // with V -> 1 ("softmax 13 -> 3, softmax*V 3 -> 2" in 2025-991.pdf table 3, which counts levels from 1)
    printf("output at chain index %zu\n", context.get_context_data(out[0].parms_id())->chain_index());
    CHECK(context.get_context_data(out[0].parms_id())->chain_index() == 1);

    // ---- the same attention in the clear, with MOAI's approximations
    const double q0 = (double)context.first_context_data()->parms().coeff_modulus()[0].value();
    const double r = scale / q0, slope = bootstrapper.mod_reducer->inverse_sin_polynomial.chebcoeff[1];
    auto boot_transfer = [&](double m) { return slope * sin(2 * M_PI * r * m) / r; };
    auto approx_exp = [](double x) { return pow(1 + x * 0.0078125, 128); };
    auto goldschmidt = [&](double x) {
        double y = 1 - x, res = 1 + y;
        for (int i = 0; i < iter; i++)
        {
            y = y * y;
            res *= 1 + y;
        }
        return res;
    };
    double worst = 0, worst_vs_true = 0, smin = 1e9, smax = -1e9, summax = 0;
    vector<vector<double>> dec(col_W);
    for (int c = 0; c < col_W; c++)
    {
        Plaintext p;
        decryptor.decrypt(out[c], p);
        encoder.decode(p, dec[c]);
    }
    for (int j = 0; j < 3; j++)
    {
        vector<vector<double>> Q(input_num, vector<double>(col_W)), Km(input_num, vector<double>(col_W)), V(input_num, vector<double>(col_W));
        for (int k = 0; k < input_num; k++)
            for (int c = 0; c < col_W; c++)
            {
                double q = bQ[c], kk = bK[c], v = bV[c];
                for (int i = 0; i < num_col; i++)
                {
                    q += X[j][k][i] * WQ[i][c];
                    kk += X[j][k][i] * WK[i][c];
                    v += X[j][k][i] * WV[i][c];
                }
                Q[k][c] = q;
                Km[k][c] = kk;
                V[k][c] = v;
            }
        for (int k = 0; k < input_num; k++)
        {
            vector<double> e(input_num), et(input_num);
            double sum = 0, sumt = 0;
            for (int k2 = 0; k2 < input_num; k2++)
            {
                double s = 0;
                for (int c = 0; c < col_W; c++) s += Q[k][c] * Km[k2][c];
                smin = min(smin, s);
                smax = max(smax, s);
                e[k2] = approx_exp(s - minus_index);
                et[k2] = exp(s - minus_index);
                sum += e[k2];
                sumt += et[k2];
            }
            summax = max(summax, sum);
            const double inv = goldschmidt(boot_transfer(sum + 0.00001));
            for (int c = 0; c < col_W; c++)
            {
                double want = 0, truth = 0;
                for (int k2 = 0; k2 < input_num; k2++)
                {
                    want += e[k2] * inv * V[k2][c];
                    truth += et[k2] / sumt * V[k2][c];
                }
                const double got = dec[c][(size_t)num_X * k + j];
                worst = max(worst, fabs(got - want));
                worst_vs_true = max(worst_vs_true, fabs(got - truth));
            }
        }
    }
    printf("scores in [%.2f, %.2f] (shifted by %.1f), largest sum of exponentials %.3f\n", smin, smax, minus_index, summax);
    printf("max |decrypted - attention with MOAI's approximations| = %.3e ; against the exact softmax attention %.3e\n", worst, worst_vs_true);
    CHECK(smax < minus_index && summax < 1.9); // inside the domain the reference's approximations are built for
    CHECK(worst < 2e-3);
    CHECK(worst_vs_true < 5e-3);
    const auto gs = bootstrapper.gather_statistics();
    printf("bootstrap_3 calls: %zu in %zu runs\n", gs.second, gs.first);
    CHECK(gs.second == 1);
    if (!g_fail)
    {
        printf("ALL PASS\n");
    }
    return g_fail ? 1 : 0;
}
