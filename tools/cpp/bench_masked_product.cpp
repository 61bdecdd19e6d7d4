// bench_masked_product.cpp -- the final feed-forward product of a layer alone (moai_fused::ct_pt_matrix_mul_wo_pre_w_mask,
// the replacement of include/source/matrix_mul/Ct_pt_matrix_mul.hpp:103-170): 3072 input ciphertexts at chain index 1 (two
// data primes, test_full_scheme.hpp:806-824), masked vector weights, `cols` output columns (128 by default; the layer has 768).
// usage: bench_masked_product [rows] [cols]
#include "seal/seal.h"
#include "seal/moai_fused.h"

#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

using namespace std;
using namespace seal;

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 3072, cols = argc > 2 ? atoi(argv[2]) : 128;
    EncryptionParameters parms(scheme_type::ckks);
    const size_t n = 65536;
    vector<int> bits;
    bits.push_back(51);
    for (int i = 0; i < 20; i++) bits.push_back(46);
    for (int i = 0; i < 14; i++) bits.push_back(51);
    bits.push_back(58);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, bits));
    parms.set_secret_key_hamming_weight(192);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 46);
    const size_t slots = encoder.slot_count();
    vector<double> v(slots, 0.25);
    Plaintext p;
    encoder.encode(v, scale, p);
    Ciphertext c;
    encryptor.encrypt(p, c);
    while (context.get_context_data(c.parms_id())->chain_index() > 1) evaluator.mod_switch_to_next_inplace(c);
    vector<Ciphertext> x(rows, c);
    mt19937_64 rng(1);
    uniform_real_distribution<double> wd(-0.05, 0.05);
    vector<vector<double>> W(rows, vector<double>(cols));
    for (auto &r : W)
        for (auto &w : r) w = wd(rng);
    vector<int> b_vec(slots, 0);
    for (size_t i = 0; i < slots; i++) b_vec[i] = (i / 256) % 128 < 5 ? 1 : 0;
    for (int rep = 0; rep < 2; rep++)
    {
        context.sync();
        const double t0 = now_s();
        vector<Ciphertext> out = moai_fused::ct_pt_matrix_mul_wo_pre_w_mask(x, W, b_vec, rows, cols, rows, context);
        context.sync();
        const double t = now_s() - t0;
        printf("masked product %d x %d at chain index 1: %.3f s = %.2f ms per column (768 columns: %.2f s); output chain index %zu\n", rows,
               cols, t, 1e3 * t / cols, t / cols * 768, context.get_context_data(out[0].parms_id())->chain_index());
    }
    return 0;
}
