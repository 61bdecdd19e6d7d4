// bench_shim.cpp -- cost of MOAI's per-ciphertext call pattern through the seal:: shim at real size
// (N = 2^16, 16 primes): Ct_pt_matrix_mul.hpp:19-38's inner loop, and a rotate + relinearize.
#include <chrono>
#include <cstdio>

#include "seal/seal.h"

using namespace seal;
using namespace std;

int main()
{
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 65536;
    parms.set_poly_modulus_degree(n);
    vector<int> bits{ 51 };
    for (int i = 0; i < 15; i++) bits.push_back(46);
    bits.push_back(58);
    parms.set_coeff_modulus(CoeffModulus::Create(n, bits));
    parms.set_secret_key_hamming_weight(192);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys rk;
    keygen.create_relin_keys(rk);
    GaloisKeys gk;
    keygen.create_galois_keys(vector<int>{ 1 }, gk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    double scale = pow(2.0, 46);
    vector<double> v(encoder.slot_count(), 0.25);
    Plaintext pt;
    encoder.encode(v, scale, pt);
    const int rows = 64;
    vector<Ciphertext> X(rows);
    for (auto &c : X) encryptor.encrypt(pt, c);
    context.sync();
    auto t0 = chrono::steady_clock::now();
    const int reps = 4;
    for (int rep = 0; rep < reps; rep++)
    {
        Ciphertext out;
        Plaintext w;
        encoder.encode(0.5, X[0].parms_id(), X[0].scale(), w);
        evaluator.multiply_plain(X[0], w, out);
        for (int j = 1; j < rows; j++)
        {
            Plaintext wj;
            encoder.encode(0.01 * j, X[j].parms_id(), X[j].scale(), wj);
            Ciphertext temp;
            evaluator.multiply_plain(X[j], wj, temp);
            evaluator.add_inplace(out, temp);
        }
        evaluator.rescale_to_next_inplace(out);
        out.scale() = scale;
    }
    context.sync();
    double ms = chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count() / reps;
    printf("ct-pt inner loop, %d rows: %.2f ms per output column (%.1f us per multiply_plain+add)\n", rows, ms, ms * 1e3 / rows);
    Ciphertext r = X[0];
    context.sync();
    t0 = chrono::steady_clock::now();
    for (int i = 0; i < 20; i++)
    {
        evaluator.rotate_vector_inplace(r, 1, gk);
    }
    context.sync();
    printf("rotate_vector_inplace: %.3f ms\n", chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count() / 20);
    Ciphertext m;
    t0 = chrono::steady_clock::now();
    for (int i = 0; i < 20; i++)
    {
        evaluator.multiply(X[0], X[1], m);
        evaluator.relinearize_inplace(m, rk);
        evaluator.rescale_to_next_inplace(m);
    }
    context.sync();
    printf("multiply + relinearize + rescale: %.3f ms\n", chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count() / 20);
    return 0;
}
