// test_bootstrap_lt.cpp -- moai_fused::BsgsLinearTransform (batched, cached diagonals) against the sequence of
// evaluator calls Bootstrapper::bsgs_linear_transform / ::rotated_bsgs_linear_transform make
// (include/source/bootstrapping/Bootstrapper.cpp:1997-2129), transcribed below call for call through the
// seal:: shim, and against the plaintext meaning of the transform.  The Bootstrapper itself cannot be
// compiled here (its headers need NTL); the two functions use nothing of it but Nh, the evaluator and the keys.
#include <complex>
#include <cstdio>
#include <random>

#include "seal/moai_bootstrap_lt.h"
#include "seal/seal.h"

using namespace seal;
using namespace std;

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

#include "ref_bootstrapper.h" // the reference's own transform routines, sliced from the checkout at build time

using namespace seal;
using namespace std;

int main()
{
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 4096;
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, { 51, 46, 46, 46, 51, 58 }));
    parms.set_secret_key_hamming_weight(64);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    SecretKey sk = keygen.secret_key();
    PublicKey pk;
    keygen.create_public_key(pk);
    GaloisKeys gal_keys;
    keygen.create_galois_keys(gal_keys); // powers of two: other steps take the NAF path, as in MOAI
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, sk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 46);
    const int Nh = (int)encoder.slot_count();
    refslice::Bootstrapper ref(11, 11, scale, context, encoder, evaluator, gal_keys);
    auto ref_bsgs = [&](Evaluator &, GaloisKeys &, int, Ciphertext &out, Ciphertext &in, int totlen, int basicstep, int coeff_logn,
                        const vector<vector<complex<double>>> &coeff) { ref.bsgs_linear_transform(out, in, totlen, basicstep, coeff_logn, coeff); };
    auto ref_rotated_bsgs = [&](Evaluator &, GaloisKeys &, int, Ciphertext &out, Ciphertext &in, int totlen, int basicstep, int coeff_logn,
                                const vector<vector<complex<double>>> &coeff) {
        ref.rotated_bsgs_linear_transform(out, in, totlen, basicstep, coeff_logn, coeff);
    };
    mt19937_64 rng(5);
    uniform_real_distribution<double> ud(-1.0, 1.0);

    const int B = 3;
    vector<vector<complex<double>>> msg(B, vector<complex<double>>(Nh));
    vector<Ciphertext> cts(B);
    for (int b = 0; b < B; b++)
    {
        for (auto &z : msg[b]) z = { ud(rng), ud(rng) };
        Plaintext p;
        encoder.encode(msg[b], scale, p);
        encryptor.encrypt(p, cts[b]);
        evaluator.mod_switch_to_next_inplace(cts[b]); // not the top level: the diagonals must drop rows too
    }

    struct Case
    {
        int totlen, basicstep, coeff_logn;
        bool rotated;
    };
    const Case cases[] = { { 7, 1, 11, false }, { 3, 8, 11, false }, { 15, 4, 10, false }, { 7, 32, 11, true }, { 5, 2, 9, true }, { 1, 1, 11, false } };
    for (const Case &cs : cases)
    {
        const int slotlen = 1 << cs.coeff_logn;
        const int nd = cs.rotated ? cs.totlen + 1 : 2 * cs.totlen + 1;
        vector<vector<complex<double>>> coeff(nd, vector<complex<double>>(slotlen));
        for (auto &d : coeff)
            for (auto &z : d) z = { ud(rng), ud(rng) };
        moai_fused::BsgsLinearTransform lt(context, Nh, cs.totlen, cs.basicstep, cs.coeff_logn, coeff, cs.rotated);
        vector<Ciphertext> got;
        lt.apply(cts, got, gal_keys);
        CHECK(got.size() == (size_t)B);
        for (int b = 0; b < B; b++)
        {
            Ciphertext want;
            if (cs.rotated)
                ref_rotated_bsgs(evaluator, gal_keys, Nh, want, cts[b], cs.totlen, cs.basicstep, cs.coeff_logn, coeff);
            else
                ref_bsgs(evaluator, gal_keys, Nh, want, cts[b], cs.totlen, cs.basicstep, cs.coeff_logn, coeff);
            CHECK(got[b].parms_id() == want.parms_id());
            CHECK(got[b].scale() == want.scale());
            CHECK(got[b].size() == want.size());
            CHECK(got[b].download() == want.download());
        }
        // a second application reuses the cached diagonals and must not change the result
        vector<Ciphertext> again;
        lt.apply(cts, again, gal_keys);
        CHECK(again[1].download() == got[1].download());
        // meaning: out[s] = sum_d coeff_d[s mod slotlen] * x[(s + d * basicstep) mod Nh]
        {
            Plaintext p;
            vector<complex<double>> dec;
            decryptor.decrypt(got[0], p);
            encoder.decode(p, dec);
            double err = 0;
            for (int s = 0; s < Nh; s += 37)
            {
                complex<double> e = 0;
                for (int k = 0; k < nd; k++)
                {
                    int d = cs.rotated ? k : k - cs.totlen;
                    e += coeff[k][s % slotlen] * msg[0][((s + d * cs.basicstep) % Nh + Nh) % Nh];
                }
                err = max(err, abs(dec[s] - e));
            }
            printf("totlen %2d step %2d logn %2d %s: %zu diagonals, %zu key switches per ciphertext, max |error| %.2e\n", cs.totlen,
                   cs.basicstep, cs.coeff_logn, cs.rotated ? "rotated" : "plain  ", lt.diagonal_count(),
                   lt.key_switches_per_ciphertext(gal_keys), err);
            CHECK(err < 1e-6);
        }
    }
    // a batch at another level and scale gets its own encodings
    {
        vector<vector<complex<double>>> coeff(3, vector<complex<double>>(Nh, complex<double>(0.5, 0.0)));
        moai_fused::BsgsLinearTransform lt(context, Nh, 1, 1, 11, coeff, false);
        vector<Ciphertext> a, b2;
        lt.apply(cts, a, gal_keys);
        vector<Ciphertext> lower = cts;
        for (auto &c : lower) evaluator.mod_switch_to_next_inplace(c);
        lt.apply(lower, b2, gal_keys);
        Ciphertext want;
        ref_bsgs(evaluator, gal_keys, Nh, want, lower[2], 1, 1, 11, coeff);
        CHECK(b2[2].download() == want.download());
        ref_bsgs(evaluator, gal_keys, Nh, want, cts[2], 1, 1, 11, coeff);
        CHECK(a[2].download() == want.download());
        // mixed levels are refused
        vector<Ciphertext> mixed{ cts[0], lower[1] };
        bool threw = false;
        try
        {
            lt.apply(mixed, a, gal_keys);
        }
        catch (const std::invalid_argument &)
        {
            threw = true;
        }
        CHECK(threw);
    }
    if (!g_fail)
    {
        printf("ALL PASS\n");
    }
    return g_fail ? 1 : 0;
}
