// test_seal_shim.cpp -- exercises the seal:: surface (seal_shim/seal/seal.h) end to end on the GPU the
// way the reference's own evaluator tests do: encode -> encrypt -> evaluate -> decrypt -> decode and
// compare with the plain computation (native/tests/seal/evaluator.cpp:2971-4293), plus the exception
// behaviour MOAI relies on.  Prints "ALL PASS" on success.
#include <cstdio>
#include <iostream>

#include "seal/seal.h"
#include "seal/moai_fused.h"

#include <omp.h>
#include <random>

using namespace seal;
using namespace std;

static int g_checks = 0, g_fail = 0;
#define CHECK(cond)                                                          \
    do                                                                       \
    {                                                                        \
        g_checks++;                                                          \
        if (!(cond))                                                         \
        {                                                                    \
            g_fail++;                                                        \
            printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
        }                                                                    \
    } while (0)
#define CHECK_THROWS(expr, extype)                                           \
    do                                                                       \
    {                                                                        \
        g_checks++;                                                          \
        bool caught = false;                                                 \
        try                                                                  \
        {                                                                    \
            expr;                                                            \
        }                                                                    \
        catch (const extype &)                                               \
        {                                                                    \
            caught = true;                                                   \
        }                                                                    \
        catch (...)                                                          \
        {}                                                                   \
        if (!caught)                                                         \
        {                                                                    \
            g_fail++;                                                        \
            printf("CHECK_THROWS FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); \
        }                                                                    \
    } while (0)

static double max_err(const vector<double> &a, const vector<double> &b, size_t count)
{
    double m = 0;
    for (size_t i = 0; i < count; i++)
    {
        m = max(m, fabs(a[i] - b[i]));
    }
    return m;
}

// BASELINE.json configs[0] / SURVEY.md 8(d) config 1
static void config1()
{
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 8192;
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, { 60, 40, 60 }));
    SEALContext context(parms, true, sec_level_type::none);
    CHECK(parms.coeff_modulus()[0].value() == 1152921504606748673ULL);
    CHECK(parms.coeff_modulus()[1].value() == 1099511480321ULL);
    CHECK(parms.coeff_modulus()[2].value() == 1152921504606830593ULL);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context, encoder);
    double scale = pow(2.0, 40);
    vector<double> v(encoder.slot_count());
    for (size_t i = 0; i < v.size(); i++)
    {
        v[i] = i * 1e-3;
    }
    Plaintext pt, w;
    encoder.encode(v, scale, pt);
    Ciphertext ct;
    encryptor.encrypt(pt, ct);
    CHECK(context.get_context_data(ct.parms_id())->chain_index() == 1);
    encoder.encode(0.5, ct.parms_id(), scale, w);
    evaluator.multiply_plain_inplace(ct, w);
    evaluator.rescale_to_next_inplace(ct);
    CHECK(context.get_context_data(ct.parms_id())->chain_index() == 0);
    Plaintext res;
    decryptor.decrypt(ct, res);
    vector<double> out;
    encoder.decode(res, out);
    CHECK(fabs(out[1000] - 0.5) < 1e-5);
    vector<double> expect(v.size());
    for (size_t i = 0; i < v.size(); i++)
    {
        expect[i] = v[i] * 0.5;
    }
    CHECK(max_err(out, expect, v.size()) < 1e-4);
    CHECK_THROWS(evaluator.rescale_to_next_inplace(ct), invalid_argument); // end of chain
}

static void evaluator_ops()
{
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 4096;
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, { 60, 40, 40, 40, 60 }));
    parms.set_secret_key_hamming_weight(64);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    RelinKeys rk;
    GaloisKeys gk;
    keygen.create_public_key(pk);
    keygen.create_relin_keys(rk);
    keygen.create_galois_keys(gk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context, encoder);
    const size_t slots = encoder.slot_count();
    const double scale = pow(2.0, 40);
    vector<double> a(slots), b(slots), out;
    for (size_t i = 0; i < slots; i++)
    {
        a[i] = sin(0.01 * i) + 0.5;
        b[i] = cos(0.02 * i) - 0.25;
    }
    Plaintext pa, pb, pr;
    encoder.encode(a, scale, pa);
    encoder.encode(b, scale, pb);
    Ciphertext ca, cb, c;
    encryptor.encrypt(pa, ca);
    encryptor.encrypt(pb, cb);
    auto dec = [&](const Ciphertext &x) {
        decryptor.decrypt(x, pr);
        encoder.decode(pr, out);
    };
    vector<double> e(slots);

    // encode/decode and encrypt/decrypt round trip
    dec(ca);
    CHECK(max_err(out, a, slots) < 1e-6);

    // add / sub / negate
    evaluator.add(ca, cb, c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] + b[i];
    CHECK(max_err(out, e, slots) < 1e-6);
    evaluator.sub(ca, cb, c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] - b[i];
    CHECK(max_err(out, e, slots) < 1e-6);
    evaluator.negate(ca, c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = -a[i];
    CHECK(max_err(out, e, slots) < 1e-6);

    // multiply -> size 3 decrypts; relinearize; rescale
    evaluator.multiply(ca, cb, c);
    CHECK(c.size() == 3);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] * b[i];
    CHECK(max_err(out, e, slots) < 1e-5);
    // products of larger ciphertexts (the dest_size != 3 branch, SEAL/evaluator.cpp:862-900): 3 x 2 -> 4, the square of a
    // size-3 ciphertext -> 5; relinearizing them needs the keys of s^3, s^4, which create_relin_keys does not make
    {
        Ciphertext c3, c4, c4b, c5;
        vector<double> e2(slots);
        evaluator.multiply(ca, cb, c3);
        evaluator.multiply(c3, ca, c4);
        CHECK(c4.size() == 4);
        CHECK(fabs(c4.scale() / pow(2.0, 120) - 1.0) < 1e-9);
        dec(c4);
        for (size_t i = 0; i < slots; i++) e2[i] = a[i] * b[i] * a[i];
        CHECK(max_err(out, e2, slots) < 1e-4);
        evaluator.multiply(ca, c3, c4b);
        CHECK(c4b.download() == c4.download());
        evaluator.square(c3, c5);
        CHECK(c5.size() == 5);
        dec(c5);
        for (size_t i = 0; i < slots; i++) e2[i] = a[i] * b[i] * a[i] * b[i];
        CHECK(max_err(out, e2, slots) < 1e-4);
        bool threw = false;
        try
        {
            evaluator.relinearize_inplace(c4, rk);
        }
        catch (const std::invalid_argument &)
        {
            threw = true;
        }
        CHECK(threw);
    }
    evaluator.relinearize_inplace(c, rk);
    CHECK(c.size() == 2);
    evaluator.rescale_to_next_inplace(c);
    CHECK(context.get_context_data(c.parms_id())->chain_index() == 2);
    CHECK(fabs(c.scale() / pow(2.0, 40) - 1.0) < 1e-4);
    dec(c);
    CHECK(max_err(out, e, slots) < 1e-5);

    // sum of size-3 products then one relinearization (Ct_ct_matrix_mul.hpp:33-46)
    {
        Ciphertext acc, t;
        evaluator.multiply(ca, cb, acc);
        evaluator.multiply(cb, cb, t);
        evaluator.add_inplace(acc, t);
        evaluator.relinearize_inplace(acc, rk);
        evaluator.rescale_to_next_inplace(acc);
        dec(acc);
        for (size_t i = 0; i < slots; i++) e[i] = a[i] * b[i] + b[i] * b[i];
        CHECK(max_err(out, e, slots) < 1e-5);
    }

    // square
    evaluator.square(ca, c);
    evaluator.relinearize_inplace(c, rk);
    evaluator.rescale_to_next_inplace(c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] * a[i];
    CHECK(max_err(out, e, slots) < 1e-5);

    // rotations: power of two (key present), 3 (NAF: 4 - 1), negative, conjugate
    for (int steps : { 1, 3, -5, 256, 700 })
    {
        evaluator.rotate_vector(ca, steps, gk, c);
        dec(c);
        for (size_t i = 0; i < slots; i++) e[i] = a[(i + slots + steps) % slots];
        CHECK(max_err(out, e, slots) < 1e-5);
    }
    evaluator.complex_conjugate(ca, gk, c);
    dec(c);
    CHECK(max_err(out, a, slots) < 1e-5); // real input
    {
        GaloisKeys only1;
        keygen.create_galois_keys(vector<int>{ 1 }, only1);
        Ciphertext t = ca;
        CHECK_THROWS(evaluator.rotate_vector_inplace(t, 2, only1), invalid_argument); // Galois key not present
        evaluator.rotate_vector_inplace(t, 1, only1);
        dec(t);
        for (size_t i = 0; i < slots; i++) e[i] = a[(i + 1) % slots];
        CHECK(max_err(out, e, slots) < 1e-5);
    }

    // plain ops: vector and scalar plaintexts
    evaluator.add_plain(ca, pb, c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] + b[i];
    CHECK(max_err(out, e, slots) < 1e-6);
    evaluator.sub_plain(ca, pb, c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] - b[i];
    CHECK(max_err(out, e, slots) < 1e-6);
    evaluator.multiply_plain(ca, pb, c);
    evaluator.rescale_to_next_inplace(c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] * b[i];
    CHECK(max_err(out, e, slots) < 1e-5);
    {
        Plaintext s;
        encoder.encode(-1.75, ca.parms_id(), ca.scale(), s);
        evaluator.add_plain(ca, s, c);
        dec(c);
        for (size_t i = 0; i < slots; i++) e[i] = a[i] - 1.75;
        CHECK(max_err(out, e, slots) < 1e-6);
        evaluator.sub_plain(ca, s, c);
        dec(c);
        for (size_t i = 0; i < slots; i++) e[i] = a[i] + 1.75;
        CHECK(max_err(out, e, slots) < 1e-6);
        evaluator.multiply_plain(ca, s, c);
        evaluator.rescale_to_next_inplace(c);
        dec(c);
        for (size_t i = 0; i < slots; i++) e[i] = a[i] * -1.75;
        CHECK(max_err(out, e, slots) < 1e-5);
        vector<double> sd;
        encoder.decode(s, sd);
        CHECK(fabs(sd[0] + 1.75) < 1e-9 && fabs(sd[slots - 1] + 1.75) < 1e-9);
    }

    // level management
    {
        Ciphertext t = ca;
        auto last_id = context.last_parms_id();
        evaluator.mod_switch_to_inplace(t, last_id);
        CHECK(t.coeff_modulus_size() == 1 && t.parms_id() == last_id);
        dec(t);
        CHECK(max_err(out, a, slots) < 1e-6);
        CHECK_THROWS(evaluator.mod_switch_to_next_inplace(t), invalid_argument);
        CHECK_THROWS(evaluator.mod_switch_to_inplace(t, context.first_parms_id()), invalid_argument);
        Plaintext p2 = pa;
        evaluator.mod_switch_to_inplace(p2, last_id);
        CHECK_THROWS(evaluator.add_plain_inplace(c = ca, p2), invalid_argument); // parameter mismatch
        evaluator.add_plain_inplace(t, p2);
        dec(t);
        for (size_t i = 0; i < slots; i++) e[i] = 2 * a[i];
        CHECK(max_err(out, e, slots) < 1e-6);
    }

    // error behaviour MOAI depends on: scale mismatch is an exception (Ct_pt_matrix_mul.hpp:41 resets scale)
    {
        Ciphertext t;
        evaluator.multiply_plain(ca, pb, t);
        evaluator.rescale_to_next_inplace(t);
        Ciphertext u = ca;
        evaluator.mod_switch_to_next_inplace(u);
        CHECK_THROWS(evaluator.add_inplace(t, u), invalid_argument); // scale mismatch
        t.scale() = u.scale();
        evaluator.add_inplace(t, u);
        CHECK_THROWS(evaluator.add_inplace(t, ca), invalid_argument); // parms mismatch
    }

    // fork additions
    evaluator.add_const(ca, 2.5, c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] + 2.5;
    CHECK(max_err(out, e, slots) < 1e-6);
    evaluator.multiply_const(ca, -0.5, c);
    evaluator.rescale_to_next_inplace(c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] * -0.5;
    CHECK(max_err(out, e, slots) < 1e-5);
    evaluator.multiply_vector_reduced_error(ca, b, c);
    evaluator.rescale_to_next_inplace(c);
    dec(c);
    for (size_t i = 0; i < slots; i++) e[i] = a[i] * b[i];
    CHECK(max_err(out, e, slots) < 1e-5);
    {
        // operands at different levels
        Ciphertext lo;
        evaluator.multiply_const(cb, 1.0, lo);
        evaluator.rescale_to_next_inplace(lo); // one level below ca, scale ~2^40
        Ciphertext r;
        evaluator.add_reduced_error(ca, lo, r);
        dec(r);
        for (size_t i = 0; i < slots; i++) e[i] = a[i] + b[i];
        CHECK(max_err(out, e, slots) < 1e-4);
        // ... and bit for bit against the definition of add_inplace_reduced_error (fork, SEAL/evaluator.cpp:447-480) spelled
        // with the primitive calls: the shim lets the final addition ride on the rescale of the level adjustment
        // (moai_mul_scalar_rescale_add), in both operand orders
        {
            auto spelled = [&](const Ciphertext &high, const Ciphertext &low, bool high_first) {
                const double q_last = (double)context.get_context_data(high.parms_id())->parms().coeff_modulus().back().value();
                const double adjust = low.scale() * q_last / (high.scale() * high.scale());
                Plaintext padj;
                encoder.encode(adjust, high.scale(), padj);
                evaluator.mod_switch_to_inplace(padj, high.parms_id());
                Ciphertext t;
                evaluator.multiply_plain(high, padj, t);
                t.scale() = low.scale() * q_last;
                evaluator.rescale_to_next_inplace(t);
                Ciphertext res;
                if (high_first)
                {
                    // encrypted1 is the higher one: adjusted takes encrypted2's scale, then += encrypted2
                    t.scale() = low.scale();
                    evaluator.add_inplace(t, low);
                    res = t;
                }
                else
                {
                    // encrypted1 is the lower one: it takes adjusted's scale, then += adjusted
                    res = low;
                    res.scale() = t.scale();
                    evaluator.add_inplace(res, t);
                }
                return res;
            };
            Ciphertext want = spelled(ca, lo, true), got;
            evaluator.add_reduced_error(ca, lo, got);
            CHECK(got.parms_id() == want.parms_id() && got.scale() == want.scale());
            CHECK(got.download() == want.download());
            want = spelled(ca, lo, false);
            evaluator.add_reduced_error(lo, ca, got);
            CHECK(got.parms_id() == want.parms_id() && got.scale() == want.scale());
            CHECK(got.download() == want.download());
            // the accumulate form used by the polynomial evaluations: acc += rescale(x * c), and acc = rescale(x) + acc
            Ciphertext acc = lo, t2;
            evaluator.multiply_const_rescale(ca, 0.75, t2);
            Ciphertext ref = lo;
            ref.scale() = t2.scale();
            evaluator.add_inplace(ref, t2);
            CHECK(evaluator.rides_on_rescale_of(acc, ca));
            evaluator.multiply_const_rescale(ca, 0.75, acc, 0, &acc);
            CHECK(acc.scale() == ref.scale() && acc.download() == ref.download());
            acc = lo;
            evaluator.rescale_to_next(ca, t2);
            ref = lo;
            ref.scale() = t2.scale();
            evaluator.add_inplace(ref, t2);
            evaluator.rescale_to_next_add_inplace(ca, acc);
            CHECK(acc.scale() == ref.scale() && acc.download() == ref.download());
            CHECK(!evaluator.rides_on_rescale_of(ca, ca) && !evaluator.rides_on_rescale_of(ca, lo));
        }
        evaluator.sub_reduced_error(lo, ca, r);
        dec(r);
        for (size_t i = 0; i < slots; i++) e[i] = b[i] - a[i];
        CHECK(max_err(out, e, slots) < 1e-4);
        evaluator.multiply_reduced_error(ca, lo, rk, r);
        evaluator.rescale_to_next_inplace(r);
        dec(r);
        for (size_t i = 0; i < slots; i++) e[i] = a[i] * b[i];
        CHECK(max_err(out, e, slots) < 1e-4);
        evaluator.double_inplace(r);
        dec(r);
        for (size_t i = 0; i < slots; i++) e[i] = 2 * a[i] * b[i];
        CHECK(max_err(out, e, slots) < 2e-4);
    }

    // NTT form round trip on a ciphertext (Bootstrapper::modraise_inplace uses these)
    {
        Ciphertext t = ca;
        auto before = t.download();
        evaluator.transform_from_ntt_inplace(t);
        CHECK(!t.is_ntt_form());
        evaluator.transform_to_ntt_inplace(t);
        CHECK(t.download() == before);
        CHECK_THROWS(evaluator.transform_to_ntt_inplace(t), invalid_argument);
    }
}

// Concurrent callers: the shim coalesces key switches that arrive together (seal/moai_combiner.h).  Twelve host
// threads rotate, square + relinearize and rescale their own ciphertexts at the same time, several rounds; every
// result must equal the one computed by a single thread.
static void concurrent_callers()
{
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 4096;
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, { 51, 46, 46, 46, 58 }));
    parms.set_secret_key_hamming_weight(64);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys rk;
    keygen.create_relin_keys(rk);
    GaloisKeys gk;
    keygen.create_galois_keys(gk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    const int T = 12;
    const double scale = pow(2.0, 40);
    vector<Ciphertext> in(T);
    for (int t = 0; t < T; t++)
    {
        vector<double> v(encoder.slot_count());
        for (size_t s = 0; s < v.size(); s++) v[s] = 0.01 * t + 1e-4 * (double)(s % 97);
        Plaintext p;
        encoder.encode(v, scale, p);
        encryptor.encrypt(p, in[t]);
    }
    auto work = [&](const Ciphertext &x, int t) {
        Ciphertext a = x;
        evaluator.rotate_vector_inplace(a, 1, gk);            // same element for everybody
        evaluator.rotate_vector_inplace(a, 3 + (t % 2), gk);  // NAF path, two different sequences
        Ciphertext sq;
        evaluator.square(a, sq);
        evaluator.relinearize_inplace(sq, rk);
        evaluator.rescale_to_next_inplace(sq);
        evaluator.complex_conjugate_inplace(sq, gk);
        return sq;
    };
    vector<vector<uint64_t>> want(T);
    for (int t = 0; t < T; t++) want[t] = work(in[t], t).download();
    for (int round = 0; round < 5; round++)
    {
        util::RotationCache::instance().clear(); // the rotations below must reach the combiner, not the cache
        vector<vector<uint64_t>> got(T);
#pragma omp parallel for num_threads(T)
        for (int t = 0; t < T; t++)
        {
            got[t] = work(in[t], t).download();
        }
        for (int t = 0; t < T; t++) CHECK(got[t] == want[t]);
    }
    // what the gathering window costs a caller who is alone: nothing -- a leader only waits for company when several threads
    // have been calling lately (moai_combiner.h).  A hundred key switches from one thread must not take a window (500 us) each.
    std::this_thread::sleep_for(std::chrono::milliseconds(50)); // the parallel rounds above are "lately"
    {
        Ciphertext a = in[0], out;
        evaluator.rotate_vector(a, 1, gk, out);
        context.sync();
        const auto t0 = std::chrono::steady_clock::now();
        const int calls = 100;
        for (int i = 0; i < calls; i++)
        {
            util::RotationCache::instance().clear();
            evaluator.rotate_vector(a, 1, gk, out);
            (void)out.block_id(); // read it: a rotation nobody reads is never made
        }
        context.sync();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
        printf("single caller: %.0f us per key switch at N = 4096 (gathering window 500 us)\n", us);
        CHECK(us < 400.0);
    }
    util::RotationCache::instance().clear();
}

// Packed ciphertexts (moai_fused::pack): a random program of evaluator calls applied to a pack of four must give,
// for every member, what the same program gives on that member alone.
static void packed_random_program()
{
    EncryptionParameters parms(scheme_type::ckks);
    size_t n = 4096;
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, { 51, 46, 46, 46, 46, 51, 58 }));
    parms.set_secret_key_hamming_weight(64);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys rk;
    keygen.create_relin_keys(rk);
    GaloisKeys gk;
    keygen.create_galois_keys(gk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 46);
    const int B = 4;
    vector<Ciphertext> in(B);
    for (int b = 0; b < B; b++)
    {
        vector<double> v(encoder.slot_count());
        for (size_t s = 0; s < v.size(); s++) v[s] = 0.3 + 0.05 * b + 1e-4 * (double)(s % 31);
        Plaintext p;
        encoder.encode(v, scale, p);
        encryptor.encrypt(p, in[b]);
    }
    vector<double> pv(encoder.slot_count());
    for (size_t s = 0; s < pv.size(); s++) pv[s] = 0.5 - 1e-3 * (double)(s % 7);
    std::mt19937 rng(123);
    for (int trial = 0; trial < 6; trial++)
    {
        vector<int> prog(10);
        for (auto &o : prog) o = (int)(rng() % 9);
        auto run = [&](Ciphertext x) {
            Ciphertext y = x;
            for (int o : prog)
            {
                auto cd = context.get_context_data(x.parms_id());
                const bool can_rescale = cd->chain_index() > 1;
                switch (o)
                {
                case 0: evaluator.add_inplace(x, y); break;
                case 1: evaluator.sub_inplace(x, y); evaluator.add_inplace(x, y); evaluator.negate_inplace(x); evaluator.negate_inplace(x); break;
                case 2:
                    if (can_rescale)
                    {
                        evaluator.multiply_inplace(x, y);
                        evaluator.relinearize_inplace(x, rk);
                        evaluator.rescale_to_next_inplace(x);
                        x.scale() = scale;
                        evaluator.mod_switch_to_inplace(y, x.parms_id());
                    }
                    break;
                case 3: evaluator.rotate_vector_inplace(x, 1 + (int)(trial % 5), gk); break;
                case 4: evaluator.rotate_vector_inplace(x, -3, gk); break;
                case 5:
                    if (can_rescale)
                    {
                        Plaintext p;
                        encoder.encode(pv, x.parms_id(), x.scale(), p);
                        evaluator.multiply_plain_inplace(x, p);
                        evaluator.rescale_to_next_inplace(x);
                        x.scale() = scale;
                        evaluator.mod_switch_to_inplace(y, x.parms_id());
                    }
                    break;
                case 6:
                {
                    Plaintext p;
                    encoder.encode(0.125, x.parms_id(), x.scale(), p);
                    evaluator.add_plain_inplace(x, p);
                    Plaintext pvv;
                    encoder.encode(pv, x.parms_id(), x.scale(), pvv);
                    evaluator.sub_plain_inplace(x, pvv);
                    break;
                }
                case 7:
                    if (can_rescale)
                    {
                        evaluator.square_inplace(x);
                        evaluator.relinearize_inplace(x, rk);
                        evaluator.rescale_to_next_inplace(x);
                        x.scale() = scale;
                        evaluator.mod_switch_to_inplace(y, x.parms_id());
                    }
                    break;
                default: evaluator.complex_conjugate_inplace(x, gk); break;
                }
            }
            return x;
        };
        Ciphertext packed = moai_fused::pack(in, context);
        Ciphertext rp = run(packed);
        vector<Ciphertext> parts;
        moai_fused::unpack(rp, context, parts);
        CHECK(parts.size() == (size_t)B);
        for (int b = 0; b < B; b++)
        {
            Ciphertext r1 = run(in[b]);
            CHECK(r1.parms_id() == parts[b].parms_id());
            CHECK(r1.scale() == parts[b].scale());
            CHECK(r1.download() == parts[b].download());
        }
    }
}

// util::DevicePool: same-stream reuse, and the order in which a shortage gives blocks back (least recently released first)
static void device_pool()
{
    auto &pool = seal::util::DevicePool::instance();
    pool.trim();
    CHECK(pool.cached_bytes() == 0);
    const size_t MB = size_t(1) << 20;
    size_t ga = 0, gb = 0, gc = 0;
    void *a = pool.acquire(MB, nullptr, &ga), *b = pool.acquire(2 * MB, nullptr, &gb), *c = pool.acquire(MB, nullptr, &gc);
    CHECK(ga == MB && gb == 2 * MB && gc == MB);
    pool.release(a, ga, nullptr);
    pool.release(b, gb, nullptr);
    pool.release(c, gc, nullptr);
    CHECK(pool.cached_bytes() == 4 * MB);
    CHECK(pool.trim(MB) == MB); // a, the oldest, goes
    CHECK(pool.cached_bytes() == 3 * MB);
    size_t g1 = 0;
    void *c2 = pool.acquire(MB, nullptr, &g1);
    CHECK(c2 == c && g1 == MB); // c is still cached and is what a request of its size gets
    CHECK(pool.cached_bytes() == 2 * MB);
    size_t g2 = 0;
    void *b2 = pool.acquire(3 * MB / 2, nullptr, &g2); // best fit within 1.5 x the request
    CHECK(b2 == b && g2 == 2 * MB);
    CHECK(pool.cached_bytes() == 0);
    pool.release(c2, g1, nullptr);
    pool.release(b2, g2, nullptr);
    CHECK(pool.trim(MB + 1) == 3 * MB); // whole blocks, oldest first, until at least that much is back
    CHECK(pool.trim() == 0);
    // order of giving back: small blocks whose free list is not serving requests, then the blocks of 256 MiB and more, then the
    // small blocks of lists that served a request within the last second (what a running loop is cycling through)
    const size_t big = size_t(256) << 20;
    size_t gl = 0, gs1 = 0, gs2 = 0;
    std::this_thread::sleep_for(std::chrono::milliseconds(1200)); // the lists used above become idle
    void *s1 = pool.acquire(MB, nullptr, &gs1), *l = pool.acquire(big, nullptr, &gl), *s2 = pool.acquire(2 * MB, nullptr, &gs2);
    pool.release(s1, gs1, nullptr);
    pool.release(l, gl, nullptr);
    pool.release(s2, gs2, nullptr);
    s2 = pool.acquire(2 * MB, nullptr, &gs2); // the 2 MiB list serves a request: it is in use
    pool.release(s2, gs2, nullptr);
    CHECK(pool.trim(1) == MB); // the idle small list first
    CHECK(pool.cached_bytes() == big + 2 * MB);
    CHECK(pool.trim(1) == big); // then the large block
    CHECK(pool.cached_bytes() == 2 * MB);
    CHECK(pool.trim(1) == 2 * MB);
    CHECK(pool.cached_bytes() == 0);
    // ... and an idle list stops being protected
    s2 = pool.acquire(2 * MB, nullptr, &gs2);
    l = pool.acquire(big, nullptr, &gl);
    pool.release(s2, gs2, nullptr);
    s2 = pool.acquire(2 * MB, nullptr, &gs2);
    pool.release(s2, gs2, nullptr);
    pool.release(l, gl, nullptr);
    std::this_thread::sleep_for(std::chrono::milliseconds(1200));
    CHECK(pool.trim(1) == 2 * MB);
    CHECK(pool.trim() == big);
}

// Keys regenerated into an object that already served hoisted rotations (legal SEAL usage: KeyGenerator::create_galois_keys
// overwrites its destination).  The per-(key, level) constant of a hoisted rotation is cached in the key object
// (KSwitchKeys::hoist_correction); it is a function of the key's bits, so the new keys must not find the old keys' constants.
static void regenerated_keys_and_hoisting()
{
    EncryptionParameters parms(scheme_type::ckks);
    const size_t N = 4096;
    parms.set_poly_modulus_degree(N);
    parms.set_coeff_modulus(CoeffModulus::Create(N, { 51, 46, 46, 58 }));
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    Encryptor encryptor(context, pk);
    CKKSEncoder encoder(context);
    Evaluator evaluator(context, encoder);
    vector<double> v(encoder.slot_count());
    for (size_t i = 0; i < v.size(); i++) v[i] = 0.001 * (double)(i % 97);
    Plaintext p;
    encoder.encode(v, pow(2.0, 40), p);
    Ciphertext ct;
    encryptor.encrypt(p, ct);
    const size_t L = ct.coeff_modulus_size();
    const vector<int> steps = { 1, 2, 5 };
    GaloisKeys gk;
    auto hoisted_equals_separate = [&](const GaloisKeys &keys) {
        vector<Ciphertext> hoisted(steps.size());
        vector<uint32_t> elts;
        vector<const uint64_t *> kptr, cptr;
        vector<uint64_t *> optr;
        for (size_t r = 0; r < steps.size(); r++)
        {
            hoisted[r].resize(context, ct.parms_id(), 2);
            const uint32_t e = moai_galois_elt_from_step(context.device(), steps[r]);
            elts.push_back(e);
            kptr.push_back(keys.device_key(GaloisKeys::get_index(e)));
            cptr.push_back(keys.hoist_correction(context, GaloisKeys::get_index(e), e, L));
            optr.push_back(hoisted[r].device_data());
        }
        int fell_back = 0;
        util::hip_check(moai_apply_galois_hoisted(context.device(), ct.device_data(), optr.data(), L, elts.data(), kptr.data(), cptr.data(),
                                                  steps.size(), 1, &fell_back, context.stream()));
        CHECK(!fell_back);
        bool same = true;
        for (size_t r = 0; r < steps.size(); r++)
        {
            Ciphertext separate;
            evaluator.rotate_vector(ct, steps[r], keys, separate);
            same = same && separate.download() == hoisted[r].download();
        }
        return same;
    };
    keygen.create_galois_keys(steps, gk);
    CHECK(hoisted_equals_separate(gk));
    GaloisKeys earlier_copy = gk; // keeps the first keys AND their constants
    keygen.create_galois_keys(steps, gk); // fresh randomness: other key bits for the same secret
    CHECK(hoisted_equals_separate(gk));
    CHECK(hoisted_equals_separate(earlier_copy));
    GaloisKeys later_copy = gk;
    CHECK(hoisted_equals_separate(later_copy));
}

// Level-trimmed key residency (KSwitchKeys::limit_to_chain_index): rotations and relinearizations at or below the limit give
// the bits they gave with the full keys (they read the same words); one above the limit brings the full key back from its host
// copy -- same bits again -- and the device footprint is what the formula says.
static void trimmed_keys()
{
    EncryptionParameters parms(scheme_type::ckks);
    const size_t N = 4096;
    parms.set_poly_modulus_degree(N);
    parms.set_coeff_modulus(CoeffModulus::Create(N, { 51, 46, 46, 46, 51, 58 }));
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    Encryptor encryptor(context, pk);
    CKKSEncoder encoder(context);
    Evaluator evaluator(context, encoder);
    RelinKeys rk;
    keygen.create_relin_keys(rk);
    GaloisKeys gk;
    keygen.create_galois_keys(vector<int>{ 1, 2, 3, 8, -1 }, gk); // 7 = 8 - 1 in non-adjacent form
    vector<double> v(encoder.slot_count());
    for (size_t i = 0; i < v.size(); i++) v[i] = 0.001 * (double)(i % 89);
    Plaintext p;
    encoder.encode(v, pow(2.0, 40), p);
    Ciphertext top;
    encryptor.encrypt(p, top); // chain index 4: five data primes
    auto at_index = [&](size_t index) {
        Ciphertext c = top;
        while (context.get_context_data(c.parms_id())->chain_index() > index) evaluator.mod_switch_to_next_inplace(c);
        return c;
    };
    auto program = [&](size_t index) {
        // rotate by a step with its own key, by one without (NAF: 1 + 2 + ... ), square + relinearize
        Ciphertext c = at_index(index), r1, r2, sq;
        evaluator.rotate_vector(c, 3, gk, r1);
        evaluator.rotate_vector(c, 7, gk, r2);
        evaluator.square(c, sq);
        evaluator.relinearize_inplace(sq, rk);
        vector<vector<uint64_t>> out = { r1.download(), r2.download(), sq.download() };
        return out;
    };
    const auto full_low = program(1), full_high = program(3);
    const size_t bytes_full = gk.device_bytes();
    gk.limit_to_chain_index(context, 1);
    rk.limit_to_chain_index(context, 1);
    // 2 levels of 5 digits, 3 of 6 rows: 2 * 3 / (5 * 6) of the full size per key
    CHECK(gk.device_bytes() * 5 == bytes_full);
    CHECK(program(1) == full_low);
    CHECK(gk.regrown_count() == 0 && rk.regrown_count() == 0);
    // the hoisted path takes the trimmed key and a correction computed from it
    {
        Ciphertext c = at_index(1);
        const size_t L = c.coeff_modulus_size();
        vector<Ciphertext> hoisted(2);
        vector<uint32_t> elts;
        vector<const uint64_t *> kptr, cptr;
        vector<uint64_t *> optr;
        const int steps[2] = { 1, 2 };
        for (int r = 0; r < 2; r++)
        {
            hoisted[r].resize(context, c.parms_id(), 2);
            const uint32_t e = moai_galois_elt_from_step(context.device(), steps[r]);
            elts.push_back(e);
            kptr.push_back(gk.device_key(GaloisKeys::get_index(e), L));
            cptr.push_back(gk.hoist_correction(context, GaloisKeys::get_index(e), e, L));
            optr.push_back(hoisted[r].device_data());
        }
        int fell_back = 0;
        util::hip_check(moai_apply_galois_hoisted(context.device(), c.device_data(), optr.data(), L, elts.data(), kptr.data(), cptr.data(), 2, 1,
                                                  &fell_back, context.stream()));
        for (int r = 0; r < 2; r++)
        {
            Ciphertext separate;
            evaluator.rotate_vector(c, steps[r], gk, separate);
            CHECK(separate.download() == hoisted[r].download());
        }
    }
    // above the limit: the keys that are asked come back whole, the others stay trimmed
    CHECK(program(3) == full_high);
    CHECK(gk.regrown_count() == 3 && rk.regrown_count() == 1); // the keys of steps 3, 8 and -1; those of 1 and 2 were not asked
    CHECK(program(1) == full_low);
    // without a host copy a higher level is an error, not a wrong result
    GaloisKeys gk2;
    keygen.create_galois_keys(vector<int>{ 1 }, gk2);
    gk2.limit_to_chain_index(context, 0, false);
    Ciphertext c = at_index(2), out;
    CHECK_THROWS(evaluator.rotate_vector(c, 1, gk2, out), std::logic_error);
    c = at_index(0);
    evaluator.rotate_vector(c, 1, gk2, out);
}

// Copies share their device block until one of them is written through (copy on write), and single-ciphertext rotations are
// remembered by (source block, element, keys): both must be invisible -- every value is what the reference's deep copies and
// from-scratch rotations give.
static void shared_blocks_and_rotation_cache()
{
    EncryptionParameters parms(scheme_type::ckks);
    const size_t N = 4096;
    parms.set_poly_modulus_degree(N);
    parms.set_coeff_modulus(CoeffModulus::Create(N, { 51, 46, 46, 58 }));
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    Encryptor encryptor(context, pk);
    CKKSEncoder encoder(context);
    Evaluator evaluator(context, encoder);
    GaloisKeys gk;
    keygen.create_galois_keys(gk); // powers of two: other steps go through the non-adjacent form
    vector<double> v(encoder.slot_count());
    for (size_t i = 0; i < v.size(); i++) v[i] = 0.001 * (double)(i % 101);
    Plaintext p;
    encoder.encode(v, pow(2.0, 40), p);
    Ciphertext a;
    encryptor.encrypt(p, a);
    const vector<uint64_t> a_bits = a.download();

    // ---- copy on write ----
    Ciphertext b = a;
    CHECK(b.block_id() == a.block_id());            // shared
    evaluator.add_inplace(b, a);                      // b written: gets its own block, a untouched
    CHECK(b.block_id() != a.block_id());
    CHECK(a.download() == a_bits);
    Ciphertext c = a, d = a;
    evaluator.negate_inplace(c);                      // the written object moves away; the others keep the block
    CHECK(a.download() == a_bits && d.download() == a_bits && c.download() != a_bits);
    Ciphertext e = a;
    evaluator.mod_switch_to_next_inplace(e);
    CHECK(a.download() == a_bits && a.coeff_modulus_size() == e.coeff_modulus_size() + 1);
    {
        Ciphertext twice;
        evaluator.add(a, a, twice);
        CHECK(b.download() == twice.download());
    }

    // ---- rotation cache ----
    auto &cache = util::RotationCache::instance();
    if (cache.enabled())
    {
        cache.clear();
        const auto s0 = cache.statistics();
        Ciphertext r1, r2, r3;
        evaluator.rotate_vector(a, 7, gk, r1); // 7 = 8 - 1: two key switches, both misses
        CHECK(r1.is_deferred());               // ... once somebody reads the result (rotations are deferred, util::RotState)
        (void)r1.block_id();
        const auto s1 = cache.statistics();
        CHECK(s1.second - s0.second == 2 && s1.first == s0.first);
        evaluator.rotate_vector(a, 7, gk, r2); // the same chain from the same block: two hits, no device work
        (void)r2.block_id();
        const auto s2 = cache.statistics();
        CHECK(s2.first - s1.first == 2 && s2.second == s1.second);
        CHECK(r2.block_id() == r1.block_id() && r1.download() == r2.download());
        Ciphertext a_copy = a;
        evaluator.rotate_vector_inplace(a_copy, 9, gk); // 9 = 8 + 1: shares the first step (-1? no: +1) of nothing above, but...
        evaluator.rotate_vector(a, 15, gk, r3);          // 15 = 16 - 1: its first step (-1) was computed for 7
        (void)a_copy.block_id();
        (void)r3.block_id();
        const auto s3 = cache.statistics();
        CHECK(s3.first - s2.first >= 1);
        // writing into a rotation's result must not reach the cached block
        const vector<uint64_t> r1_bits = r1.download();
        evaluator.negate_inplace(r1);
        Ciphertext again;
        evaluator.rotate_vector(a, 7, gk, again);
        CHECK(again.download() == r1_bits && r2.download() == r1_bits);
        // ... and the values are what a computation from scratch gives
        cache.clear();
        Ciphertext fresh7, fresh15, fresh9 = a;
        evaluator.rotate_vector(a, 7, gk, fresh7);
        evaluator.rotate_vector(a, 15, gk, fresh15);
        cache.clear();
        evaluator.rotate_vector_inplace(fresh9, 9, gk);
        CHECK(fresh7.download() == r1_bits && fresh15.download() == r3.download() && fresh9.download() == a_copy.download());
        // several ciphertexts rotated by the same step and read afterwards (MOAI's Q K^T loop): made in one batched call
        {
            cache.clear();
            vector<Ciphertext> many(5), single(5);
            for (int i = 0; i < 5; i++)
            {
                vector<double> w(encoder.slot_count(), 0.01 * (i + 1));
                Plaintext pw;
                encoder.encode(w, pow(2.0, 40), pw);
                encryptor.encrypt(pw, many[i]);
                single[i] = many[i];
            }
            for (int i = 0; i < 5; i++)
            {
                evaluator.rotate_vector_inplace(single[i], 11, gk); // 11 = 8 + 2 + 1 (NAF: -1, -4, 16): one at a time
                (void)single[i].block_id();
            }
            cache.clear();
            for (int i = 0; i < 5; i++) evaluator.rotate_vector_inplace(many[i], 11, gk); // all pending ...
            for (int i = 0; i < 5; i++) CHECK(many[i].is_deferred());
            for (int i = 0; i < 5; i++) CHECK(many[i].download() == single[i].download()); // ... made together at the first read
            cache.clear();
        }
        // new keys in the same object: nothing computed with the old ones may come back
        evaluator.rotate_vector(a, 1, gk, r1);
        const vector<uint64_t> old_key_bits = r1.download();
        keygen.create_galois_keys(gk);
        evaluator.rotate_vector(a, 1, gk, r2);
        CHECK(r2.download() != old_key_bits);
        Plaintext pd;
        Decryptor decryptor(context, keygen.secret_key());
        decryptor.decrypt(r2, pd);
        vector<double> back;
        encoder.decode(pd, back);
        CHECK(fabs(back[0] - v[1]) < 1e-6 && fabs(back[5] - v[6]) < 1e-6);
        cache.clear();
    }
}

// Deferred scalar products (Ciphertext::LazyTerm): multiply_plain by a scalar-encoded plaintext followed by add_inplace -- the
// inner loop of MOAI's ct x pt products -- is recorded and computed when the sum is needed.  Whatever the program does in
// between, every value must be the one the eager sequence gives (in-place products are always eager: they serve as the comparator).
static void deferred_scalar_products()
{
    EncryptionParameters parms(scheme_type::ckks);
    const size_t N = 4096;
    parms.set_poly_modulus_degree(N);
    parms.set_coeff_modulus(CoeffModulus::Create(N, { 51, 46, 46, 58 }));
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    PublicKey pk;
    keygen.create_public_key(pk);
    Encryptor encryptor(context, pk);
    CKKSEncoder encoder(context);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 40);
    const int rows = 37; // not a multiple of the sixteen terms a pass takes
    vector<Ciphertext> X(rows);
    vector<Plaintext> W(rows);
    mt19937_64 rng(9);
    uniform_real_distribution<double> ud(-1.0, 1.0);
    for (int j = 0; j < rows; j++)
    {
        vector<double> v(encoder.slot_count());
        for (auto &x : v) x = ud(rng);
        Plaintext p;
        encoder.encode(v, scale, p);
        encryptor.encrypt(p, X[j]);
        encoder.encode(0.37 * ud(rng), X[j].parms_id(), X[j].scale(), W[j]); // scalar encode: constant rows
    }
    // eager comparator: in-place products
    Ciphertext eager;
    for (int j = 0; j < rows; j++)
    {
        Ciphertext t = X[j];
        evaluator.multiply_plain_inplace(t, W[j]);
        if (j == 0)
        {
            eager = t;
        }
        else
        {
            evaluator.add_inplace(eager, t);
        }
    }
    const vector<uint64_t> want = eager.download();
    // MOAI's loop (Ct_pt_matrix_mul.hpp:19-42), deferred
    Ciphertext out;
    evaluator.multiply_plain(X[0], W[0], out);
    CHECK(out.is_deferred());
    for (int j = 1; j < rows; j++)
    {
        Ciphertext temp;
        evaluator.multiply_plain(X[j], W[j], temp);
        evaluator.add_inplace(out, temp);
    }
    CHECK(out.is_deferred());
    Ciphertext copy_before = out; // a copy of a deferred value is deferred too, independently
    // a source changes after it was recorded: the record keeps the old residues
    const vector<uint64_t> x3 = X[3].download();
    evaluator.negate_inplace(X[3]);
    CHECK(X[3].download() != x3);
    CHECK(out.download() == want && !out.is_deferred());
    CHECK(copy_before.is_deferred());
    // read from several threads at once
    int equal = 0;
#pragma omp parallel for reduction(+ : equal) num_threads(8)
    for (int t = 0; t < 8; t++)
    {
        equal += (copy_before.download() == want) ? 1 : 0;
    }
    CHECK(equal == 8);
    // a deferred sum added to a ciphertext that already has residues; and what follows (rescale) sees the sum
    Ciphertext base = eager, part;
    evaluator.multiply_plain(X[5], W[5], part);
    evaluator.add_inplace(base, part);
    Ciphertext t5 = X[5];
    evaluator.multiply_plain_inplace(t5, W[5]);
    Ciphertext base_eager = eager;
    evaluator.add_inplace(base_eager, t5);
    CHECK(base.download() == base_eager.download());
    Ciphertext r1 = out, r2 = eager;
    evaluator.rescale_to_next_inplace(r1);
    evaluator.rescale_to_next_inplace(r2);
    CHECK(r1.download() == r2.download());
    // a product that is never added is still its value
    Ciphertext lone;
    evaluator.multiply_plain(X[7], W[7], lone);
    Ciphertext t7 = X[7];
    evaluator.multiply_plain_inplace(t7, W[7]);
    CHECK(lone.is_deferred() && lone.download() == t7.download());

    // ---- ciphertext x ciphertext: multiply + add_inplace chains (Ct_ct_matrix_mul.hpp:32-41) are deferred as well ----
    {
        const int pairs = 19;
        Ciphertext acc;
        evaluator.multiply(X[0], X[1], acc);
        CHECK(acc.is_deferred() && acc.size() == 3);
        for (int j = 1; j < pairs; j++)
        {
            Ciphertext temp;
            evaluator.multiply(X[j], X[j + 1], temp);
            evaluator.add_inplace(acc, temp);
        }
        CHECK(acc.is_deferred());
        // the comparator: the library's eager entry points on the same blocks
        const size_t L = X[0].coeff_modulus_size(), words3 = 3 * L * N;
        util::DeviceArray sum(words3, context.stream()), prod(words3, context.stream());
        for (int j = 0; j < pairs; j++)
        {
            const Ciphertext &a = X[j], &b = X[j + 1];
            util::hip_check(moai_ct_multiply(context.device(), a.device_data(), b.device_data(), j == 0 ? sum.get() : prod.get(), L, 1, context.stream()));
            if (j > 0)
            {
                util::hip_check(moai_add(context.device(), sum.get(), prod.get(), sum.get(), 3, L, context.stream()));
            }
        }
        vector<uint64_t> want3(words3);
        util::hip_check(moai_memcpy_d2h(want3.data(), sum.get(), words3 * 8, context.stream()));
        context.sync();
        CHECK(acc.download() == want3);
        // a square is never deferred, and a product of a ciphertext with a copy of itself is a square
        Ciphertext sq, same = X[2];
        evaluator.multiply(X[2], same, sq);
        CHECK(!sq.is_deferred());
    }

    // ---- the same with MOAI's masked weights: every weight a VECTOR  w * mask  (Ct_pt_matrix_mul.hpp:124-146) ----
    vector<int> mask(encoder.slot_count(), 0);
    for (size_t i = 0; i < mask.size(); i += 3) mask[i] = 1;
    auto masked = [&](double w) {
        vector<double> v(encoder.slot_count(), 0.0);
        for (size_t i = 0; i < v.size(); i++)
            if (mask[i]) v[i] = w;
        return v;
    };
    vector<double> weights(rows);
    for (auto &w : weights) w = 0.4 * ud(rng);
    Ciphertext eager_m;
    for (int j = 0; j < rows; j++)
    {
        Plaintext pw;
        encoder.encode(masked(weights[j]), X[j].parms_id(), X[j].scale(), pw);
        CHECK(pw.is_masked_constant());
        Ciphertext t = X[j];
        evaluator.multiply_plain_inplace(t, pw); // in place: the plaintext is transformed now, alone
        CHECK(!pw.is_masked_constant());
        if (j == 0)
        {
            eager_m = t;
        }
        else
        {
            evaluator.add_inplace(eager_m, t);
        }
    }
    Ciphertext out_m;
    for (int j = 0; j < rows; j++)
    {
        Plaintext pw;
        encoder.encode(masked(weights[j]), X[j].parms_id(), X[j].scale(), pw);
        if (j == 0)
        {
            evaluator.multiply_plain(X[j], pw, out_m);
        }
        else
        {
            Ciphertext temp;
            evaluator.multiply_plain(X[j], pw, temp);
            evaluator.add_inplace(out_m, temp);
        }
    } // every plaintext is gone by now: the records carry what the transforms need
    CHECK(out_m.is_deferred());
    CHECK(out_m.download() == eager_m.download());
    // a vector that is not of that form takes the ordinary path, and a masked plaintext used for an addition is its value
    {
        vector<double> general = masked(0.25);
        general[1] = 0.125;
        Plaintext pg, pm;
        encoder.encode(general, scale, pg);
        CHECK(!pg.is_masked_constant());
        encoder.encode(masked(0.25), scale, pm);
        CHECK(pm.is_masked_constant());
        Ciphertext zero;
        encryptor.encrypt_zero(zero);
        zero.scale() = scale;
        evaluator.add_plain_inplace(zero, pm);
        Plaintext back;
        Decryptor decryptor(context, keygen.secret_key());
        decryptor.decrypt(zero, back);
        vector<double> dec;
        encoder.decode(back, dec);
        CHECK(fabs(dec[0] - 0.25) < 1e-6 && fabs(dec[1]) < 1e-6 && fabs(dec[3] - 0.25) < 1e-6);
    }
}

int main()
{
    try
    {
        device_pool();
        config1();
        evaluator_ops();
        concurrent_callers();
        packed_random_program();
        regenerated_keys_and_hoisting();
        trimmed_keys();
        shared_blocks_and_rotation_cache();
        deferred_scalar_products();
    }
    catch (const std::exception &e)
    {
        printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
    printf("%d checks, %d failed\n", g_checks, g_fail);
    if (!g_fail)
    {
        printf("ALL PASS\n");
    }
    return g_fail ? 1 : 0;
}
