// test_bootstrap_eval.cpp -- seal/moai_bootstrap_eval.h: the evaluation half of Bootstrapper::bootstrap_full_3 on
// packed ciphertexts.  The reference's Bootstrapper cannot be built here (NTL), so nothing below compares with its
// output ("parity unpinned" for the constants); what is checked:
//   1. host only (also run by the CPU test suite with --host-only): babycount's choice for MOAI's degree 59, the
//      shape of the quotient / remainder heap, and that recombining the leaves gives back the polynomial;
//   2. ModularReducer3::modular_reduction decrypts to the value of the polynomial with its double-angle steps,
//      which is sin(2 pi x) / (2 pi) near the integers -- the function bootstrapping needs;
//   3. the same call on a packed ciphertext is bit-identical to the call on each ciphertext;
//   4. PackedBootstrapper3::bootstrap_3 on a pack is bit-identical to the Bootstrapper's sequence of evaluator
//      calls made per ciphertext (linear transforms through rotate_vector / multiply_vector_reduced_error /
//      add_inplace_reduced_error as in Bootstrapper.cpp:1997-2129, everything else as in :2460-2777, :3231-3251).
#include <complex>
#include <cstdio>
#include <cstring>
#include <random>

#include "seal/moai_bootstrap_eval.h"
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

using moai_fused::ChebyshevHeap;
using moai_fused::ModularReducer3;

static vector<double> cosine_coefficients(long K, long deg, long r)
{
    // RemezCos::function_value (include/source/bootstrapping/RemezCos.h:12-17) in the variable x / K (Remez.cpp:196)
    const double two_pi = 2 * M_PI, sf = (double)(1L << r);
    return moai_fused::chebyshev_interpolant([=](double t) { return cos(two_pi * (K * t - 0.25) / sf); }, deg, 4 * deg);
}

static void host_checks()
{
    long k = 0, m = 0;
    moai_fused::babycount(k, m, 59);
    CHECK(k == 8 && m == 3); // MOAI's deg = 59 (test_full_scheme.hpp:346)
    moai_fused::babycount(k, m, 7);
    CHECK((k << m) >= 7);
    for (long deg : { 59L, 31L, 30L, 12L, 7L, 5L, 4L })
    {
        ChebyshevHeap heap(cosine_coefficients(6, deg, 2));
        CHECK((heap.heap_k() << heap.heap_m()) >= deg);
        const auto &nodes = heap.nodes();
        CHECK(nodes.size() == ((size_t)1 << (heap.heap_m() + 1)) - 1);
        // a leaf has degree at most heap_k (T_k itself is giant[0], common/Polynomial.cpp:474-481)
        for (size_t i = ((size_t)1 << heap.heap_m()) - 1; i < nodes.size(); i++)
        {
            if (nodes[i].present) CHECK(nodes[i].deg() <= heap.heap_k());
        }
        double err = 0;
        for (int s = 0; s <= 200; s++)
        {
            double x = -1.0 + s * 0.01;
            err = max(err, fabs(heap.heap_value(x) - heap.value(x)));
        }
        printf("degree %2ld: k = %ld, m = %ld, heap recombination error %.2e\n", deg, heap.heap_k(), heap.heap_m(), err);
        CHECK(err < 1e-12);
    }
    {
        ChebyshevHeap heap(cosine_coefficients(25, 59, 2));
        const auto &nodes = heap.nodes();
        // 59 = 27 * T32 + 31; 27 -> 11, 15; 31 -> 15, 15; then 3 / 7 and 7 / 7 (common/Polynomial.cpp:183-204)
        CHECK(nodes[1].deg() == 27 && nodes[2].deg() == 31);
        CHECK(nodes[3].deg() == 11 && nodes[4].deg() == 15 && nodes[5].deg() == 15 && nodes[6].deg() == 15);
        CHECK(nodes[7].deg() == 3 && nodes[8].deg() == 7);
    }
    // the stand-in polynomial with its double-angle steps is the function bootstrapping needs
    {
        const long K = 12;
        ModularReducer3 red(cosine_coefficients(K, 59, 2), 1 / (2 * M_PI), 2);
        double err = 0;
        for (long I = -K + 1; I < K; I++)
            for (double e : { -0.01, -0.001, 0.0, 0.002, 0.01 })
            {
                double x = I + e;
                err = max(err, fabs(red.value(x / K) - sin(2 * M_PI * x) / (2 * M_PI)));
            }
        printf("modular reduction polynomial (K = %ld, degree 59, 2 double-angle steps) vs sin(2 pi x) / 2 pi: %.2e\n", K, err);
        CHECK(err < 1e-9);
    }
    // key list of addLeftRotKeys_Linear_to_vector_3 at MOAI's shape: 15 powers of two and the conjugation entry come
    // from the caller (test_full_scheme.hpp:436-441); the function adds the steps the transforms use directly
    {
        vector<int> steps{ 0 };
        for (int i = 0; i < 15; i++) steps.push_back(1 << i);
        moai_fused::boot_rotation_steps_3(15, 15, steps);
        printf("rotation keys for bootstrap_3 at logn = 15: %zu (16 from the caller)\n", steps.size());
        CHECK(steps.size() > 16 && steps.size() < 80);
        for (int s : steps) CHECK(s >= 0 && s < (1 << 15));
    }
}

int main(int argc, char **argv)
{
    host_checks();
    if (argc > 1 && !strcmp(argv[1], "--host-only"))
    {
        if (!g_fail) printf("ALL PASS\n");
        return g_fail ? 1 : 0;
    }

    EncryptionParameters parms(scheme_type::ckks);
    const size_t n = 1024;
    const int logn = 9;
    parms.set_poly_modulus_degree(n);
    vector<int> bits(20, 42);
    bits.push_back(50);
    parms.set_coeff_modulus(CoeffModulus::Create(n, bits));
    parms.set_secret_key_hamming_weight(32);
    SEALContext context(parms, true, sec_level_type::none);
    KeyGenerator keygen(context);
    SecretKey sk = keygen.secret_key();
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys relin_keys;
    keygen.create_relin_keys(relin_keys);
    vector<int> steps{ 0 };
    for (int i = 0; i < logn; i++) steps.push_back(1 << i);
    moai_fused::boot_rotation_steps_3(logn, logn, steps);
    GaloisKeys gal_keys;
    keygen.create_galois_keys(steps, gal_keys);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, sk);
    Evaluator evaluator(context, encoder);
    const double scale = pow(2.0, 42);
    const int Nh = (int)encoder.slot_count();
    refslice::Bootstrapper ref(logn, logn, scale, context, encoder, evaluator, gal_keys);
    auto ref_bsgs = [&](Evaluator &, GaloisKeys &, int, Ciphertext &out, Ciphertext &in, int totlen, int basicstep, int coeff_logn,
                        const vector<vector<complex<double>>> &coeff) { ref.bsgs_linear_transform(out, in, totlen, basicstep, coeff_logn, coeff); };
    auto ref_rotated_bsgs = [&](Evaluator &, GaloisKeys &, int, Ciphertext &out, Ciphertext &in, int totlen, int basicstep, int coeff_logn,
                                const vector<vector<complex<double>>> &coeff) {
        ref.rotated_bsgs_linear_transform(out, in, totlen, basicstep, coeff_logn, coeff);
    };
    mt19937_64 rng(11);
    uniform_real_distribution<double> ud(-1.0, 1.0);

    // ---- 2, 3: modular reduction ---------------------------------------------------------------------------
    const long K = 12;
    ModularReducer3 reducer(cosine_coefficients(K, 59, 2), 1 / (2 * M_PI), 2);
    {
        const int B = 3;
        vector<vector<double>> x(B, vector<double>(Nh));
        vector<Ciphertext> cts(B);
        for (int b = 0; b < B; b++)
        {
            for (int s = 0; s < Nh; s++)
            {
                long I = (long)(rng() % (2 * K - 1)) - (K - 1);
                x[b][s] = (I + ud(rng) * 0.004) / K;
            }
            Plaintext p;
            encoder.encode(x[b], scale, p);
            encryptor.encrypt(p, cts[b]);
        }
        // the fused multiply_const + rescale the evaluation uses is the two calls, bit for bit (also with the scale
        // override of the *_reduced_error compositions, and on a pack)
        {
            Ciphertext two, one, pk2, pk1;
            evaluator.multiply_const(cts[0], -0.37, two);
            evaluator.rescale_to_next_inplace(two);
            evaluator.multiply_const_rescale(cts[0], -0.37, one);
            CHECK(one.parms_id() == two.parms_id() && one.scale() == two.scale() && one.download() == two.download());
            evaluator.multiply_const(cts[1], 1.25e-3, two);
            two.scale() = scale * 3.0;
            evaluator.rescale_to_next_inplace(two);
            evaluator.multiply_const_rescale(cts[1], 1.25e-3, one, scale * 3.0);
            CHECK(one.scale() == two.scale() && one.download() == two.download());
            Ciphertext packed = moai_fused::pack(cts, context);
            evaluator.multiply_const(packed, 0.5, pk2);
            evaluator.rescale_to_next_inplace(pk2);
            evaluator.multiply_const_rescale(packed, 0.5, pk1);
            CHECK(pk1.batch() == pk2.batch() && pk1.download() == pk2.download());
        }
        vector<Ciphertext> single(B);
        for (int b = 0; b < B; b++) reducer.modular_reduction(evaluator, relin_keys, single[b], cts[b]);
        Ciphertext packed = moai_fused::pack(cts, context), packed_out;
        reducer.modular_reduction(evaluator, relin_keys, packed_out, packed);
        vector<Ciphertext> unpacked;
        moai_fused::unpack(packed_out, context, unpacked);
        CHECK(unpacked.size() == (size_t)B);
        for (int b = 0; b < B; b++)
        {
            CHECK(unpacked[b].parms_id() == single[b].parms_id());
            CHECK(unpacked[b].scale() == single[b].scale());
            CHECK(unpacked[b].download() == single[b].download());
        }
        const size_t used = context.first_context_data()->chain_index() - context.get_context_data(single[0].parms_id())->chain_index();
        // degree 59 as k = 8, m = 3 costs 6 levels (the quotient leaves have degree 3, so the deepest product is
        // one level shallower than k and m suggest), the two double-angle steps 2: with 3 + 3 for the linear parts
        // that is MOAI's boot_level = 14 (test_full_scheme.hpp:364)
        CHECK(used == 8);
        double err = 0, err_fn = 0;
        for (int b = 0; b < B; b++)
        {
            Plaintext p;
            vector<double> dec;
            decryptor.decrypt(single[b], p);
            encoder.decode(p, dec);
            for (int s = 0; s < Nh; s++)
            {
                err = max(err, fabs(dec[s] - reducer.value(x[b][s])));
                err_fn = max(err_fn, fabs(dec[s] - sin(2 * M_PI * K * x[b][s]) / (2 * M_PI)));
            }
        }
        printf("modular_reduction: %zu levels, max |decrypted - polynomial| %.2e, max |decrypted - sin(2 pi x)/(2 pi)| %.2e\n", used, err, err_fn);
        CHECK(err < 1e-4);
        CHECK(err_fn < 1e-4);
    }

    // ---- 4: the whole bootstrap_3 sequence, packed against per-ciphertext calls -----------------------------------
    {
        const int p = logn / 3, totlen = (1 << p) - 1, slotlen = 1 << logn;
        auto random_set = [&](int count) {
            vector<vector<complex<double>>> c(count, vector<complex<double>>(slotlen));
            for (auto &d : c)
                for (auto &z : d) z = { ud(rng) * 0.3, ud(rng) * 0.3 };
            return c;
        };
        moai_fused::BootDiagonals3 dg;
        dg.invfftcoeff1 = random_set(2 * totlen + 1);
        dg.invfftcoeff2 = random_set(2 * totlen + 1);
        dg.invfftcoeff3 = random_set(2 * totlen + 1);
        dg.fftcoeff1 = random_set(2 * totlen + 1);
        dg.fftcoeff2 = random_set(2 * totlen + 1);
        dg.fftcoeff3 = random_set(2 * totlen + 1);
        const double final_scale = scale;
        moai_fused::PackedBootstrapper3 boot(context, encoder, evaluator, relin_keys, gal_keys, logn, logn, final_scale, dg, reducer);

        const int B = 2;
        vector<Ciphertext> cts(B);
        for (int b = 0; b < B; b++)
        {
            vector<complex<double>> v(Nh);
            for (auto &z : v) z = { ud(rng) * 0.01, ud(rng) * 0.01 };
            Plaintext pl;
            encoder.encode(v, scale, pl);
            encryptor.encrypt(pl, cts[b]);
            evaluator.mod_switch_to_inplace(cts[b], context.last_parms_id());
        }
        Ciphertext packed = moai_fused::pack(cts, context), packed_out;
        boot.bootstrap_3(packed_out, packed);
        vector<Ciphertext> got;
        moai_fused::unpack(packed_out, context, got);

        // the reference's sequence on one ciphertext
        const auto &modulus = context.first_context_data()->parms().coeff_modulus();
        const int bs_inv[3] = { 1 << (logn - p), 1 << (logn - 2 * p), 1 }, bs_fwd[3] = { 1, 1 << p, 1 << (2 * p) };
        for (int b = 0; b < B; b++)
        {
            Ciphertext cipher = cts[b];
            const double initial_scale = cipher.scale();
            boot.modraise_inplace(cipher); // moai_modraise has its own parity test against the oracle
            cipher.scale() = (double)modulus[0].value();
            // coefftoslot_full_3
            Ciphertext t1, t2, t3, t4, rtn1, rtn2, a, c;
            ref_rotated_bsgs(evaluator, gal_keys, Nh, a, cipher, totlen, bs_inv[0], logn, dg.invfftcoeff1);
            evaluator.rescale_to_next_inplace(a);
            ref_bsgs(evaluator, gal_keys, Nh, c, a, totlen, bs_inv[1], logn, dg.invfftcoeff2);
            evaluator.rescale_to_next_inplace(c);
            ref_bsgs(evaluator, gal_keys, Nh, t1, c, totlen, bs_inv[2], logn, dg.invfftcoeff3);
            evaluator.rescale_to_next_inplace(t1);
            {
                vector<complex<double>> tmpvec(Nh, 0);
                for (auto &z : tmpvec) z -= complex<double>(0.0, 1.0);
                Plaintext tmpplain;
                encoder.encode(tmpvec, 1.0, tmpplain);
                evaluator.mod_switch_to_inplace(tmpplain, t1.parms_id());
                evaluator.multiply_plain(t1, tmpplain, t2);
            }
            evaluator.complex_conjugate(t2, gal_keys, t3);
            evaluator.complex_conjugate(t1, gal_keys, t4);
            evaluator.add_reduced_error(t1, t4, rtn1);
            evaluator.add_reduced_error(t2, t3, rtn2);
            Ciphertext m1, m2;
            reducer.modular_reduction(evaluator, relin_keys, m1, rtn1);
            reducer.modular_reduction(evaluator, relin_keys, m2, rtn2);
            // slottocoeff_full_3
            Ciphertext s1, s3;
            {
                vector<complex<double>> tmpvec(Nh, 0);
                for (auto &z : tmpvec) z += complex<double>(0.0, 1.0);
                Plaintext tmpplain;
                encoder.encode(tmpvec, 1.0, tmpplain);
                evaluator.mod_switch_to_inplace(tmpplain, m2.parms_id());
                evaluator.multiply_plain(m2, tmpplain, s1);
            }
            evaluator.add_reduced_error(m1, s1, s3);
            Ciphertext f1, f2, want;
            ref_bsgs(evaluator, gal_keys, Nh, f1, s3, totlen, bs_fwd[0], logn, dg.fftcoeff1);
            evaluator.rescale_to_next_inplace(f1);
            ref_bsgs(evaluator, gal_keys, Nh, f2, f1, totlen, bs_fwd[1], logn, dg.fftcoeff2);
            evaluator.rescale_to_next_inplace(f2);
            {
                auto curr_level = context.get_context_data(f2.parms_id())->chain_index();
                double mod_zero = (double)modulus[0].value();
                double curr_mod = (double)modulus[curr_level].value();
                vector<vector<complex<double>>> fftcoeff3_scale(2 * totlen + 1);
                for (int i = 0; i < totlen + 1; i++) fftcoeff3_scale[i].resize(slotlen);
                for (int i = 0; i < totlen + 1; i++)
                    for (int j = 0; j < slotlen; j++)
                        fftcoeff3_scale[i][j] = dg.fftcoeff3[i][j] * curr_mod * mod_zero * final_scale / (f2.scale() * f2.scale() * initial_scale);
                ref_rotated_bsgs(evaluator, gal_keys, Nh, want, f2, totlen, bs_fwd[2], logn, fftcoeff3_scale);
            }
            evaluator.rescale_to_next_inplace(want);
            want.scale() = final_scale;
            CHECK(got[b].parms_id() == want.parms_id());
            CHECK(got[b].scale() == want.scale());
            CHECK(got[b].download() == want.download());
            if (b == 0)
            {
                CHECK(context.first_context_data()->chain_index() - context.get_context_data(want.parms_id())->chain_index() == 14);
                printf("bootstrap_3 sequence: chain index %zu -> %zu (%zu levels)\n", context.first_context_data()->chain_index(),
                       context.get_context_data(want.parms_id())->chain_index(),
                       context.first_context_data()->chain_index() - context.get_context_data(want.parms_id())->chain_index());
            }
        }
        // refusals, as in the reference (Bootstrapper.cpp:2939-2945) and for shapes it leaves undefined
        {
            auto throws = [&](auto &&f) {
                try
                {
                    f();
                }
                catch (const std::invalid_argument &)
                {
                    return true;
                }
                return false;
            };
            Ciphertext top;
            {
                Plaintext pl;
                encoder.encode(0.5, scale, pl);
                encryptor.encrypt(pl, top);
            }
            CHECK(throws([&] { boot.modraise_inplace(top); })); // not at the lowest level
            Ciphertext out;
            CHECK(throws([&] { boot.bootstrap_3(out, top); }));
            CHECK(throws([&] {
                moai_fused::PackedBootstrapper3 sparse(context, encoder, evaluator, relin_keys, gal_keys, logn - 1, logn, final_scale, dg, reducer);
            })); // the sparse-slot driver (bootstrap_sparse_3) is not provided
            CHECK(throws([&] { ChebyshevHeap bad(vector<double>{ 1.0 }); }));
        }
        // a second pack reuses every cached diagonal set, including the rescaled third one
        Ciphertext packed2 = moai_fused::pack(cts, context), packed_out2;
        boot.bootstrap_3(packed_out2, packed2);
        CHECK(packed_out2.download() == packed_out.download());
    }
    if (!g_fail)
    {
        printf("ALL PASS\n");
    }
    return g_fail ? 1 : 0;
}
