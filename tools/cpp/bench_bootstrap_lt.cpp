// bench_bootstrap_lt.cpp -- the three linear transforms of MOAI's coefficient-to-slot step
// (Bootstrapper::sfl_full_3, include/source/bootstrapping/Bootstrapper.cpp:2460-2497: logn = 15 split 5 + 5 + 5,
// 63 + 63 + 32 diagonals, a rescale after each) at N = 2^16 on the 36-prime chain, starting at the top level
// like the ciphertext modraise hands over.  Timed twice on the same inputs:
//   per-op   the evaluator-call sequence the reference's Bootstrapper issues, one ciphertext per OpenMP thread
//            (rotate_vector, encode + mod_switch_to + multiply_plain per diagonal, add_inplace_reduced_error)
//   batched  moai_fused::BsgsLinearTransform: batched key switches, cached diagonals, one MAC pass per giant step
// and the two results are compared bit for bit.  Diagonals are random: only the shape matters for the timing.
#include <omp.h>

#include <chrono>
#include <complex>
#include <cstdio>
#include <random>
#include <set>

#include "seal/moai_bootstrap_lt.h"
#include "seal/seal.h"

using namespace seal;
using namespace std;
using moai_fused::giantstep;
using moai_fused::rotation;

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

// Bootstrapper.cpp:1997-2062 (rotated = false) and :2064-2129 (rotated = true), call for call
static void ref_lt(Evaluator &evaluator, const GaloisKeys &gal_keys, int Nh, Ciphertext &rtncipher, const Ciphertext &cipher, int totlen,
                   int basicstep, int coeff_logn, const vector<vector<complex<double>>> &fftcoeff, bool rotated)
{
    int gs = rotated ? giantstep(totlen + 1) : giantstep(2 * totlen + 1);
    int basicstart = rotated ? 0 : -totlen + gs * (int)floor((totlen + 0.0) / (gs + 0.0));
    int giantfirst = rotated ? 0 : -(int)floor((totlen + 0.0) / (gs + 0.0));
    int giantlast = rotated ? (int)floor((totlen + 0.0) / (gs + 0.0)) : (int)floor((2 * totlen + 0.0) / (gs + 0.0)) + giantfirst;
    int offset = rotated ? 0 : totlen;
    vector<Ciphertext> babyct(gs);
    Ciphertext giantct, tmpct, tmptmpct;
    bool tmpctbool = false;
    vector<complex<double>> rotatedcoeff;
    for (int i = basicstart; i < basicstart + gs; i++)
    {
        if (i == 0)
            babyct[i - basicstart] = cipher;
        else
            evaluator.rotate_vector(cipher, (Nh + i * basicstep) % Nh, gal_keys, babyct[i - basicstart]);
    }
    for (int i = giantfirst; i <= giantlast; i++)
    {
        bool giantbool = false;
        int jlast = i != giantlast ? basicstart + gs - 1 : totlen - i * gs;
        for (int j = basicstart; j <= jlast; j++)
        {
            rotation(coeff_logn, Nh, (-i) * gs * basicstep, fftcoeff[(i * gs + j) + offset], rotatedcoeff);
            evaluator.multiply_vector_reduced_error(babyct[j - basicstart], rotatedcoeff, tmptmpct);
            if (!giantbool)
            {
                giantct = tmptmpct;
                giantbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(giantct, tmptmpct);
        }
        if (i != 0)
        {
            evaluator.rotate_vector(giantct, (Nh + i * gs * basicstep) % Nh, gal_keys, tmptmpct);
            if (!tmpctbool)
            {
                tmpct = tmptmpct;
                tmpctbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(tmpct, tmptmpct);
        }
        else
        {
            if (!tmpctbool)
            {
                tmpct = giantct;
                tmpctbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(tmpct, giantct);
        }
    }
    rtncipher = tmpct;
}

static void collect_steps(int Nh, int totlen, int basicstep, bool rotated, set<int> &steps)
{
    int gs = rotated ? giantstep(totlen + 1) : giantstep(2 * totlen + 1);
    int basicstart = rotated ? 0 : -totlen + gs * (int)floor((totlen + 0.0) / (gs + 0.0));
    int giantfirst = rotated ? 0 : -(int)floor((totlen + 0.0) / (gs + 0.0));
    int giantlast = rotated ? (int)floor((totlen + 0.0) / (gs + 0.0)) : (int)floor((2 * totlen + 0.0) / (gs + 0.0)) + giantfirst;
    for (int i = basicstart; i < basicstart + gs; i++)
        if (i) steps.insert((Nh + i * basicstep) % Nh);
    for (int i = giantfirst; i <= giantlast; i++)
        if (i) steps.insert((Nh + i * gs * basicstep) % Nh);
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
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
    PublicKey pk;
    keygen.create_public_key(pk);
    CKKSEncoder encoder(context);
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context, encoder);
    const int Nh = (int)encoder.slot_count();
    const int logn = 15;
    // sfl_full_3's split (Bootstrapper.cpp:2461-2471)
    const int div_part3 = (int)floor(logn / 3.0), div_part2 = (int)floor((logn - div_part3) / 2.0), div_part1 = logn - div_part3 - div_part2;
    const int totlen[3] = { (1 << div_part1) - 1, (1 << div_part2) - 1, (1 << div_part3) - 1 };
    const int basicstep[3] = { 1, 1 << div_part1, 1 << (div_part1 + div_part2) };
    const bool rotated[3] = { false, false, true };
    set<int> steps;
    for (int s = 0; s < 3; s++) collect_steps(Nh, totlen[s], basicstep[s], rotated[s], steps);
    double t0 = now_s();
    GaloisKeys gal_keys;
    keygen.create_galois_keys(vector<int>(steps.begin(), steps.end()), gal_keys); // what addBootKeys_3 does for its steps
    context.sync();
    printf("%zu Galois keys for the transform's rotations: %.1f s\n", steps.size(), now_s() - t0);

    mt19937_64 rng(3);
    uniform_real_distribution<double> ud(-1.0, 1.0);
    vector<vector<vector<complex<double>>>> coeff(3);
    vector<unique_ptr<moai_fused::BsgsLinearTransform>> lt;
    for (int s = 0; s < 3; s++)
    {
        int nd = rotated[s] ? totlen[s] + 1 : 2 * totlen[s] + 1;
        coeff[s].assign(nd, vector<complex<double>>((size_t)1 << logn));
        for (auto &d : coeff[s])
            for (auto &z : d) z = { ud(rng) * 0.1, ud(rng) * 0.1 };
        lt.emplace_back(new moai_fused::BsgsLinearTransform(context, Nh, totlen[s], basicstep[s], logn, coeff[s], rotated[s]));
        printf("stage %d: totlen %d, basic step %d, %zu diagonals, %zu key switches per ciphertext\n", s + 1, totlen[s], basicstep[s],
               lt[s]->diagonal_count(), lt[s]->key_switches_per_ciphertext(gal_keys));
    }
    const double scale = pow(2.0, 46);
    vector<Ciphertext> in(B);
    {
        vector<complex<double>> v(Nh);
        for (auto &z : v) z = { ud(rng), ud(rng) };
        Plaintext p;
        encoder.encode(v, scale, p);
        Ciphertext c;
        encryptor.encrypt(p, c);
        for (int b = 0; b < B; b++) in[b] = c;
    }
    context.sync();

    // ---- batched: first call encodes the diagonals, second call is the steady state ---------------------------
    vector<Ciphertext> outb;
    double t_first = 0, t_steady = 0;
    for (int rep = 0; rep < 2; rep++)
    {
        t0 = now_s();
        vector<Ciphertext> cur = in, nxt;
        for (int s = 0; s < 3; s++)
        {
            lt[s]->apply(cur, nxt, gal_keys);
            for (auto &c : nxt) evaluator.rescale_to_next_inplace(c);
            cur.swap(nxt);
        }
        context.sync();
        (rep == 0 ? t_first : t_steady) = now_s() - t0;
        outb = cur;
    }
    printf("batched, %d ciphertexts: first call %.3f s (encodes and caches %zu diagonals), then %.3f s = %.2f ms per ciphertext\n", B, t_first,
           lt[0]->diagonal_count() + lt[1]->diagonal_count() + lt[2]->diagonal_count(), t_steady, t_steady * 1e3 / B);

    // ---- per-op, as the reference's Bootstrapper issues it, one ciphertext per thread ---------------------------
    const int Bref = min(B, 16);
    vector<Ciphertext> outr(Bref);
    t0 = now_s();
#pragma omp parallel for
    for (int b = 0; b < Bref; b++)
    {
        Ciphertext cur = in[b], nxt;
        for (int s = 0; s < 3; s++)
        {
            ref_lt(evaluator, gal_keys, Nh, nxt, cur, totlen[s], basicstep[s], logn, coeff[s], rotated[s]);
            evaluator.rescale_to_next_inplace(nxt);
            cur = nxt;
        }
        outr[b] = cur;
    }
    context.sync();
    double t_ref = now_s() - t0;
    printf("per-op (the Bootstrapper's call sequence, %d OpenMP threads), %d ciphertexts: %.3f s = %.2f ms per ciphertext\n", threads, Bref,
           t_ref, t_ref * 1e3 / Bref);
    bool same = true;
    for (int b = 0; b < Bref; b++) same = same && outr[b].download() == outb[b].download() && outr[b].scale() == outb[b].scale();
    printf("results: %s; output chain index %zu\n", same ? "bit-identical" : "DIFFERENT",
           context.get_context_data(outb[0].parms_id())->chain_index());
    return same ? 0 : 1;
}
