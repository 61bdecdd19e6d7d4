"""Scheme-level checks of the CPU oracle by decrypt-and-compare (the reference's own strategy for
these functions, native/tests/seal/evaluator.cpp:2971-4293), so that the key-switch / rescale /
Galois restatements are pinned by semantics on top of the unit KATs.  CPU only."""
import numpy as np
import pytest

import oracle as O
from ckks_toy import ToyClient, _negacyclic_mul, decode_slots, encode_slots, galois_coeffs

LOGN = 6
N = 1 << LOGN


@pytest.fixture(scope="module")
def env():
    # {51, 46, 46, 51, 58}: the bit sizes of MOAI's chain (test_full_scheme.hpp:356-378), so the
    # special prime is larger than every data prime (qk > qi branch, evaluator.cpp:2978-2986) and
    # data primes of both orders meet in the I/J loop (key_modulus[J] <= key_modulus[I], :2846).
    primes = O.coeff_modulus_create(N, [51, 46, 46, 51, 58])
    ctx = O.Context(LOGN, primes)
    cl = ToyClient(ctx, seed=7)
    return ctx, cl


def small_msg(rng, scale_bits):
    return [int(x) for x in rng.integers(-(1 << scale_bits), 1 << scale_bits, size=N)]


def test_encrypt_decrypt(env):
    ctx, cl = env
    rng = np.random.default_rng(1)
    m = small_msg(rng, 40)
    ct = cl.encrypt(m, 4)
    d = cl.decrypt(ct, 2, 4)
    assert max(abs(a - b) for a, b in zip(d, m)) < 64


def test_multiply_relinearize_rescale(env):
    ctx, cl = env
    rng = np.random.default_rng(2)
    L = 4
    m1, m2 = small_msg(rng, 40), small_msg(rng, 40)
    c1, c2 = cl.encrypt(m1, L), cl.encrypt(m2, L)
    c3 = ctx.multiply(c1, c2, L)
    prod = _negacyclic_mul(m1, m2, N)
    d3 = cl.decrypt(c3, 3, L)
    assert max(abs(a - b) for a, b in zip(d3, prod)) < (1 << 52)  # noise ~ N * 2^40 * e
    rk = cl.relin_key()
    c2r = ctx.relinearize(c3, rk, L)
    d2 = cl.decrypt(c2r, 2, L)
    # key-switch noise is ~ sum_J q_J * e / p: tiny against 2^80 products
    assert max(abs(a - b) for a, b in zip(d2, d3)) < (1 << 20)
    rs = ctx.rescale(c2r, 2, L)
    dr = cl.decrypt(rs, 2, L - 1)
    ql = ctx.primes[L - 1]
    assert max(abs(a - round(b / ql)) for a, b in zip(dr, d2)) <= N + 2

    sq = ctx.square(c1, L)
    dsq = cl.decrypt(sq, 3, L)
    assert max(abs(a - b) for a, b in zip(dsq, _negacyclic_mul(m1, m1, N))) < (1 << 52)


def test_general_size_multiply_and_relinearize(env):
    """Evaluator::multiply beyond 2 x 2 (SEAL/evaluator.cpp:862-900) and relinearize_internal's loop over sizes (:1385-1393),
    by decryption -- the reference's own strategy for them -- and against the 2 x 2 restatement"""
    ctx, cl = env
    rng = np.random.default_rng(12)
    L = 4
    m1, m2, m3 = small_msg(rng, 30), small_msg(rng, 30), small_msg(rng, 30)
    c1, c2, c3 = cl.encrypt(m1, L), cl.encrypt(m2, L), cl.encrypt(m3, L)
    assert (ctx.multiply_general(c1, 2, c2, 2, L) == ctx.multiply(c1, c2, L)).all()
    c12 = ctx.multiply(c1, c2, L)
    c123 = ctx.multiply_general(c12, 3, c3, 2, L)
    assert c123.shape == (4, L, N)
    assert (ctx.multiply_general(c3, 2, c12, 3, L) == c123).all()  # commutative: exact sums mod q
    want = _negacyclic_mul(_negacyclic_mul(m1, m2, N), m3, N)
    d4 = cl.decrypt(c123, 4, L)
    assert max(abs(a - b) for a, b in zip(d4, want)) < (1 << 80)  # noise ~ N^2 * 2^60 * e against 2^100 products
    # (3 x 3): the square of a size-3 ciphertext is what ckks_square falls back to (:1237-1241)
    c1212 = ctx.multiply_general(c12, 3, c12, 3, L)
    p12 = _negacyclic_mul(m1, m2, N)
    d5 = cl.decrypt(c1212, 5, L)
    assert max(abs(a - b) for a, b in zip(d5, _negacyclic_mul(p12, p12, N))) < (1 << 110)
    # size 4 -> 2 with the keys of s^2 and s^3 (RelinKeys::get_index(k) = k - 2, SEAL/relinkeys.h)
    s2 = cl._dyadic(cl.s_ntt, cl.s_ntt, cl.k)
    s3 = cl._dyadic(s2, cl.s_ntt, cl.k)
    keys = [cl.kswitch_key(s2), cl.kswitch_key(s3)]
    r2 = ctx.relinearize_general(c123, 4, 2, keys, L)
    d2 = cl.decrypt(r2, 2, L)
    assert max(abs(a - b) for a, b in zip(d2, d4)) < (1 << 20)
    r3 = ctx.relinearize_general(c123, 4, 3, keys, L)
    assert max(abs(a - b) for a, b in zip(cl.decrypt(r3, 3, L), d4)) < (1 << 20)
    # the size-3 case of the loop is relinearize
    assert (ctx.relinearize_general(c12, 3, 2, keys, L) == ctx.relinearize(c12, keys[0], L)).all()


@pytest.mark.parametrize("L", [4, 3, 2, 1])
def test_apply_galois_all_levels(env, L):
    ctx, cl = env
    rng = np.random.default_rng(3 + L)
    m = small_msg(rng, 30)
    ct = cl.encrypt(m, L)
    for elt in (5, 25, 2 * N - 1, O.galois_elt_from_step(LOGN, -1)):
        gk = cl.galois_key(elt)
        out = ctx.apply_galois(ct, L, elt, gk)
        d = cl.decrypt(out, 2, L)
        exp = galois_coeffs(m, elt, N)
        assert max(abs(a - b) for a, b in zip(d, exp)) < (1 << 16), (L, elt)


def test_rotation_convention_generator_5(env):
    """rotate_vector(step) with elt = 5^step must shift slots left by `step` (galois.cpp:53-95,
    ckks.cpp:36-50)."""
    ctx, cl = env
    L = 3
    scale = float(1 << 30)
    z = np.arange(1, N // 2 + 1, dtype=np.float64) + 0.5j
    m = encode_slots(z, N, scale)
    assert np.abs(decode_slots(m, N, scale) - z).max() < 1e-6
    ct = cl.encrypt(m, L)
    for step in (1, 3, -2):
        elt = O.galois_elt_from_step(LOGN, step)
        out = ctx.apply_galois(ct, L, elt, cl.galois_key(elt))
        got = decode_slots(cl.decrypt(out, 2, L), N, scale)
        assert np.abs(got - np.roll(z, -step)).max() < 1e-3, step
    out = ctx.apply_galois(ct, L, 2 * N - 1, cl.galois_key(2 * N - 1))
    got = decode_slots(cl.decrypt(out, 2, L), N, scale)
    assert np.abs(got - np.conj(z)).max() < 1e-3


def test_multiply_plain_add_sub_negate(env):
    ctx, cl = env
    rng = np.random.default_rng(11)
    L = 3
    m1, m2 = small_msg(rng, 20), small_msg(rng, 20)
    c1, c2 = cl.encrypt(m1, L), cl.encrypt(m2, L)
    pt = cl._to_ntt(m2, L)
    d = cl.decrypt(ctx.multiply_plain(c1, 2, L, pt), 2, L)
    assert max(abs(a - b) for a, b in zip(d, _negacyclic_mul(m1, m2, N))) < (1 << 32)
    d = cl.decrypt(ctx.add(c1, c2, 2, L), 2, L)
    assert max(abs(a - (x + y)) for a, x, y in zip(d, m1, m2)) < 64
    d = cl.decrypt(ctx.sub(c1, c2, 2, L), 2, L)
    assert max(abs(a - (x - y)) for a, x, y in zip(d, m1, m2)) < 64
    d = cl.decrypt(ctx.negate(c1, 2, L), 2, L)
    assert max(abs(a + x) for a, x in zip(d, m1)) < 64


def test_mod_drop_and_modraise(env):
    ctx, cl = env
    rng = np.random.default_rng(12)
    m = small_msg(rng, 20)
    ct = cl.encrypt(m, 4)
    low = ctx.mod_drop(ct, 2, 4, 3)
    assert low.shape == (2, 1, N) and (low[:, 0] == ct[:, 0]).all()
    d = cl.decrypt(low, 2, 1)
    assert max(abs(a - b) for a, b in zip(d, m)) < 64
    up = ctx.modraise(low, 4)
    # modraise lifts each polynomial's centred residue mod q0 to the larger modulus, so the raised
    # ciphertext decrypts to m + q0 * I(X) with a small integer polynomial I (Bootstrapper.cpp:2938-2992)
    du = cl.decrypt(up, 2, 4)
    q0 = ctx.primes[0]
    for a, b in zip(du, m):
        r = (a - b) % q0
        assert min(r, q0 - r) < 64
        assert abs(a) < q0 * (N + 2)
    # row 0 is unchanged by the lift
    assert (up[:, 0] == low[:, 0]).all()


def test_switch_key_is_linear_in_ct_and_matches_formula(env):
    """Independent restatement of evaluator.cpp:2724-3020 in exact integer arithmetic."""
    ctx, cl = env
    rng = np.random.default_rng(13)
    k = ctx.k
    for L in (4, 2):
        primes = ctx.primes
        key = O.uniform_rns(rng, primes, (k - 1, 2), N)
        target = O.uniform_rns(rng, primes[:L], (), N)
        ct = O.uniform_rns(rng, primes[:L], (2,), N)
        got = ctx.switch_key(ct, target, key, L)
        p = primes[k - 1]
        tabs = [O.Tables(LOGN, q) for q in primes]
        t_coef = [tabs[j].intt(target[j]) for j in range(L)]
        exp = np.empty_like(ct)
        for kk in range(2):
            acc = {}
            for I in list(range(L)) + [k - 1]:
                q = primes[I]
                tot = np.zeros(N, dtype=object)
                for J in range(L):
                    op = tabs[I].ntt(t_coef[J] % np.uint64(q)) if J != I else target[J]
                    tot = (tot + op.astype(object) * key[J, kk, I].astype(object)) % q
                acc[I] = tot
            last = tabs[k - 1].intt(np.array(acc[k - 1], dtype=np.uint64)).astype(object)
            last = (last + (p >> 1)) % p
            for i in range(L):
                q = primes[i]
                t = (last % q + (q - (p >> 1) % q)) % q
                t = tabs[i].ntt(np.array(t, dtype=np.uint64)).astype(object)
                v = ((acc[i] - t) * pow(p, -1, q)) % q
                exp[kk, i] = np.array((v + ct[kk, i].astype(object)) % q, dtype=np.uint64)
        assert (got == exp).all(), L
