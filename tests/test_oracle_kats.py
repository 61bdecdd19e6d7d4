"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(tests/golden/seal_kats.json, transcribed from SEAL-4.1-bs native/tests/seal/**) and against the
constants SURVEY.md 8(c) captured from the real library.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle as O


def I(x):
    return int(x)


def arr(xs):
    return np.array([int(x) for x in xs], dtype=np.uint64)


def test_modulus_const_ratio(kats):
    for c in kats["modulus_const_ratio"]["cases"]:
        m = O.modulus(I(c["value"]))
        assert m.bit_count == c["bit_count"]
        assert [int(m.const_ratio[i]) for i in range(3)] == [I(x) for x in c["const_ratio"]]
        assert bool(O.lib().mo_is_prime(I(c["value"]))) == c["is_prime"]


def test_coeff_modulus_create(kats):
    for c in kats["coeff_modulus_create"]["cases"]:
        assert O.coeff_modulus_create(c["n"], c["bits"]) == [I(p) for p in c["primes"]]
    # modulus.cpp:247-258: bit sizes honoured, primes = 1 mod 2n
    cm = O.coeff_modulus_create(32, [30, 40, 30, 30, 40])
    assert [p.bit_length() for p in cm] == [30, 40, 30, 30, 40]
    assert all(p % 64 == 1 for p in cm)
    assert len(set(cm)) == 5


def test_is_prime(kats):
    for v, exp in kats["is_prime"]["cases"]:
        assert bool(O.lib().mo_is_prime(I(v))) == exp, v


def test_naf(kats):
    for v, cnt in kats["naf"]["cases"]:
        d = O.naf(v)
        assert sum(d) == v
        if cnt is not None:
            assert len(d) == cnt
        # non-adjacent: every digit is +-2^k, no two adjacent powers
        ks = sorted(abs(x).bit_length() - 1 for x in d)
        assert all(abs(x) == 1 << (abs(x).bit_length() - 1) for x in d)
        assert all(b - a >= 2 for a, b in zip(ks, ks[1:]))


def test_primitive_roots(kats):
    for q, r, deg, exp in kats["is_primitive_root"]["cases"]:
        m = O.modulus(I(q))
        assert bool(O.lib().mo_is_primitive_root(I(r), deg, C.byref(m))) == exp
    for q, deg, exp in kats["minimal_primitive_root"]["cases"]:
        m = O.modulus(I(q))
        out = C.c_uint64(0)
        assert O.lib().mo_try_minimal_primitive_root(deg, C.byref(m), C.byref(out)) == 1
        assert out.value == I(exp)


def test_barrett_and_mulmod(kats):
    L = O.lib()
    for q, lo, hi, exp in kats["barrett_reduce_128"]["cases"]:
        m = O.modulus(I(q))
        w = (C.c_uint64 * 2)(I(lo), I(hi))
        assert L.mo_barrett_reduce_128(w, C.byref(m)) == I(exp)
        assert ((I(hi) << 64) | I(lo)) % I(q) == I(exp)
    for q, a, b, exp in kats["multiply_uint_mod"]["cases"]:
        m = O.modulus(I(q))
        assert L.mo_multiply_uint_mod(I(a), I(b), C.byref(m)) == I(exp)
    for q, op, quo in kats["mulop_quotient"]["cases"]:
        m = O.modulus(I(q))
        y = O.mulop(I(op), m)
        assert y.operand == I(op) and y.quotient == I(quo)
    for q, x, yv, exp in kats["multiply_uint_mod_operand"]["cases"]:
        m = O.modulus(I(q))
        assert L.mo_multiply_uint_mod_op(I(x), O.mulop(I(yv), m), C.byref(m)) == I(exp)
    for q, x, yv, exp in kats["multiply_uint_mod_lazy"]["cases"]:
        m = O.modulus(I(q))
        assert L.mo_multiply_uint_mod_lazy(I(x), O.mulop(I(yv), m), C.byref(m)) == I(exp)


def test_barrett_64_random():
    rng = np.random.default_rng(5)
    L = O.lib()
    for q in [3, 13, 0xFFFFFFFFFFC0001, 70368698171393, 288230376147386369, (1 << 61) - 1]:
        m = O.modulus(q)
        for x in rng.integers(0, 1 << 63, size=200, dtype=np.uint64):
            x = int(x) * 2 + 1
            assert L.mo_barrett_reduce_64(x, C.byref(m)) == x % q


def test_poly_ops(kats):
    L = O.lib()
    P = kats["poly_ops"]
    for c in P["modulo"]:
        m = O.modulus(I(c["mod"]))
        a = arr(c["in"])
        r = np.empty_like(a)
        L.mo_modulo_poly_coeffs(O.ptr(a), a.size, C.byref(m), O.ptr(r))
        assert r.tolist() == arr(c["out"]).tolist()
    for c in P["negate"]:
        m = O.modulus(I(c["mod"]))
        a = arr(c["in"])
        r = np.empty_like(a)
        L.mo_negate_poly_coeffmod(O.ptr(a), a.size, C.byref(m), O.ptr(r))
        assert r.tolist() == arr(c["out"]).tolist()
    for name, fn in (("add", L.mo_add_poly_coeffmod), ("sub", L.mo_sub_poly_coeffmod),
                     ("dyadic", L.mo_dyadic_product_coeffmod)):
        for c in P[name]:
            m = O.modulus(I(c["mod"]))
            a, b = arr(c["a"]), arr(c["b"])
            r = np.empty_like(a)
            fn(O.ptr(a), O.ptr(b), a.size, C.byref(m), O.ptr(r))
            assert r.tolist() == arr(c["out"]).tolist(), name
    for c in P["mul_scalar"]:
        m = O.modulus(I(c["mod"]))
        a = arr(c["in"])
        r = np.empty_like(a)
        L.mo_multiply_poly_scalar_coeffmod(O.ptr(a), a.size, I(c["scalar"]), C.byref(m), O.ptr(r))
        assert r.tolist() == arr(c["out"]).tolist()


def test_ntt_root_powers(kats):
    K = kats["ntt_root_powers"]
    for c in K["cases"]:
        t = O.Tables(c["coeff_count_power"], I(K["modulus"]))
        assert t.root_powers() == [I(x) for x in c["root_powers"]]
        if c["coeff_count_power"] == 1:  # ntt.cpp:63-65 asserts this for N=2 only
            inv = C.c_uint64(0)
            assert O.lib().mo_try_invert_uint_mod(t.t.root_powers[1].operand, t.q, C.byref(inv)) == 1
            assert inv.value == t.t.inv_root_powers[1].operand


def test_ntt_forward_kat(kats):
    K = kats["ntt_forward"]
    t = O.Tables(K["coeff_count_power"], I(K["modulus"]))
    for c in K["cases"]:
        assert t.ntt(arr(c["in"])).tolist() == arr(c["out"]).tolist()


def test_ntt_roundtrip(kats):
    K = kats["ntt_roundtrip"]
    t = O.Tables(K["coeff_count_power"], I(K["modulus"]))
    rng = np.random.default_rng(0)
    assert t.intt(np.zeros(t.n, dtype=np.uint64)).tolist() == [0] * t.n
    for _ in range(100):
        x = rng.integers(0, t.q, size=t.n, dtype=np.uint64)
        assert t.intt(t.ntt(x)).tolist() == x.tolist()


@pytest.mark.parametrize("logn,bits", [(3, 20), (6, 46), (10, 51), (12, 58), (13, 60), (11, 61)])
def test_ntt_is_negacyclic_evaluation(logn, bits):
    """NTT output [bitrev(i)] must be the polynomial evaluated at psi^(2i+1) (ntt.cpp:269-278)."""
    n = 1 << logn
    q = O.coeff_modulus_create(n, [bits])[0]
    t = O.Tables(logn, q)
    psi = int(t.t.root)
    assert pow(psi, n, q) == q - 1
    rng = np.random.default_rng(logn)
    x = rng.integers(0, q, size=n, dtype=np.uint64)
    y = t.ntt(x)
    lazy = t.ntt(x, lazy=True)
    assert (lazy < np.uint64(4 * q)).all() and ((lazy % np.uint64(q)) == y).all()
    xs = [int(v) for v in x]
    for i in list(range(4)) + [n // 2, n - 1]:
        rev = int(format(i, "0%db" % logn)[::-1], 2)
        w = pow(psi, 2 * rev + 1, q)
        acc = 0
        for c in reversed(xs):
            acc = (acc * w + c) % q
        assert int(y[i]) == acc
    il = t.intt(y, lazy=True)
    assert (il < np.uint64(2 * q)).all() and ((il % np.uint64(q)) == x).all()


def test_divide_and_round_kat(kats):
    K = kats["divide_and_round_q_last_ntt"]
    primes = [I(p) for p in K["primes"]]
    logn = K["coeff_count_power"]
    n = 1 << logn
    # context with a dummy special prime is not needed: the function only touches primes[0..L)
    ctx = O.Context(logn, primes)
    tabs = [O.Tables(logn, p) for p in primes]
    for c in K["cases"]:
        rows = np.stack([tabs[i].ntt(arr(c["in"][i])) for i in range(2)])
        poly = np.ascontiguousarray(rows)
        O.lib().mo_divide_and_round_q_last_ntt_inplace(ctx.h, O.ptr(poly), 2)
        out = tabs[0].intt(poly[0])
        for j in range(n):
            d = (53 + I(c["expect"][j]) - int(out[j])) % 53
            assert d <= 1
            if c["exact"]:
                assert int(out[j]) == I(c["expect"][j])


def test_galois_kats(kats):
    G = kats["galois"]
    logn = G["coeff_count_power"]
    gen = G["generator_in_kats"]
    for step, elt in G["elt_from_step"]:
        assert O.galois_elt_from_step(logn, step, gen) == elt
    assert O.galois_elts_all(logn, gen) == G["elts_all"]
    for elt, idx in G["index_from_elt"]:
        assert (elt - 1) >> 1 == idx  # galoiskeys.h:48
    a = G["apply_galois"]
    m = O.modulus(I(a["modulus"]))
    x = arr(a["in"])
    r = np.empty_like(x)
    O.lib().mo_apply_galois(O.ptr(x), logn, a["elt"], C.byref(m), O.ptr(r))
    assert r.tolist() == arr(a["out"]).tolist()
    a = G["apply_galois_ntt"]
    x = arr(a["in"])
    r = np.empty_like(x)
    tab = O.galois_table_ntt(logn, a["elt"])
    O.lib().mo_apply_galois_ntt(O.ptr(x), O.ptr(tab), x.size, O.ptr(r))
    assert r.tolist() == arr(a["out"]).tolist()


def test_galois_fork_generator_is_5():
    # native/src/seal/util/galois.h:169; SURVEY.md appendix A: 31 keys at N = 2^16
    assert O.galois_elt_from_step(16, 1) == 5
    assert O.galois_elt_from_step(16, 2) == 25
    assert O.galois_elt_from_step(16, -1) == pow(5, 32768 - 1, 131072)
    elts = O.galois_elts_all(16)
    assert len(elts) == 31 and elts[0] == 131071 and elts[1] == 5
    with pytest.raises(ValueError):
        O.galois_elt_from_step(3, 4)


def test_survey_constants(kats):
    S = kats["survey_oracle_constants"]
    mc = S["moai_chain"]
    primes = O.coeff_modulus_create(mc["n"], mc["bits"])
    assert len(primes) == 36 and len(set(primes)) == 36
    for idx, (p, root) in mc["primes_and_roots"].items():
        assert primes[int(idx)] == I(p)
        m = O.modulus(I(p))
        out = C.c_uint64(0)
        assert O.lib().mo_try_minimal_primitive_root(2 * mc["n"], C.byref(m), C.byref(out)) == 1
        assert out.value == I(root), idx
    assert sum(p.bit_length() for p in primes) == 1743  # 2025-991.pdf section 6: 1743-bit modulus
    p4 = O.coeff_modulus_create(65536, [60] * 4)
    assert p4[0] == I(S["n65536_4x60"]["first_prime"])
    m = O.modulus(p4[0])
    out = C.c_uint64(0)
    O.lib().mo_try_minimal_primitive_root(131072, C.byref(m), C.byref(out))
    assert out.value == I(S["n65536_4x60"]["root"])
    p44 = O.coeff_modulus_create(65536, [60] * 44)
    assert p44[0] == I(S["n65536_44x60"]["first_prime"]) and p44[-1] == I(S["n65536_44x60"]["last_prime"])
    assert O.coeff_modulus_create(8192, [60, 40, 60]) == [I(x) for x in S["config1_n8192_60_40_60"]]
