"""CKKSEncoder restatement in the CPU oracle (oracle/moai_oracle.c mo_ckks_*; SEAL/ckks.h:457-637,
ckks.cpp:13-216).  The reference's own encoder tests (native/tests/seal/ckks.cpp:18-345) are
encode -> decode round trips with |error| < 0.5 and hold no residue vectors, so the floating-point
bit pattern of this restatement is pinned only by construction ("parity unpinned" for bits); what
is pinned here: the round trips of those tests, decoded by an independent Vandermonde evaluation
with exact big-integer CRT, the slot order (generator 5), and the exactness of the integer
decomposition in all three size branches.  CPU only."""
from functools import reduce

import numpy as np
import pytest

import oracle as O
from ckks_toy import decode_slots


def crt_centered(res, primes):
    Q = reduce(lambda a, b: a * b, primes, 1)
    out = []
    for i in range(res.shape[1]):
        x = 0
        for j, q in enumerate(primes):
            Qj = Q // q
            x += int(res[j, i]) * Qj * pow(Qj, -1, q)
        x %= Q
        out.append(x - Q if x > Q // 2 else x)
    return out


def independent_decode(ctx, res, L, scale):
    coeffs = crt_centered(ctx.ntt(res, L, inverse=True).reshape(L, ctx.n), ctx.primes[:L])
    return decode_slots(coeffs, ctx.n, scale)


# (slots, prime bits, delta, data bound): native/tests/seal/ckks.cpp:20-137
ROUND_TRIPS = [
    (32, [40, 40, 40, 40], 2.0**16, 1),
    (32, [60, 60, 60, 60], 2.0**40, 1 << 30),
    (64, [60, 60, 60], 2.0**40, 1 << 30),
    (64, [30, 30, 30, 30, 30], 2.0**40, 1 << 30),
    (32, [30, 30, 30, 30, 30], 2.0**40, 1 << 30),
]


@pytest.mark.parametrize("slots,bits,delta,bound", ROUND_TRIPS)
def test_encode_vector_round_trip(slots, bits, delta, bound):
    n = 2 * slots
    logn = n.bit_length() - 1
    ctx = O.Context(logn, O.coeff_modulus_create(n, bits))
    enc = O.CkksEncoder(ctx)
    rng = np.random.default_rng(slots + len(bits))
    vals = rng.integers(0, bound, size=slots).astype(np.float64) + 0j
    res = enc.encode(vals, len(bits), delta)
    back = independent_decode(ctx, res, len(bits), delta)
    assert np.max(np.abs(back - vals)) < 0.5
    # real input takes the same path (conj of a real is itself)
    assert np.array_equal(enc.encode(vals.real.copy(), len(bits), delta), res)


def test_short_input_is_zero_padded():
    ctx = O.Context(6, O.coeff_modulus_create(64, [40, 40, 40]))
    enc = O.CkksEncoder(ctx)
    v = np.array([1.5, -2.25, 3.0])
    full = np.zeros(32)
    full[:3] = v
    assert np.array_equal(enc.encode(v, 3, 2.0**30), enc.encode(full, 3, 2.0**30))
    back = independent_decode(ctx, enc.encode(v, 3, 2.0**30), 3, 2.0**30)
    assert np.allclose(back[:3], v, atol=1e-6) and np.allclose(back[3:], 0, atol=1e-6)


def test_complex_values_and_levels():
    primes = O.coeff_modulus_create(128, [51, 46, 46, 58])
    ctx = O.Context(7, primes)
    enc = O.CkksEncoder(ctx)
    rng = np.random.default_rng(5)
    z = rng.normal(size=64) + 1j * rng.normal(size=64)
    for L in (3, 2, 1):
        res = enc.encode(z, L, 2.0**40)
        back = independent_decode(ctx, res, L, 2.0**40)
        assert np.max(np.abs(back - z)) < 1e-8
    # lower level = the first rows of the higher level (same coefficients, fewer residues)
    assert np.array_equal(enc.encode(z, 2, 2.0**40), enc.encode(z, 3, 2.0**40)[:2])


def test_tables_follow_the_reference_layout():
    ctx = O.Context(5, O.coeff_modulus_create(32, [40]))
    enc = O.CkksEncoder(ctx)
    n, m = 32, 64
    # matrix_reps_index_map_: slot i sits at bitrev((5^i mod 2n - 1) / 2), its conjugate at the
    # mirrored odd power (ckks.cpp:36-50)
    rev = lambda x, b: int(format(x, "0%db" % b)[::-1], 2)
    pos = 1
    for i in range(n // 2):
        assert enc.index_map[i] == rev((pos - 1) >> 1, 5)
        assert enc.index_map[n // 2 + i] == rev((m - pos - 1) >> 1, 5)
        pos = pos * 5 % m
    # root_powers_[i] = zeta^bitrev(i), inv_root_powers_[i] = conj(zeta^(bitrev(i-1)+1)) (ckks.cpp:58-62)
    for i in range(1, n):
        w = np.exp(2j * np.pi * rev(i, 5) / m)
        wi = np.conj(np.exp(2j * np.pi * (rev(i - 1, 5) + 1) / m))
        assert abs(complex(*enc.root_powers[i]) - w) < 1e-15
        assert abs(complex(*enc.inv_root_powers[i]) - wi) < 1e-15
    # 8-fold symmetry is exact: zeta^(m/8+1) is zeta^(m/8-1) with its components swapped, and
    # zeta^(m/2-1) is -conj(zeta^1) (croots.cpp:44-75)
    at = {rev(i, 5): enc.root_powers[i] for i in range(1, n)}
    assert at[m // 8 + 1][0] == at[m // 8 - 1][1] and at[m // 8 + 1][1] == at[m // 8 - 1][0]
    assert at[m // 2 - 1][0] == -at[1][0] and at[m // 2 - 1][1] == at[1][1]


def test_fft_pair_inverts():
    ctx = O.Context(8, O.coeff_modulus_create(256, [40]))
    enc = O.CkksEncoder(ctx)
    rng = np.random.default_rng(9)
    z = rng.normal(size=256) + 1j * rng.normal(size=256)
    back = enc.fft_to_rev(enc.fft_from_rev(z, 1.0 / 256))
    assert np.max(np.abs(back - z)) < 1e-12


def test_decomposition_branches_agree_with_exact_integers():
    # <= 64, <= 128 and the multi-word branch (ckks.h:549-629) must all return the exact integer
    # round(c) mod q; exercise them through the scale
    primes = O.coeff_modulus_create(64, [60, 60, 60, 60, 60])
    ctx = O.Context(6, primes)
    enc = O.CkksEncoder(ctx)
    rng = np.random.default_rng(3)
    v = rng.normal(size=32)
    seen = set()
    for sb in (30, 70, 100, 150, 200):
        res, bits = enc.encode(v, 5, 2.0**sb, return_bits=True)
        seen.add(0 if bits <= 64 else 1 if bits <= 128 else 2)
        coeffs = crt_centered(ctx.ntt(res, 5, inverse=True).reshape(5, 64), primes)
        # the coefficients are integers of about sb bits whose decode returns v
        back = decode_slots(coeffs, 64, 1.0) / 2.0**sb
        assert np.max(np.abs(back - v)) < max(64 * 2.0**-sb, 1e-11)
        assert max(abs(c) for c in coeffs).bit_length() <= bits
    assert seen == {0, 1, 2}


def test_scalar_encode_matches_constant_vector_semantics():
    # ckks.cpp:77-216: a scalar encodes to the constant polynomial round(value*scale); in NTT form
    # every word of a row equals that residue
    primes = O.coeff_modulus_create(64, [40, 40, 40, 40])
    ctx = O.Context(6, primes)
    enc = O.CkksEncoder(ctx)
    for value, sb in ((0.5, 16), (-3.25, 40), (123456.0, 16), (-7.0, 100), (1.0, 140), (0.0, 30)):
        rows = enc.encode_scalar(value, 4, 2.0**sb)
        want = round(value * 2.0**sb)
        assert [int(r) for r in rows] == [want % q for q in primes]


def test_error_codes():
    ctx = O.Context(6, O.coeff_modulus_create(64, [30, 30]))
    enc = O.CkksEncoder(ctx)
    with pytest.raises(ValueError, match="values_size is too large"):
        enc.encode(np.zeros(33), 2, 2.0**20)
    with pytest.raises(ValueError, match="scale out of bounds"):
        enc.encode(np.zeros(4), 2, 2.0**60)
    with pytest.raises(ValueError, match="scale out of bounds"):
        enc.encode(np.zeros(4), 2, -1.0)
    with pytest.raises(ValueError, match="encoded values are too large"):
        enc.encode(np.full(32, 1e12), 2, 2.0**30)
    with pytest.raises(ValueError, match="encoded values are too large"):
        enc.encode_scalar(1e12, 2, 2.0**30)


def test_config1_plumbing_on_the_cpu_path():
    """BASELINE.json configs[0] / SURVEY.md 8(d) config 1, on the CPU oracle end to end: N = 8192, primes
    {60, 40, 60} from CoeffModulus::Create, scale 2^40, v[i] = i * 1e-3, weight 0.5:
    encode -> encrypt -> multiply_plain -> rescale_to_next -> decrypt -> decode; out[1000] = 0.5, chain index 0."""
    from ckks_toy import ToyClient

    logn, n = 13, 8192
    primes = O.coeff_modulus_create(n, [60, 40, 60])
    assert primes == [1152921504606748673, 1099511480321, 1152921504606830593]
    ctx = O.Context(logn, primes)
    enc = O.CkksEncoder(ctx)
    cl = ToyClient(ctx, seed=11)
    L, scale = 2, 2.0**40  # two data primes; the third is the key-switching prime
    v = np.arange(n // 2) * 1e-3
    ct = cl.encrypt_zero_symmetric(L)
    m = enc.encode(v, L, scale)
    for i in range(L):
        ct[0, i] = (ct[0, i] + m[i]) % np.uint64(primes[i])
    w = enc.encode_scalar(0.5, L, scale)  # encoder.encode(0.5, parms_id, scale, plain): constant rows
    plain = np.stack([np.full(n, w[i], dtype=np.uint64) for i in range(L)])
    prod = ctx.multiply_plain(ct, 2, L, plain)
    out = ctx.rescale(prod, 2, L)  # [2][1][N]: chain index 0
    assert out.shape == (2, L - 1, n)
    new_scale = scale * scale / primes[L - 1]
    coeffs = cl.decrypt(out, 2, L - 1)
    # decode with the oracle's forward transform (CKKSEncoder::decode_internal, SEAL/ckks.h:644-760)
    z = np.array([c / new_scale for c in coeffs], dtype=np.complex128)
    slots = enc.fft_to_rev(z)[enc.index_map[: n // 2]]
    assert abs(slots[1000].real - 0.5) < 1e-5
    assert np.max(np.abs(slots.real - 0.5 * v)) < 1e-5 and np.max(np.abs(slots.imag)) < 1e-5
