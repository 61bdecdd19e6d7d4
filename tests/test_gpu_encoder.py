"""GPU parity of the device CKKS encoder (moai_ckks_encode) against the oracle's restatement of
CKKSEncoder::encode_internal (SEAL/ckks.h:457-637): bit-exact residues -- the FP64 transform performs
the reference's operations per butterfly with no contraction, so no tolerance is needed or allowed.
The tables are compared bit for bit too (both sides call the host libm as util/croots.cpp does).
The reference holds no residue vectors for the encoder (its tests are round trips, mirrored in
tests/test_oracle_encoder.py), so the floating-point bits are pinned to the restatement only."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

MOAI_BITS = [51] + [46] * 20 + [51] * 14 + [58]  # include/test/test_full_scheme.hpp:356-378


@pytest.mark.parametrize("logn,bits", [
    (3, [40, 40]), (6, [40, 40, 40, 40]), (7, [60, 60, 60]), (10, [51, 46, 58]), (12, [51, 46, 46, 58]),
    (13, [60, 40, 60]), (14, [46, 51]), (15, [51, 46, 46, 51]), (16, [51, 46, 46, 46, 51, 58]),
])
def test_encode_matches_oracle(moai, logn, bits):
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    enc = O.CkksEncoder(octx)
    idx, roots = ctx.ckks_tables()
    assert (idx == enc.index_map).all()
    assert (roots.view(np.uint64) == enc.inv_root_powers.view(np.uint64)).all()
    rng = np.random.default_rng(logn)
    L = len(primes)
    slots = n // 2
    batch = [rng.normal(size=slots) + 1j * rng.normal(size=slots),
             rng.uniform(-100, 100, size=slots) + 0j,
             np.zeros(slots, dtype=np.complex128)]
    batch[2][0] = 1.0  # a delta: every coefficient is a root of unity times scale / n
    scale = 2.0**40
    out, mx = ctx.ckks_encode(np.stack(batch), L, scale)
    got = out.to_numpy((3, L, n))
    for b in range(3):
        want, bits_ = enc.encode(batch[b], L, scale, return_bits=True)
        assert (got[b] == want).all()
        assert int(np.ceil(np.log2(max(mx[b], 1.0)))) + 1 == bits_
    # real input, fewer values than slots, lower level with explicit rows
    v = rng.normal(size=max(slots // 3, 1))
    pidx = list(range(L - 1))[::-1] if L > 1 else [0]
    out, _ = ctx.ckks_encode(v, len(pidx), 2.0**30, prime_index=pidx)
    assert (out.to_numpy((len(pidx), n)) == enc.encode(v, len(pidx), 2.0**30, prime_index=pidx)).all()


def test_encode_large_coefficients(moai):
    # the reference's <= 128-bit and multi-word decomposition branches (ckks.h:575-629)
    logn, bits = 13, [60, 60, 60, 60, 60]
    primes = O.coeff_modulus_create(1 << logn, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    enc = O.CkksEncoder(octx)
    v = np.random.default_rng(1).normal(size=4096)
    seen = set()
    for sb in (50, 62, 70, 100, 127, 150, 250):
        want, nb = enc.encode(v, 5, 2.0**sb, return_bits=True)
        seen.add(0 if nb <= 64 else 1 if nb <= 128 else 2)
        out, _ = ctx.ckks_encode(v, 5, 2.0**sb)
        assert (out.to_numpy((5, 1 << logn)) == want).all(), sb
    assert seen == {0, 1, 2}


def test_encode_halfway_and_signed_zero(moai):
    # std::round is half away from zero (not rint); a slot vector whose coefficients are exact halves:
    # a constant c encodes to coefficient 0 = c * scale, the rest exactly 0 (up to sign of zero)
    logn, bits = 10, [40, 40]
    primes = O.coeff_modulus_create(1 << logn, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    enc = O.CkksEncoder(octx)
    for c in (0.5, -0.5, 1.5, -2.5, 0.0, -0.0):
        v = np.full(512, c)
        want = enc.encode(v, 2, 1.0)
        out, _ = ctx.ckks_encode(v, 2, 1.0)
        assert (out.to_numpy((2, 1024)) == want).all(), c


def test_encode_moai_parameters(moai):
    # N = 2^16 on MOAI's chain at the levels the bias / mask encodes run at (single_att_block.hpp:30-47)
    logn = 16
    primes = O.coeff_modulus_create(1 << logn, MOAI_BITS)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    enc = O.CkksEncoder(octx)
    rng = np.random.default_rng(16)
    vals = np.stack([rng.normal(size=32768), np.where(np.arange(32768) % 128 == 0, 0.37, 0.0)])
    for L in (35, 16, 1):
        out, _ = ctx.ckks_encode(vals, L, 2.0**46)
        got = out.to_numpy((2, L, 1 << logn))
        for b in range(2):
            assert (got[b] == enc.encode(vals[b], L, 2.0**46)).all()


def test_encode_errors(moai):
    primes = O.coeff_modulus_create(64, [30, 30])
    ctx = moai.Context(6, primes)
    assert ctx.total_coeff_modulus_bit_count(2) == (primes[0] * primes[1]).bit_length()
    with pytest.raises(moai.MoaiError, match="values_size is too large"):
        ctx.ckks_encode(np.zeros(33), 2, 2.0**20)
    with pytest.raises(moai.MoaiError, match="scale out of bounds"):
        ctx.ckks_encode(np.zeros(4), 2, 2.0**60)
    with pytest.raises(moai.MoaiError, match="encoded values are too large"):
        ctx.ckks_encode(np.full(32, 1e12), 2, 2.0**30)
    out, mx = ctx.ckks_encode(np.zeros((0, 4)), 2, 2.0**20)  # empty batch
    assert mx.size == 0


@pytest.mark.parametrize("logn,bits", [(6, [40, 40]), (12, [51, 46, 58]), (16, [51, 46, 46])])
def test_encode_masked_constants(moai, logn, bits):
    # the vectors MOAI's masked matrix product encodes (Ct_pt_matrix_mul.hpp:124-146): w on the slots with
    # bias_vec == 1, zero elsewhere; other mask values (0, 2, -1) count as "not 1" like the reference's test
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    enc = O.CkksEncoder(octx)
    rng = np.random.default_rng(logn)
    slots = n // 2
    mask = rng.integers(-1, 3, size=slots).astype(np.int32)
    w = np.concatenate([rng.normal(scale=0.02, size=5), [0.0, -0.0, 1.0]])
    L = len(primes) - 1
    out, mx = ctx.ckks_encode_masked(w, mask, L, 2.0**30)
    got = out.to_numpy((w.size, L, n))
    for b in range(w.size):
        assert (got[b] == enc.encode(np.where(mask == 1, w[b], 0.0), L, 2.0**30)).all()
    # a shorter mask leaves the remaining slots zero
    out, _ = ctx.ckks_encode_masked(w[:2], mask[: slots // 2], L, 2.0**30)
    got = out.to_numpy((2, L, n))
    v = np.zeros(slots)
    v[: slots // 2] = np.where(mask[: slots // 2] == 1, w[1], 0.0)
    assert (got[1] == enc.encode(v, L, 2.0**30)).all()
