"""GPU parity: every C-ABI entry point against the CPU oracle on the same seeded inputs, bit-exact
(integer path: no tolerance).  Runs on the MI355X box only (-m gpu)."""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

MOAI_BITS = [51] + [46] * 20 + [51] * 14 + [58]  # include/test/test_full_scheme.hpp:356-378


def up(m, a):
    return m.DeviceBuffer.from_numpy(a)


@pytest.fixture(params=["fp64", "int64"])
def ks_arith(request, moai):
    """Key-switch tests run twice: with the FP64 arithmetic modes forced on for every prime below 2^51 (the
    library only picks them from 16 digit rows per call) and with the integer units only."""
    moai.hip.set_tuning("MOAI_KS_FP_MIN_ROWS", 0 if request.param == "fp64" else 1 << 40)
    moai.hip.set_tuning("MOAI_MD_FP_MIN_ROWS", 0 if request.param == "fp64" else 1 << 40)  # the mod-down tail and rescale
    yield request.param
    moai.hip.set_tuning("MOAI_KS_FP_MIN_ROWS", 16)
    moai.hip.set_tuning("MOAI_MD_FP_MIN_ROWS", 256)


@pytest.fixture(scope="module")
def env12(moai):
    logn = 12
    primes = O.coeff_modulus_create(1 << logn, [51, 46, 46, 51, 58])
    return logn, primes, O.Context(logn, primes), moai.Context(logn, primes)


@pytest.mark.parametrize("logn,bits", [
    (1, [60]), (2, [50, 30]), (3, [60, 20]), (5, [46, 51]), (8, [58, 61]), (10, [60, 40, 61]), (11, [46, 51, 58]),
    (12, [61, 46, 20]), (13, [60, 40, 60]), (14, [51, 58, 30]), (15, [60, 40, 40, 60]), (16, [60, 51, 46, 58, 61]),
])
def test_ntt_matches_oracle(moai, logn, bits):
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    for i in range(len(primes)):
        assert ctx.root(i) == int(octx_root(logn, primes[i]))
    rng = np.random.default_rng(logn)
    L, npoly = len(primes), 3
    x = O.uniform_rns(rng, primes, (npoly,), n)
    # edge values: 0, q-1 rows
    x[0, :, 0] = 0
    for i, q in enumerate(primes):
        x[1, i, :4 if n >= 4 else n] = q - 1
    d = up(moai, x)
    ctx.ntt_forward(d, npoly, L)
    got = d.to_numpy(x.shape)
    assert (got == octx.ntt(x, L)).all()
    ctx.ntt_inverse(d, npoly, L)
    assert (d.to_numpy(x.shape) == x).all()
    # inverse on arbitrary data vs oracle
    ctx.ntt_inverse(d, npoly, L)
    assert (d.to_numpy(x.shape) == octx.ntt(x, L, inverse=True)).all()
    # explicit prime_index (rows under a permuted / repeated prime choice)
    pidx = [L - 1 - i for i in range(L)]
    y = np.stack([rng.integers(0, primes[p], size=n, dtype=np.uint64) for p in pidx])[None]
    d2 = up(moai, y)
    ctx.ntt_forward(d2, 1, L, prime_index=pidx)
    assert (d2.to_numpy(y.shape) == octx.ntt(y, L, prime_index=pidx)).all()


def octx_root(logn, q):
    return O.Tables(logn, q).t.root


@pytest.mark.parametrize("logn", [4, 12, 13, 16])
def test_ntt_naive_stage_path_agrees(moai, logn, monkeypatch):
    """the per-stage global-memory kernels (MOAI_NTT_NAIVE=1) and the tiled kernels are two
    implementations of the same transform; both must equal the oracle."""
    import subprocess, sys
    code = r'''
import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle as O, __graft_entry__ as g
m = g.load_package()
logn = int(sys.argv[1]); n = 1 << logn
primes = O.coeff_modulus_create(n, [60, 46])
octx, ctx = O.Context(logn, primes), m.Context(logn, primes)
x = O.uniform_rns(np.random.default_rng(1), primes, (2,), n)
d = m.DeviceBuffer.from_numpy(x)
ctx.ntt_forward(d, 2, 2); assert (d.to_numpy(x.shape) == octx.ntt(x, 2)).all()
ctx.ntt_inverse(d, 2, 2); assert (d.to_numpy(x.shape) == x).all()
print("ok")
'''
    env = dict(os.environ, MOAI_NTT_NAIVE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, str(logn)], cwd=root, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_elementwise(moai, env12):
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(3)
    L, npoly = 4, 3
    a = O.uniform_rns(rng, primes[:L], (npoly,), n)
    b = O.uniform_rns(rng, primes[:L], (npoly,), n)
    a[0, :, :8] = 0
    b[0, :, :4] = 0
    for i in range(L):
        a[1, i, :8] = primes[i] - 1
        b[1, i, :8] = primes[i] - 1
    da, db = up(moai, a), up(moai, b)
    do = moai.DeviceBuffer(a.size)
    ctx.add(da, db, do, npoly, L)
    assert (do.to_numpy(a.shape) == octx.add(a, b, npoly, L)).all()
    ctx.sub(da, db, do, npoly, L)
    assert (do.to_numpy(a.shape) == octx.sub(a, b, npoly, L)).all()
    ctx.negate(da, do, npoly, L)
    assert (do.to_numpy(a.shape) == octx.negate(a, npoly, L)).all()
    ctx.dyadic_mul(da, db, do, npoly, npoly, L)
    exp = np.stack([octx.multiply_plain(a[p], 1, L, b[p]).reshape(L, n) for p in range(npoly)])
    assert (do.to_numpy(a.shape) == exp).all()
    # broadcast plaintext (multiply_plain)
    ctx.dyadic_mul(da, db, do, npoly, 1, L)
    assert (do.to_numpy(a.shape) == octx.multiply_plain(a, npoly, L, b[0])).all()
    # in place
    ctx.add(da, db, da, npoly, L)
    assert (da.to_numpy(a.shape) == octx.add(a, b, npoly, L)).all()
    # scalar rows == multiply by a constant-row plaintext (ckks.cpp:131-150)
    da = up(moai, a)
    scal = [int(rng.integers(0, 1 << 63)) * 2 + 1 for _ in range(L)]
    ctx.mul_scalar_rows(da, scal, do, npoly, L)
    pt = np.stack([np.full(n, scal[i] % primes[i], dtype=np.uint64) for i in range(L)])
    assert (do.to_numpy(a.shape) == octx.multiply_plain(a, npoly, L, pt)).all()
    ctx.add_scalar_rows(da, scal, do, npoly, L)
    ptb = np.broadcast_to(pt, a.shape)
    assert (do.to_numpy(a.shape) == octx.add(a, np.ascontiguousarray(ptb), npoly, L)).all()


def test_ct_multiply_square(moai, env12):
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(4)
    L, B = 4, 2
    x = O.uniform_rns(rng, primes[:L], (B, 2), n)
    y = O.uniform_rns(rng, primes[:L], (B, 2), n)
    dx, dy = up(moai, x), up(moai, y)
    do = moai.DeviceBuffer(B * 3 * L * n)
    ctx.ct_multiply(dx, dy, do, L, B)
    got = do.to_numpy((B, 3, L, n))
    for b in range(B):
        assert (got[b] == octx.multiply(x[b], y[b], L)).all()
    ctx.ct_square(dx, do, L, B)
    got = do.to_numpy((B, 3, L, n))
    for b in range(B):
        assert (got[b] == octx.square(x[b], L)).all()


@pytest.mark.parametrize("sx,sy", [(2, 2), (3, 2), (2, 3), (3, 3), (4, 2), (5, 4)])
def test_ct_multiply_general_sizes(moai, env12, sx, sy):
    """Evaluator::multiply for operands that are not both of size 2 (SEAL/evaluator.cpp:862-900), batch 2, against the
    oracle's restatement of that branch; edge residues 0 and q-1 in every polynomial"""
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(10 * sx + sy)
    L, B = 3, 2
    x = O.uniform_rns(rng, primes[:L], (B, sx), n)
    y = O.uniform_rns(rng, primes[:L], (B, sy), n)
    x[:, :, :, :4] = 0
    for r in range(L):
        x[:, :, r, 4:8] = primes[r] - 1
        y[:, :, r, 4:12] = primes[r] - 1
    do = moai.DeviceBuffer(B * (sx + sy - 1) * L * n)
    ctx.ct_multiply_general(up(moai, x), sx, up(moai, y), sy, do, L, B)
    got = do.to_numpy((B, sx + sy - 1, L, n))
    for b in range(B):
        assert (got[b] == octx.multiply_general(x[b], sx, y[b], sy, L)).all()
    if (sx, sy) == (2, 2):
        for b in range(B):
            assert (got[b] == octx.multiply(x[b], y[b], L)).all()


def test_ct_multiply_general_refuses_bad_sizes(moai, env12):
    logn, primes, octx, ctx = env12
    n = 1 << logn
    buf = moai.DeviceBuffer(16 * 2 * n)
    out = moai.DeviceBuffer(17 * 2 * n)
    for sx, sy in ((1, 2), (2, 1), (9, 9), (16, 2)):
        with pytest.raises(moai.hip.MoaiError):
            ctx.ct_multiply_general(buf, sx, buf, sy, out, 2, 1)


@pytest.mark.parametrize("count,bits", [(1, [51, 46, 58]), (17, [61, 60, 40]), (64, [51, 46, 46]), (100, [61, 61])])
def test_ct_dot_matches_multiply_add_chain(moai, count, bits):
    # sum_j multiply(x[j], y[j]) as the reference issues it (Ct_ct_matrix_mul.hpp:33-42): one multiply and one
    # add_inplace per term; all-(q-1) rows push the lazy accumulators to their bound (61-bit primes, 100 terms)
    logn = 10
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    L = len(primes)
    rng = np.random.default_rng(count)
    x = O.uniform_rns(rng, primes, (count, 2), n)
    y = O.uniform_rns(rng, primes, (count, 2), n)
    for r, q in enumerate(primes):
        x[:, :, r, :8] = q - 1
        y[:, :, r, :8] = q - 1
    want = octx.multiply(x[0], y[0], L)
    for j in range(1, count):
        want = octx.add(want, octx.multiply(x[j], y[j], L), 3, L)
    dx, dy = up(moai, x), up(moai, y)
    do = moai.DeviceBuffer(3 * L * n)
    ctx.ct_dot(dx, dy, do, count, L)
    assert (do.to_numpy((3, L, n)) == want).all()


@pytest.mark.parametrize("terms,bits", [(1, [51, 46]), (9, [61, 60, 40]), (40, [61, 61]), (64, [51, 46, 58])])
def test_ct_pt_dot_matches_multiply_plain_add_chain(moai, terms, bits):
    # sum_t multiply_plain(baby[xi[t]], plain[pi[t]]) as Bootstrapper::bsgs_linear_transform issues it
    # (Bootstrapper.cpp:2028-2046): one multiply_plain and one add_inplace per term, for a batch of 3
    logn = 10
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    L = len(primes)
    rng = np.random.default_rng(terms)
    n_ops, n_pt, B = 5, 7, 3
    x = O.uniform_rns(rng, primes, (n_ops, B, 2), n)
    p = O.uniform_rns(rng, primes, (n_pt,), n)
    for r, q in enumerate(primes):
        x[:, :, :, r, :8] = q - 1
        p[:, r, :8] = q - 1
    xi = rng.integers(0, n_ops, size=terms)
    pi = rng.integers(0, n_pt, size=terms)
    dx, dp = up(moai, x), up(moai, p)
    do = moai.DeviceBuffer(B * 2 * L * n)
    ctx.ct_pt_dot(dx, dp, do, xi, pi, B * 2, L)
    got = do.to_numpy((B, 2, L, n))
    for b in range(B):
        want = octx.multiply_plain(x[xi[0], b], 2, L, p[pi[0]])
        for t in range(1, terms):
            want = octx.add(want, octx.multiply_plain(x[xi[t], b], 2, L, p[pi[t]]), 2, L)
        assert (got[b] == want).all()
    # two sums over the same operands in one pass (two giant steps of a transform): the second over the leading terms only
    for t2 in sorted({1, max(1, terms // 2), terms}):
        pi2 = rng.integers(0, n_pt, size=t2)
        do2 = moai.DeviceBuffer(B * 2 * L * n)
        ctx.ct_pt_dot2(dx, dp, do, do2, xi, pi, pi2, B * 2, L)
        assert (do.to_numpy((B, 2, L, n)) == got).all()
        got2 = do2.to_numpy((B, 2, L, n))
        ctx.ct_pt_dot(dx, dp, do, xi[:t2], pi2, B * 2, L)
        assert (got2 == do.to_numpy((B, 2, L, n))).all(), t2
        for b in range(B):
            want = octx.multiply_plain(x[xi[0], b], 2, L, p[pi2[0]])
            for t in range(1, t2):
                want = octx.add(want, octx.multiply_plain(x[xi[t], b], 2, L, p[pi2[t]]), 2, L)
            assert (got2[b] == want).all(), (t2, b)
    with pytest.raises(moai.MoaiError):
        ctx.ct_pt_dot2(dx, dp, do, do, xi, pi, pi, B * 2, L)  # the two outputs must differ
    # a single polynomial per operand takes the one-polynomial-per-thread kernel
    one, one2, ref = moai.DeviceBuffer(L * n), moai.DeviceBuffer(L * n), moai.DeviceBuffer(L * n)
    ctx.ct_pt_dot2(dx, dp, one, one2, xi, pi, pi[:1], 1, L)
    ctx.ct_pt_dot(dx, dp, ref, xi, pi, 1, L)
    assert (one.to_numpy() == ref.to_numpy()).all()
    ctx.ct_pt_dot(dx, dp, ref, xi[:1], pi[:1], 1, L)
    assert (one2.to_numpy() == ref.to_numpy()).all()


@pytest.mark.parametrize("rows,bits,logn", [(1, [51, 46], 10), (70, [61, 60, 40], 10), (333, [61, 61], 11), (3072, [51, 46], 10)])
def test_ct_pt_dot_rows_matches_multiply_plain_add_chain(moai, rows, bits, logn):
    """one column of the masked ciphertext x plaintext product (Ct_pt_matrix_mul.hpp:120-150): sum over all rows of
    multiply_plain(x[r], p[r]), and two columns in one pass; all-(q-1) residues in the first coefficients push the lazy
    accumulators to their bound; 3072 rows is the final feed-forward product's count"""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    L = len(primes)
    rng = np.random.default_rng(rows)
    x = O.uniform_rns(rng, primes, (rows, 2), n)
    p = O.uniform_rns(rng, primes, (rows,), n)
    p2 = O.uniform_rns(rng, primes, (rows,), n)
    for r, q in enumerate(primes):
        x[:, :, r, :8] = q - 1
        p[:, r, :8] = q - 1
    def column(pl):
        acc = np.zeros((2, L, n), dtype=object)
        xo, po = x.astype(object), pl.astype(object)
        for r in range(L):
            acc[:, r] = (xo[:, :, r] * po[:, None, r]).sum(axis=0) % primes[r]
        return acc.astype(np.uint64)
    want, want2 = column(p), column(p2)
    if rows <= 70:  # and the oracle's own call chain on the small cases
        ref = octx.multiply_plain(x[0], 2, L, p[0])
        for r in range(1, rows):
            ref = octx.add(ref, octx.multiply_plain(x[r], 2, L, p[r]), 2, L)
        assert (ref.reshape(2, L, n) == want).all()
    dx, dp, dp2 = up(moai, x), up(moai, p), up(moai, p2)
    o1, o2 = moai.DeviceBuffer(2 * L * n), moai.DeviceBuffer(2 * L * n)
    ctx.ct_pt_dot_rows(dx, dp, None, o1, None, rows, 2, L)
    assert (o1.to_numpy((2, L, n)) == want).all()
    ctx.ct_pt_dot_rows(dx, dp2, dp, o2, o1, rows, 2, L)
    assert (o2.to_numpy((2, L, n)) == want2).all()
    assert (o1.to_numpy((2, L, n)) == want).all()
    with pytest.raises(moai.MoaiError):
        ctx.ct_pt_dot_rows(dx, dp, dp2, o1, None, rows, 2, L)
    with pytest.raises(moai.MoaiError):
        ctx.ct_pt_dot_rows(dx, dp, None, o1, None, 0, 2, L)


@pytest.mark.parametrize("L", [5, 4, 2])
def test_rescale_and_drop(moai, env12, L, ks_arith):
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(5 + L)
    B, size = 2, 2
    x = O.uniform_rns(rng, primes[:L], (B, size), n)
    dx = up(moai, x)
    do = moai.DeviceBuffer(B * size * (L - 1) * n)
    ctx.rescale(dx, do, size, L, B)
    got = do.to_numpy((B, size, L - 1, n))
    for b in range(B):
        assert (got[b] == octx.rescale(x[b], size, L)).all()
    assert (dx.to_numpy(x.shape) == x).all()  # input preserved
    for drop in range(1, L):
        ctx.mod_drop(dx, do, size, L, drop, B)
        got = do.to_numpy((B, size, L - drop, n), words=B * size * (L - drop) * n)
        assert (got == x[:, :, : L - drop]).all()
    with pytest.raises(moai.MoaiError):
        ctx.mod_drop(dx, do, size, L, L, B)
    if L == 2:
        one = moai.DeviceBuffer(B * size * n)
        ctx.mod_drop(dx, one, size, L, 1, B)
        with pytest.raises(moai.MoaiError):  # end of modulus switching chain (evaluator.cpp:1693)
            ctx.rescale(one, do, size, 1, B)


@pytest.mark.parametrize("logn,bits", [(12, [51, 46, 46, 51, 58]), (13, [60, 40, 61, 30]), (10, [50, 40, 45])])
def test_scalar_product_fused_into_rescale(moai, logn, bits, ks_arith):
    """moai_mul_scalar_rescale = moai_mul_scalar_rows followed by moai_rescale (multiply_const + rescale_to_next of
    the fork), residues identical to the two calls and to the oracle; scalars above the modulus are reduced first."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    rng = np.random.default_rng(31 + logn)
    for L in range(len(primes), 1, -1):
        B, size = 3, 2
        x = O.uniform_rns(rng, primes[:L], (B, size), n)
        scalars = [int(rng.integers(0, 1 << 62)) for _ in range(L)]
        scalars[0] = primes[0] - 1
        if L > 2:
            scalars[1] = 0
        dx = up(moai, x)
        dprod = moai.DeviceBuffer(x.size)
        two = moai.DeviceBuffer(B * size * (L - 1) * n)
        one = moai.DeviceBuffer(B * size * (L - 1) * n)
        ctx.mul_scalar_rows(dx, scalars, dprod, B * size, L)
        ctx.rescale(dprod, two, size, L, B)
        ctx.mul_scalar_rescale(dx, scalars, one, size, L, B)
        want = two.to_numpy((B, size, L - 1, n))
        assert (one.to_numpy((B, size, L - 1, n)) == want).all(), L
        assert (dx.to_numpy(x.shape) == x).all()
        # and against the oracle: rescale of the canonical products
        prod = x.copy()
        for i in range(L):
            q = primes[i]
            prod[:, :, i, :] = (prod[:, :, i, :].astype(object) * (scalars[i] % q) % q).astype(np.uint64)
        for b in range(B):
            assert (want[b] == octx.rescale(prod[b], size, L)).all(), (L, b)
        # the same with the addition that follows (add_inplace of the running sum, SEAL/evaluator.cpp:155-240) done by the
        # rescale's last kernel: separate destination, and accumulating in place; edge residues q-1 in the addend
        acc = O.uniform_rns(rng, primes[: L - 1], (B, size), n)
        for i in range(L - 1):
            acc[:, :, i, :16] = primes[i] - 1
        expect = np.stack([octx.add(want[b], acc[b], size, L - 1).reshape(size, L - 1, n) for b in range(B)])
        dacc, out = up(moai, acc), moai.DeviceBuffer(acc.size)
        ctx.mul_scalar_rescale_add(dx, scalars, dacc, out, size, L, B)
        assert (out.to_numpy(acc.shape) == expect).all(), L
        assert (dacc.to_numpy(acc.shape) == acc).all()
        ctx.mul_scalar_rescale_add(dx, scalars, dacc, dacc, size, L, B)
        assert (dacc.to_numpy(acc.shape) == expect).all(), L
        dacc = up(moai, acc)
        ctx.rescale_add(dprod, dacc, out, size, L, B)
        assert (out.to_numpy(acc.shape) == expect).all(), L
        ctx.rescale_add(dprod, dacc, dacc, size, L, B)
        assert (dacc.to_numpy(acc.shape) == expect).all(), L


def test_galois_permute(moai, env12):
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(6)
    L = 3
    x = O.uniform_rns(rng, primes[:L], (2,), n)
    dx = up(moai, x)
    do = moai.DeviceBuffer(x.size)
    for step in (1, -1, 5, 0, 100):
        elt = ctx.galois_elt_from_step(step)
        assert elt == O.galois_elt_from_step(logn, step)
        ctx.galois_permute(dx, do, 2, L, elt)
        tab = O.galois_table_ntt(logn, elt)
        assert (do.to_numpy(x.shape) == x[:, :, tab]).all()
    with pytest.raises(moai.MoaiError):
        ctx.galois_elt_from_step(n // 2)
    with pytest.raises(moai.MoaiError):
        ctx.galois_permute(dx, do, 2, L, 4)  # even element is not valid


@pytest.mark.parametrize("L", [4, 3, 1])
def test_switch_key_relin_galois(moai, env12, L, ks_arith):
    logn, primes, octx, ctx = env12
    n, k = 1 << logn, len(primes)
    rng = np.random.default_rng(7 + L)
    B = 3
    key = O.uniform_rns(rng, primes, (k - 1, 2), n)
    dkey = up(moai, key)
    ct = O.uniform_rns(rng, primes[:L], (B, 2), n)
    tgt = O.uniform_rns(rng, primes[:L], (B,), n)
    dct, dt = up(moai, ct), up(moai, tgt)
    ctx.switch_key(dct, dt, dkey, L, B)
    got = dct.to_numpy(ct.shape)
    for b in range(B):
        assert (got[b] == octx.switch_key(ct[b], tgt[b], key, L).reshape(2, L, n)).all(), b
    assert (dt.to_numpy(tgt.shape) == tgt).all()
    # relinearize
    ct3 = O.uniform_rns(rng, primes[:L], (B, 3), n)
    d3 = up(moai, ct3)
    do = moai.DeviceBuffer(B * 2 * L * n)
    ctx.relinearize(d3, dkey, do, L, B)
    got = do.to_numpy((B, 2, L, n))
    for b in range(B):
        assert (got[b] == octx.relinearize(ct3[b], key, L)).all()
    # rotate / conjugate
    for elt in (ctx.galois_elt_from_step(1), ctx.galois_elt_from_step(-3), 2 * n - 1):
        dct = up(moai, ct)
        ctx.apply_galois(dct, L, elt, dkey, B)
        got = dct.to_numpy(ct.shape)
        for b in range(B):
            assert (got[b] == octx.apply_galois(ct[b], L, elt, key).reshape(2, L, n)).all()
        # accumulated into a running sum by the key switch's last kernel (moai_apply_galois_acc): rotation + add_inplace
        run = O.uniform_rns(rng, primes[:L], (B, 2), n)
        drun, dsrc = up(moai, run), up(moai, ct)
        ctx.apply_galois_acc(dsrc, drun, L, elt, dkey, B)
        got_acc = drun.to_numpy(ct.shape)
        for b in range(B):
            want_acc = octx.add(run[b], octx.apply_galois(ct[b], L, elt, key).reshape(2, L, n), 2, L).reshape(2, L, n)
            assert (got_acc[b] == want_acc).all(), (elt, b)
        assert (dsrc.to_numpy(ct.shape) == ct).all()
        # separate destination: same result, source untouched
        dsrc, ddst = up(moai, ct), moai.DeviceBuffer(ct.size)
        ctx.apply_galois_to(dsrc, ddst, L, elt, dkey, B)
        assert (ddst.to_numpy(ct.shape) == got).all()
        assert (dsrc.to_numpy(ct.shape) == ct).all()
    assert (d3.to_numpy(ct3.shape) == ct3).all()  # relinearize reads c0, c1 in its last kernel and leaves its input alone

def test_relinearize_loop_over_sizes(moai, env12):
    """relinearize_internal for a size-4 ciphertext (SEAL/evaluator.cpp:1385-1393) as the seal:: shim issues it: one
    moai_switch_key per dropped polynomial, the key of s^3 first -- against the oracle's restatement of the loop"""
    logn, primes, octx, ctx = env12
    n, k = 1 << logn, len(primes)
    rng = np.random.default_rng(44)
    L = 3
    keys = [O.uniform_rns(rng, primes, (k - 1, 2), n) for _ in range(2)]
    dkeys = [up(moai, kk) for kk in keys]
    ct4 = O.uniform_rns(rng, primes[:L], (4,), n)
    want = octx.relinearize_general(ct4, 4, 2, keys, L)
    d = up(moai, ct4)
    rn = L * n
    for size in (4, 3):
        ctx.switch_key(d.ptr, d.ptr + (size - 1) * rn * 8, dkeys[size - 3], L, 1)
    got = d.to_numpy((4, L, n))
    assert (got[:2] == want).all()
    assert (got[2:] == ct4[2:]).all()



@pytest.mark.parametrize("bits", [[60, 50, 60, 61], [46, 58, 51, 58]])
def test_switch_key_guarded_and_lazy_arithmetic(moai, bits, ks_arith):
    """60/61-bit primes take the guarded butterflies and normalised MAC; <= 58-bit chains (MOAI) the
    unguarded ones with the lazy 128-bit MAC.  Both must reproduce the oracle bit for bit."""
    logn = 13
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k = len(primes)
    rng = np.random.default_rng(sum(bits))
    key = O.uniform_rns(rng, primes, (k - 1, 2), n)
    # worst-case magnitudes: every residue q-1
    for J in range(k - 1):
        for K in range(2):
            for i in range(k):
                key[J, K, i, :64] = primes[i] - 1
    dkey = up(moai, key)
    for L in (3, 2):
        ct = O.uniform_rns(rng, primes[:L], (2, 2), n)
        tgt = O.uniform_rns(rng, primes[:L], (2,), n)
        for i in range(L):
            tgt[0, i, :64] = primes[i] - 1
        dct, dt = up(moai, ct), up(moai, tgt)
        ctx.switch_key(dct, dt, dkey, L, 2)
        got = dct.to_numpy(ct.shape)
        for b in range(2):
            assert (got[b] == octx.switch_key(ct[b], tgt[b], key, L).reshape(2, L, n)).all(), (bits, L, b)
    # the plain NTT under both disciplines, lazy [0,4q) input included in the unguarded bound
    x = O.uniform_rns(rng, primes, (2,), n)
    d = up(moai, x)
    ctx.ntt_forward(d, 2, k)
    assert (d.to_numpy(x.shape) == octx.ntt(x, k)).all()


def test_keyswitch_decrypts(moai, ks_arith):
    """semantic end-to-end: encrypt -> rotate on the GPU -> decrypt gives the rotated message."""
    from ckks_toy import ToyClient, galois_coeffs
    logn = 6
    n = 1 << logn
    primes = O.coeff_modulus_create(n, [51, 46, 46, 51, 58])
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    cl = ToyClient(octx, seed=3)
    rng = np.random.default_rng(1)
    m = [int(v) for v in rng.integers(-(1 << 30), 1 << 30, size=n)]
    L = 4
    ct = cl.encrypt(m, L)
    elt = ctx.galois_elt_from_step(1)
    gk = cl.galois_key(elt)
    dct = up(moai, ct)
    ctx.apply_galois(dct, L, elt, up(moai, gk), 1)
    out = dct.to_numpy(ct.shape)
    d = cl.decrypt(out, 2, L)
    assert max(abs(a - b) for a, b in zip(d, galois_coeffs(m, elt, n))) < (1 << 16)
    assert (out == octx.apply_galois(ct, L, elt, gk).reshape(out.shape)).all()


def test_modraise(moai, env12):
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(9)
    B, Lout = 2, 4
    x = O.uniform_rns(rng, primes[:1], (B, 2), n)
    x[0, 0, 0, :4] = [0, primes[0] // 2, primes[0] // 2 + 1, primes[0] - 1]
    dx = up(moai, x)
    do = moai.DeviceBuffer(B * 2 * Lout * n)
    ctx.modraise(dx, do, Lout, B)
    got = do.to_numpy((B, 2, Lout, n))
    for b in range(B):
        assert (got[b] == octx.modraise(x[b], Lout)).all()


def test_moai_chain_level_ops_n16(moai, ks_arith):
    """MOAI's real parameters (N = 2^16, the 36-prime chain): NTT on all 36 primes, rescale and one
    key switch at a low level, against the oracle."""
    logn = 16
    n = 1 << logn
    primes = O.coeff_modulus_create(n, MOAI_BITS)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    rng = np.random.default_rng(16)
    k = len(primes)
    x = O.uniform_rns(rng, primes, (1,), n)
    d = up(moai, x)
    ctx.ntt_forward(d, 1, k)
    assert (d.to_numpy(x.shape) == octx.ntt(x, k, batch=True)).all()
    ctx.ntt_inverse(d, 1, k)
    assert (d.to_numpy(x.shape) == x).all()
    L = 3
    key = O.uniform_rns(rng, primes, (k - 1, 2), n)  # 1.3 GB, the reference's key size
    ct = O.uniform_rns(rng, primes[:L], (1, 2), n)
    dct = up(moai, ct)
    elt = ctx.galois_elt_from_step(1)
    ctx.apply_galois(dct, L, elt, up(moai, key), 1)
    got = dct.to_numpy(ct.shape)
    assert (got[0] == octx.apply_galois(ct[0], L, elt, key).reshape(2, L, n)).all()
    do = moai.DeviceBuffer(2 * (L - 1) * n)
    ctx.rescale(dct, do, 2, L, 1)
    assert (do.to_numpy((2, L - 1, n)) == octx.rescale(got[0], 2, L)).all()


def test_config2_shape_properties(moai):
    """BASELINE config 2 shape (N = 2^16, 44 x 60-bit primes), reduced batch: forward result checked
    against the oracle on every row of one ciphertext and by round trip + linearity on the batch."""
    logn, L = 16, 44
    n = 1 << logn
    primes = O.coeff_modulus_create(n, [60] * L)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    rng = np.random.default_rng(1)
    B = 4
    x = O.uniform_rns(rng, primes, (B, 2), n)
    y = O.uniform_rns(rng, primes, (B, 2), n)
    dx, dy = up(moai, x), up(moai, y)
    ds = moai.DeviceBuffer(x.size)
    ctx.add(dx, dy, ds, B * 2, L)
    ctx.ntt_forward(dx, B * 2, L)
    ctx.ntt_forward(dy, B * 2, L)
    ctx.ntt_forward(ds, B * 2, L)
    fx = dx.to_numpy(x.shape)
    assert (fx[0] == octx.ntt(x[0], L, batch=True)).all()
    # linearity: NTT(x + y) == NTT(x) + NTT(y)
    dl = moai.DeviceBuffer(x.size)
    ctx.add(dx, dy, dl, B * 2, L)
    assert (dl.to_numpy() == ds.to_numpy()).all()
    ctx.ntt_inverse(dx, B * 2, L)
    assert (dx.to_numpy(x.shape) == x).all()


def test_config2_full_batch_properties(moai):
    """BASELINE configs[1] at its FULL size -- N = 2^16, 44 x 60-bit primes, 256 ciphertexts x 2 polynomials = 11.8 GB resident --
    through size-independent properties, all checked on the device: the round trip INTT(NTT(x)) = x on every one of the 22 528
    rows, linearity NTT(x + y) = NTT(x) + NTT(y) on every row of a second half-batch, and the oracle on the 44 rows of the last
    ciphertext's second polynomial (the far end of the buffer).  Residues generated on the device like bench.py's."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs torch on the GPU box for device-side generation")
    logn, L, B = 16, 44, 256
    n = 1 << logn
    primes = O.coeff_modulus_create(n, [60] * L)
    ctx = moai.Context(logn, primes)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
    for i, q in enumerate(primes):
        x[:, :, i, :] = torch.randint(0, q, (B, 2, n), dtype=torch.int64, device=dev, generator=gen)
    x[0, 0, :, :8] = 0
    for i, q in enumerate(primes):
        x[0, 1, i, :8] = q - 1
    ref = x.clone()
    ctx.ntt_forward(x.data_ptr(), B * 2, L, stream=st)
    torch.cuda.synchronize()
    assert not torch.equal(x, ref)
    last = x[B - 1, 1].cpu().numpy().view(np.uint64)
    want = O.Context(logn, primes).ntt(ref[B - 1, 1].cpu().numpy().view(np.uint64).reshape(1, L, n), L)[0]
    assert (last == want).all()
    # linearity on the first half against the second half, row by row: NTT(a) + NTT(b) == NTT(a + b)  (mod q, canonical)
    H = B // 2
    s = torch.empty((H, 2, L, n), dtype=torch.int64, device=dev)
    ctx.add(ref[:H].contiguous().data_ptr(), ref[H:].contiguous().data_ptr(), s.data_ptr(), H * 2, L, stream=st)
    ctx.ntt_forward(s.data_ptr(), H * 2, L, stream=st)
    t = torch.empty_like(s)
    ctx.add(x[:H].contiguous().data_ptr(), x[H:].contiguous().data_ptr(), t.data_ptr(), H * 2, L, stream=st)
    torch.cuda.synchronize()
    assert torch.equal(s, t)
    del s, t
    ctx.ntt_inverse(x.data_ptr(), B * 2, L, stream=st)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    del x, ref
    ctx.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("logn", [12, 13, 15, 16])
def test_forward_ntt_guard_every_second_stage(moai, logn):
    """59..61-bit primes take the integer butterflies with one guard per two stages (modarith.hip.h M_GUARD2, values up
    to 8q < 2^64): canonical extremes and lazy inputs up to 4q - 1, the documented input range, against the oracle."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, [61, 61, 60, 59])
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k = len(primes)
    rng = np.random.default_rng(100 + logn)
    x = O.uniform_rns(rng, primes, (4,), n)
    for i, q in enumerate(primes):
        x[0, i, :] = q - 1
        x[1, i, ::2] = 0
        x[1, i, 1::2] = q - 1
    want = octx.ntt(x, k)
    d = up(moai, x)
    ctx.ntt_forward(d, 4, k)
    assert (d.to_numpy(x.shape) == want).all()
    lazy = x.copy()
    for i, q in enumerate(primes):
        lazy[:, i, :] += np.uint64(3 * q)  # [3q, 4q): 61-bit primes keep this below 2^63
    d = up(moai, lazy)
    ctx.ntt_forward(d, 4, k)
    assert (d.to_numpy(x.shape) == want).all()


def test_failed_allocation_does_not_poison_later_launches(moai, env12):
    """A refused hipMalloc is reported through the return value only: callers (the shim's block pool) free memory
    and retry, and the launch checks that follow must not trip over a stale "out of memory"."""
    import ctypes as C

    logn, primes, octx, ctx = env12
    p = C.c_void_p()
    rc = moai.hip.lib().moai_malloc(C.byref(p), C.c_size_t(1 << 46))  # 64 TiB
    assert rc != 0 and not p.value
    n = 1 << logn
    x = O.uniform_rns(np.random.default_rng(3), primes[:2], (1,), n)
    d = up(moai, x)
    ctx.ntt_forward(d, 1, 2)
    assert (d.to_numpy(x.shape) == octx.ntt(x, 2)).all()


def test_empty_and_degenerate_calls(moai, env12):
    """zero-sized batches are no-ops; L = 1 (last level) works; invalid levels are refused."""
    logn, primes, octx, ctx = env12
    n = 1 << logn
    rng = np.random.default_rng(21)
    x = O.uniform_rns(rng, primes[:1], (2,), n)
    d = up(moai, x)
    ctx.ntt_forward(d, 0, 1)
    ctx.add(d, d, d, 0, 1)
    ctx.switch_key(d, d, d, 1, 0)
    assert (d.to_numpy(x.shape) == x).all()
    ctx.ntt_forward(d, 2, 1)
    assert (d.to_numpy(x.shape) == octx.ntt(x, 1)).all()
    with pytest.raises(moai.MoaiError):
        ctx.ntt_forward(d, 1, 1, prime_index=[len(primes)])  # prime index out of range
    with pytest.raises(moai.MoaiError):
        ctx.add(d, d, d, 1, len(primes) + 1)  # more rows than the context has primes
    with pytest.raises(moai.MoaiError):
        ctx.switch_key(d, d, d, len(primes), 1)  # L exceeds the key's decomposition size
    # the fused sums: an empty sum and more than 64 terms are refused, a batch of zero polynomials is a no-op
    with pytest.raises(moai.MoaiError, match="empty sum"):
        ctx.ct_dot(d, d, d, 0, 1)
    with pytest.raises(moai.MoaiError, match="terms"):
        ctx.ct_pt_dot(d, d, d, [0] * 65, [0] * 65, 1, 1)
    with pytest.raises(moai.MoaiError, match="terms"):
        ctx.ct_pt_dot(d, d, d, [], [], 1, 1)
    ctx.ct_pt_dot(d, d, d, [0], [0], 0, 1)
    assert (d.to_numpy(x.shape) == octx.ntt(x, 1)).all()
    with pytest.raises(moai.MoaiError):
        moai.hip._check(moai.hip.lib().moai_set_tuning(None, 1))


def test_coop_single_launch_ntt_agrees(moai):
    """the opt-in single-launch transform (MOAI_NTT_COOP=1, per-XCD queues) against the oracle"""
    import subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle as O, __graft_entry__ as g
m = g.load_package()
for logn, bits in ((12, [60, 46]), (14, [58, 51, 46]), (16, [60, 51, 46, 58])):
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), m.Context(logn, primes)
    x = O.uniform_rns(np.random.default_rng(logn), primes, (5,), n)
    d = m.DeviceBuffer.from_numpy(x)
    ctx.ntt_forward(d, 5, len(primes)); assert (d.to_numpy(x.shape) == octx.ntt(x, len(primes))).all()
    ctx.ntt_inverse(d, 5, len(primes)); assert (d.to_numpy(x.shape) == x).all()
print("ok")
'''
    env = dict(os.environ, MOAI_NTT_COOP="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("bits,rows,cols", [([51, 46, 46, 58], 70, 19), ([60, 61, 60], 33, 5), ([46, 51], 3, 16), ([51, 51, 47, 58], 48, 33)])
def test_ct_pt_matmul_matches_multiply_plain_loop(moai, bits, rows, cols):
    """moai_ct_pt_matmul == the reference loop of multiply_plain(scalar plaintext) + add_inplace
    (Ct_pt_matrix_mul.hpp:19-38), including the 32-term accumulator folds and ragged column groups."""
    logn = 12
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    L = len(primes) - 1
    rng = np.random.default_rng(rows * cols)
    x = O.uniform_rns(rng, primes[:L], (rows, 2), n)
    w = np.empty((L, rows, cols), dtype=np.uint64)
    for r in range(L):
        w[r] = rng.integers(0, primes[r], size=(rows, cols), dtype=np.uint64)
    # worst-case magnitudes: q-1 in the first coefficients of EVERY row against a column of q-1 weights
    w[:, :, 0] = np.array([q - 1 for q in primes[:L]], dtype=np.uint64)[:, None]
    x[:, :, :, :8] = np.array([q - 1 for q in primes[:L]], dtype=np.uint64)[None, None, :, None]
    dx, dw = up(moai, x), up(moai, w)
    dout = moai.DeviceBuffer(cols * 2 * L * n)
    ctx.ct_pt_matmul(dx, dw, dout, rows, cols, 2, L)
    got = dout.to_numpy((cols, 2, L, n))
    # primes below 2^51 take the exact-FP64 kernel, the others the integer one; with the FP64 kernel switched off: the same bits
    moai.hip.set_tuning("MOAI_MATMUL_FP", 0)
    try:
        ctx.ct_pt_matmul(dx, dw, dout, rows, cols, 2, L)
        assert (dout.to_numpy((cols, 2, L, n)) == got).all()
    finally:
        moai.hip.set_tuning("MOAI_MATMUL_FP", 1)
    for c in (0, cols // 2, cols - 1):
        acc = np.zeros((2, L, n), dtype=np.uint64)
        for j in range(rows):
            pt = np.stack([np.full(n, w[r, j, c], dtype=np.uint64) for r in range(L)])
            acc = octx.add(acc, octx.multiply_plain(x[j], 2, L, pt), 2, L)
        assert (got[c] == acc).all(), c


def test_key_switch_replays_from_a_hip_graph(moai, ks_arith):
    """the ~80 launches of one rotate captured once into a hipGraph (through torch's capture API, which
    is plumbing here) and replayed on new data: same bits as the oracle, one launch per call."""
    torch = pytest.importorskip("torch")
    logn = 13
    n = 1 << logn
    primes = O.coeff_modulus_create(n, [51, 46, 46, 46, 58])
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k, L = len(primes), 4
    rng = np.random.default_rng(5)
    key = O.uniform_rns(rng, primes, (k - 1, 2), n)
    dkey = up(moai, key)
    elt = ctx.galois_elt_from_step(1)
    ct_static = torch.zeros((2, L, n), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # warm-up on the capture stream: builds the Galois table and sizes this stream's arena
        ctx.apply_galois(ct_static.data_ptr(), L, elt, dkey, 1, stream=side.cuda_stream)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        ctx.apply_galois(ct_static.data_ptr(), L, elt, dkey, 1, stream=torch.cuda.current_stream().cuda_stream)
    for seed in (1, 2):
        ct = O.uniform_rns(np.random.default_rng(seed), primes[:L], (2,), n)
        ct_static.copy_(torch.from_numpy(ct.view(np.int64)))
        graph.replay()
        torch.cuda.synchronize()
        got = ct_static.cpu().numpy().view(np.uint64)
        assert (got == octx.apply_galois(ct, L, elt, key).reshape(2, L, n)).all()


def test_fp64_modes_at_their_size_limits(moai):
    """Primes at the edges of the FP64 arithmetic modes (modarith.hip.h): the largest ones below 2^51 (M_FPR), around
    2^52 / 33 where M_FPN ends, and a 52-bit one that must stay on the integer units; worst-case magnitudes
    (all residues q - 1, key and digits alike) and random data, forward NTT and key switch against the oracle."""
    logn = 12
    n = 1 << logn
    step = 2 * n

    def prime_below(limit, count):
        out, v = [], (limit - 1) // step * step + 1
        while len(out) < count:
            if O.lib().mo_is_prime(v):
                out.append(v)
            v -= step
        return out

    fpn_limit = (1 << 52) // 33
    primes = prime_below(1 << 51, 2) + prime_below(fpn_limit, 1) + prime_below(fpn_limit + 40 * step, 1)[:1] \
        + prime_below(1 << 46, 1) + prime_below(1 << 52, 1) + prime_below(1 << 58, 1)
    assert len(set(primes)) == len(primes)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k = len(primes)
    rng = np.random.default_rng(77)
    moai.hip.set_tuning("MOAI_KS_FP_MIN_ROWS", 0)
    moai.hip.set_tuning("MOAI_MD_FP_MIN_ROWS", 0)
    try:
        # forward NTT: canonical extremes and lazy inputs up to 4q - 1
        x = O.uniform_rns(rng, primes, (3,), n)
        for i, q in enumerate(primes):
            x[0, i, :] = q - 1
            x[1, i, ::2] = 0
            x[1, i, 1::2] = q - 1
        d = up(moai, x)
        ctx.ntt_forward(d, 3, k)
        assert (d.to_numpy(x.shape) == octx.ntt(x, k)).all()
        lazy = x.copy()
        for i, q in enumerate(primes):
            lazy[2, i, :] = (lazy[2, i, :] + np.uint64(3 * q)) if q < (1 << 51) else lazy[2, i, :]
        d = up(moai, lazy)
        ctx.ntt_forward(d, 3, k)
        assert (d.to_numpy(x.shape) == octx.ntt(x, k)).all()  # same residues as the canonical input
        # inverse NTT (FP64 Gentleman-Sande butterflies below 2^51): the same extremes, and lazy inputs below 2q
        d = up(moai, x)
        ctx.ntt_inverse(d, 3, k)
        assert (d.to_numpy(x.shape) == octx.ntt(x, k, inverse=True)).all()
        lazy = x.copy()
        for i, q in enumerate(primes):
            lazy[2, i, :] = (lazy[2, i, :] + np.uint64(q)) if q < (1 << 51) else lazy[2, i, :]
        d = up(moai, lazy)
        ctx.ntt_inverse(d, 3, k)
        assert (d.to_numpy(x.shape) == octx.ntt(x, k, inverse=True)).all()
        # key switch with every digit and key residue at q - 1, then random
        for trial in range(3):
            key = O.uniform_rns(rng, primes, (k - 1, 2), n)
            L = k - 1
            ct = O.uniform_rns(rng, primes[:L], (2, 2), n)
            tgt = O.uniform_rns(rng, primes[:L], (2,), n)
            if trial == 0:
                for i in range(k):
                    key[:, :, i, :] = primes[i] - 1
                # the target is in NTT form: choose it so that its INTT (the digits) is all q - 1
                coeff = np.stack([np.full(n, primes[i] - 1, dtype=np.uint64) for i in range(L)])
                tgt[0] = octx.ntt(coeff[None], L)[0]
            dct, dt, dkey = up(moai, ct), up(moai, tgt), up(moai, key)
            ctx.switch_key(dct, dt, dkey, L, 2)
            got = dct.to_numpy(ct.shape)
            for b in range(2):
                assert (got[b] == octx.switch_key(ct[b], tgt[b], key, L).reshape(2, L, n)).all(), (trial, b)
        # rescale (the mod-down tail under the FP64 modes): every level of this chain, the dropped row's
        # coefficients at 0, q_last - 1 and around q_last / 2 (the rounding boundary), the kept rows at q - 1
        for L in range(k, 1, -1):
            x = O.uniform_rns(rng, primes[:L], (3, 2), n)
            ql = primes[L - 1]
            pattern = np.array([0, ql - 1, ql // 2, ql // 2 + 1, ql // 2 - 1, 1], dtype=np.uint64)
            coeff = np.resize(pattern, n)
            x[0, :, L - 1, :] = octx.ntt(coeff, 1, prime_index=[L - 1]).reshape(n)  # NTT form of that coefficient row
            for i in range(L - 1):
                x[1, :, i, :] = primes[i] - 1
            dx = up(moai, x)
            do = moai.DeviceBuffer(3 * 2 * (L - 1) * n)
            ctx.rescale(dx, do, 2, L, 3)
            got = do.to_numpy((3, 2, L - 1, n))
            for b in range(3):
                assert (got[b] == octx.rescale(x[b], 2, L)).all(), (L, b)
    finally:
        moai.hip.set_tuning("MOAI_KS_FP_MIN_ROWS", 16)
        moai.hip.set_tuning("MOAI_MD_FP_MIN_ROWS", 256)


@pytest.mark.parametrize("logn,bits,L,B", [(12, [51, 46, 46, 51, 58], 4, 3), (12, [60, 46, 51, 61], 3, 2), (13, [46, 46, 51, 58], 2, 1)])
def test_hoisted_rotations_equal_separate_rotations(moai, logn, bits, L, B, ks_arith):
    """moai_apply_galois_hoisted: R rotations of one ciphertext with ONE digit decomposition (the baby steps of MOAI's
    bootstrapping transforms) against the oracle's apply_galois per rotation (SEAL/evaluator.cpp:2563-2665 over
    :2724-3020), bit for bit; steps with positive and negative sign patterns and the conjugation; a ciphertext with a
    zero coefficient in INTT(c1) must take the fallback and still match."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k = len(primes)
    rng = np.random.default_rng(logn * 10 + L)
    steps = [1, 3, -2, 64, 0]  # 0 = conjugation (galois element 2N - 1)
    elts = [ctx.galois_elt_from_step(s) if s else 2 * n - 1 for s in steps]
    keys = [O.uniform_rns(rng, primes, (k - 1, 2), n) for _ in steps]
    dkeys = [up(moai, kk) for kk in keys]
    corrs = [ctx.hoist_correction(dk, e, L) for dk, e in zip(dkeys, elts)]
    ct = O.uniform_rns(rng, primes[:L], (B, 2), n)
    dct = up(moai, ct)
    dout = moai.DeviceBuffer(len(steps) * B * 2 * L * n)
    fell_back = ctx.apply_galois_hoisted(dct, dout, L, elts, dkeys, corrs, B)
    assert not fell_back
    got = dout.to_numpy((len(steps), B, 2, L, n))
    for r, (e, kk) in enumerate(zip(elts, keys)):
        for b in range(B):
            assert (got[r, b] == octx.apply_galois(ct[b], L, e, kk).reshape(2, L, n)).all(), (r, b)
    assert (dct.to_numpy(ct.shape) == ct).all()  # the input is left alone
    # the FP64 modes take four (or two) rotations per pass over the digits; two at most, and one per pass: the same bits
    try:
        for per_pass in (2, 0):
            moai.hip.set_tuning("MOAI_KS_HOIST_PAIR", per_pass)
            assert not ctx.apply_galois_hoisted(dct, dout, L, elts, dkeys, corrs, B)
            assert (dout.to_numpy((len(steps), B, 2, L, n)) == got).all(), per_pass
    finally:
        moai.hip.set_tuning("MOAI_KS_HOIST_PAIR", 4)
    # a transparent-looking input: c1 = NTT(polynomial with zero coefficients) -> the identity does not hold -> fallback
    ct0 = ct.copy()
    sparse = np.zeros((1, L, n), dtype=np.uint64)
    sparse[0, :, 1] = 5
    ct0[0, 1] = octx.ntt(sparse, L)[0]
    dct0 = up(moai, ct0)
    assert ctx.apply_galois_hoisted(dct0, dout, L, elts, dkeys, corrs, B)
    got = dout.to_numpy((len(steps), B, 2, L, n))
    for r, (e, kk) in enumerate(zip(elts, keys)):
        assert (got[r, 0] == octx.apply_galois(ct0[0], L, e, kk).reshape(2, L, n)).all(), r


@pytest.mark.gpu
def test_hoisted_rotations_more_than_one_pass_holds(moai):
    """R = 65 rotations of one ciphertext in one moai_apply_galois_hoisted call: the accumulators of a pass hold 64, so the
    call makes two passes (a GaloisKeys with a dedicated key per step hands MOAI's Q K^T loop 127 children of one node,
    Ct_ct_matrix_mul.hpp:22-31); every rotation against the oracle's apply_galois, bit for bit."""
    logn, bits, L, B = 12, [46, 46, 58], 2, 1
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k = len(primes)
    rng = np.random.default_rng(65)
    steps = list(range(1, 66))
    elts = [ctx.galois_elt_from_step(s) for s in steps]
    # two distinct keys, shared round-robin (the call takes one pointer per rotation; 65 uploads would only cost time)
    keys = [O.uniform_rns(rng, primes, (k - 1, 2), n) for _ in range(2)]
    dkeys2 = [up(moai, kk) for kk in keys]
    dkeys = [dkeys2[r % 2] for r in range(len(steps))]
    corrs = [ctx.hoist_correction(dkeys[r], elts[r], L) for r in range(len(steps))]
    ct = O.uniform_rns(rng, primes[:L], (B, 2), n)
    dct = up(moai, ct)
    dout = moai.DeviceBuffer(len(steps) * B * 2 * L * n)
    assert not ctx.apply_galois_hoisted(dct, dout, L, elts, dkeys, corrs, B)
    got = dout.to_numpy((len(steps), B, 2, L, n))
    for r in (0, 1, 31, 63, 64):
        assert (got[r, 0] == octx.apply_galois(ct[0], L, elts[r], keys[r % 2]).reshape(2, L, n)).all(), r
    # and all 65 against the single-rotation entry point of the same library
    for r in range(len(steps)):
        d1 = up(moai, ct)
        ctx.apply_galois(d1, L, elts[r], dkeys[r], B)
        assert (d1.to_numpy(ct.shape)[0] == got[r, 0]).all(), r


@pytest.mark.gpu
def test_stream_audit_refuses_a_block_on_another_stream(moai):
    """The debug audit behind the shim's stream-ordered block cache (include/moai_hip.h, "stream audit"): a labelled block
    is accepted on its own stream, refused (MOAI_ELOGIC, before anything is enqueued) on any other stream and after it was
    released to the cache; unlabelled memory is never refused."""
    import ctypes as C

    L = moai.hip.lib()
    logn, primes = 10, O.coeff_modulus_create(1 << 10, [46, 46])
    ctx = moai.Context(logn, primes)
    n = 1 << logn
    rng = np.random.default_rng(3)
    a = O.uniform_rns(rng, primes, (1,), n)
    da, db, dout = up(moai, a), up(moai, a), moai.DeviceBuffer(2 * n)
    s1, s2 = C.c_void_p(), C.c_void_p()
    assert L.moai_stream_create(C.byref(s1)) == 0 and L.moai_stream_create(C.byref(s2)) == 0
    was = L.moai_debug_stream_audit(1)
    try:
        ctx.add(da, db, dout, 1, 2, stream=s1)  # nothing labelled: accepted
        L.moai_debug_block_label(da.ptr, 2 * n * 8, s1, 1)
        ctx.add(da, db, dout, 1, 2, stream=s1)  # on its own stream
        with pytest.raises(moai.hip.MoaiError) as e:
            ctx.add(da, db, dout, 1, 2, stream=s2)
        assert e.value.code == -2 and "stream audit" in str(e.value)
        with pytest.raises(moai.hip.MoaiError):
            ctx.add(db, da, dout, 1, 2, stream=None)  # the legacy stream is another stream, too
        with pytest.raises(moai.hip.MoaiError):  # a pointer INSIDE the block
            L_rc = L.moai_memcpy_d2d(dout.ptr, da.ptr + 64, 64, s2)
            moai.hip._check(L_rc)
        L.moai_debug_block_label(da.ptr, 2 * n * 8, s1, 2)  # released to the cache of s1
        with pytest.raises(moai.hip.MoaiError) as e:
            ctx.add(da, db, dout, 1, 2, stream=s1)
        assert "released" in str(e.value)
        L.moai_debug_block_label(da.ptr, 0, None, 0)  # forgotten
        ctx.add(da, db, dout, 1, 2, stream=s2)
        L.moai_stream_sync(s1)
        L.moai_stream_sync(s2)
        assert (dout.to_numpy((1, 2, n)) == octx_add(primes, a)).all()
    finally:
        L.moai_debug_block_label(da.ptr, 0, None, 0)
        L.moai_debug_stream_audit(was)
        L.moai_stream_destroy(s1)
        L.moai_stream_destroy(s2)


def octx_add(primes, a):
    q = np.array(primes, dtype=np.uint64)[None, :, None]
    return (a + a) % q


@pytest.mark.gpu
@pytest.mark.parametrize("logn,bits,levels", [(12, [51, 46, 46, 51, 46, 58], 3), (10, [46, 46, 51, 58], 2)])
def test_level_trimmed_key_gives_the_oracles_bits(moai, logn, bits, levels, ks_arith):
    """moai_key_trim: a key cut down to the digits and rows a switch at <= `levels` data primes reads (SEAL/evaluator.cpp:2818,
    2831) gives the oracle's bits in every key-switch entry point -- apply_galois, switch_key, relinearize, the hoisted
    rotations and their correction -- at its level and below; a higher level is refused (MOAI_ERANGE), and the full key it was
    cut from serves that level (re-materialisation = going back to the full key)."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    k = len(primes)
    rng = np.random.default_rng(logn + levels)
    key = O.uniform_rns(rng, primes, (k - 1, 2), n)
    dfull = up(moai, key)
    dtrim = ctx.key_trim(dfull, levels)
    assert dtrim.n_words == levels * 2 * (levels + 1) * n < dfull.n_words
    elt = ctx.galois_elt_from_step(3)
    for L in range(1, levels + 1):
        ct = O.uniform_rns(rng, primes[:L], (2, 2), n)
        want = [octx.apply_galois(ct[b], L, elt, key).reshape(2, L, n) for b in range(2)]
        for dk in (dtrim, dfull):
            d = up(moai, ct)
            ctx.apply_galois(d, L, elt, dk, 2)
            got = d.to_numpy(ct.shape)
            assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), L
        # relinearize: a size-3 ciphertext with the trimmed key as the relinearization key
        ct3 = O.uniform_rns(rng, primes[:L], (1, 3), n)
        dout = moai.DeviceBuffer(2 * L * n)
        ctx.relinearize(up(moai, ct3), dtrim, dout, L, 1)
        assert (dout.to_numpy((2, L, n)) == octx.relinearize(ct3[0], key, L)).all(), L
    # hoisted rotations with the trimmed key and a correction computed FROM the trimmed key
    if logn >= 12:
        L = levels
        elts = [ctx.galois_elt_from_step(s) for s in (1, 2, 5)]
        corr_t = [ctx.hoist_correction(dtrim, e, L) for e in elts]
        corr_f = [ctx.hoist_correction(dfull, e, L) for e in elts]
        for a, b in zip(corr_t, corr_f):
            assert (a.to_numpy() == b.to_numpy()).all()
        ct = O.uniform_rns(rng, primes[:L], (1, 2), n)
        dct = up(moai, ct)
        dout = moai.DeviceBuffer(3 * 2 * L * n)
        assert not ctx.apply_galois_hoisted(dct, dout, L, elts, [dtrim, dfull, dtrim], corr_t, 1)
        got = dout.to_numpy((3, 1, 2, L, n))
        for r, e in enumerate(elts):
            assert (got[r, 0] == octx.apply_galois(ct[0], L, e, key).reshape(2, L, n)).all(), r
    # one level above what the trimmed key holds
    if levels < k - 1:
        L = levels + 1
        ct = O.uniform_rns(rng, primes[:L], (1, 2), n)
        d = up(moai, ct)
        with pytest.raises(moai.hip.MoaiError) as e:
            ctx.apply_galois(d, L, elt, dtrim, 1)
        assert e.value.code == -3 and "trimmed" in str(e.value)
        assert (d.to_numpy(ct.shape) == ct).all()  # refused before anything was enqueued
        ctx.apply_galois(d, L, elt, dfull, 1)
        assert (d.to_numpy(ct.shape)[0] == octx.apply_galois(ct[0], L, elt, key).reshape(2, L, n)).all()
    # a forgotten record: the pointer is a plain key again (and too short to be one -- so only forget before freeing)
    ctx.key_forget(dtrim)


@pytest.mark.gpu
@pytest.mark.parametrize("logn,bits,L,terms", [(12, [51, 46, 46, 58], 3, 37), (10, [60, 61, 46], 3, 16), (13, [46] * 30 + [58], 30, 25)])
def test_scalar_dot_is_the_chain_of_multiply_plain_and_add(moai, logn, bits, L, terms):
    """moai_scalar_dot against the oracle's chain out = x_0 * s_0; out += x_t * s_t (Evaluator::multiply_plain with a scalar-encoded
    plaintext + add_inplace, Ct_pt_matrix_mul.hpp:19-42), bit for bit: term counts that are not a multiple of the sixteen per launch,
    more than 27 rows (fewer terms per launch), 60/61-bit primes, edge scalars 0 and q - 1, with and without a base, in place."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    rng = np.random.default_rng(terms)
    xs = [O.uniform_rns(rng, primes[:L], (2,), n) for _ in range(terms)]
    sc = np.stack([np.array([int(rng.integers(0, q)) for q in primes[:L]], dtype=np.uint64) for _ in range(terms)])
    sc[0, :] = 0
    sc[1, :] = np.array([q - 1 for q in primes[:L]], dtype=np.uint64)
    q = np.array(primes[:L], dtype=object)[None, :, None]
    want = np.zeros((2, L, n), dtype=object)
    for t in range(terms):
        want = (want + xs[t].astype(object) * sc[t].astype(object)[None, :, None]) % q
    dxs = [up(moai, x) for x in xs]
    dout = moai.DeviceBuffer(2 * L * n)
    ctx.scalar_dot(dxs, sc, None, dout, 2, L)
    assert (dout.to_numpy((2, L, n)).astype(object) == want).all()
    # the oracle's own multiply_plain / add chain on the first polynomial row agrees with the big-integer sum
    pt = np.broadcast_to(sc[2][:, None], (L, n)).astype(np.uint64).copy()
    assert (octx.multiply_plain(xs[2], 2, L, pt).astype(object) == (xs[2].astype(object) * sc[2].astype(object)[None, :, None]) % q).all()
    # with a base, accumulating in place
    base = O.uniform_rns(rng, primes[:L], (2,), n)
    dacc = up(moai, base)
    ctx.scalar_dot(dxs, sc, dacc, dacc, 2, L)
    assert (dacc.to_numpy((2, L, n)).astype(object) == (want + base.astype(object)) % q).all()
    with pytest.raises(moai.hip.MoaiError):
        ctx.scalar_dot([dacc], sc[:1], None, dacc, 2, L)  # a term must not be the output
    bad = sc[:1].copy()
    bad[0, 0] = primes[0]
    with pytest.raises(moai.hip.MoaiError):
        ctx.scalar_dot(dxs[:1], bad, None, dout, 2, L)


@pytest.mark.gpu
@pytest.mark.parametrize("logn,bits,L,terms", [(12, [51, 46, 58], 2, 21), (11, [60, 61, 46, 46], 4, 33)])
def test_vector_dot_is_the_chain_of_dyadic_products_and_adds(moai, logn, bits, L, terms):
    """moai_vector_dot: out = base + sum_t x_t (*) p_t against the oracle's multiply_plain + add chain (MOAI's masked products,
    Ct_pt_matrix_mul.hpp:103-170), bit for bit, term counts off the sixteen per launch, with a base, in place."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    rng = np.random.default_rng(terms + L)
    xs = [O.uniform_rns(rng, primes[:L], (2,), n) for _ in range(terms)]
    ps = O.uniform_rns(rng, primes[:L], (terms,), n)
    want = octx.multiply_plain(xs[0], 2, L, ps[0])
    for t in range(1, terms):
        want = octx.add(want, octx.multiply_plain(xs[t], 2, L, ps[t]), 2, L)
    dxs = [up(moai, x) for x in xs]
    dp = up(moai, ps)
    dout = moai.DeviceBuffer(2 * L * n)
    ctx.vector_dot(dxs, dp, None, dout, 2, L)
    assert (dout.to_numpy((2, L, n)) == np.asarray(want).reshape(2, L, n)).all()
    base = O.uniform_rns(rng, primes[:L], (2,), n)
    dacc = up(moai, base)
    ctx.vector_dot(dxs, dp, dacc, dacc, 2, L)
    assert (dacc.to_numpy((2, L, n)) == np.asarray(octx.add(base, want, 2, L)).reshape(2, L, n)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("logn,bits,L,terms", [(12, [51, 46, 58], 2, 19), (11, [60, 61, 46, 46], 4, 33)])
def test_ct_dot_ptrs_is_the_chain_of_multiply_and_add(moai, logn, bits, L, terms):
    """moai_ct_dot_ptrs: out = base + sum_t multiply(x_t, y_t) over ciphertexts in separate buffers against the oracle's ckks_multiply +
    add chain (Ct_ct_matrix_mul.hpp:32-41), bit for bit: pair counts off the sixteen per launch, 60/61-bit primes, a base, in place."""
    n = 1 << logn
    primes = O.coeff_modulus_create(n, bits)
    octx, ctx = O.Context(logn, primes), moai.Context(logn, primes)
    rng = np.random.default_rng(terms * 3 + L)
    xs = [O.uniform_rns(rng, primes[:L], (2,), n) for _ in range(terms)]
    ys = [O.uniform_rns(rng, primes[:L], (2,), n) for _ in range(terms)]
    want = np.asarray(octx.multiply(xs[0], ys[0], L)).reshape(3, L, n)
    for t in range(1, terms):
        want = np.asarray(octx.add(want, np.asarray(octx.multiply(xs[t], ys[t], L)).reshape(3, L, n), 3, L)).reshape(3, L, n)
    dxs, dys = [up(moai, x) for x in xs], [up(moai, y) for y in ys]
    dout = moai.DeviceBuffer(3 * L * n)
    ctx.ct_dot_ptrs(dxs, dys, None, dout, L)
    assert (dout.to_numpy((3, L, n)) == want).all()
    base = O.uniform_rns(rng, primes[:L], (3,), n)
    dacc = up(moai, base)
    ctx.ct_dot_ptrs(dxs, dys, dacc, dacc, L)
    assert (dacc.to_numpy((3, L, n)) == np.asarray(octx.add(base, want, 3, L)).reshape(3, L, n)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_blocks,words", [(1, 2), (7, 4096 * 3), (64, 2 * 65536)])
def test_gather_and_scatter_blocks_copy_every_word(moai, n_blocks, words):
    """moai_gather_blocks / moai_scatter_blocks (the call combiner's packing around a batched operation): separate blocks -> one
    packed array and back in one launch each, every word; 65 blocks and an odd word count are refused."""
    primes = O.coeff_modulus_create(4096, [40, 40])
    ctx = moai.Context(12, primes)
    rng = np.random.default_rng(n_blocks)
    data = [rng.integers(0, 1 << 63, size=words, dtype=np.uint64) for _ in range(n_blocks)]
    blocks = [up(moai, d) for d in data]
    packed = moai.DeviceBuffer(n_blocks * words)
    ctx.gather_blocks(blocks, packed, words)
    assert (packed.to_numpy((n_blocks, words)) == np.stack(data)).all()
    fresh = rng.integers(0, 1 << 63, size=(n_blocks, words), dtype=np.uint64)
    packed2 = up(moai, fresh)
    ctx.scatter_blocks(packed2, blocks, words)
    for i in range(n_blocks):
        assert (blocks[i].to_numpy((words,)) == fresh[i]).all()
    with pytest.raises(Exception):
        ctx.gather_blocks(blocks[:1] * 65, packed, 2)
    with pytest.raises(Exception):
        ctx.gather_blocks(blocks[:1], packed, 3)
