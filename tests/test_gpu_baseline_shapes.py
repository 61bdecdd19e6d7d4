"""GPU parity at the shapes BASELINE.json quotes (configs[2] and configs[3]), through the C ABI, bit-exact against the
CPU oracle (integer path: no tolerance).

configs[2]: Evaluator::rotate_vector's key switch (SEAL/evaluator.cpp:2563-2665 -> :2724-3020) at N = 2^16 on
MOAI's 36-prime chain (include/test/test_full_scheme.hpp:356-378), l = 35 and l = 15 data primes, a batch of 8
ciphertexts per call, key uniform in [0, q_i), under both arithmetic disciplines of the fused kernels and with the
scratch budget lowered so that the digits of 1 < G < l+1 and of G = 1 output moduli are in flight per launch (the
branch batch 64..256 takes in production, keyswitch.hip ks_group_size).  "dnum=3" has no counterpart in the
reference: the reference's per-prime decomposition is what is compared.

configs[3]: one row of the attention block's Q.K^T (include/source/matrix_mul/Ct_ct_matrix_mul.hpp:22-51: 64
rotations by i*256 through the NAF fallback of rotate_internal, SEAL/evaluator.cpp:2699-2721, 64 multiplies, adds,
one relinearize, one rescale) and columns of its X.W product (Ct_pt_matrix_mul.hpp:19-42) at N = 2^16 and chain
index 15 (16 data primes), against the oracle's composition of the same reference calls.

The oracle needs about a second per key switch at l = 35; its calls run on a thread pool (ctypes releases the GIL).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

MOAI_BITS = [51] + [46] * 20 + [51] * 14 + [58]  # include/test/test_full_scheme.hpp:356-378
LOGN = 16
N = 1 << LOGN


def up(m, a):
    return m.DeviceBuffer.from_numpy(a)


def pool_map(fn, items):
    workers = max(1, min(16, (os.cpu_count() or 2)))
    O.lib().mo_set_threads(1)
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


class Env16:
    def __init__(self, moai):
        self.primes = O.coeff_modulus_create(N, MOAI_BITS)
        self.octx = O.Context(LOGN, self.primes)
        self.ctx = moai.Context(LOGN, self.primes)
        self.k = len(self.primes)
        self._keys = {}
        self.moai = moai

    def key(self, seed):
        """a switching key's layout ([k-1 digits][2][k][N], SEAL/kswitchkeys.h:340) filled uniformly mod each prime"""
        if seed not in self._keys:
            rng = np.random.default_rng(1000 + seed)
            host = O.uniform_rns(rng, self.primes, (self.k - 1, 2), N)
            self._keys[seed] = (host, up(self.moai, host))
        return self._keys[seed]


@pytest.fixture(scope="module")
def env16(moai):
    e = Env16(moai)
    yield e
    e._keys.clear()


def set_arith(moai, arith):
    moai.hip.set_tuning("MOAI_KS_FP_MIN_ROWS", 0 if arith == "fp64" else 1 << 40)
    moai.hip.set_tuning("MOAI_MD_FP_MIN_ROWS", 0 if arith == "fp64" else 1 << 40)


def reset_tuning(moai):
    moai.hip.set_tuning("MOAI_KS_FP_MIN_ROWS", 16)
    moai.hip.set_tuning("MOAI_MD_FP_MIN_ROWS", 256)
    moai.hip.set_tuning("MOAI_KS_TMP_MB", 8192)
    moai.hip.set_tuning("MOAI_KS_P1_ITEMS", 8)


@pytest.mark.parametrize("L", [35, 15])
def test_config2_rotate_key_switch_at_moai_levels(moai, env16, L):
    e = env16
    B = 8
    rng = np.random.default_rng(L)
    ct = O.uniform_rns(rng, e.primes[:L], (B, 2), N)
    # edge residues: 0 and q-1 runs in both polynomials of the first ciphertext
    ct[0, :, :, :16] = 0
    for i in range(L):
        ct[0, :, i, 16:32] = e.primes[i] - 1
    key, dkey = e.key(0)
    elt = e.ctx.galois_elt_from_step(1)
    want = pool_map(lambda b: e.octx.apply_galois(ct[b], L, elt, key).reshape(2, L, N), range(B))
    # scratch budgets: all l+1 output moduli in one pair of launches; a few per launch; one per launch (the last one is also too
    # little work for the strided pass's eight-tiles-per-workgroup form, which the first two take in the FP64 arithmetic);
    # and once more with that form switched off at the full budget: the same bits from both forms of the pass
    per_modulus_mb = B * L * N * 8 / (1 << 20)
    budgets = [(8192, 8), (int(per_modulus_mb * 5.5), 8), (int(per_modulus_mb * 1.5), 8), (8192, 1)]
    try:
        for arith in ("fp64", "int64"):
            set_arith(moai, arith)
            for mb, items in budgets:
                moai.hip.set_tuning("MOAI_KS_TMP_MB", mb)
                moai.hip.set_tuning("MOAI_KS_P1_ITEMS", items)
                d = up(moai, ct)
                e.ctx.apply_galois(d, L, elt, dkey, B)
                got = d.to_numpy(ct.shape)
                for b in range(B):
                    assert (got[b] == want[b]).all(), (arith, mb, items, b)
                d.free()
    finally:
        reset_tuning(moai)


def test_config2_relinearize_at_moai_level(moai, env16):
    """relinearize_internal (SEAL/evaluator.cpp:1345-1400) on size-3 ciphertexts, l = 15, batch 4, grouped scratch"""
    e = env16
    L, B = 15, 4
    rng = np.random.default_rng(77)
    ct3 = O.uniform_rns(rng, e.primes[:L], (B, 3), N)
    key, dkey = e.key(0)
    want = pool_map(lambda b: e.octx.relinearize(ct3[b], key, L), range(B))
    try:
        for arith, mb in (("fp64", 8192), ("int64", int(B * L * N * 8 / (1 << 20) * 2.5))):
            set_arith(moai, arith)
            moai.hip.set_tuning("MOAI_KS_TMP_MB", mb)
            dout = moai.DeviceBuffer(B * 2 * L * N)
            e.ctx.relinearize(up(moai, ct3), dkey, dout, L, B)
            got = dout.to_numpy((B, 2, L, N))
            for b in range(B):
                assert (got[b] == want[b]).all(), (arith, b)
    finally:
        reset_tuning(moai)


def naf_steps(step):
    """rotate_internal's fallback (SEAL/evaluator.cpp:2699-2721): the non-zero NAF digits of the step, in NAF order"""
    return [s for s in O.naf(step) if s != 0]


def test_config3_one_row_of_q_kt(moai, env16):
    e = env16
    L = 16  # chain index 15
    cols = 64
    row = 3  # rotation by 3 * 256 = 768 = 1024 - 256: two key switches per rotation
    step = row * 256
    steps = naf_steps(step)
    assert len(steps) == 2 and sum(steps) == step
    rng = np.random.default_rng(3)
    q = O.uniform_rns(rng, e.primes[:L], (cols, 2), N)
    kk = O.uniform_rns(rng, e.primes[:L], (cols, 2), N)
    keys = {s: e.key(10 + i) for i, s in enumerate(steps)}
    relin, drelin = e.key(20)
    elts = {s: e.ctx.galois_elt_from_step(s) for s in steps}

    def ref_product(j):
        r = kk[j]
        for s in steps:
            r = e.octx.apply_galois(r, L, elts[s], keys[s][0]).reshape(2, L, N)
        return e.octx.multiply(q[j], r, L).reshape(3, L, N)

    prods = pool_map(ref_product, range(cols))
    acc = prods[0]
    for j in range(1, cols):
        acc = e.octx.add(acc, prods[j], 3, L)
    want = e.octx.rescale(e.octx.relinearize(acc, relin, L), 2, L)

    try:
        for arith in ("fp64", "int64"):
            set_arith(moai, arith)
            dk = up(moai, kk)
            for s in steps:
                e.ctx.apply_galois(dk, L, elts[s], keys[s][1], cols)
            dq = up(moai, q)
            # the fused sum of products (moai_ct_dot) ...
            dsum = moai.DeviceBuffer(3 * L * N)
            e.ctx.ct_dot(dq, dk, dsum, cols, L)
            got3 = dsum.to_numpy((3, L, N))
            assert (got3 == acc).all(), arith
            # ... and the reference's own multiply + add chain on the first columns
            dp = moai.DeviceBuffer(4 * 3 * L * N)
            e.ctx.ct_multiply(dq, dk, dp, L, 4)
            gp = dp.to_numpy((4, 3, L, N))
            for j in range(4):
                assert (gp[j] == prods[j]).all(), (arith, j)
            d2 = moai.DeviceBuffer(2 * L * N)
            e.ctx.relinearize(dsum, drelin, d2, L, 1)
            dout = moai.DeviceBuffer(2 * (L - 1) * N)
            e.ctx.rescale(d2, dout, 2, L, 1)
            assert (dout.to_numpy((2, L - 1, N)) == want).all(), arith
    finally:
        reset_tuning(moai)


def test_config3_ct_pt_columns(moai, env16):
    """columns of X.W at chain index 15: sum_r multiply_plain(X[r], encode(w[r][c])) then rescale
    (Ct_pt_matrix_mul.hpp:19-42); scalar plaintexts are constant rows (SEAL/ckks.cpp:131-150)"""
    e = env16
    L, rows, cols = 16, 96, 16
    rng = np.random.default_rng(5)
    x = O.uniform_rns(rng, e.primes[:L], (rows, 2), N)
    w = np.empty((L, rows, cols), dtype=np.uint64)
    for r in range(L):
        w[r] = rng.integers(0, e.primes[r], size=(rows, cols), dtype=np.uint64)

    def ref_col(c):
        acc = np.zeros((2, L, N), dtype=np.uint64)
        for j in range(rows):
            pt = np.repeat(w[:, j, c][:, None], N, axis=1)
            acc = e.octx.add(acc, e.octx.multiply_plain(x[j], 2, L, pt), 2, L)
        return acc, e.octx.rescale(acc, 2, L)

    check = [0, 7, 15]
    want = dict(zip(check, pool_map(ref_col, check)))
    dout = moai.DeviceBuffer(cols * 2 * L * N)
    e.ctx.ct_pt_matmul(up(moai, x), up(moai, w), dout, rows, cols, 2, L)
    got = dout.to_numpy((cols, 2, L, N))
    dres = moai.DeviceBuffer(cols * 2 * (L - 1) * N)
    e.ctx.rescale(dout, dres, 2, L, cols)
    res = dres.to_numpy((cols, 2, L - 1, N))
    for c in check:
        assert (got[c] == want[c][0]).all(), c
        assert (res[c] == want[c][1]).all(), c


def test_hoisted_baby_step_rotations_at_moai_top_level(moai, env16):
    """the baby steps of a bootstrapping transform as ONE hoisted call at N = 2^16, l = 35, batch 2: three rotations of the
    same ciphertexts against the oracle's separate rotate_vector calls, under both arithmetics"""
    e = env16
    L, B = 35, 2
    rng = np.random.default_rng(350)
    ct = O.uniform_rns(rng, e.primes[:L], (B, 2), N)
    steps = [1, 1024, -4]
    elts = [e.ctx.galois_elt_from_step(s) for s in steps]
    keys = [e.key(30 + i) for i in range(len(steps))]
    want = pool_map(lambda rb: e.octx.apply_galois(ct[rb[1]], L, elts[rb[0]], keys[rb[0]][0]).reshape(2, L, N),
                    [(r, b) for r in range(len(steps)) for b in range(B)])
    corrs = [e.ctx.hoist_correction(keys[r][1], elts[r], L) for r in range(len(steps))]
    try:
        for arith in ("fp64", "int64"):
            set_arith(moai, arith)
            dout = moai.DeviceBuffer(len(steps) * B * 2 * L * N)
            assert not e.ctx.apply_galois_hoisted(up(moai, ct), dout, L, elts, [kk[1] for kk in keys], corrs, B)
            got = dout.to_numpy((len(steps), B, 2, L, N))
            for r in range(len(steps)):
                for b in range(B):
                    assert (got[r, b] == want[r * B + b]).all(), (arith, r, b)
            dout.free()
    finally:
        reset_tuning(moai)
