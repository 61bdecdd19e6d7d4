"""A tiny CKKS *client* (secret key, symmetric encryption, key-switch keys, decryption) used only by
tests to give the evaluator path meaningful inputs and to check results by decrypt-and-compare, the
way the reference's own evaluator tests do (native/tests/seal/evaluator.cpp:2971-4293).

Key layout follows the reference: a key-switch key is vector<PublicKey> of length k-1, each a
size-2 ciphertext at the key level, i.e. uint64[k-1][2][k][N]
(native/src/seal/kswitchkeys.h:340, keygenerator.cpp:303-336).  TEST INFRASTRUCTURE.
"""
import numpy as np

import oracle as O


def _negacyclic_mul(a, b, n):
    """schoolbook negacyclic product of integer coefficient lists (small n only)."""
    res = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k >= n:
                res[k - n] -= ai * bj
            else:
                res[k] += ai * bj
    return res


def galois_coeffs(m, elt, n):
    """m(X) -> m(X^elt) mod X^n+1 on integer coefficients (galois.cpp:147-190)."""
    out = [0] * n
    for i, c in enumerate(m):
        raw = i * elt
        idx = raw % n
        out[idx] = -c if (raw // n) & 1 else c
    return out


class ToyClient:
    def __init__(self, ctx, seed=1, hamming_weight=None):
        self.ctx = ctx
        self.n = ctx.n
        self.k = ctx.k
        self.primes = ctx.primes
        self.rng = np.random.default_rng(seed)
        n = self.n
        if hamming_weight is None:
            s = self.rng.integers(-1, 2, size=n)
        else:  # sparse ternary secret (fork: util/rlwe.cpp:40-97)
            s = np.zeros(n, dtype=np.int64)
            pos = self.rng.choice(n, size=hamming_weight, replace=False)
            s[pos] = self.rng.choice([-1, 1], size=hamming_weight)
        self.s = [int(x) for x in s]
        self.s_ntt = self._to_ntt(self.s, self.k)  # [k][N]

    # -- helpers -------------------------------------------------------------------------------
    def _to_rns(self, coeffs, L, prime_index=None):
        idx = list(range(L)) if prime_index is None else prime_index
        out = np.empty((len(idx), self.n), dtype=np.uint64)
        for r, i in enumerate(idx):
            q = self.primes[i]
            out[r] = np.array([c % q for c in coeffs], dtype=np.uint64)
        return out

    def _to_ntt(self, coeffs, L, prime_index=None):
        return self.ctx.ntt(self._to_rns(coeffs, L, prime_index)[None], len(prime_index) if prime_index else L,
                            prime_index=prime_index)[0]

    def _dyadic(self, a, b, L, prime_index=None):
        out = np.empty_like(a)
        idx = list(range(L)) if prime_index is None else prime_index
        for r, i in enumerate(idx):
            q = self.primes[i]
            out[r] = np.array([(int(x) * int(y)) % q for x, y in zip(a[r], b[r])], dtype=np.uint64)
        return out

    def _noise(self):
        return [int(round(x)) for x in self.rng.normal(0, 3.2, size=self.n)]

    # -- encryption ------------------------------------------------------------------------------
    def encrypt_zero_symmetric(self, L, prime_index=None):
        """(c0, c1) = (-(a s) + e, a), NTT form, [2][L][N] (util/rlwe.cpp encrypt_zero_symmetric)."""
        idx = list(range(L)) if prime_index is None else prime_index
        a = O.uniform_rns(self.rng, [self.primes[i] for i in idx], (), self.n)
        e = self._to_ntt(self._noise(), len(idx), prime_index=idx)
        s = self.s_ntt[idx]
        c0 = np.empty_like(a)
        for r, i in enumerate(idx):
            q = self.primes[i]
            c0[r] = np.array([(int(ev) - int(av) * int(sv)) % q for av, sv, ev in zip(a[r], s[r], e[r])],
                             dtype=np.uint64)
        return np.stack([c0, a])

    def encrypt(self, m_coeffs, L):
        ct = self.encrypt_zero_symmetric(L)
        m = self._to_ntt(m_coeffs, L)
        for i in range(L):
            q = self.primes[i]
            ct[0, i] = (ct[0, i] + m[i]) % np.uint64(q)
        return ct

    def decrypt(self, ct, size, L):
        """returns centred integer coefficients of c0 + c1 s + c2 s^2 mod Q_L (python ints)."""
        ct = np.asarray(ct, dtype=np.uint64).reshape(size, L, self.n)
        acc = [[int(x) for x in ct[0, i]] for i in range(L)]
        spow = [[int(x) for x in self.s_ntt[i]] for i in range(L)]
        for p in range(1, size):
            for i in range(L):
                q = self.primes[i]
                acc[i] = [(a + int(c) * sp) % q for a, c, sp in zip(acc[i], ct[p, i], spow[i])]
                spow[i] = [(sp * int(s)) % q for sp, s in zip(spow[i], self.s_ntt[i])]
        rows = np.array(acc, dtype=np.uint64)
        coeff = self.ctx.ntt(rows[None], L, inverse=True)[0]
        # CRT lift
        Q = 1
        for i in range(L):
            Q *= self.primes[i]
        out = [0] * self.n
        for i in range(L):
            q = self.primes[i]
            Qi = Q // q
            inv = pow(Qi % q, -1, q)
            f = (Qi * inv) % Q
            for j in range(self.n):
                out[j] = (out[j] + int(coeff[i, j]) * f) % Q
        return [v - Q if v > Q // 2 else v for v in out]

    # -- keys ------------------------------------------------------------------------------------
    def kswitch_key(self, new_key_ntt):
        """keygenerator.cpp:303-336: key[J] = Enc_keylevel(0) with c0[J] += (p mod q_J) * new_key[J]."""
        k = self.k
        key = np.empty((k - 1, 2, k, self.n), dtype=np.uint64)
        p = self.primes[k - 1]
        for J in range(k - 1):
            ct = self.encrypt_zero_symmetric(k)
            q = self.primes[J]
            factor = p % q
            ct[0, J] = np.array([(int(c) + factor * int(nk)) % q for c, nk in zip(ct[0, J], new_key_ntt[J])],
                                dtype=np.uint64)
            key[J] = ct
        return key

    def relin_key(self):
        s2 = self._dyadic(self.s_ntt, self.s_ntt, self.k)
        return self.kswitch_key(s2)

    def galois_key(self, elt):
        tab = O.galois_table_ntt(self.ctx.logn, elt)
        sg = self.s_ntt[:, tab]
        return self.kswitch_key(np.ascontiguousarray(sg))


# -- a slow but independent CKKS encoder for semantic checks of the rotation convention ----------
def slot_roots(n, gen=5):
    """zeta^(gen^j) for j < n/2 with zeta = exp(i pi / n) (ckks.cpp:36-50: gen = 5 in the fork)."""
    m = 2 * n
    pos = 1
    out = []
    for _ in range(n // 2):
        out.append(np.exp(1j * np.pi * pos / n))
        pos = (pos * gen) % m
    return np.array(out)


def decode_slots(coeffs, n, scale, gen=5):
    r = slot_roots(n, gen)
    c = np.array([float(x) for x in coeffs])
    powers = np.vander(r, n, increasing=True)
    return powers @ c / scale


def encode_slots(z, n, scale, gen=5):
    """least-squares inverse of decode (small n only)."""
    r = slot_roots(n, gen)
    V = np.vander(r, n, increasing=True)
    A = np.vstack([V, np.conj(V)])
    b = np.concatenate([z, np.conj(z)])
    c = np.linalg.solve(A, b).real
    return [int(round(x * scale)) for x in c]
