"""ctypes binding of oracle/libmoai_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's hot path (oracle/moai_oracle.h).  It is the
checker for the HIP path; nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ODIR, "libmoai_oracle.so")

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


class Modulus(C.Structure):
    _fields_ = [("value", u64), ("const_ratio", u64 * 3), ("bit_count", C.c_int)]


class MulOp(C.Structure):
    _fields_ = [("operand", u64), ("quotient", u64)]


class NttTables(C.Structure):
    _fields_ = [
        ("coeff_count_power", C.c_int),
        ("coeff_count", C.c_size_t),
        ("modulus", Modulus),
        ("root", u64),
        ("inv_root", u64),
        ("root_powers", C.POINTER(MulOp)),
        ("inv_root_powers", C.POINTER(MulOp)),
        ("inv_degree_modulo", MulOp),
    ]


def build():
    """(Re)build the oracle shared library with its own Makefile (gcc only)."""
    src_m = max(os.path.getmtime(os.path.join(_ODIR, f)) for f in ("moai_oracle.c", "moai_oracle.h", "Makefile"))
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.check_call(["make", "-C", _ODIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _declare(_lib)
    return _lib


def _declare(L):
    mp = C.POINTER(Modulus)
    tp = C.POINTER(NttTables)
    sz = C.c_size_t
    vp = C.c_void_p
    sig = {
        "mo_modulus_init": (None, [mp, u64]),
        "mo_barrett_reduce_64": (u64, [u64, mp]),
        "mo_barrett_reduce_128": (u64, [u64p, mp]),
        "mo_multiply_uint_mod": (u64, [u64, u64, mp]),
        "mo_mulop_set": (None, [C.POINTER(MulOp), u64, mp]),
        "mo_multiply_uint_mod_op": (u64, [u64, MulOp, mp]),
        "mo_multiply_uint_mod_lazy": (u64, [u64, MulOp, mp]),
        "mo_exponentiate_uint_mod": (u64, [u64, u64, mp]),
        "mo_try_invert_uint_mod": (C.c_int, [u64, u64, u64p]),
        "mo_is_prime": (C.c_int, [u64]),
        "mo_get_primes": (C.c_int, [u64, C.c_int, sz, u64p]),
        "mo_coeff_modulus_create": (C.c_int, [sz, C.POINTER(C.c_int), sz, u64p]),
        "mo_is_primitive_root": (C.c_int, [u64, u64, mp]),
        "mo_try_minimal_primitive_root": (C.c_int, [u64, mp, u64p]),
        "mo_naf": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int]),
        "mo_ntt_tables_init": (C.c_int, [tp, C.c_int, u64]),
        "mo_ntt_tables_free": (None, [tp]),
        "mo_ntt_negacyclic_harvey_lazy": (None, [vp, tp]),
        "mo_ntt_negacyclic_harvey": (None, [vp, tp]),
        "mo_inverse_ntt_negacyclic_harvey_lazy": (None, [vp, tp]),
        "mo_inverse_ntt_negacyclic_harvey": (None, [vp, tp]),
        "mo_modulo_poly_coeffs": (None, [vp, sz, mp, vp]),
        "mo_add_poly_coeffmod": (None, [vp, vp, sz, mp, vp]),
        "mo_sub_poly_coeffmod": (None, [vp, vp, sz, mp, vp]),
        "mo_negate_poly_coeffmod": (None, [vp, sz, mp, vp]),
        "mo_add_poly_scalar_coeffmod": (None, [vp, sz, u64, mp, vp]),
        "mo_multiply_poly_scalar_coeffmod": (None, [vp, sz, u64, mp, vp]),
        "mo_dyadic_product_coeffmod": (None, [vp, vp, sz, mp, vp]),
        "mo_galois_elt_from_step": (C.c_uint32, [C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_int)]),
        "mo_galois_elts_all": (C.c_int, [C.c_int, C.c_uint32, u32p]),
        "mo_galois_table_ntt": (None, [C.c_int, C.c_uint32, vp]),
        "mo_apply_galois_ntt": (None, [vp, vp, sz, vp]),
        "mo_apply_galois": (None, [vp, C.c_int, C.c_uint32, mp, vp]),
        "mo_context_create": (vp, [C.c_int, u64p, sz]),
        "mo_context_destroy": (None, [vp]),
        "mo_ntt_rns": (None, [vp, vp, sz, sz, vp, C.c_int]),
        "mo_batch_ntt": (None, [vp, vp, sz, sz, vp, C.c_int]),
        "mo_divide_and_round_q_last_ntt_inplace": (None, [vp, vp, sz]),
        "mo_rescale_to_next": (None, [vp, vp, sz, sz, vp]),
        "mo_mod_switch_drop": (None, [vp, vp, sz, sz, sz, vp]),
        "mo_ckks_multiply": (None, [vp, vp, vp, sz]),
        "mo_ckks_square": (None, [vp, vp, sz]),
        "mo_ckks_multiply_general": (None, [vp, vp, sz, vp, sz, sz, vp]),
        "mo_relinearize_general": (None, [vp, vp, sz, sz, vp, sz]),
        "mo_multiply_plain": (None, [vp, vp, sz, sz, vp]),
        "mo_ct_add": (None, [vp, vp, vp, sz, sz, vp]),
        "mo_ct_sub": (None, [vp, vp, vp, sz, sz, vp]),
        "mo_ct_negate": (None, [vp, vp, sz, sz, vp]),
        "mo_switch_key_inplace": (None, [vp, vp, vp, vp, sz]),
        "mo_batch_switch_key": (None, [vp, vp, vp, vp, sz, sz]),
        "mo_relinearize": (None, [vp, vp, vp, sz]),
        "mo_apply_galois_inplace": (None, [vp, vp, sz, C.c_uint32, vp]),
        "mo_modraise": (None, [vp, vp, sz, vp]),
        "mo_ckks_tables_create": (vp, [C.c_int]),
        "mo_ckks_tables_free": (None, [vp]),
        "mo_fft_transform_from_rev": (None, [vp, C.c_int, vp, C.POINTER(C.c_double)]),
        "mo_fft_transform_to_rev": (None, [vp, C.c_int, vp]),
        "mo_ckks_encode": (C.c_int, [vp, vp, vp, C.c_int, sz, sz, vp, C.c_double, C.c_int, vp,
                                     C.POINTER(C.c_int)]),
        "mo_ckks_encode_scalar": (C.c_int, [vp, C.c_double, sz, vp, C.c_double, C.c_int, vp]),
        "mo_max_threads": (C.c_int, []),
        "mo_set_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


# ------------------------------------------------------------------------------------------------
# pythonic helpers
# ------------------------------------------------------------------------------------------------
def modulus(value):
    m = Modulus()
    lib().mo_modulus_init(C.byref(m), int(value))
    return m


def mulop(operand, m):
    y = MulOp()
    lib().mo_mulop_set(C.byref(y), int(operand), C.byref(m))
    return y


def ptr(a):
    assert a.dtype in (np.uint64, np.uint32, np.float64, np.complex128)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def coeff_modulus_create(n, bits):
    arr = (C.c_int * len(bits))(*bits)
    out = (u64 * len(bits))()
    rc = lib().mo_coeff_modulus_create(n, arr, len(bits), out)
    if rc != 0:
        raise RuntimeError("failed to find enough qualifying primes")
    return [int(x) for x in out]


class Tables:
    def __init__(self, logn, q):
        self.t = NttTables()
        if lib().mo_ntt_tables_init(C.byref(self.t), logn, int(q)) != 0:
            raise ValueError("invalid modulus")
        self.n = 1 << logn
        self.q = int(q)

    def __del__(self):
        try:
            lib().mo_ntt_tables_free(C.byref(self.t))
        except Exception:
            pass

    def root_powers(self):
        return [self.t.root_powers[i].operand for i in range(self.n)]

    def ntt(self, a, lazy=False):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        fn = lib().mo_ntt_negacyclic_harvey_lazy if lazy else lib().mo_ntt_negacyclic_harvey
        fn(ptr(a), C.byref(self.t))
        return a

    def intt(self, a, lazy=False):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        fn = lib().mo_inverse_ntt_negacyclic_harvey_lazy if lazy else lib().mo_inverse_ntt_negacyclic_harvey
        fn(ptr(a), C.byref(self.t))
        return a


class Context:
    """The modulus chain: primes[0..k), last one the special prime (oracle/moai_oracle.h mo_context)."""

    def __init__(self, logn, primes):
        self.logn = logn
        self.n = 1 << logn
        self.primes = [int(p) for p in primes]
        self.k = len(primes)
        arr = (u64 * self.k)(*self.primes)
        self.h = lib().mo_context_create(logn, arr, self.k)
        if not self.h:
            raise ValueError("invalid modulus")

    def __del__(self):
        try:
            lib().mo_context_destroy(self.h)
        except Exception:
            pass

    # data [npoly][L][N]
    def ntt(self, data, L, prime_index=None, inverse=False, batch=False):
        d = np.ascontiguousarray(data, dtype=np.uint64).copy()
        npoly = d.size // (L * self.n)
        pi = None
        if prime_index is not None:
            pi_arr = np.ascontiguousarray(prime_index, dtype=np.uint32)
            pi = ptr(pi_arr)
        fn = lib().mo_batch_ntt if batch else lib().mo_ntt_rns
        fn(self.h, ptr(d), npoly, L, pi, 1 if inverse else 0)
        return d

    def rescale(self, ct, size, L):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.empty((size, L - 1, self.n), dtype=np.uint64)
        lib().mo_rescale_to_next(self.h, ptr(ct), size, L, ptr(out))
        return out

    def mod_drop(self, ct, size, L, drop):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.empty((size, L - drop, self.n), dtype=np.uint64)
        lib().mo_mod_switch_drop(self.h, ptr(ct), size, L, drop, ptr(out))
        return out

    def multiply(self, x, y, L):
        out = np.zeros((3, L, self.n), dtype=np.uint64)
        out[:2] = np.asarray(x, dtype=np.uint64).reshape(2, L, self.n)
        y = np.ascontiguousarray(y, dtype=np.uint64)
        lib().mo_ckks_multiply(self.h, ptr(out), ptr(y), L)
        return out

    def multiply_general(self, x, size_x, y, size_y, L):
        """Evaluator::multiply for any sizes (SEAL/evaluator.cpp:862-900)"""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        y = np.ascontiguousarray(y, dtype=np.uint64)
        out = np.empty((size_x + size_y - 1, L, self.n), dtype=np.uint64)
        lib().mo_ckks_multiply_general(self.h, ptr(x), size_x, ptr(y), size_y, L, ptr(out))
        return out

    def relinearize_general(self, ct, size, dest_size, keys, L):
        """Evaluator::relinearize_internal for any size; keys[t] is the switching key of s^(t+2)"""
        ct = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        keys = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys]
        arr = (C.c_void_p * len(keys))(*[k.ctypes.data for k in keys])
        lib().mo_relinearize_general(self.h, ptr(ct), size, dest_size, arr, L)
        return ct.reshape(size, L, self.n)[:dest_size].copy()

    def square(self, x, L):
        out = np.zeros((3, L, self.n), dtype=np.uint64)
        out[:2] = np.asarray(x, dtype=np.uint64).reshape(2, L, self.n)
        lib().mo_ckks_square(self.h, ptr(out), L)
        return out

    def multiply_plain(self, ct, size, L, plain):
        ct = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        lib().mo_multiply_plain(self.h, ptr(ct), size, L, ptr(plain))
        return ct

    def add(self, a, b, size, L):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.empty_like(a)
        lib().mo_ct_add(self.h, ptr(a), ptr(b), size, L, ptr(r))
        return r

    def sub(self, a, b, size, L):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.empty_like(a)
        lib().mo_ct_sub(self.h, ptr(a), ptr(b), size, L, ptr(r))
        return r

    def negate(self, a, size, L):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        r = np.empty_like(a)
        lib().mo_ct_negate(self.h, ptr(a), size, L, ptr(r))
        return r

    def switch_key(self, ct, target, key, L):
        ct = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        target = np.ascontiguousarray(target, dtype=np.uint64)
        key = np.ascontiguousarray(key, dtype=np.uint64)
        assert key.size == (self.k - 1) * 2 * self.k * self.n
        lib().mo_switch_key_inplace(self.h, ptr(ct), ptr(target), ptr(key), L)
        return ct

    def batch_switch_key(self, cts, targets, key, L, batch):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).copy()
        targets = np.ascontiguousarray(targets, dtype=np.uint64)
        key = np.ascontiguousarray(key, dtype=np.uint64)
        lib().mo_batch_switch_key(self.h, ptr(cts), ptr(targets), ptr(key), L, batch)
        return cts

    def relinearize(self, ct3, key, L):
        ct3 = np.ascontiguousarray(ct3, dtype=np.uint64).copy()
        key = np.ascontiguousarray(key, dtype=np.uint64)
        lib().mo_relinearize(self.h, ptr(ct3), ptr(key), L)
        return ct3.reshape(3, L, self.n)[:2].copy()

    def apply_galois(self, ct, L, elt, key):
        ct = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        key = np.ascontiguousarray(key, dtype=np.uint64)
        lib().mo_apply_galois_inplace(self.h, ptr(ct), L, int(elt), ptr(key))
        return ct

    def modraise(self, ct, Lout):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.empty((2, Lout, self.n), dtype=np.uint64)
        lib().mo_modraise(self.h, ptr(ct), Lout, ptr(out))
        return out


class CkksTablesStruct(C.Structure):
    _fields_ = [("logn", C.c_int), ("n", C.c_size_t), ("slots", C.c_size_t),
                ("index_map", C.POINTER(C.c_uint32)), ("root_powers", C.POINTER(C.c_double)),
                ("inv_root_powers", C.POINTER(C.c_double))]


class CkksEncoder:
    """CKKSEncoder restatement (oracle/moai_oracle.h mo_ckks_*; SEAL/ckks.h:457-637, ckks.cpp:13-216)."""

    ERRORS = {-1: "values_size is too large", -2: "scale out of bounds", -3: "encoded values are too large"}

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = lib().mo_ckks_tables_create(ctx.logn)
        st = C.cast(self.h, C.POINTER(CkksTablesStruct)).contents
        n = ctx.n
        self.index_map = np.ctypeslib.as_array(st.index_map, shape=(n,)).copy()
        self.root_powers = np.ctypeslib.as_array(st.root_powers, shape=(n, 2)).copy()
        self.inv_root_powers = np.ctypeslib.as_array(st.inv_root_powers, shape=(n, 2)).copy()

    def __del__(self):
        try:
            lib().mo_ckks_tables_free(self.h)
        except Exception:
            pass

    def total_bits(self, L, prime_index=None):
        idx = range(L) if prime_index is None else prime_index
        prod = 1
        for i in idx:
            prod *= self.ctx.primes[int(i)]
        return prod.bit_length()

    def encode(self, values, L, scale, prime_index=None, return_bits=False):
        """values: real or complex vector (<= n/2 entries).  Returns [L][N] NTT-form residues."""
        v = np.asarray(values)
        is_complex = np.iscomplexobj(v)
        v = np.ascontiguousarray(v, dtype=np.complex128 if is_complex else np.float64)
        out = np.zeros((L, self.ctx.n), dtype=np.uint64)
        pi = None
        if prime_index is not None:
            pi_arr = np.ascontiguousarray(prime_index, dtype=np.uint32)
            pi = ptr(pi_arr)
        bits = C.c_int(0)
        rc = lib().mo_ckks_encode(self.ctx.h, self.h, ptr(v), 1 if is_complex else 0, v.size, L, pi,
                                  float(scale), self.total_bits(L, prime_index), ptr(out), C.byref(bits))
        if rc:
            raise ValueError(self.ERRORS[rc])
        return (out, bits.value) if return_bits else out

    def encode_scalar(self, value, L, scale, prime_index=None):
        out = np.zeros(L, dtype=np.uint64)
        pi = None
        if prime_index is not None:
            pi_arr = np.ascontiguousarray(prime_index, dtype=np.uint32)
            pi = ptr(pi_arr)
        rc = lib().mo_ckks_encode_scalar(self.ctx.h, float(value), L, pi, float(scale),
                                         self.total_bits(L, prime_index), ptr(out))
        if rc:
            raise ValueError(self.ERRORS[rc])
        return out

    def fft_from_rev(self, z, scalar=None):
        a = np.ascontiguousarray(z, dtype=np.complex128).copy()
        s = C.byref(C.c_double(scalar)) if scalar is not None else None
        lib().mo_fft_transform_from_rev(ptr(a), self.ctx.logn, ptr(self.inv_root_powers), s)
        return a

    def fft_to_rev(self, z):
        a = np.ascontiguousarray(z, dtype=np.complex128).copy()
        lib().mo_fft_transform_to_rev(ptr(a), self.ctx.logn, ptr(self.root_powers))
        return a


def galois_elt_from_step(logn, step, generator=5):
    err = C.c_int(0)
    e = lib().mo_galois_elt_from_step(logn, step, generator, C.byref(err))
    if err.value:
        raise ValueError("step count too large")
    return int(e)


def galois_elts_all(logn, generator=5):
    out = (C.c_uint32 * (2 * logn))()
    cnt = lib().mo_galois_elts_all(logn, generator, out)
    return [int(out[i]) for i in range(cnt)]


def galois_table_ntt(logn, elt):
    t = np.empty(1 << logn, dtype=np.uint32)
    lib().mo_galois_table_ntt(logn, int(elt), ptr(t))
    return t


def naf(v):
    out = (C.c_int * 40)()
    cnt = lib().mo_naf(int(v), out, 40)
    return [int(out[i]) for i in range(cnt)]


def uniform_rns(rng, primes, shape_prefix, n):
    """uint64[*shape_prefix][len(primes)][n] with row i uniform in [0, q_i)."""
    out = np.empty(tuple(shape_prefix) + (len(primes), n), dtype=np.uint64)
    for i, q in enumerate(primes):
        out[..., i, :] = rng.integers(0, int(q), size=tuple(shape_prefix) + (n,), dtype=np.uint64)
    return out
