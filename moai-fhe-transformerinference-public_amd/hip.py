"""ctypes binding of libmoai_hip.so -- the C ABI declared in include/moai_hip.h.

This is plumbing, not product logic: every call goes straight to the HIP library.  Residue data is
numpy uint64 on the host and raw device pointers on the GPU, in the reference's layout
[poly][rns prime][coefficient] (SEAL/ciphertext.h:337-349).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MOAI_HIP_LIB") or os.path.join(_HERE, "libmoai_hip.so")  # override: diagnostic builds only

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p
sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/moai_hip.h declares
SYMBOLS = {
    "moai_last_error": (C.c_char_p, []),
    "moai_version": (C.c_int, []),
    "moai_ctx_create": (C.c_int, [C.c_int, u64p, sz, C.c_int, C.POINTER(vp)]),
    "moai_ctx_destroy": (None, [vp]),
    "moai_ctx_reserve": (C.c_int, [vp, sz]),
    "moai_ctx_reserve_stream": (C.c_int, [vp, vp, sz]),
    "moai_ctx_coeff_count": (sz, [vp]),
    "moai_ctx_prime_count": (sz, [vp]),
    "moai_ctx_root": (C.c_uint64, [vp, sz]),
    "moai_malloc": (C.c_int, [C.POINTER(vp), sz]),
    "moai_free": (C.c_int, [vp]),
    "moai_memcpy_h2d": (C.c_int, [vp, vp, sz, vp]),
    "moai_memcpy_d2h": (C.c_int, [vp, vp, sz, vp]),
    "moai_memcpy_d2d": (C.c_int, [vp, vp, sz, vp]),
    "moai_memset_zero": (C.c_int, [vp, sz, vp]),
    "moai_stream_create": (C.c_int, [C.POINTER(vp)]),
    "moai_stream_destroy": (C.c_int, [vp]),
    "moai_stream_sync": (C.c_int, [vp]),
    "moai_ntt_forward": (C.c_int, [vp, vp, sz, sz, u32p, vp]),
    "moai_ntt_inverse": (C.c_int, [vp, vp, sz, sz, u32p, vp]),
    "moai_add": (C.c_int, [vp, vp, vp, vp, sz, sz, vp]),
    "moai_sub": (C.c_int, [vp, vp, vp, vp, sz, sz, vp]),
    "moai_negate": (C.c_int, [vp, vp, vp, sz, sz, vp]),
    "moai_dyadic_mul": (C.c_int, [vp, vp, vp, vp, sz, sz, sz, vp]),
    "moai_mul_scalar_rows": (C.c_int, [vp, vp, u64p, vp, sz, sz, vp]),
    "moai_add_scalar_rows": (C.c_int, [vp, vp, u64p, vp, sz, sz, vp]),
    "moai_ct_multiply": (C.c_int, [vp, vp, vp, vp, sz, sz, vp]),
    "moai_ct_square": (C.c_int, [vp, vp, vp, sz, sz, vp]),
    "moai_ct_multiply_general": (C.c_int, [vp, vp, sz, vp, sz, vp, sz, sz, vp]),
    "moai_ct_dot": (C.c_int, [vp, vp, vp, vp, sz, sz, vp]),
    "moai_ct_pt_dot": (C.c_int, [vp, vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), sz, sz, sz, vp]),
    "moai_ct_pt_dot_rows": (C.c_int, [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp]),
    "moai_ct_pt_dot2": (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), sz, sz, sz, sz, vp]),
    "moai_ct_pt_matmul": (C.c_int, [vp, vp, vp, vp, sz, sz, sz, sz, vp]),
    "moai_rescale": (C.c_int, [vp, vp, vp, sz, sz, sz, vp]),
    "moai_mul_scalar_rescale": (C.c_int, [vp, vp, u64p, vp, sz, sz, sz, vp]),
    "moai_rescale_add": (C.c_int, [vp, vp, vp, vp, sz, sz, sz, vp]),
    "moai_mul_scalar_rescale_add": (C.c_int, [vp, vp, u64p, vp, vp, sz, sz, sz, vp]),
    "moai_mod_drop": (C.c_int, [vp, vp, vp, sz, sz, sz, sz, vp]),
    "moai_galois_permute": (C.c_int, [vp, vp, vp, sz, sz, C.c_uint32, vp]),
    "moai_galois_elt_from_step": (C.c_uint32, [vp, C.c_int]),
    "moai_switch_key": (C.c_int, [vp, vp, vp, vp, sz, sz, vp]),
    "moai_relinearize": (C.c_int, [vp, vp, vp, vp, sz, sz, vp]),
    "moai_apply_galois": (C.c_int, [vp, vp, sz, C.c_uint32, vp, sz, vp]),
    "moai_apply_galois_to": (C.c_int, [vp, vp, vp, sz, C.c_uint32, vp, sz, vp]),
    "moai_apply_galois_acc": (C.c_int, [vp, vp, vp, sz, C.c_uint32, vp, sz, vp]),
    "moai_modraise": (C.c_int, [vp, vp, vp, sz, sz, vp]),
    "moai_hoist_correction": (C.c_int, [vp, vp, C.c_uint32, sz, vp, vp]),
    "moai_apply_galois_hoisted": (C.c_int, [vp, vp, C.POINTER(vp), sz, C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(vp), sz, sz, C.POINTER(C.c_int), vp]),
    "moai_ckks_encode": (C.c_int, [vp, vp, C.c_int, sz, sz, vp, sz, C.POINTER(C.c_uint32), C.c_double, vp, vp]),
    "moai_ckks_encode_masked": (C.c_int, [vp, vp, vp, sz, sz, vp, sz, C.POINTER(C.c_uint32), C.c_double, vp, vp]),
    "moai_total_coeff_modulus_bit_count": (C.c_int, [vp, sz, C.POINTER(C.c_uint32)]),
    "moai_ckks_tables": (C.c_int, [vp, vp, vp]),
    "moai_set_tuning": (C.c_int, [C.c_char_p, C.c_long]),
    "moai_mem_info": (C.c_int, [C.POINTER(sz), C.POINTER(sz)]),
    "moai_op_trace": (C.c_int, [C.c_int]),
    "moai_op_trace_dump": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "moai_scalar_dot": (C.c_int, [vp, C.POINTER(vp), u64p, sz, vp, vp, sz, sz, vp]),
    "moai_vector_dot": (C.c_int, [vp, C.POINTER(vp), vp, sz, vp, vp, sz, sz, vp]),
    "moai_ct_dot_ptrs": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), sz, vp, vp, sz, vp]),
    "moai_gather_blocks": (C.c_int, [vp, C.POINTER(vp), vp, sz, sz, vp]),
    "moai_scatter_blocks": (C.c_int, [vp, vp, C.POINTER(vp), sz, sz, vp]),
    "moai_key_words": (sz, [vp, sz]),
    "moai_key_trim": (C.c_int, [vp, vp, sz, vp, vp]),
    "moai_key_forget": (C.c_int, [vp, vp]),
    "moai_debug_stream_audit": (C.c_int, [C.c_int]),
    "moai_debug_block_label": (None, [vp, sz, vp, C.c_int]),
    "moai_debug_stream_audit_counts": (None, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "moai_device_info": (C.c_int, [C.c_int, C.c_char_p, sz, C.POINTER(C.c_int), C.POINTER(sz)]),
    "moai_event_create": (C.c_int, [C.POINTER(vp)]),
    "moai_event_destroy": (C.c_int, [vp]),
    "moai_event_record": (C.c_int, [vp, vp]),
    "moai_event_synchronize": (C.c_int, [vp]),
    "moai_host_malloc": (C.c_int, [C.POINTER(vp), sz]),
    "moai_host_free": (C.c_int, [vp]),
    "moai_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
}


MOAI_EINVAL = -1  # include/moai_hip.h


class MoaiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("moai_hip error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    return _SO


_lib = None


def lib():
    """Load libmoai_hip.so.  Raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')" % _SO
            )
        L = C.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MoaiError(rc, lib().moai_last_error().decode())


class DeviceBuffer:
    """A block of device memory holding uint64 residues (moai_malloc / moai_free)."""

    def __init__(self, n_words):
        self.n_words = int(n_words)
        p = vp()
        _check(lib().moai_malloc(C.byref(p), self.n_words * 8))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = cls(a.size)
        b.upload(a, stream)
        return b

    def upload(self, a, stream=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        assert a.size <= self.n_words
        _check(lib().moai_memcpy_h2d(self.ptr, a.ctypes.data, a.size * 8, stream))
        _check(lib().moai_stream_sync(stream))

    def to_numpy(self, shape=None, stream=None, words=None):
        words = self.n_words if words is None else words
        out = np.empty(words, dtype=np.uint64)
        _check(lib().moai_stream_sync(stream))
        _check(lib().moai_memcpy_d2h(out.ctypes.data, self.ptr, words * 8, stream))
        _check(lib().moai_stream_sync(stream))
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            lib().moai_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return int(x)  # raw device pointer (e.g. torch.Tensor.data_ptr())


class Context:
    """moai_ctx: per-prime NTT tables and constants on the device, shared by all levels."""

    def __init__(self, coeff_count_power, primes, device=0):
        self.logn = int(coeff_count_power)
        self.n = 1 << self.logn
        self.primes = [int(p) for p in primes]
        self.k = len(self.primes)
        arr = (C.c_uint64 * self.k)(*self.primes)
        h = vp()
        _check(lib().moai_ctx_create(self.logn, arr, self.k, device, C.byref(h)))
        self.h = h.value

    def close(self):
        if getattr(self, "h", None):
            lib().moai_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, nbytes, stream=None):
        if stream is None:
            _check(lib().moai_ctx_reserve(self.h, int(nbytes)))
        else:
            _check(lib().moai_ctx_reserve_stream(self.h, stream, int(nbytes)))

    def root(self, i):
        return int(lib().moai_ctx_root(self.h, i))

    @staticmethod
    def _pidx(prime_index):
        if prime_index is None:
            return None
        return (C.c_uint32 * len(prime_index))(*[int(x) for x in prime_index])

    # --- raw-pointer API (device buffers) -----------------------------------------------------------
    def ntt_forward(self, data, n_poly, L, prime_index=None, stream=None):
        _check(lib().moai_ntt_forward(self.h, _ptr(data), n_poly, L, self._pidx(prime_index), stream))

    def ntt_inverse(self, data, n_poly, L, prime_index=None, stream=None):
        _check(lib().moai_ntt_inverse(self.h, _ptr(data), n_poly, L, self._pidx(prime_index), stream))

    def add(self, a, b, out, n_poly, L, stream=None):
        _check(lib().moai_add(self.h, _ptr(a), _ptr(b), _ptr(out), n_poly, L, stream))

    def sub(self, a, b, out, n_poly, L, stream=None):
        _check(lib().moai_sub(self.h, _ptr(a), _ptr(b), _ptr(out), n_poly, L, stream))

    def negate(self, a, out, n_poly, L, stream=None):
        _check(lib().moai_negate(self.h, _ptr(a), _ptr(out), n_poly, L, stream))

    def dyadic_mul(self, a, b, out, n_poly, n_poly_b, L, stream=None):
        _check(lib().moai_dyadic_mul(self.h, _ptr(a), _ptr(b), _ptr(out), n_poly, n_poly_b, L, stream))

    def mul_scalar_rows(self, a, scalars, out, n_poly, L, stream=None):
        s = (C.c_uint64 * L)(*[int(x) for x in scalars])
        _check(lib().moai_mul_scalar_rows(self.h, _ptr(a), s, _ptr(out), n_poly, L, stream))

    def add_scalar_rows(self, a, scalars, out, n_poly, L, stream=None):
        s = (C.c_uint64 * L)(*[int(x) for x in scalars])
        _check(lib().moai_add_scalar_rows(self.h, _ptr(a), s, _ptr(out), n_poly, L, stream))

    def ct_multiply(self, x, y, out, L, batch, stream=None):
        _check(lib().moai_ct_multiply(self.h, _ptr(x), _ptr(y), _ptr(out), L, batch, stream))

    def ct_multiply_general(self, x, size_x, y, size_y, out, L, batch, stream=None):
        _check(lib().moai_ct_multiply_general(self.h, _ptr(x), size_x, _ptr(y), size_y, _ptr(out), L, batch, stream))

    def ct_square(self, x, out, L, batch, stream=None):
        _check(lib().moai_ct_square(self.h, _ptr(x), _ptr(out), L, batch, stream))

    def ct_dot(self, x, y, out, count, L, stream=None):
        _check(lib().moai_ct_dot(self.h, _ptr(x), _ptr(y), _ptr(out), count, L, stream))

    def ct_pt_dot(self, x, p, out, x_index, p_index, n_poly, L, stream=None):
        xi = (C.c_uint32 * len(x_index))(*[int(v) for v in x_index])
        pi = (C.c_uint32 * len(p_index))(*[int(v) for v in p_index])
        _check(lib().moai_ct_pt_dot(self.h, _ptr(x), _ptr(p), _ptr(out), xi, pi, len(x_index), n_poly, L, stream))

    def ct_pt_dot2(self, x, p, out, out2, x_index, p_index, p_index2, n_poly, L, stream=None):
        xi = (C.c_uint32 * len(x_index))(*[int(v) for v in x_index])
        pi = (C.c_uint32 * len(p_index))(*[int(v) for v in p_index])
        pi2 = (C.c_uint32 * max(1, len(p_index2)))(*[int(v) for v in p_index2])
        _check(lib().moai_ct_pt_dot2(self.h, _ptr(x), _ptr(p), _ptr(out), _ptr(out2), xi, pi, pi2, len(x_index), len(p_index2), n_poly, L,
                                     stream))

    def ct_pt_dot_rows(self, x, p, p2, out, out2, rows, n_poly, L, stream=None):
        _check(lib().moai_ct_pt_dot_rows(self.h, _ptr(x), _ptr(p), _ptr(p2) if p2 is not None else None, _ptr(out),
                                         _ptr(out2) if out2 is not None else None, rows, n_poly, L, stream))

    def ct_pt_matmul(self, x, w, out, rows, cols, size, L, stream=None):
        _check(lib().moai_ct_pt_matmul(self.h, _ptr(x), _ptr(w), _ptr(out), rows, cols, size, L, stream))

    def rescale(self, src, out, size, L, batch, stream=None):
        _check(lib().moai_rescale(self.h, _ptr(src), _ptr(out), size, L, batch, stream))

    def mul_scalar_rescale(self, src, scalars, out, size, L, batch, stream=None):
        s = (C.c_uint64 * L)(*[int(x) for x in scalars])
        _check(lib().moai_mul_scalar_rescale(self.h, _ptr(src), s, _ptr(out), size, L, batch, stream))

    def rescale_add(self, src, addend, out, size, L, batch, stream=None):
        _check(lib().moai_rescale_add(self.h, _ptr(src), _ptr(addend), _ptr(out), size, L, batch, stream))

    def mul_scalar_rescale_add(self, src, scalars, addend, out, size, L, batch, stream=None):
        s = (C.c_uint64 * L)(*[int(x) for x in scalars])
        _check(lib().moai_mul_scalar_rescale_add(self.h, _ptr(src), s, _ptr(addend), _ptr(out), size, L, batch, stream))

    def mod_drop(self, src, out, size, L, drop, batch, stream=None):
        _check(lib().moai_mod_drop(self.h, _ptr(src), _ptr(out), size, L, drop, batch, stream))

    def galois_permute(self, src, out, n_poly, L, elt, stream=None):
        _check(lib().moai_galois_permute(self.h, _ptr(src), _ptr(out), n_poly, L, int(elt), stream))

    def galois_elt_from_step(self, step):
        e = lib().moai_galois_elt_from_step(self.h, int(step))
        if e == 0:
            raise MoaiError(-1, lib().moai_last_error().decode())
        return int(e)

    def switch_key(self, ct, target, key, L, batch, stream=None):
        _check(lib().moai_switch_key(self.h, _ptr(ct), _ptr(target), _ptr(key), L, batch, stream))

    def relinearize(self, ct3, key, out, L, batch, stream=None):
        _check(lib().moai_relinearize(self.h, _ptr(ct3), _ptr(key), _ptr(out), L, batch, stream))

    def apply_galois_acc(self, src, acc, L, elt, key, batch, stream=None):
        _check(lib().moai_apply_galois_acc(self.h, _ptr(src), _ptr(acc), L, int(elt), _ptr(key), batch, stream))

    def apply_galois(self, ct, L, elt, key, batch, stream=None):
        _check(lib().moai_apply_galois(self.h, _ptr(ct), L, int(elt), _ptr(key), batch, stream))

    def apply_galois_to(self, src, dst, L, elt, key, batch, stream=None):
        _check(lib().moai_apply_galois_to(self.h, _ptr(src), _ptr(dst), L, int(elt), _ptr(key), batch, stream))

    def scalar_dot(self, xs, scalars, base, out, size, L, stream=None):
        """out = base + sum_t xs[t] (*) scalars[t] (scalars: numpy uint64 [terms][L], reduced)"""
        T = len(xs)
        arr = (vp * T)(*[_ptr(x) for x in xs])
        sc = np.ascontiguousarray(scalars, dtype=np.uint64)
        _check(lib().moai_scalar_dot(self.h, arr, sc.ctypes.data_as(u64p), T, _ptr(base), _ptr(out), size, L, stream))

    def vector_dot(self, xs, plains, base, out, size, L, stream=None):
        """out = base + sum_t xs[t] (*) plains[t] (plains: one device buffer [terms][L][N])"""
        T = len(xs)
        arr = (vp * T)(*[_ptr(x) for x in xs])
        _check(lib().moai_vector_dot(self.h, arr, _ptr(plains), T, _ptr(base), _ptr(out), size, L, stream))

    def ct_dot_ptrs(self, xs, ys, base, out, L, stream=None):
        """out[3][L][N] = base + sum_t multiply(xs[t], ys[t]) for size-2 ciphertexts in separate buffers"""
        T = len(xs)
        ax = (vp * T)(*[_ptr(x) for x in xs])
        ay = (vp * T)(*[_ptr(y) for y in ys])
        _check(lib().moai_ct_dot_ptrs(self.h, ax, ay, T, _ptr(base), _ptr(out), L, stream))

    def gather_blocks(self, blocks, packed, words, stream=None):
        """packed[i] = blocks[i] (`words` 64-bit words each) in one launch"""
        n = len(blocks)
        arr = (vp * n)(*[_ptr(b) for b in blocks])
        _check(lib().moai_gather_blocks(self.h, arr, _ptr(packed), n, words, stream))

    def scatter_blocks(self, packed, blocks, words, stream=None):
        """blocks[i] = packed[i] in one launch"""
        n = len(blocks)
        arr = (vp * n)(*[_ptr(b) for b in blocks])
        _check(lib().moai_scatter_blocks(self.h, _ptr(packed), arr, n, words, stream))

    def key_trim(self, full_key, levels, stream=None):
        """the part of a key a switch at <= `levels` data primes reads, as a DeviceBuffer [levels][2][levels+1][N] whose layout the
        context knows (moai_key_trim); pass it wherever a key is expected"""
        out = DeviceBuffer(lib().moai_key_words(self.h, levels))
        _check(lib().moai_key_trim(self.h, _ptr(full_key), levels, out.ptr, stream))
        _check(lib().moai_stream_sync(stream))
        return out

    def key_forget(self, key):
        _check(lib().moai_key_forget(self.h, _ptr(key)))

    def hoist_correction(self, key, elt, L, stream=None):
        """the per-(key, level) constant of the hoisted rotations: DeviceBuffer [2][L+1][N]"""
        out = DeviceBuffer(2 * (L + 1) * self.n)
        _check(lib().moai_hoist_correction(self.h, _ptr(key), int(elt), L, _ptr(out), stream))
        return out

    def apply_galois_hoisted(self, src, dst, L, elts, keys, corrections, batch, stream=None):
        """dst[r] = apply_galois(src, elts[r], keys[r]) for every r with one digit decomposition (dst: one buffer
        [R][batch][2][L][N] or a list of R buffers); returns True when the library had to fall back to separate calls (a
        zero coefficient in INTT(c1))"""
        R = len(elts)
        if isinstance(dst, (list, tuple)):
            op = (vp * R)(*[_ptr(d) for d in dst])
        else:
            words = batch * 2 * L * self.n * 8
            op = (vp * R)(*[_ptr(dst) + r * words for r in range(R)])
        e = (C.c_uint32 * R)(*[int(x) for x in elts])
        kp = (vp * R)(*[_ptr(k) for k in keys])
        cp = (vp * R)(*[_ptr(k) for k in corrections])
        fb = C.c_int(0)
        _check(lib().moai_apply_galois_hoisted(self.h, _ptr(src), op, L, e, kp, cp, R, batch, C.byref(fb), stream))
        return bool(fb.value)

    def modraise(self, src, out, L_out, batch, stream=None):
        _check(lib().moai_modraise(self.h, _ptr(src), _ptr(out), L_out, batch, stream))

    def ckks_encode(self, values, L, scale, prime_index=None, stream=None):
        """CKKSEncoder::encode for a batch of vectors: values [n_batch][count] float64 or complex128 (host).
        Returns (DeviceBuffer [n_batch][L][N] in NTT form, max |coefficient| per vector); raises like
        the reference when a vector does not fit the level's modulus (SEAL/ckks.h:527-538)."""
        v = np.asarray(values)
        is_complex = np.iscomplexobj(v)
        v = np.ascontiguousarray(v, dtype=np.complex128 if is_complex else np.float64)
        if v.ndim == 1:
            v = v[None, :]
        n_batch, count = v.shape
        words = v.size * (2 if is_complex else 1)
        dv = DeviceBuffer(max(words, 1) + n_batch)
        _check(lib().moai_memcpy_h2d(dv.ptr, v.ctypes.data, words * 8, stream))
        out = DeviceBuffer(n_batch * L * self.n)
        mx_ptr = dv.ptr + max(words, 1) * 8
        _check(lib().moai_ckks_encode(self.h, dv.ptr, 1 if is_complex else 0, count, n_batch, out.ptr, L,
                                      self._pidx(prime_index), float(scale), mx_ptr, stream))
        mx = np.empty(n_batch, dtype=np.float64)
        _check(lib().moai_stream_sync(stream))
        _check(lib().moai_memcpy_d2h(mx.ctypes.data, mx_ptr, n_batch * 8, stream))
        _check(lib().moai_stream_sync(stream))
        total = self.total_coeff_modulus_bit_count(L, prime_index)
        bits = np.ceil(np.log2(np.maximum(mx, 1.0))).astype(int) + 1
        if np.any(bits >= total) or not np.all(np.isfinite(mx)):
            raise MoaiError(MOAI_EINVAL, "encoded values are too large")
        return out, mx

    def ckks_encode_masked(self, constants, mask, L, scale, prime_index=None, stream=None):
        """moai_ckks_encode_masked: vector b = constants[b] on the slots where mask == 1, 0 elsewhere."""
        cst = np.ascontiguousarray(constants, dtype=np.float64)
        msk = np.ascontiguousarray(mask, dtype=np.int32)
        n_batch = cst.size
        dc = DeviceBuffer(2 * n_batch + 1)
        dm = DeviceBuffer((msk.size + 1) // 2 + 1)
        _check(lib().moai_memcpy_h2d(dc.ptr, cst.ctypes.data, n_batch * 8, stream))
        _check(lib().moai_memcpy_h2d(dm.ptr, msk.ctypes.data, msk.size * 4, stream))
        out = DeviceBuffer(n_batch * L * self.n)
        mx_ptr = dc.ptr + n_batch * 8
        _check(lib().moai_ckks_encode_masked(self.h, dc.ptr, dm.ptr, msk.size, n_batch, out.ptr, L,
                                             self._pidx(prime_index), float(scale), mx_ptr, stream))
        mx = np.empty(n_batch, dtype=np.float64)
        _check(lib().moai_stream_sync(stream))
        _check(lib().moai_memcpy_d2h(mx.ctypes.data, mx_ptr, n_batch * 8, stream))
        _check(lib().moai_stream_sync(stream))
        return out, mx

    def total_coeff_modulus_bit_count(self, L, prime_index=None):
        r = lib().moai_total_coeff_modulus_bit_count(self.h, L, self._pidx(prime_index))
        if r == 0:
            _check(MOAI_EINVAL)
        return r

    def ckks_tables(self):
        idx = np.empty(self.n, dtype=np.uint32)
        roots = np.empty((self.n, 2), dtype=np.float64)
        _check(lib().moai_ckks_tables(self.h, idx.ctypes.data, roots.ctypes.data))
        return idx, roots

    def sync(self, stream=None):
        _check(lib().moai_stream_sync(stream))


def set_tuning(name, value):
    _check(lib().moai_set_tuning(name.encode(), int(value)))


def op_trace(enable):
    """start (True: clears the counters) or stop the census of operations the entry points are asked to perform"""
    _check(lib().moai_op_trace(1 if enable else 0))


def op_trace_counts():
    """{(entry point, level): units} counted since op_trace(True)"""
    need = lib().moai_op_trace_dump(None, 0)
    buf = C.create_string_buffer(need + 16)
    lib().moai_op_trace_dump(buf, need + 16)
    out = {}
    for line in buf.value.decode().splitlines():
        name, level, count = line.split()
        out[(name, int(level))] = int(count)
    return out


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    hbm = sz(0)
    _check(lib().moai_device_info(device, name, 256, C.byref(cus), C.byref(hbm)))
    return name.value.decode(), cus.value, hbm.value


class Event:
    def __init__(self):
        p = vp()
        _check(lib().moai_event_create(C.byref(p)))
        self.ptr = p.value

    def record(self, stream=None):
        _check(lib().moai_event_record(self.ptr, stream))

    def elapsed_ms_since(self, start):
        ms = C.c_float(0)
        _check(lib().moai_event_elapsed_ms(start.ptr, self.ptr, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().moai_event_destroy(self.ptr)
        except Exception:
            pass
