"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/moai_hip.h declares, and validates arguments before touching the GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(moai):
    hdr = open(os.path.join(ROOT, "include", "moai_hip.h")).read()
    declared = set(re.findall(r"\b(moai_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    L = C.CDLL(moai.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), name
    # the python binding covers the whole header, no more, no less
    assert declared == set(moai.hip.SYMBOLS)


def test_header_cites_reference_for_every_entry_point():
    hdr = open(os.path.join(ROOT, "include", "moai_hip.h")).read()
    assert hdr.count("SEAL/") > 25
    assert "Bootstrapper.cpp:2938-2992" in hdr


def test_ctx_create_validates_before_gpu(moai):
    L = moai.hip.lib()
    h = C.c_void_p()
    primes = (C.c_uint64 * 2)(1152921504606748673, 1099511480321)
    assert L.moai_ctx_create(0, primes, 2, 0, C.byref(h)) == -1
    assert L.moai_ctx_create(17, primes, 2, 0, C.byref(h)) == -1
    assert b"coeff_count_power" in L.moai_last_error()
    assert L.moai_ctx_create(13, primes, 0, 0, C.byref(h)) == -1
    bad = (C.c_uint64 * 1)(1152921504606748671)  # not prime
    assert L.moai_ctx_create(13, bad, 1, 0, C.byref(h)) == -1
    assert b"invalid modulus" in L.moai_last_error()
    notntt = (C.c_uint64 * 1)(1000003)  # prime but != 1 mod 2N
    assert L.moai_ctx_create(13, notntt, 1, 0, C.byref(h)) == -1
    dup = (C.c_uint64 * 2)(1099511480321, 1099511480321)
    assert L.moai_ctx_create(13, dup, 2, 0, C.byref(h)) == -1
    assert h.value is None


def test_no_product_dependency_on_oracle():
    """The product path may never import, link or call the oracle (and has no CPU fallback)."""
    pkg = os.path.join(ROOT, "moai-fhe-transformerinference-public_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "moai_oracle" not in txt, f
                assert "import oracle" not in txt, f


def test_missing_extension_fails_loudly(moai, monkeypatch):
    monkeypatch.setattr(moai.hip, "_lib", None)
    monkeypatch.setattr(moai.hip, "_SO", "/nonexistent/libmoai_hip.so")
    with pytest.raises(ImportError):
        moai.hip.lib()
