"""bench.py's host-side helpers (CPU): the product-side prime generation must reproduce
CoeffModulus::Create, which the oracle restates and the reference's fixtures pin."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle as O


def test_bench_primes_match_coeff_modulus_create(kats):
    import bench

    p = bench.primes_44x60()
    assert p == O.coeff_modulus_create(65536, [60] * 44)
    S = kats["survey_oracle_constants"]["n65536_44x60"]
    assert p[0] == int(S["first_prime"]) and p[-1] == int(S["last_prime"])
    assert bench.host_cores() >= 1


def test_bench_is_prime_agrees_with_oracle():
    import bench

    for v in (2, 3, 4, 221, 65537, 72307 * 59399, 36893488147419103, 36893488147419107, 1152921504606584833):
        assert bench.is_prime(v) == bool(O.lib().mo_is_prime(v))


def test_pmc_traffic_file_is_consistent():
    import json

    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert t["algorithmic_bytes"] == 22528 * 65536 * 16
    assert abs(t["ntt_forward_hbm_bytes_per_launch"] - (t["read_bytes"] + t["write_bytes"])) < 1
    assert 1.9 < t["traffic_over_algorithmic"] < 2.2  # two passes by construction
