"""The floating-point fixtures copied from the reference (tests/golden/moai_data, tests/golden/copy_moai_fixtures.py) are the
ones tests/cpp/test_moai_fixtures.cpp decrypts against under -m gpu.  CPU checks, no evaluator involved: the files are there,
have the shapes the reference's readers expect (include/test/test_full_scheme.hpp:41-337), and are mutually consistent the
way the C++ test assumes -- QKT = Q K^T / 8 per head, aftsoftmax = softmax(QKT), real_attention = aftsoftmax V,
the two LayerNorm outputs = LayerNorm(input) with the shipped gamma / beta, real_intermediate_output = GELU(input) --
so a wrong slice or transposition in the C++ packing code cannot hide behind a loose tolerance."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "moai_data")
LAYERS = [0, 11]


def _load(layer, rel):
    return np.loadtxt(os.path.join(DATA, "layer_%d" % layer, rel), delimiter=",")


@pytest.mark.parametrize("layer", LAYERS)
def test_attention_fixtures_are_consistent(layer):
    a = "Attention/BertSelfAttention/allresults/"
    q, k, v, qkt, sm, att = [_load(layer, a + f + ".csv") for f in ("Q", "K", "V", "QKT", "aftsoftmax", "real_attention")]
    assert q.shape == k.shape == v.shape == att.shape == (5, 768)
    assert qkt.shape == sm.shape == (5, 60)
    for h in range(12):
        cols = slice(64 * h, 64 * h + 64)
        s = q[:, cols] @ k[:, cols].T / 8.0  # sqrt_d = 8 (test_full_scheme.hpp:12), folded into W_Q by the reference's reader
        assert np.abs(s - qkt[:, 5 * h:5 * h + 5]).max() < 2e-6
        e = np.exp(s)
        p = e / e.sum(1, keepdims=True)
        assert np.abs(p - sm[:, 5 * h:5 * h + 5]).max() < 1e-6
        assert np.abs(p @ v[:, cols] - att[:, cols]).max() < 1e-6


@pytest.mark.parametrize("layer", LAYERS)
@pytest.mark.parametrize("where,prefix", [("Attention/SelfOutput/", "self_output"), ("Output/", "final_output")])
def test_layernorm_fixtures_are_consistent(layer, where, prefix):
    x = _load(layer, where + "allresults/%s_residual_connection_before_layernorm.csv" % prefix)
    y = _load(layer, where + "allresults/real_%s.csv" % prefix)
    g = _load(layer, where + "parms/%s_LayerNorm_weight.csv" % prefix)
    b = _load(layer, where + "parms/%s_LayerNorm_bias.csv" % prefix)
    assert x.shape == y.shape == (5, 768) and g.shape == b.shape == (768,)
    m = x.mean(1, keepdims=True)
    var = x.var(1, keepdims=True)
    assert np.abs((x - m) / np.sqrt(var + 1e-12) * g + b - y).max() < 1e-5


@pytest.mark.parametrize("layer", LAYERS)
def test_gelu_fixtures_are_consistent_and_inside_the_polynomial_range(layer):
    from scipy.special import erf

    x = _load(layer, "Intermediate/allresults/intermediate_output_after_linear.csv")
    y = _load(layer, "Intermediate/allresults/real_intermediate_output.csv")
    assert x.shape == y.shape == (5, 3072)
    assert np.abs(0.5 * x * (1 + erf(x / np.sqrt(2))) - y).max() < 2e-6
    # the range the copy script's docstring promises for these layers (gelu_v2's degree-24 polynomial diverges beyond it)
    assert x.min() > -16 and x.max() < 9
