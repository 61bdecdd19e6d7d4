#!/usr/bin/env python3
"""Copies the floating-point activation fixtures the reference holds for its own pipeline into tests/golden/moai_data/.

The reference ships, per encoder layer, the clear-text activations of ONE 5-token input (tokens 101,3374,1010,4918,102)
at every stage of the layer, together with the LayerNorm parameters (data/layer_K/**/allresults/*.csv, **/parms/*.csv;
read by include/test/test_full_scheme.hpp:41-337 and diffed by hand against the decrypted output, :1047-1065).  They
are DATA (comma-separated numbers), not source.  The stages below need no dense weight matrix (those are the files
.MISSING_LARGE_BLOBS lists): Q K^T -> softmax -> . V from Q/K/V, both LayerNorms from their residual inputs with the
real gamma / beta, GELU from the intermediate product's output.

    python3 tests/golden/copy_moai_fixtures.py [/root/reference] [layers...]

Default layers: 0 and 11 (in both, every GELU input lies inside the range MOAI's degree-24 polynomial covers; layers
2-6, 9, 10 hold outlier activations up to 122 for which gelu_v2's polynomial returns 1e15..1e27, a property of the
reference's approximation that tests/cpp/test_moai_fixtures.cpp therefore cannot pin to a value)."""
import os
import shutil
import sys

FILES = [
    "Attention/BertSelfAttention/allresults/Q.csv",
    "Attention/BertSelfAttention/allresults/K.csv",
    "Attention/BertSelfAttention/allresults/V.csv",
    "Attention/BertSelfAttention/allresults/QKT.csv",
    "Attention/BertSelfAttention/allresults/aftsoftmax.csv",
    "Attention/BertSelfAttention/allresults/real_attention.csv",
    "Attention/SelfOutput/allresults/self_output_residual_connection_before_layernorm.csv",
    "Attention/SelfOutput/allresults/real_self_output.csv",
    "Attention/SelfOutput/parms/self_output_LayerNorm_weight.csv",
    "Attention/SelfOutput/parms/self_output_LayerNorm_bias.csv",
    "Intermediate/allresults/intermediate_output_after_linear.csv",
    "Intermediate/allresults/real_intermediate_output.csv",
    "Output/allresults/final_output_residual_connection_before_layernorm.csv",
    "Output/allresults/real_final_output.csv",
    "Output/parms/final_output_LayerNorm_weight.csv",
    "Output/parms/final_output_LayerNorm_bias.csv",
]


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    layers = [int(a) for a in sys.argv[2:]] or [0, 11]
    here = os.path.dirname(os.path.abspath(__file__))
    total = 0
    for layer in layers:
        for rel in FILES:
            src = os.path.join(ref, "data", "layer_%d" % layer, rel)
            dst = os.path.join(here, "moai_data", "layer_%d" % layer, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(src, dst)
            total += os.path.getsize(dst)
    print("copied %d files, %.1f KiB" % (len(layers) * len(FILES), total / 1024.0))


if __name__ == "__main__":
    main()
