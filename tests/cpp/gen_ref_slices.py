#!/usr/bin/env python3
"""Build-time comparator source: cuts the definitions of a few functions out of the reference checkout
(/root/reference/include/source/bootstrapping/Bootstrapper.cpp and common/func.cpp) into
tests/cpp/generated/ref_bootstrapper_slices.inc, which is git-ignored and never committed.  The test binaries that
include it (through tests/cpp/ref_bootstrapper.h) are built in the container that holds the reference and travel to the
GPU box prebuilt, like test_moai_headers.  Nothing of the reference's text lives in this repository.

The sliced routines are the evaluation half of bootstrap_3 that needs no NTL: the two baby-step / giant-step transforms,
sflinv_full_3 / sfl_full_3, coefftoslot_full_3 / slottocoeff_full_3 and bootstrap_full_3.  They are compiled against the
seal:: shim inside namespace refslice and run one ciphertext at a time: the reference's own call sequence, to which
the packed device pipeline is compared bit for bit.
"""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "generated", "ref_bootstrapper_slices.inc")

BOOT = os.path.join(REF, "include/source/bootstrapping/Bootstrapper.cpp")
FUNC = os.path.join(REF, "include/source/bootstrapping/common/func.cpp")

METHODS = ["bsgs_linear_transform", "rotated_bsgs_linear_transform", "sflinv_full_3", "sfl_full_3", "coefftoslot_full_3",
           "slottocoeff_full_3", "bootstrap_full_3"]
FREE = ["giantstep", "rotation"]


def cut(text, head_regex):
    m = re.search(head_regex, text, re.M)
    if not m:
        raise SystemExit("not found: " + head_regex)
    i = text.index("{", m.end() - 1)
    depth = 0
    j = i
    while True:
        c = text[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return text[m.start():j + 1]


def main():
    if not os.path.exists(BOOT):
        raise SystemExit("reference checkout absent")
    boot = open(BOOT).read()
    func = open(FUNC).read()
    parts = ["// GENERATED at build time from the reference checkout by tests/cpp/gen_ref_slices.py -- not committed\n"]
    for name in FREE:
        parts.append(cut(func, r"^(?:int|void) %s\(" % name))
    for name in METHODS:
        parts.append(cut(boot, r"^void Bootstrapper::%s\(" % name))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write("\n\n".join(parts) + "\n")


if __name__ == "__main__":
    main()
