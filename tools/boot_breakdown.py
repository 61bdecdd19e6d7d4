#!/usr/bin/env python3
"""Kernels of ONE packed bootstrap_3 out of a rocprofv3 kernel trace of tools/cpp/bench_e2e (run on the GPU box; the trace stays there).

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- tools/cpp/bench_e2e 48 16 --no-head --no-ffn
    python3 tools/boot_breakdown.py <dir>

bench_e2e bootstraps a warm-up pack, then the timed pack, then the same ciphertexts through single calls: the dispatches between the
second modraise kernel and the third are the timed packed run."""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "modraise_kernel" in r[2]]
if len(marks) < 3:
    sys.exit("expected at least three modraise dispatches, found %d" % len(marks))
win = rows[marks[1]:marks[2]]
t0, t1 = win[0][0], max(r[1] for r in win)
acc = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for s, e, k in win:
    k = k.replace("void ", "").split("(")[0]
    a = acc[k]
    us = (e - s) / 1e3
    a[0] += 1
    a[1] += us
    a[2] = min(a[2], us)
    a[3] = max(a[3], us)
total = sum(a[1] for a in acc.values())
print("# window %.1f ms, kernel time %.1f ms (device busy %.0f %% of the wall time), %d dispatches" % ((t1 - t0) / 1e6, total / 1e3, 100 * total / ((t1 - t0) / 1e3), len(win)))
group = lambda pred: 100 * sum(a[1] for k, a in acc.items() if pred(k)) / total
print("# key switch (ks_*): %.1f %%; mod-down / rescale (moddown_*): %.1f %%; plaintext dot products: %.1f %%; additions (ew_kernel<0>): %.1f %%" % (
    group(lambda k: "::ks_" in k), group(lambda k: "moddown" in k), group(lambda k: "ct_pt_dot" in k), group(lambda k: "ew_kernel<0>" in k)))
print("%-64s %6s %10s %6s %9s %9s %9s" % ("kernel", "calls", "total ms", "%", "avg us", "min us", "max us"))
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-64s %6d %10.1f %6.1f %9.1f %9.1f %9.1f" % (k[:64], a[0], a[1] / 1e3, 100 * a[1] / total, a[1] / a[0], a[2], a[3]))
