#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database on the GPU box (the database itself is too large to copy back): kernel time per
wall-clock bin, top kernels per bin.  usage: prof_bins.py results.db [bin_seconds]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
binw = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
cur = db.cursor()
t0 = cur.execute("select min(start) from kernels").fetchone()[0]
bins = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name, start, end in cur.execute("select name, start, end from kernels"):
    b = int((start - t0) / 1e9 / binw)
    e = bins[b][name.replace("void ", "").replace("moai::", "")[:44]]
    e[0] += 1
    e[1] += (end - start) / 1e6
for b in sorted(bins):
    tot = sum(v[1] for v in bins[b].values())
    top = sorted(bins[b].items(), key=lambda kv: -kv[1][1])[:6]
    print("[%5.0f s] busy %6.0f ms | " % (b * binw, tot) + "; ".join("%s x%d %.0f ms" % (k, v[0], v[1]) for k, v in top))
