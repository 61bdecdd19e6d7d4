#!/usr/bin/env python3
"""HBM traffic of one key switch from a tools/pmc.sh summary of `tools/ks_time.py --only <l> <batch>` (separate FETCH_SIZE / WRITE_SIZE
passes; FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes): bytes per ciphertext = sum over kernels of
(mean bytes per dispatch x dispatches) / (calls x batch).  Writes the JSON bench.py's `keyswitch.roofline.traffic` reads.

    python3 tools/ks_traffic.py gpurun_out/pmc_<tag>/<tag>_pmc_summary.json <l> <batch> <calls> <commit> > profiles/r03_ks_traffic.json
"""
import json
import sys

d = json.load(open(sys.argv[1]))
L, batch, calls, commit = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
N = 65536
rd = wr = 0.0
per_kernel = {}
for k, v in d.items():
    n = v.get("_dispatches", 0)
    r, w = v.get("FETCH_SIZE", 0) * 1024 * 2 * n, v.get("WRITE_SIZE", 0) * 1024 * n
    rd += r
    wr += w
    per_kernel[k] = round((r + w) / (calls * batch) / 1e6, 2)
per_ct = (rd + wr) / (calls * batch)
alg = 5 * L * N * 8 + 2 * L * (L + 1) * N * 8 / batch
out = {
    "_about": "HBM bytes of moai_apply_galois at MOAI's parameters from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
              "tools/pmc.sh over tools/ks_time.py --only %d %d (%d profiled calls); FETCH_SIZE doubled (gfx950)" % (L, batch, calls),
    "collected_at_commit": commit,
    "l%d" % L: {"hbm_bytes_per_ciphertext": per_ct, "read_bytes_per_ciphertext": rd / (calls * batch), "write_bytes_per_ciphertext": wr / (calls * batch),
                "algorithmic_bytes_per_ciphertext_at_this_batch": alg, "traffic_over_algorithmic": per_ct / alg,
                "megabytes_per_ciphertext_by_kernel": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8])},
}
print(json.dumps(out, indent=1))
