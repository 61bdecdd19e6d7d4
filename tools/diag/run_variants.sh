#!/bin/bash
# per-kernel times of the key switch (l = 35, batch 64) under the default library and the diagnostic builds in tools/diag/
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  if [ $v = default ]; then unset MOAI_HIP_LIB; else export MOAI_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/libmoai_hip_$v.so; fi
  rm -rf /tmp/dv_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dv_$v -- python3 $GRAFT_REPO_ROOT/tools/ks_time.py --only 35 64 > /tmp/dv_$v.log 2>&1 || echo "$v failed"
  echo "== $v: $(grep 'per ciphertext' /tmp/dv_$v.log)"
  python3 - /tmp/dv_$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").split("(")[0]
        if "ks_" in n or "moddown" in n or "ntt_inv" in n:
            print("   %-44s calls %4s avg %9.1f us  total %8.2f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
