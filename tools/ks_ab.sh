#!/bin/bash
# A/B of the key-switch kernel variants on one box, same hour: ms per ciphertext at l = 35 / 15, batch 64 (tools/ks_time.py).
# usage: tools/ks_ab.sh "<ENV=VAL ...>" ...   one configuration per argument ("" = defaults)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  for L in 35 15; do
    env $cfg timeout -k 10 120 python3 tools/ks_time.py --only $L 64 2>&1 | grep "per ciphertext"
  done
done
