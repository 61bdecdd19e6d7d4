#!/bin/bash
set -o pipefail
for cfg in "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=1" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=6" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=12" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=2 MOAI_NTT_COOP_DELAY=2" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=2 MOAI_NTT_COOP_DELAY=3" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=2 MOAI_NTT_COOP_DELAY=8" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=64"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['fwd_ms'], d['roofline']['inv_ms'])"
done
export MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=2 MOAI_NTT_COOP_DELAY=3
bash tools/pmc.sh coop --batch 64 --steps 1 --warmup 1
