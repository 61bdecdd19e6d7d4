#!/bin/bash
set -o pipefail
MOAI_NTT_COOP=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "ntt or config2 or moai_chain" 2>&1 | tail -2 || exit 1
for cfg in "MOAI_NTT_COOP=0" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=2" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=4" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=6" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=8" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=12" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=3 MOAI_NTT_COOP_DELAY=5" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=2 MOAI_NTT_COOP_DELAY=3" "MOAI_NTT_COOP=1 MOAI_NTT_COOP_WPC=4 MOAI_NTT_COOP_DELAY=32"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['fwd_ms'], d['roofline']['inv_ms'])"
done
