#!/bin/bash
# sweep the Infinity-Cache chunk size of the NTT launcher on the GPU box
for mb in 0 32 64 96 128 160 192 256; do
  echo "== MOAI_NTT_CHUNK_MB=$mb"
  MOAI_NTT_CHUNK_MB=$mb timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['fwd_ms'], d['roofline']['inv_ms'])"
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
