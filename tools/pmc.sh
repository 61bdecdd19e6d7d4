#!/bin/bash
# PMC passes over a python program (rocprofv3 --pmc only; never combined with trace domains other than kernel-trace),
# then one --kernel-trace --stats pass.  usage: tools/pmc.sh <tag> <script.py> [args]   (default: bench.py, NTT only)
tag=$1; shift
if [ $# -eq 0 ]; then set -- bench.py --no-cpu-baseline --no-e2e --no-keyswitch --steps 2 --warmup 1; fi
prog=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $prog "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $prog "$@" > $out/stats.log 2>&1 || echo "stats failed"
cp $out/stats/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out > $out/${tag}_pmc_summary.json
head -c 6000 $out/${tag}_pmc_summary.json
